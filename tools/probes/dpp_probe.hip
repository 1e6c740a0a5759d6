// Probe: what does a DPP read return when the SOURCE lane is disabled by EXEC?  (bound_ctrl = 1)
// Expected (GCN3/Vega ISA, "invalid lane" = out of row or EXEC = 0): 0.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
    const int lane = threadIdx.x;
    int v = 1000 + lane;            // every lane's register holds 1000 + lane
    int shl = -1, shr = -1, shl_nb = -1;
    if ((lane & 7) != 7) {          // lane 7 of every 8-lane group is disabled inside this region
        shl = __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true);      // row_shl:1 : lane <- lane+1
        shr = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1 : lane <- lane-1
        shl_nb = __builtin_amdgcn_update_dpp(-7, v, 0x101, 0xF, 0xF, false); // bound_ctrl = 0: keep old (-7) for invalid sources
    }
    out[lane] = shl; out[64 + lane] = shr; out[128 + lane] = shl_nb;
}
int main() {
    int* d; hipMalloc(&d, 192 * sizeof(int));
    probe<<<1, 64>>>(d);
    int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row_shl:1 bound_ctrl=1 :"); for (int i = 0; i < 16; i++) printf(" %d", h[i]); printf("\n");
    printf("row_shr:1 bound_ctrl=1 :"); for (int i = 0; i < 16; i++) printf(" %d", h[64 + i]); printf("\n");
    printf("row_shl:1 bound_ctrl=0 :"); for (int i = 0; i < 16; i++) printf(" %d", h[128 + i]); printf("\n");
    return 0;
}
