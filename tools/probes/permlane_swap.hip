// Checks on the device what csrc/bp_mfma.hpp's vector-ALU exchanges rely on (gfx950):
//   v_permlane16_swap_b32 vdst, vsrc : vdst's odd 16-lane rows <-> vsrc's even rows;  v_permlane32_swap_b32 vdst, vsrc : vdst's upper half <-> vsrc's lower half
//   => mx_row_of_group<GO>(w): lane (g, c) receives lane (GO, c)'s value;   DPP row_newbcast:L => mx_row_bcast<L>(w): lane L of the lane's own row;
//   v_fmac_f32_dpp r, -r, q row_newbcast:L = fma(-bcast_L(r), q, r)
// and times dependent chains of the three exchange forms of one Gauss-Jordan pivot (ds_bpermute round trip | swaps + DPP) on one wave.
// build: hipcc --offload-arch=gfx950 -O3 -I parallel-ddp_amd/csrc -o tools/probes/permlane_swap tools/probes/permlane_swap.hip
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "bp_mfma.hpp"
using namespace pddp;

template <int GO> __device__ void rows(unsigned* o, unsigned w) { o[GO * 64 + threadIdx.x] = mx_row_of_group<GO>(w); }
__global__ void k_sem(unsigned* o, float* f) {
    const unsigned w = 1000u + threadIdx.x;
    rows<0>(o, w); rows<1>(o, w); rows<2>(o, w); rows<3>(o, w);
    o[256 + threadIdx.x] = mx_row_bcast<5>(w);
    o[320 + threadIdx.x] = mx_row_bcast<12>(w);
    const float r = 0.37f * threadIdx.x + 1.f, q = 1.f / (3.f + threadIdx.x);
    f[threadIdx.x] = Mx<float>::fnma_row_bcast<9>(r, q);
    f[64 + threadIdx.x] = __builtin_fmaf(-Mx<float>::row_bcast<9>(r), q, r);
}
// MODE 0: ds_bpermute pivot row + two ds_bpermute column entries (round 4); 1: swaps + DPP moves; 2: swaps + v_fmac_dpp
template <int MODE> __global__ void k_chain(float* o, int n, long long* cyc) {
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    float R0 = 1.f + 0.01f * lane, R1 = 2.f - 0.01f * lane;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        const float piv = Mx<float>::readlane(R0, 17);
        float rp = Mx<float>::recip(piv + 3.f);
        float q, n0, n1;
        if (MODE == 0) {
            float prow = Mx<float>::from_lane_off<64>(R0, 4 * c);
            float c0 = Mx<float>::from_lane_off<16>(R0, 64 * g), c1 = Mx<float>::from_lane_off<16>(R1, 64 * g);
            Mx<float>::lanes_arrived(prow, c0, c1, rp);
            q = prow * rp; n0 = __builtin_fmaf(-c0, q, R0); n1 = __builtin_fmaf(-c1, q, R1);
        } else if (MODE == 1) {
            q = Mx<float>::row_of_group<1>(R0) * rp;
            n0 = __builtin_fmaf(-Mx<float>::row_bcast<4>(R0), q, R0); n1 = __builtin_fmaf(-Mx<float>::row_bcast<4>(R1), q, R1);
        } else {
            q = Mx<float>::row_of_group<1>(R0) * rp;
            n0 = Mx<float>::fnma_row_bcast<4>(R0, q); n1 = Mx<float>::fnma_row_bcast<4>(R1, q);
        }
        R0 = (g == 1) ? q : n0; R1 = n1;
    }
    const long long t1 = clock64();
    o[lane] = R0 + R1;
    if (lane == 0) cyc[MODE] = t1 - t0;
}
int main() {
    unsigned* o; float* f; long long* cyc;
    hipMalloc(&o, 384 * 4); hipMalloc(&f, 128 * 4); hipMalloc(&cyc, 64);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, o, f);
    std::vector<unsigned> h(384); std::vector<float> hf(128);
    hipMemcpy(h.data(), o, 384 * 4, hipMemcpyDeviceToHost); hipMemcpy(hf.data(), f, 128 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int go = 0; go < 4; go++) for (int l = 0; l < 64; l++) bad += h[go * 64 + l] != 1000u + 16 * go + (l & 15);
    for (int l = 0; l < 64; l++) { bad += h[256 + l] != 1000u + (l & 48) + 5; bad += h[320 + l] != 1000u + (l & 48) + 12; bad += hf[l] != hf[64 + l]; }
    printf("permlane_swap semantics: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
    const int n = 200000;
    float* d; hipMalloc(&d, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[3];
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, d, n, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, d, n, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, d, n, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[2], e0, e1);
    }
    printf("one wave, %d dependent pivots (HIP events; cycles at 2.4 GHz): ds_bpermute %.1f | swaps + dpp mov %.1f | swaps + fmac_dpp %.1f cycles per pivot\n", n,
           ms[0] * 2.4e6 / n, ms[1] * 2.4e6 / n, ms[2] * 2.4e6 / n);
    return bad != 0;
}
