// Multi-GPU exchanges of the C ABI (include/pddp.h, "multi-GPU"): RCCL collectives on the solver's own stream.  gfx950 / ROCm only.
// Written against the public C ABI of the solver (pddp_stream, pddp_array_ptr, pddp_get_config) -- it needs nothing private of a handle.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/pddp.h"
#include "solver_state.hpp"

using namespace pddp;

extern int pddp_internal_fail(int code, const std::string& msg);    // sets pddp_last_error (pddp_api.hip)

static_assert(PDDP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the rendezvous blob is an ncclUniqueId");

struct pddp_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    int* d_flag = nullptr;            // [1] this rank's "still running", reduced in place
    double* d_costs = nullptr;        // [world][batch][2], grown on demand
    size_t cost_cap = 0;
    // the per-iteration cost table (pddp_comm_cost_table_begin / _end): its own stream, so that the solver's next sweep does not queue behind the exchange
    hipStream_t side = nullptr;
    ncclComm_t side_comm = nullptr;   // a duplicate of `comm` (ncclCommSplit, every rank one colour) for that exchange: RCCL serialises the operations of ONE communicator across streams,
                                      // so a pddp_comm_all_done poll issued while a table is in flight would otherwise queue behind the gather
    hipEvent_t ev_ls = nullptr, ev_done = nullptr;
    double* d_table = nullptr;        // [world + 1][batch][A]: gathered + this rank's send buffer
    double* h_table = nullptr;        // pinned, [world][batch][A]
    size_t table_cap = 0; int table_B = 0, table_A = 0; bool table_pending = false;
};

#define COMM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return pddp_internal_fail(PDDP_ENODEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define COMM_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return pddp_internal_fail(PDDP_ENODEVICE, std::string(#call) + ": " + ncclGetErrorString(r_)); } while (0)

// flag = 1 when some problem of this handle has not exited yet
template <typename T>
__global__ void k_comm_running(const SolverState<T>* st, int batch, int* flag) {
    int running = 0;
    for (int b = threadIdx.x; b < batch; b += blockDim.x) running |= (st[b].done == 0);
    running = __any(running);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (running && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
// (J_initial, J at the last iteration) of every local problem, as doubles
template <typename T>
__global__ void k_comm_costs(const SolverState<T>* st, const T* Jout, int stride, int batch, double* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    out[2 * b] = (double)Jout[(size_t)b * stride];
    // a problem that exited stopped at `iter` (the slot of its last iteration); one that is still running has `iter` pointing at the NEXT, unwritten slot
    const int last = st[b].done ? st[b].iter : (st[b].iter > 0 ? st[b].iter - 1 : 0);
    out[2 * b + 1] = (double)Jout[(size_t)b * stride + last];
}

// J[B][A] of the last line search as doubles
template <typename T>
__global__ void k_comm_table(const T* J, int count, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (double)J[i];
}

extern "C" int pddp_comm_unique_id(void* id) {
    if (!id) return pddp_internal_fail(PDDP_EINVAL, "pddp_comm_unique_id: null argument");
    ncclUniqueId u;
    COMM_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return 0;
}
extern "C" int pddp_comm_init(pddp_comm_handle* out, int rank, int world, const void* id, int device) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return pddp_internal_fail(PDDP_EINVAL, "pddp_comm_init: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return pddp_internal_fail(PDDP_ENODEVICE, "pddp_comm_init: no such HIP device");
    COMM_HIP(hipSetDevice(device));
    pddp_comm* c = new pddp_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return pddp_internal_fail(PDDP_ENODEVICE, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    if (hipMalloc((void**)&c->d_flag, sizeof(int)) != hipSuccess) { ncclCommDestroy(c->comm); delete c; return pddp_internal_fail(PDDP_ENOMEM, "pddp_comm_init: hipMalloc"); }
    *out = c;
    return 0;
}
extern "C" int pddp_comm_destroy(pddp_comm_handle c) {
    if (!c) return 0;
    if (c->d_flag) hipFree(c->d_flag);
    if (c->d_costs) hipFree(c->d_costs);
    if (c->d_table) hipFree(c->d_table);
    if (c->h_table) hipHostFree(c->h_table);
    if (c->ev_ls) hipEventDestroy(c->ev_ls);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->side) hipStreamDestroy(c->side);
    if (c->side_comm) ncclCommDestroy(c->side_comm);
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
    return 0;
}
extern "C" int pddp_comm_ranks(pddp_comm_handle c, int* rank, int* world) {
    if (!c) return pddp_internal_fail(PDDP_EINVAL, "null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return 0;
}

extern "C" int pddp_comm_allreduce_max(pddp_comm_handle c, double* value) {
    if (!c || !value) return pddp_internal_fail(PDDP_EINVAL, "pddp_comm_allreduce_max: null argument");
    COMM_HIP(hipSetDevice(c->device));
    if (c->cost_cap < 1) {
        if (hipMalloc((void**)&c->d_costs, 16 * sizeof(double)) != hipSuccess) return pddp_internal_fail(PDDP_ENOMEM, "pddp_comm_allreduce_max: hipMalloc");
        c->cost_cap = 16;
    }
    COMM_HIP(hipMemcpy(c->d_costs, value, sizeof(double), hipMemcpyHostToDevice));
    COMM_NCCL(ncclAllReduce(c->d_costs, c->d_costs, 1, ncclDouble, ncclMax, c->comm, 0));
    COMM_HIP(hipStreamSynchronize(0));
    COMM_HIP(hipMemcpy(value, c->d_costs, sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

static int solver_views(pddp_handle h, pddp_config& cfg, hipStream_t& stream, void*& state, void*& Jout) {
    int rc;
    if ((rc = pddp_get_config(h, &cfg))) return rc;
    void* st = nullptr; size_t nb = 0;
    if ((rc = pddp_stream(h, &st))) return rc;
    stream = (hipStream_t)st;
    if ((rc = pddp_array_ptr(h, "state", &state, &nb))) return rc;
    if ((rc = pddp_array_ptr(h, "Jout", &Jout, &nb))) return rc;
    return 0;
}

extern "C" int pddp_comm_all_done(pddp_comm_handle c, pddp_handle h, int* all_done) {
    if (!c || !h || !all_done) return pddp_internal_fail(PDDP_EINVAL, "pddp_comm_all_done: null argument");
    COMM_HIP(hipSetDevice(c->device));
    pddp_config cfg; hipStream_t s; void *state, *Jout;
    int rc = solver_views(h, cfg, s, state, Jout);
    if (rc) return rc;
    if (cfg.dtype == 1) hipLaunchKernelGGL((k_comm_running<double>), dim3(1), dim3(256), 0, s, (const SolverState<double>*)state, cfg.batch, c->d_flag);
    else hipLaunchKernelGGL((k_comm_running<float>), dim3(1), dim3(256), 0, s, (const SolverState<float>*)state, cfg.batch, c->d_flag);
    COMM_HIP(hipGetLastError());
    COMM_NCCL(ncclAllReduce(c->d_flag, c->d_flag, 1, ncclInt32, ncclMax, c->comm, s));
    int flag = 1;
    COMM_HIP(hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    COMM_HIP(hipStreamSynchronize(s));
    *all_done = flag ? 0 : 1;
    return 0;
}

extern "C" int pddp_comm_allgather_costs(pddp_comm_handle c, pddp_handle h, double* costs) {
    if (!c || !h || !costs) return pddp_internal_fail(PDDP_EINVAL, "pddp_comm_allgather_costs: null argument");
    COMM_HIP(hipSetDevice(c->device));
    pddp_config cfg; hipStream_t s; void *state, *Jout;
    int rc = solver_views(h, cfg, s, state, Jout);
    if (rc) return rc;
    const size_t B = cfg.batch, need = (size_t)(c->world + 1) * B * 2;          // [world][B][2] gathered + this rank's [B][2] send buffer behind it
    if (need > c->cost_cap) {
        if (c->d_costs) hipFree(c->d_costs);
        c->d_costs = nullptr; c->cost_cap = 0;
        if (hipMalloc((void**)&c->d_costs, need * sizeof(double)) != hipSuccess) return pddp_internal_fail(PDDP_ENOMEM, "pddp_comm_allgather_costs: hipMalloc");
        c->cost_cap = need;
    }
    double* send = c->d_costs + (size_t)c->world * B * 2;
    const int stride = cfg.max_iter + 2;
    if (cfg.dtype == 1) hipLaunchKernelGGL((k_comm_costs<double>), dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, (const SolverState<double>*)state, (const double*)Jout, stride, (int)B, send);
    else hipLaunchKernelGGL((k_comm_costs<float>), dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, (const SolverState<float>*)state, (const float*)Jout, stride, (int)B, send);
    COMM_HIP(hipGetLastError());
    COMM_NCCL(ncclAllGather(send, c->d_costs, B * 2, ncclDouble, c->comm, s));
    std::vector<double> rank_major((size_t)c->world * B * 2);
    COMM_HIP(hipMemcpyAsync(rank_major.data(), c->d_costs, rank_major.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    COMM_HIP(hipStreamSynchronize(s));
    for (int r = 0; r < c->world; r++)                                           // [rank][local] -> global problem id g = local * world + rank
        for (size_t l = 0; l < B; l++) {
            const size_t g = l * c->world + r;
            costs[2 * g] = rank_major[((size_t)r * B + l) * 2]; costs[2 * g + 1] = rank_major[((size_t)r * B + l) * 2 + 1];
        }
    return 0;
}

// The per-iteration cost-table exchange north_star names ("an RCCL all-reduce of the per-alpha cost over xGMI"; SURVEY.md section 8(e) mode R: one all-gather of the [B x A]
// table per iteration so that every rank can report / select globally).  The solves of different ranks are independent, so nothing on the DATA path needs it: it is
// optional, and it is kept OFF the sweep's critical path -- _begin copies the table out behind what the solver's stream holds (one small kernel after the line search of the last
// enqueued sweep), records an event, and gathers on the communicator's own stream; the solver's next sweeps run meanwhile.  _end waits for the exchange and returns the table in global problem order.
extern "C" int pddp_comm_cost_table_begin(pddp_comm_handle c, pddp_handle h) {
    if (!c || !h) return pddp_internal_fail(PDDP_EINVAL, "pddp_comm_cost_table_begin: null argument");
    if (c->table_pending) return pddp_internal_fail(PDDP_EINVAL, "pddp_comm_cost_table_begin: the previous exchange has not been collected (pddp_comm_cost_table_end)");
    COMM_HIP(hipSetDevice(c->device));
    pddp_config cfg; hipStream_t s; void *state, *Jout;
    int rc = solver_views(h, cfg, s, state, Jout);
    if (rc) return rc;
    void* J = nullptr; size_t nb = 0;
    if ((rc = pddp_array_ptr(h, "J", &J, &nb))) return rc;
    const size_t B = cfg.batch, A = cfg.A, need = (size_t)(c->world + 1) * B * A;
    if (!c->side_comm) {
        // the FIRST _begin of a communicator is collective beyond the gather itself: every rank duplicates the communicator here (all ranks call _begin once per iteration
        // anyway).  Should the duplicate not be available the exchange runs on `comm` -- correct, only ordered with the other pddp_comm_* calls instead of beside them.
        COMM_HIP(hipStreamSynchronize(s));
        if (ncclCommSplit(c->comm, 0, c->rank, &c->side_comm, nullptr) != ncclSuccess) c->side_comm = nullptr;
    }
    if (!c->side) { COMM_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking)); COMM_HIP(hipEventCreateWithFlags(&c->ev_ls, hipEventDisableTiming)); COMM_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming)); }
    if (need > c->table_cap) {
        if (c->d_table) hipFree(c->d_table);
        if (c->h_table) hipHostFree(c->h_table);
        c->d_table = nullptr; c->h_table = nullptr; c->table_cap = 0;
        if (hipMalloc((void**)&c->d_table, need * sizeof(double)) != hipSuccess || hipHostMalloc((void**)&c->h_table, (size_t)c->world * B * A * sizeof(double)) != hipSuccess)
            return pddp_internal_fail(PDDP_ENOMEM, "pddp_comm_cost_table_begin: allocation");
        c->table_cap = need;
    }
    c->table_B = (int)B; c->table_A = (int)A;
    double* send = c->d_table + (size_t)c->world * B * A;
    // the table is COPIED OUT on the solver's own stream (one small kernel behind the last enqueued sweep's line search: the next sweep overwrites J), ...
    const unsigned blocks = (unsigned)((B * A + 255) / 256);
    if (cfg.dtype == 1) hipLaunchKernelGGL((k_comm_table<double>), dim3(blocks), dim3(256), 0, s, (const double*)J, (int)(B * A), send);
    else hipLaunchKernelGGL((k_comm_table<float>), dim3(blocks), dim3(256), 0, s, (const float*)J, (int)(B * A), send);
    COMM_HIP(hipGetLastError());
    COMM_HIP(hipEventRecord(c->ev_ls, s));
    COMM_HIP(hipStreamWaitEvent(c->side, c->ev_ls, 0));                          // ... the exchange itself runs beside the solver's stream
    COMM_NCCL(ncclAllGather(send, c->d_table, B * A, ncclDouble, c->side_comm ? c->side_comm : c->comm, c->side));
    COMM_HIP(hipMemcpyAsync(c->h_table, c->d_table, (size_t)c->world * B * A * sizeof(double), hipMemcpyDeviceToHost, c->side));
    COMM_HIP(hipEventRecord(c->ev_done, c->side));
    c->table_pending = true;
    return 0;
}
extern "C" int pddp_comm_cost_table_end(pddp_comm_handle c, double* table) {
    if (!c || !table) return pddp_internal_fail(PDDP_EINVAL, "pddp_comm_cost_table_end: null argument");
    if (!c->table_pending) return pddp_internal_fail(PDDP_EINVAL, "pddp_comm_cost_table_end: no exchange in flight (pddp_comm_cost_table_begin)");
    COMM_HIP(hipSetDevice(c->device));
    COMM_HIP(hipEventSynchronize(c->ev_done));
    c->table_pending = false;
    const size_t B = c->table_B, A = c->table_A;
    for (int r = 0; r < c->world; r++)                                           // [rank][local][A] -> global problem id g = local * world + rank
        for (size_t l = 0; l < B; l++) std::memcpy(table + ((size_t)l * c->world + r) * A, c->h_table + ((size_t)r * B + l) * A, A * sizeof(double));
    return 0;
}
