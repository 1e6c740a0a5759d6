// Solver<Plant, Integrator, T>: buffers, kernel selection and launches of one handle (the host side of allocateMemory_GPU / runiLQR_GPU for a batch).  Included by the
// per-plant translation units only (pddp_plant_*.hip); the C ABI sees SolverBase (solver_base.hpp).
#pragma once
#include "solver_base.hpp"
#include "iiwa14_model_data.h"
#include "kernels.hpp"
#include "tl_launch.hpp"
#include "mx_launch.hpp"

using namespace pddp;

// (problem, block) pairs from which the matrix-core backward pass is the default for float handles of the arm (profiles/r02_path_sweep_mx.txt)
static constexpr size_t kBpMfmaMinBlocks = 1;
// pddp_kernel_selection values by name (0 = nullptr: the library's choice).  The selection logic below is written against these names -- they were the values of the
// PDDP_BP / PDDP_FP / ... environment variables it read until round 4.
static const char* ksel(int v, std::initializer_list<const char*> names) { return (v > 0 && (size_t)v <= names.size()) ? *(names.begin() + (v - 1)) : nullptr; }
static const char* ksel_bp(const pddp_config& c) { return ksel(c.kernels.bp, {"mx", "lg", "coop", "wide"}); }
static const char* ksel_fp(const pddp_config& c) { return ksel(c.kernels.fp, {"tl", "lg", "coop", "tl2", "tl4"}); }
static const char* ksel_sweep(const pddp_config& c) { return ksel(c.kernels.sweep, {"alpha", "st", "wg", "maps"}); }
static const char* ksel_ls(const pddp_config& c) { return ksel(c.kernels.ls, {"many", "wg"}); }
static const char* ksel_ab(const pddp_config& c) { return ksel(c.kernels.ab, {"full"}); }
static const char* ksel_cf(const pddp_config& c) { return ksel(c.kernels.cf, {"ts", "coop"}); }
static const char* ksel_cf_bp(const pddp_config& c) { return ksel(c.kernels.cf_bp, {"ts", "coop", "gl", "gl32", "cl", "mq"}); }
static const char* ksel_cf_fp(const pddp_config& c) { return ksel(c.kernels.cf_fp, {"ts", "coop", "cf"}); }
static const char* ksel_cf_nis(const pddp_config& c) { return ksel(c.kernels.cf_nis, {"ts", "coop", "gl", "gl8", "kb16", "kb32", "kb64", "kb20"}); }
template <typename T> static void fill_model(ArmModel<T>& m, const pddp_config& c) {
    const int v = c.wafr_urdf ? 1 : 0;
    for (int b = 0; b < 7; b++) {
        for (int i = 0; i < 36; i++) m.I[36 * b + i] = (T)IIWA14_SPATIAL_INERTIA[v][b][i];
        for (int i = 0; i < 16; i++) m.F[16 * b + i] = (T)IIWA14_JOINT_FRAME[v][b][i];
    }
    m.grav = (T)(c.mpc_mode ? 0.0 : 9.81);   // plants/dynamics_arm.cuh:42-46
    arm_model_apply_ee_type(m, c.wafr_urdf, c.ee_type);
}
static void fill_model(EmptyModel& m, const pddp_config&) { m.unused = 0; }

template <typename P, int INTEG, typename T>
struct Solver : SolverBase {
    static constexpr int NX = P::NX, NU = P::NU, NM = NX + NU, NP = P::NPOS;
    Buffers<T> b{};
    MpcBuffers<T> mb{};
    T* d_xActual = nullptr; T* d_goal_in = nullptr; int* d_shift = nullptr;
    unsigned char* d_mpc_out = nullptr;      // MPC outputs of a control cycle packed per problem by k_mpc_store (one transfer)
    unsigned char* h_state = nullptr;                              // pinned copy target of the solver states (status polls)
    unsigned char* h_stage = nullptr; size_t h_stage_bytes = 0;     // pinned host staging of the MPC call (inputs, then outputs): its transfers are asynchronous, one sync per control cycle
    Dims dm{};
    SolverParams sp{};
    CostWeights<T> cw{};
    T dt{};
    std::map<std::string, std::pair<void*, size_t>> arrays;
    std::vector<void*> allocs;
    // backward pass of the arm: the lane-group kernel carries 8 (problem, block) pairs per wave and wins once the GPU is
    // full; the wave-cooperative kernel has the shorter critical path for a handful of problems.  PDDP_BP=lg|coop overrides.
    bool bp_lane_groups = false;
    bool fp_coop = false;          // PDDP_FP=coop
    static constexpr int kNisTl7MaxBatch = 511;   // measured crossover against k_nis_lg (profiles/)
    bool fp_split = false;         // rollouts of a lane-group handle on the split thread-lane kernels (k_fp_tl4 / k_fp_tl2)
    bool fp_two_wave = false;
    bool ls_many = false;          // line search one thread per problem (k_ls_many): from 2048 problems in flight
    FpPath fp_path = kFpLg;        // the arm's forward pass / next-iteration setup (fp_tl.hpp select_fp_path)
    int tl_variant = -1;           // which built-in robot model the handle's tables equal (the thread-lane kernels fold it into literals); -1: neither
    T tl_grav = T(0);
    void derive_tl_model(const ArmModel<T>& hm) {
        ArmTlModel<T> m;
        tl_variant = -1;
        if (arm_tl_model_from_tables(m, hm))
            for (int v = 0; v < 2; v++) if (arm_tl_models_equal(m, arm_tl_builtin<T>(v))) tl_variant = v;
        tl_grav = hm.grav;
        // USE_FINITE_DIFF: the setup runs on the wave-cooperative kernel (k_nis: any plant's `dynamics`), which adopts the winner from the candidate-major
        // xs / us / ds -- so the rollouts stay on lane groups (they write those), not on the thread-lane kernels
        fp_path = select_fp_path(ksel_fp(cfg), sizeof(T) == 4, cfg.ee_cost != 0, tl_variant >= 0 && !cfg.use_finite_diff, cfg.batch);
        if ((cfg.use_limits || cfg.use_smooth_abs) && fp_path == kFpLg) fp_path = kFpCoop;      // USE_LIMITS_FLAG / USE_SMOOTH_ABS: the lane-group family does not carry the variants
        fp_coop = (fp_path == kFpCoop);
        // few problems in flight, joint-space cost, float, built-in robot model: the rollouts run on k_fp_tl2 (every step split over two wavefronts);
        // sweep, line search and setup stay on the lane-group kernels.  PDDP_FP=lg keeps the lane-group rollouts (bit-identity tests), PDDP_FP=tl2 asks for the split.
        const char* fpenv = ksel_fp(cfg);
        fp_split = sizeof(T) == 4 && fp_path == kFpLg && !cfg.use_finite_diff && tl_variant >= 0 && !(fpenv && std::string(fpenv) == "lg");
        // PDDP_FP=tl4 on a DOUBLE handle: the same few-problem selection (k_fp_tl4 pipeline + k_nis_tl7) in its parity instantiation (tests/test_f64_benched_family.py)
        if (sizeof(T) == 8 && fpenv && std::string(fpenv) == "tl4" && fp_path == kFpLg && !cfg.use_finite_diff && tl_variant >= 0 && !cfg.use_limits && !cfg.use_smooth_abs) fp_split = true;
        // the split's current form is the four-wave pipeline (k_fp_tl4, fp_pipe.hpp; also the end-effector cost family); PDDP_FP=tl2 keeps the two-wave kernel (joint-space cost only)
        fp_two_wave = fp_split && !cfg.ee_cost && fpenv && std::string(fpenv) == "tl2";
    }
    void derive_tl_model(const EmptyModel&) {}
    // pddp_set_array("model_I" / "model_F"): re-derive what the kernels take from the model tables as launch arguments
    void drop_graph() override { if (graph) { hipGraphExecDestroy(graph); graph = nullptr; graph_mode = -1; } if (graph_n) { hipGraphExecDestroy(graph_n); graph_n = nullptr; } }
    int ab_view(int to_compact) override {
        if constexpr (P::PLANT == 4) {
            if (b.ABc) { launch_abc_convert<T>(stream, b, (int)(cfg.batch * cfg.N), cfg.N, dt, to_compact); HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(stream)); }
        }
        return 0;
    }
    int ab_keep_reference_layout() override {
        if (!b.ABc) return 0;
        int rc = ab_view(0);       // (the caller refreshed the reference-layout H from the compact position block BEFORE it wrote into it: pddp_set_array)
        b.ABc = nullptr; b.Hc = nullptr; drop_graph();
        return rc;
    }
    // end-effector handles with the compact position block: refresh the reference-layout array "H" of the running knots from it (API view)
    int h_view() override {
        if constexpr (P::PLANT == 4) {
            if (b.Hc) {
                hipLaunchKernelGGL((k_hc_expand<T>), dim3((cfg.batch * cfg.N + 63) / 64), dim3(64), 0, stream, b, (int)(cfg.batch * cfg.N), cfg.N, hw[1], hw[2]);
                HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(stream));
            }
        }
        return 0;
    }
    int cand_view(int to_records) override {
        if constexpr (P::PLANT != 4) {
            if (b.xw && cf_fp_staged) {
                const dim3 g((unsigned)(((size_t)cfg.batch * cfg.N * cfg.A + 255) / 256));
                if (to_records) hipLaunchKernelGGL((k_cand_to_xw<P, T>), g, dim3(256), 0, stream, b, dm, (int)cfg.batch);
                else hipLaunchKernelGGL((k_xw_to_cand<P, T>), g, dim3(256), 0, stream, b, dm, (int)cfg.batch);
                HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(stream));
            }
        }
        cand_stale = false;
        return 0;
    }
    int reference_views(int what) override {
        if ((what & 1) && cfg.M > 1) { int rc = ab_view(0); if (rc) return rc; }           // (the reference-layout [A B] from the compact one)
        hipLaunchKernelGGL((k_reference_views<P, T>), dim3(cfg.N, cfg.batch), dim3(64), 0, stream, b, dm, what);
        HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(stream));
        if (what & 1) fs_vars_stale = false;
        if (what & 2) cand_stale = false;                // (xs / us now hold the winner in every slot: the records must not be expanded over them by the next pddp_get_array)
        return 0;
    }
    int model_changed() override {
        typename P::Model hm;
        HIPCHK(hipMemcpy(&hm, b.model, sizeof(hm), hipMemcpyDeviceToHost));
        derive_tl_model(hm);
        drop_graph();
        return 0;
    }
    bool cf_bp = false, cf_fp = false, cf_nis = false;   // ... per phase
    bool gl_bp32 = false, gl_nis8 = false;
    bool cl_bp = false;                                  // 12 states + 4 controls: 16 lanes per block of knots, lane = column (k_bp_cl, bp_cl.hpp) instead of k_bp_gl; PDDP_CF_BP = cl | gl | gl32
    bool mq_bp = false;                                  // 12 states + 4 controls on the matrix cores (k_bp_mq, bp_mq.hpp): where k_bp_cl was the choice, for the plant's own diagonal cost Hessian; kernels.cf_bp = mq | cl
    bool mq_fused = false;                               // k_bp_mq composes the segments' forward-sweep maps itself (bp_mq.hpp FUSE) and k_sweep_maps_cf finishes: no A - B K / B du traffic, no per-knot sweep; kernels.sweep = st: k_sweep_cf
    bool cf_fp_staged = false;                           // thread-serial rollouts with the knot's operands staged through LDS once per wavefront (k_fp_cf: 16 step sizes, 12-state plants); PDDP_CF_FP = cf | ts
    int kb_nis = 0;                                      // knots per wavefront of the knot-batched setup kernel (k_nis_kb: scalar plug-ins, RK3); 0 = k_nis_gl.  PDDP_CF_NIS = kb16 | kb32 | kb64
    bool gl_bp = false, gl_nis = false;                  // 16 lanes per unit (k_bp_gl / k_nis_gl): the 12-state plants with the device full; PDDP_CF_BP / _NIS = gl
    bool cf_serial = false;        // closed-form plants with many problems in flight: thread-serial kernels (k_bp_ts / k_fp_ts / k_nis_ts); PDDP_CF=coop|ts overrides
    bool bp_wide = false;          // cooperative backward pass with a whole workgroup per block of knots (few problems in flight); PDDP_BP=wide
    bool phase_fused_sweep = false; // pddp_run_phase(PDDP_PHASE_BP_FUSED / _SWEEP_FUSED): the teacher-forcing hook runs the production sweep path (maps composed in the backward pass)
    bool sweep_fused = false;      // production sweeps: forward-sweep maps composed inside k_bp_mfma + k_sweep_maps (no A - B K / B du traffic)
    int sweep_kind = 0;            // the arm's linear sweep: 0 one lane group per candidate (k_sweep_lg), 1 two sequences on lane groups (k_sweep_st), 2 two sequences, workgroup per problem (k_sweep_wg); PDDP_SWEEP=alpha|st|wg
    bool mpc_used = false;         // pddp_mpc_solve ran on this handle: its warm start shifts every cost-to-go slot, so the backward pass keeps writing all of them
    bool lean_ctg_ran = false;     // sweeps ran that left the interior cost-to-go slots unwritten (config.boundary_cost_to_go_only): a warm-started MPC call would shift stale slots
    // every knot's P, p written (the reference's d_P / d_p) unless the caller opted out; MPC handles always keep them (MPCHelpers.cuh:602-655 shifts the whole arrays)
    bool keep_all_ctg() const { return !cfg.boundary_cost_to_go_only || cfg.mpc_mode || mpc_used; }
    bool bp_mfma = false;          // matrix-core backward pass, one wavefront per block of knots (bp_mfma.hpp): float handles of the arm; PDDP_BP=mx
    // few problems in flight on the four-wave rollout pipeline, every problem's M x A rollouts inside one wavefront: the rollout kernel ends with the line search (k_fp_tl4)
    bool ls_in_rollouts() const { return P::PLANT == 4 && fp_split && !fp_two_wave && !ls_many && cfg.kernels.ls == 0 && 64 % (cfg.M * cfg.A) == 0; }
    // ... and BEGINS with the linear forward sweep: the segment maps the backward pass composed are applied in the rollout kernel's prologue (a problem's M x A lanes sit in
    // one wavefront, 14 of them walk the maps) instead of by a k_sweep_maps launch in front of it -- one more kernel boundary of the ~115 us iteration gone.
    // kernels.sweep = maps keeps the separate kernel (A/B, tests); so do the phase hooks and the per-phase / per-kernel timing, which launch the sweep on its own.
    bool cf_records() const { return P::PLANT != 4 && cf_fp && cf_fp_staged; }      // production rollouts = k_sweep_cf + k_fp_cf (records in xw)
    bool maps_in_rollouts() const { return P::PLANT == 4 && sweep_fused && cfg.kernels.sweep == 0 && fp_split && !fp_two_wave && 64 % (cfg.M * cfg.A) == 0 && cfg.M * cfg.A >= 16; }
    hipGraphExec_t graph = nullptr;
    int graph_mode = -1;
    size_t fp_lds = 0;

    ~Solver() override {
        if (graph) hipGraphExecDestroy(graph);
        if (graph_n) hipGraphExecDestroy(graph_n);
        for (void* p : allocs) hipFree(p);
        for (void* p : scratch_buf) if (p) hipFree(p);
        if (h_stage) hipHostFree(h_stage);
        if (h_state) hipHostFree(h_state);
        if (stream) hipStreamDestroy(stream);
    }
    void register_model(void* dmodel, const ArmModel<T>&) {
        arrays["model_I"] = {dmodel, sizeof(T) * kArmNB * 36};
        arrays["model_F"] = {(char*)dmodel + offsetof(ArmModel<T>, F), sizeof(T) * kArmNB * 16};
    }
    void register_model(void*, const EmptyModel&) {}
    template <typename U> int alloc(const char* name, U** out, size_t count) {
        void* p = nullptr;
        if (hipMalloc(&p, count * sizeof(U)) != hipSuccess) return fail(PDDP_ENOMEM, std::string("hipMalloc failed for ") + name);
        if (hipMemset(p, 0, count * sizeof(U)) != hipSuccess) return fail(PDDP_ENODEVICE, "hipMemset failed");
        allocs.push_back(p); arrays[name] = {p, count * sizeof(U)}; *out = (U*)p;
        return 0;
    }
    int init() override {
        const pddp_config& c = cfg;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= c.device)
            return fail(PDDP_ENODEVICE, "no HIP device available: libpddp has no CPU fallback");
        HIPCHK(hipSetDevice(c.device));
#ifdef PDDP_REF_PLANT_FILE
        if constexpr (P::PLANT == 5) { const std::string complaint = ref_plugin_setup<T>(c.N); if (!complaint.empty()) return fail(PDDP_EINVAL, complaint); }
#endif
        if constexpr (P::PLANT == 5) {
            if (!scalar_plugin_qdd_is_dynamics<P, T>())
                return fail(PDDP_EINVAL, "plant 5: the plug-in's gradient routine returns a qdd that differs from its dynamics routine at the same state; the kernel families build the "
                                         "integrators' stage states from either one, so the two have to be the same numbers (call the dynamics routine inside the gradient routine)");
        }
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        dm.N = c.N; dm.M = c.M; dm.A = c.A; dm.NB = c.N / c.M;
        bp_lane_groups = (size_t)c.batch * c.M >= 4096;     // measured crossovers on MI355X (Kuka N=128): wide <= 256 problems < cooperative < 1024 <= lane groups
        bp_wide = (size_t)c.batch * c.M <= 1024 && P::NX >= 12;
        if (const char* v = ksel_fp(cfg)) fp_coop = (std::string(v) == "coop");       // the arm refines this below (derive_tl_model)
        // closed-form plants (and user plants): one wave per unit while a handful of problems is in flight (the shorter critical path), one thread per unit once
        // the batch fills the device (64 units per wave instead of 1); the horizon has to fit the per-thread cost table of k_fp_ts
        { const char* e = ksel_ls(cfg); ls_many = e ? std::string(e) == "many" : c.batch >= 2048; }      // PDDP_LS=many|wg
        cf_serial = P::PLANT != 4 && (size_t)c.batch * c.M >= 256 && c.N <= kTsMaxN && c.M <= kTsMaxM;
        if (const char* v = ksel_cf(cfg)) cf_serial = P::PLANT != 4 && std::string(v) == "ts" && c.N <= kTsMaxN && c.M <= kTsMaxM;
        // measured on MI355X (tools/cf_variants.py; profiles/r03_closed_form_variants.txt): the rollouts always win thread-serially once the device is full (cart-pole, 16384
        // problems: 0.74 against 15.7 ms; quadrotor, 4096: 3.0 against 57.9 ms); the derivative kernel too for the small plants (0.16 against 2.1 ms) but not for the
        // quadrotor's 12 states, whose per-thread stage scratch spills to memory (6.2 against 5.4 ms); the backward pass thread-serially only for the small plants with the
        // device full (0.42 against 0.65 ms; quadrotor: its 64-knot serial chain on private memory takes 15 ms against 1.9 ms for a wave per block)
        cf_fp = cf_serial;
        cf_nis = cf_serial && P::NX < 12;
        cf_bp = cf_serial && P::NX < 12 && (size_t)c.batch * c.M >= 8192;
        if (ksel_cf(cfg)) cf_bp = cf_nis = cf_fp;         // the override forces every phase
        // per-phase overrides (measurement): PDDP_CF_BP / PDDP_CF_FP / PDDP_CF_NIS = coop | ts
        if (const char* v = ksel_cf_bp(cfg)) cf_bp = P::PLANT != 4 && std::string(v) == "ts";
        if (const char* v = ksel_cf_fp(cfg)) cf_fp = P::PLANT != 4 && std::string(v) == "ts" && c.N <= kTsMaxN && c.M <= kTsMaxM;
        if (const char* v = ksel_cf_nis(cfg)) cf_nis = P::PLANT != 4 && std::string(v) == "ts";
        gl_nis = cf_serial && !cf_nis && P::NX + P::NU <= 16 && !ksel_cf(cfg);
        gl_bp = cf_serial && !cf_bp && P::NX + P::NU <= 16 && (size_t)c.batch * c.M >= 8192 && !ksel_cf(cfg); gl_bp32 = true;      // 32 lanes per block of knots: 1.92 -> 1.72 ms (quadrotor, 4096 problems); 16 lanes: 2.5 ms
        if (const char* v = ksel_cf_nis(cfg)) { gl_nis = P::PLANT != 4 && (std::string(v) == "gl" || std::string(v) == "gl8") && P::NX + P::NU <= 16; gl_nis8 = std::string(v) == "gl8"; }
        cl_bp = gl_bp && P::NX == 12 && P::NU == 4;
        mq_bp = cl_bp;                                                    // round 5: 4.1 -> ... ms at 16384 quadrotor problems (profiles/r05_quad_mfma.md)
        const bool cf_fits = (c.A == 16 && (64 / 16) * P::NX <= 64) || (c.A == 8 && (64 / 8) * P::NX <= 64);      // whole problems per wavefront, one state fetch per lane
        cf_fp_staged = cf_fp && cf_fits && !ksel_cf(cfg);      // (cart-pole, 16384 problems: 0.73 -> 0.69 ms; quadrotor: 8.6 -> 5.1 ms)
        if (const char* v = ksel_cf_fp(cfg)) { if (std::string(v) == "cf") { cf_fp = P::PLANT != 4 && c.N <= kTsMaxN && c.M <= kTsMaxM; cf_fp_staged = cf_fp && cf_fits; } else cf_fp_staged = false; }
        kb_nis = (gl_nis && c.integrator == 3) ? 16 : 0;      // 16 knots per wavefront: 2.15 ms (32: 2.6, 64: 3.6; the 16-lane-group kernel 5.7-7.1) at 16384 quadrotor problems -- LDS per block sets the occupancy
        if (const char* v = ksel_cf_nis(cfg)) { const std::string m(v); kb_nis = (P::PLANT != 4 && P::NX + P::NU <= 16 && c.integrator == 3) ? (m == "kb16" ? 16 : m == "kb20" ? 20 : m == "kb32" ? 32 : m == "kb64" ? 64 : 0) : 0; if (kb_nis) { gl_nis = true; cf_nis = false; } }
        if (const char* v = ksel_cf_bp(cfg)) {
            const std::string m(v);
            const bool col = (m == "cl" || m == "mq") && P::NX == 12 && P::NU == 4;
            gl_bp = P::PLANT != 4 && (m == "gl" || m == "gl32" || col) && P::NX + P::NU <= 16; gl_bp32 = m == "gl32"; cl_bp = gl_bp && col; mq_bp = cl_bp && m == "mq";
        }
        // the matrix-core backward pass of the 12-state plants with the record rollouts behind it: fused sweep maps by default (round 6: profiles/r06_quad.md)
        if constexpr (P::PLANT != 4 && P::NX == 12 && P::NU == 4 && P::kScalarPlugin) {
            mq_fused = gl_bp && cl_bp && mq_bp && !cf_bp && cf_fp && cf_fp_staged && c.M > 1 && (!ksel_sweep(cfg) || std::string(ksel_sweep(cfg)) == "maps");
        }
        if (P::PLANT == 4 && sizeof(T) == 4) {        // float handles of the arm: measured crossover (profiles/r02b_sweep_wg.txt): the staged workgroup sweep up to 512 problems
            sweep_kind = (c.batch <= 512 && c.N / c.M <= 96) ? 2 : 1;      // (a segment has to fit the 96-knot staging area of k_sweep_wg)
            if (const char* v = ksel_sweep(cfg)) sweep_kind = std::string(v) == "alpha" ? 0 : std::string(v) == "st" ? 1 : std::string(v) == "wg" ? 2 : sweep_kind;
            if (sweep_kind == 2 && c.N / c.M > 96) sweep_kind = 1;
            // default with the matrix-core backward pass: that pass composes the segments' sweep maps itself (bp_mfma.hpp kMxFuseSweep) and k_sweep_maps finishes;
            // sweep_kind stays the kernel of the phase hook, whose teacher-forced A - B K / B du must be what the sweep reads.  PDDP_SWEEP=alpha|st|wg: no fusion.
        }
        bp_mfma = (P::PLANT == 4 && sizeof(T) == 4 && (size_t)c.batch * c.M >= kBpMfmaMinBlocks);
        // PDDP_BP=mx on a double handle: the same tile algebra on v_mfma_f64_16x16x4_f64 (a test selection: float64 handles default to the lane-group family,
        // whose operation order is the reference's)
        if (const char* v = ksel_bp(cfg)) { bp_lane_groups = (std::string(v) == "lg"); bp_wide = (std::string(v) == "wide"); bp_mfma = (P::PLANT == 4 && std::string(v) == "mx"); }
        sweep_fused = bp_mfma && c.M > 1 && (!ksel_sweep(cfg) || std::string(ksel_sweep(cfg)) == "maps");
        sp.max_iter = c.max_iter; sp.out_stride = c.max_iter + 2; sp.ignore_max_rho_exit = c.ignore_max_rho_exit; sp.tol_cost = c.tol_cost;
        sp.exp_red_min = c.exp_red_min; sp.exp_red_max = c.exp_red_max; sp.max_defect = c.max_defect; sp.rho_init = c.rho_init; sp.ee_initial_cost_fix = c.ee_initial_cost_fix;
        cw.Q1 = (T)c.Q1; cw.Q2 = (T)c.Q2; cw.R = (T)c.R; cw.QF1 = (T)c.QF1; cw.QF2 = (T)c.QF2;
        cw.ee = c.ee_cost; cw.Q_EE1 = (T)c.Q_EE1; cw.Q_EE2 = (T)c.Q_EE2; cw.QF_EE1 = (T)c.QF_EE1; cw.QF_EE2 = (T)c.QF_EE2; cw.R_EE = (T)c.R_EE;
        cw.Q_xEE = (T)c.Q_xEE; cw.QF_xEE = (T)c.QF_xEE; cw.Q_xdEE = (T)c.Q_xdEE; cw.QF_xdEE = (T)c.QF_xdEE; cw.ee_z = (T)c.ee_on_link_z;
        cw.fd_eps = c.use_finite_diff ? c.finite_diff_epsilon : 0.0;
        cw.limits = (P::PLANT == 4) ? c.use_limits : 0;
        cw.smooth_abs = (P::PLANT == 4 && c.ee_cost) ? c.use_smooth_abs : 0; cw.sa = (T)c.smooth_abs_alpha; cw.sa2 = (T)(c.smooth_abs_alpha * c.smooth_abs_alpha);
        dt = (T)(c.total_time / (c.N - 1));                       // TIME_STEP, config.cuh:136
        const size_t B = c.batch, N = c.N, A = c.A, M = c.M;
        int rc = 0;
#define AL(name, count) if ((rc = alloc(#name, &b.name, (count)))) return rc
        AL(xs, B * A * N * NX); AL(us, B * A * N * NU); AL(ds, B * A * N * NX);
        AL(xb, B * 2 * N * NX); AL(ucur, B * N * NU); AL(dcur, B * N * NX);
        AL(P, 2 * B * N * NX * NX); AL(p, 2 * B * N * NX);     // double buffers: the second half is Pp / pp
        AL(AB, B * N * NX * NM); AL(H, B * N * NM * NM); AL(g, B * N * NM);
        AL(KT, B * N * NX * NU); AL(du, B * N * NU); AL(ApBK, B * N * NX * NX); AL(Bdu, B * N * NX);
        AL(J, B * A); AL(dmax, B * A); AL(dJexp, B * 2 * M); AL(alpha, A); AL(xGoal, B * NX);
        AL(Jout, B * (c.max_iter + 2)); AL(err, B * M); AL(alphaOut, B * (c.max_iter + 2)); AL(state, B);
#undef AL
        b.Pp = b.P + B * N * NX * NX; b.pp = b.p + B * N * NX;
        arrays["P"].second /= 2; arrays["p"].second /= 2;
        arrays["Pp"] = {b.Pp, arrays["P"].second}; arrays["pp"] = {b.pp, arrays["p"].second};
        if ((rc = alloc("x_old", &mb.x_old, B * N * NX)) || (rc = alloc("u_old", &mb.u_old, B * N * NU)) || (rc = alloc("KT_old", &mb.KT_old, B * N * NX * NU))) return rc;
        // MPC inputs of a control cycle in ONE device run (one transfer): measured states | goals | shifts
        if ((rc = alloc("mpc_in", &d_xActual, 2 * B * NX + B * sizeof(int) / sizeof(T) + 2))) return rc;
        d_goal_in = d_xActual + B * NX; d_shift = reinterpret_cast<int*>(d_goal_in + B * NX);
        arrays["xActual"] = {d_xActual, B * NX * sizeof(T)}; arrays["shift"] = {d_shift, B * sizeof(int)};      // the views the facade's GPUVars name (MPCHelpers.hpp)
        if ((rc = alloc("xTarget", &b.xTarget, B * NX)) || (rc = alloc("costk", &b.costk, B * N)) || (rc = alloc("tshift", &b.tshift, B))) return rc;
        std::vector<T> al(A);
        for (size_t i = 0; i < A; i++) al[i] = (T)std::pow(c.alpha_base, (double)i);   // nisInitHelpers.cuh:829
        HIPCHK(hipMemcpy(b.alpha, al.data(), A * sizeof(T), hipMemcpyHostToDevice));
        typename P::Model hm; fill_model(hm, c);
        void* dmodel = nullptr;
        HIPCHK(hipMalloc(&dmodel, sizeof(hm))); allocs.push_back(dmodel);
        HIPCHK(hipMemcpy(dmodel, &hm, sizeof(hm), hipMemcpyHostToDevice));
        b.model = dmodel;
        register_model(dmodel, hm);
        derive_tl_model(hm);
        if constexpr (P::PLANT != 4) { if (cf_fp_staged) { b.xw_rec = P::NX + P::NU; if ((rc = alloc("xw", &b.xw, B * N * A * b.xw_rec))) return rc; } }   // records of the staged closed-form rollouts (k_fp_cf)
        if constexpr (P::PLANT == 4) { if (fp_path == kFpTl) { b.xw_rec = 22; if ((rc = alloc("xw", &b.xw, B * N * A * b.xw_rec))) return rc; } }   // knot-major candidate states (fp_tl.hpp)
        if constexpr (P::PLANT == 4) { if (sweep_fused) { if ((rc = alloc("segmap", &b.segmap, B * M * 256))) return rc; } }
        if (mq_fused) { if ((rc = alloc("segmap", &b.segmap, B * M * 256))) return rc; }
        if constexpr (P::PLANT == 4) {
            const char* abenv = ksel_ab(cfg);             // PDDP_AB=full: keep the reference layout (comparison runs)
            const bool full_h = c.ee_cost && c.use_limits;         // end-effector cost with USE_LIMITS_FLAG: the whole diagonal of H moves with the trajectory -> reference-layout H and [A B]
            if (bp_mfma && fp_path == kFpTl && !(abenv && abenv[0] == 'f') && !full_h) {
                if ((rc = alloc("ABc", &b.ABc, abc_floats(B * N)))) return rc;
                // end-effector cost: the Gauss-Newton Hessian's only dense part, the 7 x 7 position block, travels compact as well (bp_mfma.hpp HQQ)
                if (c.ee_cost) { if ((rc = alloc("Hc", &b.Hc, B * N * 49 + 16))) return rc; }
            }
        }
        if ((rc = alloc("Jpart", &b.Jpart, B * A * M)) || (rc = alloc("dpart", &b.dpart, B * A * M)) || (rc = alloc("parts_fresh", &b.parts_fresh, B))) return rc;
        // device tables of per-alpha pointers, the reference's d_x / d_u / d_d (nisInitHelpers.cuh:777-789,808-813)
        void** tab[3]; const char* tn[3] = {"xs_ptrs", "us_ptrs", "ds_ptrs"};
        T* base[3] = {b.xs, b.us, b.ds}; const size_t per[3] = {N * NX, N * NU, N * NX};
        for (int t = 0; t < 3; t++) {
            if ((rc = alloc(tn[t], &tab[t], B * A))) return rc;
            std::vector<void*> hp(B * A);
            for (size_t i = 0; i < B * A; i++) hp[i] = base[t] + i * per[t];
            HIPCHK(hipMemcpy(tab[t], hp.data(), B * A * sizeof(void*), hipMemcpyHostToDevice));
        }
        fp_lds = FpLds<P, T>::bytes(c.M, c.N);
        if constexpr (P::PLANT == 4) {                                 // lane-group forward pass: A * (N + M) elements of dynamic LDS per workgroup
            const size_t a_wg = (c.A > 8 && c.A % 8 == 0) ? 8 : c.A;       // candidates per workgroup (launch_fp)
            const size_t lds = a_wg * (c.N + c.M) * sizeof(T);
            if (lds > 160 * 1024) return fail(PDDP_EINVAL, "forward-pass LDS footprint (candidates per workgroup * (N + M) elements) exceeds 160 KiB: reduce A or N");
            if (lds > 48 * 1024) {
                const void* ks[6] = {reinterpret_cast<const void*>(&k_fp_lg<T, 256, false>), reinterpret_cast<const void*>(&k_fp_lg<T, 512, false>),
                                     reinterpret_cast<const void*>(&k_fp_lg<T, 1024, false>), reinterpret_cast<const void*>(&k_fp_lg<T, 256, true>),
                                     reinterpret_cast<const void*>(&k_fp_lg<T, 512, true>), reinterpret_cast<const void*>(&k_fp_lg<T, 1024, true>)};
                for (const void* k : ks) HIPCHK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
        }
        const bool uses_coop_fp = (P::PLANT != 4) || fp_coop || cfg.use_limits || cfg.use_smooth_abs;      // the arm's forward pass runs on lane groups (no per-segment LDS scratch) unless PDDP_FP=coop
        if (uses_coop_fp) {
            if (fp_lds > 160 * 1024) return fail(PDDP_EINVAL, "forward-pass LDS footprint exceeds 160 KiB: reduce M");
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fp<P, INTEG, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp_lds));
        }
        HIPCHK(hipDeviceSynchronize());
        return 0;
    }
    int load(const void* x0, const void* u0, const void* xg, const void* KT0, const void* P0, const void* p0, const void* d0, int rollout, int clear,
             int ignore_first_defect) override {
        const size_t B = cfg.batch, N = cfg.N;
        // current trajectory goes to half 0 of xb (state.cur = 0 after init): one strided copy for the whole batch
        HIPCHK(hipMemcpy2DAsync(b.xb, 2 * N * NX * sizeof(T), x0, N * NX * sizeof(T), N * NX * sizeof(T), B, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(b.ucur, u0, B * N * NU * sizeof(T), hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(b.xGoal, xg, B * NX * sizeof(T), hipMemcpyHostToDevice, stream));
        if (clear) {                                                 // clearVarsFlag, nisInitHelpers.cuh:612-619
            HIPCHK(hipMemsetAsync(b.P, 0, B * N * NX * NX * sizeof(T), stream)); HIPCHK(hipMemsetAsync(b.Pp, 0, B * N * NX * NX * sizeof(T), stream));
            HIPCHK(hipMemsetAsync(b.p, 0, B * N * NX * sizeof(T), stream)); HIPCHK(hipMemsetAsync(b.pp, 0, B * N * NX * sizeof(T), stream));
            HIPCHK(hipMemsetAsync(b.KT, 0, B * N * NX * NU * sizeof(T), stream)); HIPCHK(hipMemsetAsync(b.dcur, 0, B * N * NX * sizeof(T), stream));
        } else {                                                     // warm start (:621-628); a NULL array keeps the device values
            if (P0) { HIPCHK(hipMemcpyAsync(b.P, P0, B * N * NX * NX * sizeof(T), hipMemcpyHostToDevice, stream)); HIPCHK(hipMemcpyAsync(b.Pp, P0, B * N * NX * NX * sizeof(T), hipMemcpyHostToDevice, stream)); }
            if (p0) { HIPCHK(hipMemcpyAsync(b.p, p0, B * N * NX * sizeof(T), hipMemcpyHostToDevice, stream)); HIPCHK(hipMemcpyAsync(b.pp, p0, B * N * NX * sizeof(T), hipMemcpyHostToDevice, stream)); }
            if (KT0) HIPCHK(hipMemcpyAsync(b.KT, KT0, B * N * NX * NU * sizeof(T), hipMemcpyHostToDevice, stream));
            if (d0) HIPCHK(hipMemcpyAsync(b.dcur, d0, B * N * NX * sizeof(T), hipMemcpyHostToDevice, stream));
        }
        HIPCHK(hipMemsetAsync(b.du, 0, B * N * NU * sizeof(T), stream));   // always (:630-632)
        HIPCHK(hipMemsetAsync(b.err, 0, B * cfg.M * sizeof(int), stream));
        HIPCHK(hipMemsetAsync(b.dmax, 0, B * cfg.A * sizeof(T), stream));
        const int ee = cfg.ee_cost ? 1 : 0;                          // end-effector cost: the initial cost comes out of the setup kernel (stage 2)
        HIPCHK(hipMemsetAsync(b.tshift, 0, B * sizeof(int), stream));
        if (rollout) {                                               // forwardRolloutFlag (:642-648)
            hipLaunchKernelGGL((k_init_cost<P, T>), dim3(B), dim3(64), N * sizeof(T), stream, b, dm, cw, sp, ignore_first_defect, 1, ee, 0);   // state.cur = 0
            launch_fp(stream, 1);
            hipLaunchKernelGGL((k_adopt_slot0<P, T>), dim3(N, B), dim3(64), 0, stream, b, dm);
        }
        hipLaunchKernelGGL((k_init_cost<P, T>), dim3(B), dim3(64), N * sizeof(T), stream, b, dm, cw, sp, ignore_first_defect, rollout, ee, 0);
        launch_nis(stream, 1);
        if (ee) hipLaunchKernelGGL((k_init_cost<P, T>), dim3(B), dim3(64), N * sizeof(T), stream, b, dm, cw, sp, ignore_first_defect, rollout, 2, 0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }
    // forward pass: the arm runs on lane groups (fp_lg.hpp), the closed-form plants on the wave-cooperative kernel
    // part: -1 everything; 0 only the linear sweep kernel (when the path has a separate one); 1 only the rollout kernel (per-kernel timing; kernels that sweep
    // themselves still do); 2 the rollouts WITHOUT any sweep, from the start states in xs (PDDP_PHASE_ROLLOUT)
    bool launch_sweep_maps_cf(hipStream_t s) {                // the record rollouts' sweep from the maps the fused matrix-core backward pass left (false: not this handle's path)
        if constexpr (P::PLANT != 4 && P::NX == 12 && P::NU == 4) {
            if (mq_fused) { hipLaunchKernelGGL((k_sweep_maps_cf<P, T>), dim3((cfg.batch + 3) / 4), dim3(64), 0, s, b, dm, (int)cfg.batch); return true; }
        }
        return false;
    }
    void launch_fp(hipStream_t s, int init_rollout, int store_candidates = 0, int part = -1) {
        const unsigned B = cfg.batch;
        bool lane_groups = false;
        if constexpr (P::PLANT == 4) lane_groups = !fp_coop && !(init_rollout && (cfg.use_limits || cfg.use_smooth_abs));       // PDDP_FP=coop: the wave-cooperative forward pass / setup kernels (comparison tests); the initial rollout of a thread-lane handle with USE_LIMITS_FLAG too
        if (!lane_groups) {
            bool serial = false, records_ran = false;
            if constexpr (P::PLANT != 4) {      // (the arm has its own families: no thread-serial instantiation of its cooperative bodies)
                if (phase_fused_sweep && part == 0) { launch_sweep_maps_cf(s); return; }      // PDDP_PHASE_SWEEP_FUSED: the candidates' segment start states land in their records (xw)
                const bool records = cf_fp && cf_fp_staged && !init_rollout && part != 2 && !store_candidates;      // (the phase hook wants the reference's candidate-major arrays: k_fp_ts, then k_cand_to_xw)
                if (records) cand_stale = true;
                // two launches: the linear sweep (segment start states -> the candidates' records), then the rollouts (kernels.hpp fp_cf_body); part 0 / 1 = only the one / the other
                if constexpr (P::kScalarPlugin && (64 / 16) * P::NX <= 64) {
                    if (records && cfg.A == 16) {
                        if (part != 1 && cfg.M > 1) { if (!launch_sweep_maps_cf(s)) hipLaunchKernelGGL((k_sweep_cf<P, INTEG, T, 16>), dim3((B + 3) / 4), dim3(64), 0, s, b, dm, cw, dt, (int)B); }
                        if (part != 0) hipLaunchKernelGGL((k_fp_cf<P, INTEG, T, 16>), dim3((B + 3) / 4), dim3(64), 0, s, b, dm, cw, dt, (int)B);
                        return;
                    }
                }
                if constexpr (P::kScalarPlugin && (64 / 8) * P::NX <= 64) {
                    if (records && cfg.A == 8) {
                        if (part != 1 && cfg.M > 1) { if (!launch_sweep_maps_cf(s)) hipLaunchKernelGGL((k_sweep_cf<P, INTEG, T, 8>), dim3((B + 7) / 8), dim3(64), 0, s, b, dm, cw, dt, (int)B); }
                        if (part != 0) hipLaunchKernelGGL((k_fp_cf<P, INTEG, T, 8>), dim3((B + 7) / 8), dim3(64), 0, s, b, dm, cw, dt, (int)B);
                        return;
                    }
                }
                if (part == 0) return;                                  // (the paths below sweep inside their rollout kernel)
                if (!serial && cf_fp && !init_rollout) { hipLaunchKernelGGL((k_fp_ts<P, INTEG, T>), dim3((B * cfg.A + 63) / 64), dim3(64), 0, s, b, dm, cw, dt, (int)B, part == 2 ? 1 : 0); serial = true; }
            }
            if (part == 0) return;
            if (!serial) hipLaunchKernelGGL((k_fp<P, INTEG, T>), dim3(init_rollout ? 1 : cfg.A, B), dim3(64 * cfg.M), fp_lds, s, b, dm, cw, dt, init_rollout ? 1 : (part == 2 ? 2 : 0));
            if constexpr (P::PLANT != 4) { if (b.xw && !records_ran && !init_rollout) hipLaunchKernelGGL((k_cand_to_xw<P, T>), dim3((unsigned)(((size_t)B * cfg.N * cfg.A + 255) / 256)), dim3(256), 0, s, b, dm, (int)B); }
            return;
        }
        if constexpr (P::PLANT == 4) {
            const int A_all = init_rollout ? 1 : cfg.A;
            const unsigned chunks = (A_all > 8 && A_all % 8 == 0) ? A_all / 8 : 1;      // one workgroup per 8 candidates when they tile exactly (see k_fp_lg)
            const int A_eff = A_all / chunks;
            const unsigned waves = (A_eff * cfg.M + kLgPerWave - 1) / kLgPerWave;
            if (!init_rollout && cfg.M > 1 && part != 1 && part != 2) {
                bool st = false;
                const bool in_rollouts = maps_in_rollouts() && part == -1 && !store_candidates;      // (k_fp_tl4 begins with the sweep)
                if (in_rollouts) st = true;
                else if (sweep_fused && (!store_candidates || phase_fused_sweep)) { launch_sweep_maps<T>(s, b, dm, (int)B); st = true; }
                if constexpr (sizeof(T) == 4) {
                    if (st) {}
                    else if (sweep_kind == 2) { launch_sweep_wg(s, b, dm, (int)B); st = true; }
                    else if (sweep_kind == 1) { launch_sweep_st(s, b, dm, (int)B); st = true; }
                }
                if (!st) hipLaunchKernelGGL((k_sweep_lg<T>), dim3((cfg.A + kLgPerWave - 1) / kLgPerWave, B), dim3(64), 0, s, b, dm, dt);
            }
            if (part == 0) return;
            if (!init_rollout && fp_split) {
                bool two = false;
                if constexpr (sizeof(T) == 4) { if (fp_two_wave) { launch_fp_tl2(s, tl_variant, b, dm, cw, dt, tl_grav, (int)B); two = true; } }
                if (!two) launch_fp_tl4<T>(s, tl_variant, b, dm, cw, dt, tl_grav, (int)B, sp, (ls_in_rollouts() && !store_candidates && part != 2) ? bench_mode : -1,
                                           maps_in_rollouts() && part == -1 && !store_candidates);
                return;
            }
            if (!init_rollout && fp_path == kFpTl) {               // one thread per (candidate, segment) rollout
                launch_fp_tl<T>(s, tl_variant, b, dm, cw, dt, tl_grav, (int)B, store_candidates);
                return;
            }
            const size_t lds = (size_t)A_eff * (cfg.N + cfg.M) * sizeof(T);
            const dim3 grid(B, chunks);
            if (cfg.ee_cost) {
                if (waves <= 4) hipLaunchKernelGGL((k_fp_lg<T, 256, true>), grid, dim3(64 * waves), lds, s, b, dm, cw, dt, init_rollout);
                else if (waves <= 8) hipLaunchKernelGGL((k_fp_lg<T, 512, true>), grid, dim3(64 * waves), lds, s, b, dm, cw, dt, init_rollout);
                else hipLaunchKernelGGL((k_fp_lg<T, 1024, true>), grid, dim3(64 * waves), lds, s, b, dm, cw, dt, init_rollout);
            } else if (waves <= 4) hipLaunchKernelGGL((k_fp_lg<T, 256>), grid, dim3(64 * waves), lds, s, b, dm, cw, dt, init_rollout);
            else if (waves <= 8) hipLaunchKernelGGL((k_fp_lg<T, 512>), grid, dim3(64 * waves), lds, s, b, dm, cw, dt, init_rollout);
            else hipLaunchKernelGGL((k_fp_lg<T, 1024>), grid, dim3(64 * waves), lds, s, b, dm, cw, dt, init_rollout);
        }
    }
    // part: -1 everything; 0 nothing (slot of a former separate winner kernel in the per-kernel timing); 1 only the setup kernel
    // The running knots' cost Hessian the matrix-core backward pass does not read (diag_h) is the one the LAST setup kernel wrote -- the reference's d_H holds exactly
    // that (costGradientHessianKern runs inside the setup, nisInitHelpers.cuh:46-93).  The weights are therefore remembered at every setup launch: a pddp_set_cost /
    // pddp_set_cost_ee between a setup and the next backward pass (phase hook, re-captured graph) must not mix new weights into a Hessian whose gradient and compact
    // position block still carry the old ones.
    T hw[3] = {T(0), T(0), T(0)};
    void note_setup_weights() { const bool ee = cfg.ee_cost != 0; hw[0] = ee ? cw.Q_xEE : cw.Q1; hw[1] = ee ? cw.Q_xdEE : cw.Q2; hw[2] = ee ? cw.R_EE : cw.R; }
    void launch_nis(hipStream_t s, int mode, int part = -1) {
        const unsigned B = cfg.batch;
        if (part != 0) note_setup_weights();
        if constexpr (P::PLANT == 4) {
            if (fp_path == kFpTl) {
                if (part != 0) launch_nis_tl<T>(s, tl_variant, b, dm, cw, dt, tl_grav, mode, (int)B);   // mode 0: adopts the accepted candidate first (arm_tl_adopt_knot)
                return;
            }
            if (!fp_coop && !cfg.use_finite_diff) {
                if (part == 0) return;
                if (fp_split && cfg.batch <= kNisTl7MaxBatch) { launch_nis_tl7<T>(s, tl_variant, b, dm, cw, dt, tl_grav, mode, (int)B); return; }
                if (cfg.ee_cost) hipLaunchKernelGGL((k_nis_lg<T, true>), dim3((cfg.N + 31) / 32, B), dim3(256), 0, s, b, dm, cw, dt, mode);
                else hipLaunchKernelGGL((k_nis_lg<T>), dim3((cfg.N + 31) / 32, B), dim3(256), 0, s, b, dm, cw, dt, mode);
                return;
            }
        }
        if (part == 0) return;
        if constexpr (P::PLANT != 4) { if (cf_nis) { hipLaunchKernelGGL((k_nis_ts<P, INTEG, T>), dim3((B * cfg.N + 63) / 64), dim3(64), 0, s, b, dm, cw, dt, mode, (int)B); return; } }
        if constexpr (P::PLANT != 4 && P::NX + P::NU <= 16 && INTEG == 3 && P::kScalarPlugin) {
            if (gl_nis && kb_nis) {
                const int units = (int)(B * cfg.N);
                if (kb_nis == 16) hipLaunchKernelGGL((k_nis_kb<P, T, 16>), dim3((units + 15) / 16), dim3(64), 0, s, b, dm, cw, dt, mode, (int)B);
                else if (kb_nis == 20) hipLaunchKernelGGL((k_nis_kb<P, T, 20>), dim3((units + 19) / 20), dim3(64), 0, s, b, dm, cw, dt, mode, (int)B);
                else if (kb_nis == 64 && sizeof(T) == 4) hipLaunchKernelGGL((k_nis_kb<P, T, sizeof(T) == 4 ? 64 : 16>), dim3((units + 63) / 64), dim3(64), 0, s, b, dm, cw, dt, mode, (int)B);
                else hipLaunchKernelGGL((k_nis_kb<P, T, 32>), dim3((units + 31) / 32), dim3(64), 0, s, b, dm, cw, dt, mode, (int)B);
                return;
            }
        }
        if constexpr (P::PLANT != 4 && P::NX + P::NU <= 16) { if (gl_nis) { if (gl_nis8) hipLaunchKernelGGL((k_nis_gl<P, INTEG, T, 8>), dim3((B * cfg.N + 7) / 8), dim3(64), 0, s, b, dm, cw, dt, mode, (int)B); else hipLaunchKernelGGL((k_nis_gl<P, INTEG, T, 16>), dim3((B * cfg.N + 3) / 4), dim3(64), 0, s, b, dm, cw, dt, mode, (int)B); return; } }
        hipLaunchKernelGGL((k_nis<P, INTEG, T>), dim3(cfg.N, B), dim3(64), 0, s, b, dm, cw, dt, mode);
    }
    void launch_sweep(hipStream_t s, int only = -1, int store_candidates = 0, int part = -1) {
        const unsigned B = cfg.batch;
        if (only < 0 || only == PDDP_PHASE_BP) {
            bool lane_groups = false;
            if constexpr (P::PLANT == 4) lane_groups = bp_lane_groups || bp_mfma;
            if constexpr (P::PLANT == 4) {
                if (bp_mfma) {
                    // the running knots' cost Hessian is known without reading it: the joint-space cost's diagonal, or (end-effector cost on the compact path) the diagonal of
                    // the nominal-state / control weights + the compact position block b.Hc
                    const bool ee = cfg.ee_cost != 0;
                    const bool diag_h = !h_overridden && (!ee || b.Hc != nullptr);
                    launch_bp_mfma<T>(s, b, dm, (int)B, diag_h, hw[0], hw[1], hw[2], dt, keep_all_ctg() || store_candidates,
                                      sweep_fused && (!store_candidates || phase_fused_sweep));
                }
            }
            if constexpr (P::PLANT == 4) { if (lane_groups && !bp_mfma) hipLaunchKernelGGL((k_bp_lg<T>), dim3((B * cfg.M + kLgPerWave - 1) / kLgPerWave), dim3(64), 0, s, b, dm, (int)B); }
            if (!lane_groups) {
                bool serial = false;
                if constexpr (P::PLANT != 4) { if (cf_bp) { hipLaunchKernelGGL((k_bp_ts<P, T>), dim3((B * cfg.M + 63) / 64), dim3(64), 0, s, b, dm, (int)B); serial = true; } }
                if constexpr (P::PLANT != 4 && P::NX == 12 && P::NU == 4) {
                    if (!serial && gl_bp && cl_bp && mq_bp) {
                        const bool fu = mq_fused && (!store_candidates || phase_fused_sweep);      // (the phase hook's backward pass writes A - B K / B du: its sweep is teacher-forced from them)
                        const bool dh = !P::kPluginCost && !h_overridden;   // the plant's own diagonal cost Hessian: not read
                        if (dh && fu) hipLaunchKernelGGL((k_bp_mq<P, T, true, true>), dim3(B * cfg.M), dim3(64), 0, s, b, dm, cw, (int)B);
                        else if (dh) hipLaunchKernelGGL((k_bp_mq<P, T, true>), dim3(B * cfg.M), dim3(64), 0, s, b, dm, cw, (int)B);
                        else if (fu) hipLaunchKernelGGL((k_bp_mq<P, T, false, true>), dim3(B * cfg.M), dim3(64), 0, s, b, dm, cw, (int)B);
                        else hipLaunchKernelGGL((k_bp_mq<P, T, false>), dim3(B * cfg.M), dim3(64), 0, s, b, dm, cw, (int)B);
                        serial = true;
                    }
                }
                if constexpr (P::PLANT != 4 && P::NX == 12 && P::NU == 4) {
                    if (!serial && gl_bp && cl_bp) { hipLaunchKernelGGL((k_bp_cl<P, T>), dim3((B * cfg.M + 3) / 4), dim3(64), 0, s, b, dm, cw, (int)B, (!P::kPluginCost && !h_overridden) ? 1 : 0); serial = true; }
                }
                if constexpr (P::PLANT != 4 && P::NX + P::NU <= 16) { if (!serial && gl_bp) { if (gl_bp32) hipLaunchKernelGGL((k_bp_gl<P, T, 32>), dim3((B * cfg.M + 1) / 2), dim3(64), 0, s, b, dm, (int)B); else hipLaunchKernelGGL((k_bp_gl<P, T, 16>), dim3((B * cfg.M + 3) / 4), dim3(64), 0, s, b, dm, (int)B); serial = true; } }
                if (serial) {}
                else if (bp_wide) hipLaunchKernelGGL((k_bp_wide<P, T>), dim3(cfg.M, B), dim3(256), 0, s, b, dm);
                else hipLaunchKernelGGL((k_bp<P, T>), dim3(cfg.M, B), dim3(64), 0, s, b, dm);
            }
        }
        if (only < 0 || only == PDDP_PHASE_FP) launch_fp(s, 0, store_candidates, part);
        if ((only < 0 || only == PDDP_PHASE_LS) && !(ls_in_rollouts() && !store_candidates)) {      // (k_fp_tl4 ended with the line search of its problems)
            if (ls_many) hipLaunchKernelGGL((k_ls_many<T>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, b, dm, sp, bench_mode, (int)B);
            else hipLaunchKernelGGL((k_ls<T>), dim3(B), dim3(64), 0, s, b, dm, sp, bench_mode);
        }
        if (only < 0 || only == PDDP_PHASE_NIS) launch_nis(s, 0, part);
    }
    // The kernels of one sweep in launch order, by name, and their average duration over `sweeps` sweeps (an event after every launch, one pass).
    // Slots: 0 backward pass, 1 linear sweep, 2 rollouts, 3 line search, 4 winner re-roll, 5 next-iteration setup; a path without a separate kernel
    // for a slot leaves its name empty and its time 0.
    int time_kernels(int sweeps, float* ms, char* names, int name_stride) override {
        const bool arm = (P::PLANT == 4), tl = arm && fp_path == kFpTl, lg = arm && !fp_coop;
        const char* nm[6] = {bp_mfma ? "k_bp_mfma" : (arm && bp_lane_groups) ? "k_bp_lg" : cf_bp ? "k_bp_ts" : (gl_bp && cl_bp && mq_bp) ? "k_bp_mq" : (gl_bp && cl_bp) ? "k_bp_cl" : gl_bp ? "k_bp_gl" : bp_wide ? "k_bp_wide" : "k_bp",
                             (lg && cfg.M > 1 && !maps_in_rollouts()) ? (sweep_fused ? "k_sweep_maps" : sweep_kind == 2 ? "k_sweep_wg" : sweep_kind == 1 ? "k_sweep_st" : "k_sweep_lg") : (cf_records() && cfg.M > 1) ? (mq_fused ? "k_sweep_maps" : "k_sweep_cf") : "", tl ? "k_fp_tl" : (lg && fp_split) ? (fp_two_wave ? "k_fp_tl2" : "k_fp_tl4") : lg ? "k_fp_lg" : (cf_fp && cf_fp_staged) ? "k_fp_cf" : cf_fp ? "k_fp_ts" : "k_fp", ls_in_rollouts() ? "" : ls_many ? "k_ls_many" : "k_ls", "", tl ? "k_nis_tl" : (lg && fp_split && cfg.batch <= kNisTl7MaxBatch) ? "k_nis_tl7" : lg ? "k_nis_lg" : cf_nis ? "k_nis_ts" : (gl_nis && kb_nis) ? "k_nis_kb" : gl_nis ? "k_nis_gl" : "k_nis"};
        static const int phase_of[6] = {PDDP_PHASE_BP, PDDP_PHASE_FP, PDDP_PHASE_FP, PDDP_PHASE_LS, PDDP_PHASE_NIS, PDDP_PHASE_NIS};
        const int part_of[6] = {-1, 0, maps_in_rollouts() ? -1 : 1, -1, 0, 1};      // (a rollout kernel that begins with the sweep is timed as it runs in production)
        HIPCHK(hipStreamSynchronize(stream));
        const size_t need = 7 * (size_t)sweeps;
        while (trace_ev.size() < need) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); trace_ev.push_back(e); }
        for (int i = 0; i < sweeps; i++) {
            HIPCHK(hipEventRecord(trace_ev[7 * i], stream));
            for (int k = 0; k < 6; k++) { if (nm[k][0]) launch_sweep(stream, phase_of[k], 0, part_of[k]); HIPCHK(hipEventRecord(trace_ev[7 * i + k + 1], stream)); }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        for (int k = 0; k < 6; k++) {
            double sum = 0;
            for (int i = 0; i < sweeps; i++) { float t = 0; HIPCHK(hipEventElapsedTime(&t, trace_ev[7 * i + k], trace_ev[7 * i + k + 1])); sum += t; }
            ms[k] = nm[k][0] ? (float)(sum / sweeps) : 0.f;
            if (names) { std::strncpy(names + (size_t)k * name_stride, nm[k], name_stride - 1); names[(size_t)k * name_stride + name_stride - 1] = 0; }
        }
        return 0;
    }
    // One sweep is one graph; a second executable holds kGraphUnroll sweeps back to back: between the kernels INSIDE a graph there is no gap, between two graph
    // launches ~9 us (measured, rocprofv3 kernel trace) -- 5 % of a single problem's 175 us iteration, nothing at large batch.
    static constexpr int kGraphUnroll = 4;
    hipGraphExec_t graph_n = nullptr;
    int capture_sweeps(int count, hipGraphExec_t* out) {
        hipGraph_t gr;
        HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < count; i++) launch_sweep(stream);
        {   // a failed launch inside the capture must not leave the stream capturing
            const hipError_t le = hipGetLastError(), ce = hipStreamEndCapture(stream, &gr);
            if (le != hipSuccess || ce != hipSuccess) {
                if (ce == hipSuccess && gr) hipGraphDestroy(gr);
                return fail(PDDP_ENODEVICE, std::string("sweep capture failed: ") + hipGetErrorString(le != hipSuccess ? le : ce));
            }
        }
        HIPCHK(hipGraphInstantiate(out, gr, nullptr, nullptr, 0));
        HIPCHK(hipGraphDestroy(gr));
        return 0;
    }
    int iterate(int sweeps) override {
        if (bp_mfma && !keep_all_ctg()) lean_ctg_ran = true;
        if ((sweep_fused || mq_fused) && sweeps > 0) fs_vars_stale = true;
        if (cf_fp && cf_fp_staged && sweeps > 0) cand_stale = true;      // (here, not only in launch_fp: a hipGraph REPLAY runs the rollouts without passing through the launch function)
        if (cfg.use_graph) {
            if (!graph || graph_mode != bench_mode + 2 * sp.max_iter) {
                if (graph) { hipGraphExecDestroy(graph); graph = nullptr; }
                if (graph_n) { hipGraphExecDestroy(graph_n); graph_n = nullptr; }
                int rc = capture_sweeps(1, &graph);
                if (rc) return rc;
                graph_mode = bench_mode + 2 * sp.max_iter;
            }
            int left = sweeps;
            if (left >= kGraphUnroll && (size_t)cfg.batch * cfg.N <= 65536) {          // only where a launch gap is a visible share of a sweep
                if (!graph_n) { int rc = capture_sweeps(kGraphUnroll, &graph_n); if (rc) return rc; }
                for (; left >= kGraphUnroll; left -= kGraphUnroll) HIPCHK(hipGraphLaunch(graph_n, stream));
            }
            for (int i = 0; i < left; i++) HIPCHK(hipGraphLaunch(graph, stream));
        } else {
            for (int i = 0; i < sweeps; i++) launch_sweep(stream);
        }
        HIPCHK(hipGetLastError());
        return 0;
    }
    // `sweeps` sweeps, kernel by kernel, an event after every launch; phase_ms[ph*stride + first_sweep + i] = duration of phase ph of sweep i, FIVE rows:
    // 0 backward pass, 1 forward pass (linear sweep + rollouts), 2 line search, 3 next-iteration setup, 4 the linear forward sweep's own kernel alone (a part of row 1:
    // the reference's sweepTime[], DDPWrappers.cuh:77; 0 on paths whose rollout kernel sweeps itself)
    std::vector<hipEvent_t> trace_ev;
    int iterate_traced(int sweeps, double* phase_ms, int first_sweep, int stride) override {
        const size_t need = 6 * (size_t)sweeps;
        if ((sweep_fused || mq_fused) && sweeps > 0) fs_vars_stale = true;
        if (cf_fp && cf_fp_staged && sweeps > 0) cand_stale = true;
        while (trace_ev.size() < need) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); trace_ev.push_back(e); }
        static const int phase_of[5] = {PDDP_PHASE_BP, PDDP_PHASE_FP, PDDP_PHASE_FP, PDDP_PHASE_LS, PDDP_PHASE_NIS};
        const bool own_sweep = !maps_in_rollouts();                                  // (otherwise the rollout kernel begins with it: row 4 stays 0)
        const int part_of[5] = {-1, 0, own_sweep ? 1 : -1, -1, -1};
        for (int i = 0; i < sweeps; i++) {
            HIPCHK(hipEventRecord(trace_ev[6 * i], stream));
            for (int k = 0; k < 5; k++) { if (k != 1 || own_sweep) launch_sweep(stream, phase_of[k], 0, part_of[k]); HIPCHK(hipEventRecord(trace_ev[6 * i + k + 1], stream)); }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        for (int i = 0; i < sweeps; i++) {
            if (first_sweep + i >= stride) continue;
            float ms[5];
            for (int k = 0; k < 5; k++) HIPCHK(hipEventElapsedTime(&ms[k], trace_ev[6 * i + k], trace_ev[6 * i + k + 1]));
            if (!own_sweep) ms[1] = 0.f;
            const size_t o = (size_t)first_sweep + i;
            phase_ms[0 * (size_t)stride + o] = ms[0]; phase_ms[1 * (size_t)stride + o] = (double)ms[1] + ms[2]; phase_ms[2 * (size_t)stride + o] = ms[3];
            phase_ms[3 * (size_t)stride + o] = ms[4]; phase_ms[4 * (size_t)stride + o] = ms[1];
        }
        return 0;
    }
    int sync() override { HIPCHK(hipStreamSynchronize(stream)); return 0; }
    // runiLQR_MPC_GPU (MPCHelpers.cuh:864-1045) for the batch
    int set_cost(double Q1, double Q2, double R, double QF1, double QF2) override {
        HIPCHK(hipStreamSynchronize(stream));
        cfg.Q1 = Q1; cfg.Q2 = Q2; cfg.R = R; cfg.QF1 = QF1; cfg.QF2 = QF2;
        cw.Q1 = (T)Q1; cw.Q2 = (T)Q2; cw.R = (T)R; cw.QF1 = (T)QF1; cw.QF2 = (T)QF2;
        if (graph) { hipGraphExecDestroy(graph); graph = nullptr; graph_mode = -1; }   // the weights are kernel arguments baked into the captured sweep
        return 0;
    }
    int set_cost_ee(const double* v) override {
        if (!cfg.ee_cost) return fail(PDDP_EINVAL, "pddp_set_cost_ee: the handle was not created with ee_cost = 1");
        HIPCHK(hipStreamSynchronize(stream));
        cfg.Q_EE1 = v[0]; cfg.Q_EE2 = v[1]; cfg.QF_EE1 = v[2]; cfg.QF_EE2 = v[3]; cfg.R_EE = v[4]; cfg.Q_xEE = v[5]; cfg.QF_xEE = v[6]; cfg.Q_xdEE = v[7]; cfg.QF_xdEE = v[8];
        cw.Q_EE1 = (T)v[0]; cw.Q_EE2 = (T)v[1]; cw.QF_EE1 = (T)v[2]; cw.QF_EE2 = (T)v[3]; cw.R_EE = (T)v[4]; cw.Q_xEE = (T)v[5]; cw.QF_xEE = (T)v[6];
        cw.Q_xdEE = (T)v[7]; cw.QF_xdEE = (T)v[8];
        drop_graph();
        return 0;
    }
    int mpc_solve(const void* xActual, const void* xGoal, const int* shift, int clear_vars, int full_rollout, int ifd, int max_iter, double budget_ms,
                  int poll_every, void* x, void* u, void* KT, void* Jout, int* alphaOut, int* success, int* iters) override {
        const size_t B = cfg.batch, N = cfg.N;
        if (max_iter < 1 || max_iter > cfg.max_iter) return fail(PDDP_EINVAL, "mpc_solve: max_iter must be in [1, config.max_iter]");
        for (size_t i = 0; i < B; i++) if (shift[i] < 0 || shift[i] >= (int)N - 1) return fail(PDDP_EINVAL, "mpc_solve: shift must be in [0, N-2]");
        if (lean_ctg_ran && !clear_vars)
            return fail(PDDP_EINVAL, "mpc_solve: this handle iterated with boundary_cost_to_go_only = 1, so its interior cost-to-go slots are stale and a warm start "
                                     "(clear_vars = 0) would shift them into the block boundaries; call with clear_vars = 1 once, or create the handle without that option");
        lean_ctg_ran = false;
        if (!mpc_used) { mpc_used = true; drop_graph(); }
        const double t0 = now_ms();
        // one pinned staging area: pageable host memory would make every small transfer of the cycle a synchronous staging copy of its own
        const size_t out_stride = (size_t)cfg.max_iter + 2;
        const size_t o_state = 0, o_xb = o_state + B * sizeof(SolverState<T>), o_u = o_xb + B * 2 * N * NX * sizeof(T), o_KT = o_u + B * N * NU * sizeof(T),
                     o_J = o_KT + B * N * NX * NU * sizeof(T), o_a = o_J + B * out_stride * sizeof(T), need_bytes = o_a + B * out_stride * sizeof(int) + 16 * B;
        const size_t rec_state = (sizeof(SolverState<T>) + 15) / 16 * 16;
        const size_t rec_bytes = (rec_state + (N * NX + N * NU + N * NX * NU + out_stride) * sizeof(T) + out_stride * sizeof(int) + 15) / 16 * 16;   // <= the six separate areas' share per problem
        if (h_stage_bytes < need_bytes) {
            if (h_stage) hipHostFree(h_stage);
            h_stage = nullptr; h_stage_bytes = 0;
            HIPCHK(hipHostMalloc((void**)&h_stage, need_bytes, hipHostMallocDefault));
            h_stage_bytes = need_bytes;
        }
        {
            unsigned char* hi = h_stage;                            // inputs first (the outputs overwrite them after the solve)
            T* hx = (T*)hi; T* hg = hx + B * NX; int* hs = (int*)(hg + B * NX);
            std::memcpy(hx, xActual, B * NX * sizeof(T)); std::memcpy(hg, xGoal, B * NX * sizeof(T)); std::memcpy(hs, shift, B * sizeof(int));
            HIPCHK(hipMemcpyAsync(d_xActual, hx, 2 * B * NX * sizeof(T) + B * sizeof(int), hipMemcpyHostToDevice, stream));   // the load kernel moves goals / shifts where the sweeps read them
        }
        bool split_roll = false;
        if constexpr (P::PLANT == 4 && INTEG == 1 && sizeof(T) == 4) {                // float arm with a built-in robot model: the warm-start rollout split over two waves
            const char* fpenv = ksel_fp(cfg);
            if (tl_variant >= 0 && !(fpenv && (std::string(fpenv) == "lg" || std::string(fpenv) == "coop"))) {
                split_roll = true;
                if (tl_variant == 0) hipLaunchKernelGGL((k_mpc_load<P, INTEG, T, 0>), dim3(B), dim3(512), 0, stream, b, mb, dm, dt, d_xActual, d_shift, clear_vars, full_rollout, d_goal_in, (cfg.ee_cost && cfg.ee_cost_shift) ? 1 : 0);
                else hipLaunchKernelGGL((k_mpc_load<P, INTEG, T, 1>), dim3(B), dim3(512), 0, stream, b, mb, dm, dt, d_xActual, d_shift, clear_vars, full_rollout, d_goal_in, (cfg.ee_cost && cfg.ee_cost_shift) ? 1 : 0);
            }
        }
        if (!split_roll) hipLaunchKernelGGL((k_mpc_load<P, INTEG, T>), dim3(B), dim3(256), 0, stream, b, mb, dm, dt, d_xActual, d_shift, clear_vars, full_rollout, d_goal_in, (cfg.ee_cost && cfg.ee_cost_shift) ? 1 : 0);
        const int saved_max_iter = sp.max_iter;
        sp.max_iter = max_iter;                                  // acceptRejectTrajGPU(..., max_iter)
        const int ee = cfg.ee_cost ? 1 : 0;
        hipLaunchKernelGGL((k_init_cost<P, T>), dim3(B), dim3(64), N * sizeof(T), stream, b, dm, cw, sp, ifd, 0, ee, 1);   // keeps alphaIndex (runiLQR_MPC_GPU does not reset it)
        launch_nis(stream, 1);
        if (ee) hipLaunchKernelGGL((k_init_cost<P, T>), dim3(B), dim3(64), N * sizeof(T), stream, b, dm, cw, sp, ifd, 0, 2, 1);
        HIPCHK(hipGetLastError());
        std::vector<int> done(B);
        int rc = 0;
        const int chunk = poll_every > 0 ? poll_every : 4;
        if (budget_ms <= 0 && chunk >= max_iter) {
            // no time budget and the whole iteration limit in one chunk: every problem is done after max_iter sweeps whatever happens (the limit forces the exit), so
            // nothing has to be polled -- sweeps, fall-back kernel and ALL result transfers are enqueued back to back and the cycle synchronises once
            if ((rc = iterate(max_iter))) { sp.max_iter = saved_max_iter; return rc; }
            sp.max_iter = saved_max_iter;
            // k_mpc_store also packs every problem's results (state | x | u | K | J | step sizes) into one device run: ONE transfer back instead of six
            if (!d_mpc_out) { int arc = alloc("mpc_out", &d_mpc_out, B * rec_bytes); if (arc) return arc; }
            hipLaunchKernelGGL((k_mpc_store<P, T>), dim3(B), dim3(256), 0, stream, b, mb, dm, 1, d_mpc_out, (int)rec_bytes, (int)out_stride);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h_stage, d_mpc_out, B * rec_bytes, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            hstate.resize(B);
            for (size_t pb = 0; pb < B; pb++) {
                const unsigned char* r = h_stage + pb * rec_bytes;
                std::memcpy(&hstate[pb], r, sizeof(SolverState<T>));
                const T* rx = reinterpret_cast<const T*>(r + rec_state);
                if (x) std::memcpy((T*)x + pb * N * NX, rx, N * NX * sizeof(T));
                if (u) std::memcpy((T*)u + pb * N * NU, rx + N * NX, N * NU * sizeof(T));
                if (KT) std::memcpy((T*)KT + pb * N * NX * NU, rx + N * NX + N * NU, N * NX * NU * sizeof(T));
                if (Jout) std::memcpy((T*)Jout + pb * out_stride, rx + N * NX + N * NU + N * NX * NU, out_stride * sizeof(T));
                if (alphaOut) std::memcpy(alphaOut + pb * out_stride, rx + N * NX + N * NU + N * NX * NU + out_stride, out_stride * sizeof(int));
            }
            bool all_exited = true;
            for (size_t i = 0; i < B; i++) all_exited &= (hstate[i].done != 0);
            if (all_exited) {
                for (size_t i = 0; i < B; i++) { if (success) success[i] = hstate[i].took_step; if (iters) iters[i] = hstate[i].iter; }
                return 0;
            }
            // a sweep whose backward pass failed raises rho and repeats without advancing `iter` (backwardPassGPU's retry loop, bpHelpers.cuh:497-511): such a
            // problem is still running after max_iter sweeps -- go on in the polled loop below, like the reference would
            sp.max_iter = max_iter;
        }
        bool fresh = false;
        for (int guard = 0; guard < 100000; guard++) {
            if (budget_ms > 0 && now_ms() - t0 > budget_ms) break;   // time_budget (MPCHelpers.cuh:919,941,1001): checked between chunks of sweeps
            fresh = false;
            if ((rc = iterate(chunk))) break;
            if ((rc = status(done.data(), nullptr))) break;
            fresh = true;                                            // hstate reflects everything enqueued so far
            bool all = true;
            for (int v : done) all &= (v != 0);
            if (all) break;
        }
        sp.max_iter = saved_max_iter;
        if (rc) return rc;
        hipLaunchKernelGGL((k_mpc_store<P, T>), dim3(B), dim3(64), 0, stream, b, mb, dm, 0, (unsigned char*)nullptr, 0, 0);   // copies only: the states status() fetched above stay valid
        HIPCHK(hipGetLastError());
        if ((rc = store_impl(x, u, KT, Jout, alphaOut, nullptr, fresh))) return rc;
        for (size_t i = 0; i < B; i++) { if (success) success[i] = hstate[i].took_step; if (iters) iters[i] = hstate[i].iter; }
        return 0;
    }
    std::vector<SolverState<T>> hstate;      // the solver states as last fetched by status()
    int status(int* done, int* iters) override {
        hstate.resize(cfg.batch);
        const size_t bytes = cfg.batch * sizeof(SolverState<T>);
        if (!h_state) HIPCHK(hipHostMalloc((void**)&h_state, bytes, hipHostMallocDefault));   // pinned: the poll is one asynchronous copy + one wait
        HIPCHK(hipMemcpyAsync(h_state, b.state, bytes, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        std::memcpy(hstate.data(), h_state, bytes);
        for (int i = 0; i < cfg.batch; i++) { if (done) done[i] = hstate[i].done; if (iters) iters[i] = hstate[i].iter; }
        return 0;
    }
    int store(void* x, void* u, void* KT, void* Jout, int* alphaOut, void* dmax) override { return store_impl(x, u, KT, Jout, alphaOut, dmax, false); }
    // state_is_current: hstate was fetched after the last kernel that changes `cur` / `alphaIndex` (saves a round trip in the MPC cycle)
    int store_impl(void* x, void* u, void* KT, void* Jout, int* alphaOut, void* dmax, bool state_is_current) {
        const size_t B = cfg.batch, N = cfg.N;
        if (!state_is_current) {
            hstate.resize(B);
            HIPCHK(hipMemcpyAsync(hstate.data(), b.state, B * sizeof(SolverState<T>), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        const std::vector<SolverState<T>>& st = hstate;
        for (size_t pb = 0; pb < B; pb++) {
            if (x) HIPCHK(hipMemcpyAsync((T*)x + pb * N * NX, b.xb + (pb * 2 + st[pb].cur) * N * NX, N * NX * sizeof(T), hipMemcpyDeviceToHost, stream));
            if (dmax) HIPCHK(hipMemcpyAsync((T*)dmax + pb, b.dmax + pb * cfg.A + st[pb].alphaIndex, sizeof(T), hipMemcpyDeviceToHost, stream));
        }
        if (u) HIPCHK(hipMemcpyAsync(u, b.ucur, B * N * NU * sizeof(T), hipMemcpyDeviceToHost, stream));
        if (KT) HIPCHK(hipMemcpyAsync(KT, b.KT, B * N * NX * NU * sizeof(T), hipMemcpyDeviceToHost, stream));
        if (Jout) HIPCHK(hipMemcpyAsync(Jout, b.Jout, B * (cfg.max_iter + 2) * sizeof(T), hipMemcpyDeviceToHost, stream));
        if (alphaOut) HIPCHK(hipMemcpyAsync(alphaOut, b.alphaOut, B * (cfg.max_iter + 2) * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }
    int time_sweeps(int sweeps, float* ms_total, float* ms_phase) override {
        HIPCHK(hipStreamSynchronize(stream));
        if (!ms_phase) {                       // total only: the sweeps exactly as pddp_iterate enqueues them (graph replay if configured)
            hipEvent_t e0, e1;
            HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
            HIPCHK(hipEventRecord(e0, stream));
            int rc = iterate(sweeps);
            if (rc) return rc;
            HIPCHK(hipEventRecord(e1, stream));
            HIPCHK(hipEventSynchronize(e1));
            if (ms_total) HIPCHK(hipEventElapsedTime(ms_total, e0, e1));
            hipEventDestroy(e0); hipEventDestroy(e1);
            return 0;
        }
        // per phase: ONE pass, kernel by kernel, an event after every launch (the sweeps are not replayed a second time:
        // the state machine moves on, and later sweeps do different amounts of work)
        std::vector<double> ph(5 * (size_t)sweeps, 0.0);
        int rc = iterate_traced(sweeps, ph.data(), 0, sweeps);
        if (rc) return rc;
        double tot = 0;
        for (int k = 0; k < 4; k++) { double sum = 0; for (int i = 0; i < sweeps; i++) sum += ph[(size_t)k * sweeps + i]; ms_phase[k] = (float)sum; tot += sum; }
        if (ms_total) *ms_total = (float)tot;
        return 0;
    }
    int array(const char* name, void** ptr, size_t* bytes) override {
        auto it = arrays.find(name);
        if (it == arrays.end()) return fail(PDDP_EINVAL, std::string("unknown array ") + name);
        *ptr = it->second.first; *bytes = it->second.second;
        return 0;
    }
    int get_state(pddp_state* out) override {
        std::vector<SolverState<T>> st(cfg.batch);
        HIPCHK(hipStreamSynchronize(stream));        // the solver stream is non-blocking: order the copy after every enqueued sweep
        HIPCHK(hipMemcpy(st.data(), b.state, cfg.batch * sizeof(SolverState<T>), hipMemcpyDeviceToHost));
        for (int i = 0; i < cfg.batch; i++) {
            const auto& s = st[i]; pddp_state& o = out[i];
            o.rho = s.rho; o.drho = s.drho; o.prevJ = s.prevJ; o.dJ = s.dJ; o.z = s.z; o.iter = s.iter; o.alphaIndex = s.alphaIndex;
            o.ignore_defect = s.ignore_defect; o.accepted = s.accepted; o.done = s.done; o.cur = s.cur; o.cur2 = s.cur2; o.bp_retries = s.bp_retries; o.pw = s.pw;
        }
        return 0;
    }
    int set_state(const pddp_state* in) override {
        std::vector<SolverState<T>> st(cfg.batch);
        for (int i = 0; i < cfg.batch; i++) {
            auto& s = st[i]; const pddp_state& o = in[i];
            s.rho = (T)o.rho; s.drho = (T)o.drho; s.prevJ = (T)o.prevJ; s.dJ = (T)o.dJ; s.z = (T)o.z; s.iter = o.iter; s.alphaIndex = o.alphaIndex;
            s.ignore_defect = o.ignore_defect; s.accepted = o.accepted; s.done = o.done; s.cur = o.cur; s.cur2 = o.cur2; s.bp_retries = o.bp_retries; s.took_step = 0; s.pw = o.pw; s.win_pending = (o.accepted == 1) ? 1 : 0;
        }
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipMemcpy(b.state, st.data(), cfg.batch * sizeof(SolverState<T>), hipMemcpyHostToDevice));
        return 0;
    }
    int run_phase(int phase) override {
        const unsigned B = cfg.batch;
        if (std::getenv("PDDP_POISON_LDS")) {                       // debugging aid (tools/determinism_check.py): every CU's LDS holds NaNs when the phase starts -- a kernel that reads LDS it has not written shows
            static bool attr_set = false;
            if (!attr_set) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
            hipLaunchKernelGGL(k_poison_lds, dim3(4096), dim3(256), 160 * 1024, stream, 160 * 256);
        }
        if (phase >= 0 && phase <= 3) {
            if (phase == PDDP_PHASE_BP) fs_vars_stale = false;     // (the hook's backward pass writes A - B K / B du itself)
            // the hook's rollouts sweep from A - B K / B du: after production sweeps that composed maps instead of writing them, rebuild them from [A B], K, du first
            if (phase == PDDP_PHASE_FP && fs_vars_stale && cfg.M > 1) { int rc = reference_views(1); if (rc) return rc; }
            launch_sweep(stream, phase, 1);                         // teacher-forcing hook: the forward pass also stores every candidate trajectory
            if (phase == PDDP_PHASE_FP) hipLaunchKernelGGL((k_reduce_parts<T>), dim3((B + 63) / 64), dim3(64), 0, stream, b, dm, (int)B);   // J / dmax readable right after the phase
        }
        else if (phase == PDDP_PHASE_BP_FUSED || phase == PDDP_PHASE_SWEEP_FUSED) {
            // the production sweep path under teacher forcing: the matrix-core backward pass composes the segment maps (and writes every cost-to-go slot, not
            // A - B K / B du); then k_sweep_maps alone -- the candidates' segment start states land in xs
            if (!sweep_fused && !mq_fused) return fail(PDDP_EINVAL, "PDDP_PHASE_BP_FUSED / _SWEEP_FUSED: this handle's selection has no fused sweep (matrix-core backward pass -- the KUKA arm's, or the 12-state plants' with the record rollouts --, M > 1, no kernels.sweep override)");
            phase_fused_sweep = true;
            if (phase == PDDP_PHASE_BP_FUSED) launch_sweep(stream, PDDP_PHASE_BP, 1);
            else launch_fp(stream, 0, 1, 0);
            phase_fused_sweep = false;
        }
        else if (phase == PDDP_PHASE_ROLLOUT) {
            launch_fp(stream, 0, 1, 2);
            hipLaunchKernelGGL((k_reduce_parts<T>), dim3((B + 63) / 64), dim3(64), 0, stream, b, dm, (int)B);
        }
        else if (phase == PDDP_PHASE_BP_COOP) { fs_vars_stale = false; hipLaunchKernelGGL((k_bp<P, T>), dim3(cfg.M, B), dim3(64), 0, stream, b, dm); }
        else if (phase == PDDP_PHASE_INIT_NIS) launch_nis(stream, 1);
        else if (phase == PDDP_PHASE_INIT_COST) hipLaunchKernelGGL((k_init_cost<P, T>), dim3(B), dim3(64), cfg.N * sizeof(T), stream, b, dm, cw, sp, 1, 0, cfg.ee_cost ? 1 : 0, 0);
        else return fail(PDDP_EINVAL, "unknown phase");
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }
    // grow-only device scratch of the helper entry points below (owned by the handle: no hipMalloc / hipFree -- an implicit device synchronisation -- per call,
    // nothing to leak on an error return; released with the handle)
    void* scratch_buf[3] = {nullptr, nullptr, nullptr}; size_t scratch_cap[3] = {0, 0, 0};
    int scratch(int slot, size_t bytes, void** out) {
        if (scratch_cap[slot] < bytes) {
            if (scratch_buf[slot]) { HIPCHK(hipStreamSynchronize(stream)); hipFree(scratch_buf[slot]); scratch_buf[slot] = nullptr; scratch_cap[slot] = 0; }
            const size_t cap = bytes < 4096 ? 4096 : bytes;
            if (hipMalloc(&scratch_buf[slot], cap) != hipSuccess) return fail(PDDP_ENOMEM, "hipMalloc failed for a helper's scratch buffer");
            scratch_cap[slot] = cap;
        }
        *out = scratch_buf[slot];
        return 0;
    }
    // ---- lock-step experiment helpers
    using PD = typename P::template Rebind<double>;
    void* model_d = nullptr;                           // the plant's constants in double (the simulated robot runs in double)
    int simulate(const void* x, const void* u, const void* KT, double t0_us, double elapsed_us, int substeps, const void* goal, void* xActual,
                 double* avg_err, int* failed) override {
        if (substeps < 1 || !(elapsed_us >= 0)) return fail(PDDP_EINVAL, "pddp_simulate: substeps >= 1 and elapsed_us >= 0");
        const size_t N = cfg.N;
        if (!model_d) {
            typename PD::Model hm; fill_model(hm, cfg);
            HIPCHK(hipMalloc(&model_d, sizeof(hm))); allocs.push_back(model_d);
            HIPCHK(hipMemcpy(model_d, &hm, sizeof(hm), hipMemcpyHostToDevice));
        }
        const size_t nx = N * NX, nu = N * NU, nk = N * NX * NU;
        T* buf = nullptr; double* dout = nullptr;
        int rc;
        if ((rc = scratch(0, (nx + nu + nk + NX + 3) * sizeof(T), (void**)&buf)) || (rc = scratch(1, 2 * sizeof(double), (void**)&dout))) return rc;
        HIPCHK(hipMemcpyAsync(buf, x, nx * sizeof(T), hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(buf + nx, u, nu * sizeof(T), hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(buf + nx + nu, KT, nk * sizeof(T), hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(buf + nx + nu + nk, xActual, NX * sizeof(T), hipMemcpyHostToDevice, stream));
        if (goal) HIPCHK(hipMemcpyAsync(buf + nx + nu + nk + NX, goal, 3 * sizeof(T), hipMemcpyHostToDevice, stream));
        PlantSimArgs<T> a;
        a.x = buf; a.u = buf + nx; a.KT = buf + nx + nu; a.N = cfg.N; a.step_us = cfg.total_time / (cfg.N - 1) * 1000.0 * 1000.0;
        a.t0_us = t0_us; a.elapsed_us = elapsed_us; a.substeps = substeps; a.goal = goal ? buf + nx + nu + nk + NX : nullptr; a.ee_z = cfg.ee_on_link_z;
        a.xActual = buf + nx + nu + nk; a.out = dout;
        hipLaunchKernelGGL((k_plant_sim<PD, INTEG, T>), dim3(1), dim3(64), 0, stream, (const void*)model_d, a);
        HIPCHK(hipGetLastError());
        double ho[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(ho, dout, sizeof(ho), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(xActual, buf + nx + nu + nk, NX * sizeof(T), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (avg_err) *avg_err = ho[0];
        if (failed) *failed = (int)ho[1];
        return 0;
    }
    int ee_pos(int count, const void* x, void* out) override {
        if (P::PLANT != 4 || count <= 0) return fail(PDDP_EINVAL, "pddp_ee_pos: KUKA arm only, count >= 1");
        T *dx = nullptr, *dout = nullptr;
        int rc;
        if ((rc = scratch(0, (size_t)count * NX * sizeof(T), (void**)&dx)) || (rc = scratch(1, (size_t)count * 6 * sizeof(T), (void**)&dout))) return rc;
        HIPCHK(hipMemcpyAsync(dx, x, (size_t)count * NX * sizeof(T), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL((k_ee_pos<P, T>), dim3(count), dim3(64), 0, stream, b.model, (T)cfg.ee_on_link_z, (const T*)dx, dout);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, dout, (size_t)count * 6 * sizeof(T), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }
    int plant_eval(int what, int count, const void* x, const void* u, void* out) override {
        if (what < 0 || what > 9 || count <= 0 || (what >= 4 && P::PLANT != 4)) return fail(PDDP_EINVAL, "plant_eval: bad arguments");
        const size_t osz = what == 9 ? 48 : (what == 0 || what == 4 || what == 6 || what == 7 ? NP : (what == 1 || what == 5 || what == 8) ? NP * NM : what == 2 ? NX : NX * NM);
        T *dx, *du_, *dout;
        int rc;
        if ((rc = scratch(0, (size_t)count * NX * sizeof(T), (void**)&dx)) || (rc = scratch(1, (size_t)count * NU * sizeof(T), (void**)&du_)) ||
            (rc = scratch(2, (size_t)count * osz * sizeof(T), (void**)&dout))) return rc;
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipMemcpy(dx, x, (size_t)count * NX * sizeof(T), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(du_, u, (size_t)count * NU * sizeof(T), hipMemcpyHostToDevice));
        int grid = count < 4096 ? count : 4096;
        if (const char* g = std::getenv("PDDP_EVAL_GRID")) grid = std::atoi(g) > 0 ? std::atoi(g) : grid;   // micro-benchmarks (tools/)
        if (what >= 7) {
            if constexpr (P::PLANT == 4) {
                if (tl_variant < 0) { return fail(PDDP_EINVAL, "plant_eval: the thread-lane kernels need one of the built-in robot models"); }
                launch_plant_eval_tl<T>(stream, tl_variant, what == 9 ? (T)cfg.ee_on_link_z : tl_grav, count, dx, du_, dout, what == 9 ? 2 : what == 8 ? 1 : 0);
            }
        }
        else if (what >= 4) { if constexpr (P::PLANT == 4) hipLaunchKernelGGL((k_plant_eval_lg<T>), dim3(grid), dim3(64), 0, stream, b.model, count, dx, du_, dout, what == 5 ? 1 : (what == 6 ? 2 : 0)); }
        else hipLaunchKernelGGL((k_plant_eval<P, INTEG, T>), dim3(grid), dim3(64), 0, stream, b.model, what, count, dx, du_, dout, dt);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipMemcpy(out, dout, (size_t)count * osz * sizeof(T), hipMemcpyDeviceToHost));
        return 0;
    }
};

template <template <typename> class PT, typename T>
static SolverBase* make_integ(int integ) {
    switch (integ) {
    case 1: return new Solver<PT<T>, 1, T>();
    case 2: return new Solver<PT<T>, 2, T>();
    case 3: return new Solver<PT<T>, 3, T>();
    }
    return nullptr;
}
template <template <typename> class PT>
static SolverBase* make_solver_of(const pddp_config& c) {
    return c.dtype == 0 ? make_integ<PT, float>(c.integrator) : c.dtype == 1 ? make_integ<PT, double>(c.integrator) : nullptr;
}
