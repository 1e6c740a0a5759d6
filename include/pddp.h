/* pddp.h -- C ABI of the MI355X-native parallel DDP / iLQR hot path (libpddp.so).
 *
 * The reference (plancherb1/parallel-DDP @ v1) has no binary interface: its hot path is a set of C++
 * templates configured by preprocessor macros (config.cuh) and called from the examples as
 *     allocateMemory_GPU<T>  (DDPHelpers/nisInitHelpers.cuh:768-861)
 *     runiLQR_GPU<T>         (DDPHelpers/DDPWrappers.cuh:10-138)
 *     freeMemory_GPU<T>      (DDPHelpers/nisInitHelpers.cuh:865-882)
 * This header is what a binding of that path would call.  Each entry point names the reference
 * interface it replaces.  Plain pointers and sizes only; element type is selected by pddp_config.dtype
 * (0 = float, the reference's algType, config.cuh:74; 1 = double, config.cuh:73).
 *
 * All functions return 0 on success or a negative PDDP_E* code; pddp_last_error() gives the text.
 * Nothing here ever calls exit() (the reference's gpuAssert does, utils/cudaUtils.cu:31-37).
 * The library fails loudly (PDDP_ENODEVICE) when no HIP device is available: there is no CPU fallback.
 */
#ifndef PDDP_H
#define PDDP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDDP_EINVAL     (-1)   /* bad argument / unsupported configuration */
#define PDDP_ENODEVICE  (-2)   /* no HIP device, or HIP runtime error      */
#define PDDP_ENOMEM     (-3)
#define PDDP_ENUMERIC   (-4)   /* regulariser hit RHO_MAX (in-band in the reference: loop exit) */

typedef struct pddp_solver* pddp_handle;

/* Which kernel family a handle uses for each phase of a sweep.  0 everywhere (what pddp_default_config writes) = the library's own choice: one function of plant,
 * element type, batch, M, A and the cost family (Solver::init in csrc/pddp_api.hip; the table is asserted by tests/test_kernel_selection.py and reported back by
 * pddp_time_kernels).  Every family computes the same functions; the other values exist so that comparison tests and measurements can pin a family WITHOUT the
 * library reading the process environment (until round 4 these were environment variables).  A value that does not apply to the handle's plant is
 * ignored exactly as the variable was. */
typedef struct pddp_kernel_selection {
    int bp;       /* arm, backward pass:        1 mx (matrix cores)  2 lg (8-lane groups)  3 coop (one wave per block)  4 wide (one workgroup per block)             */
    int fp;       /* arm, rollouts + setup:     1 tl (thread lanes)  2 lg  3 coop  4 tl2 (two-wave split, few problems)  5 tl4 (four-wave pipeline, few problems)   */
    int sweep;    /* arm, linear forward sweep: 1 alpha (lane group per candidate)  2 st (two sequences)  3 wg (workgroup per problem): no fusion into bp;  4 maps: fused, applied by k_sweep_maps even where the rollout kernel would apply the maps itself.
                     12-state plants on the matrix-core backward pass with the record rollouts (cf_bp mq + cf_fp cf, M > 1): 0 / 4 the backward pass composes the maps, 1..3 it writes A - B K | B du and k_sweep_cf sweeps knot by knot */
    int ls;       /* line search:               1 many (thread per problem)  2 wg (wave per problem)                                                                */
    int ab;       /* arm, layout of [A B]:      1 full (reference layout instead of the compact one)                                                                */
    int cf;       /* closed-form plants, every phase: 1 ts (thread-serial)  2 coop                                                                                   */
    int cf_bp;    /* ... backward pass:         1 ts  2 coop  3 gl (16-lane groups)  4 gl32  5 cl (lane = column)  6 mq (matrix cores, 12 states + 4 controls)            */
    int cf_fp;    /* ... rollouts:              1 ts  2 coop  3 cf (staged per wavefront)                                                                            */
    int cf_nis;   /* ... setup:                 1 ts  2 coop  3 gl  4 gl8  5 kb16  6 kb32  7 kb64  8 kb20 (knot-batched: knots per wavefront; 16 / 20: one lane per (RK3 stage, knot)) */
} pddp_kernel_selection;

/* The reference's compile-time configuration (config.cuh) as a run-time record. */
typedef struct pddp_config {
    int plant;            /* PLANT 1 pendulum, 2 cart-pole, 3 quadrotor, 4 KUKA iiwa14      config.cuh:21-61   */
    int dtype;            /* 0 float, 1 double                                               config.cuh:72-74   */
    int N;                /* NUM_TIME_STEPS (power of two, multiple of M)                    config.cuh:133-135 */
    int M;                /* M_BLOCKS = M_BLOCKS_B = M_BLOCKS_F                               config.cuh:90-94   */
    int A;                /* NUM_ALPHA                                                       config.cuh:113-115 */
    int integrator;       /* INTEGRATOR 1 Euler, 2 midpoint, 3 RK3                           config.cuh:78-80   */
    int batch;            /* independent problems solved concurrently (new: the reference solves one)           */
    int max_iter;         /* MAX_ITER                                                        config.cuh:83      */
    int wafr_urdf;        /* USE_WAFR_URDF (arm)                                             config.cuh:182-184 */
    int mpc_mode;         /* MPC_MODE: gravity 0 (arm)                                       config.cuh:185-187 */
    int ignore_max_rho_exit; /* IGNORE_MAX_ROX_EXIT                                          config.cuh:105-107 */
    int device;           /* HIP device ordinal                                                                 */
    int use_graph;        /* replay each DDP sweep from a hipGraph instead of four launches                     */
    double total_time;    /* TOTAL_TIME                                                      config.cuh:130-132 */
    double alpha_base;    /* ALPHA_BASE                                                      config.cuh:110-112 */
    double rho_init;      /* RHO_INIT                                                        config.cuh:99-101  */
    double max_defect;    /* MAX_DEFECT_SIZE                                                 config.cuh:124-126 */
    double tol_cost;      /* TOL_COST                                                        config.cuh:85-87   */
    double exp_red_min, exp_red_max;   /* EXP_RED_MIN / EXP_RED_MAX                          config.cuh:117-122 */
    double Q1, Q2, R, QF1, QF2;        /* _Q1 _Q2 _R _QF1 _QF2 (arm joint cost)   plants/cost_arm.cuh:97-103   */
    /* End-effector cost family of the arm (plant 4; plants/cost_arm.cuh:104-115, 206-389), in the configuration of
     * examples/WAFR_MPC_examples.cu:4-37 (USE_EE_VEL_COST 0, USE_SMOOTH_ABS 0, USE_LIMITS_FLAG 0). */
    int ee_cost;          /* EE_COST: xGoal[b][0..5] = tool-point goal (x, y, z, roll, pitch, yaw)   config.cuh:165-167 */
    int ee_cost_shift;    /* use_cost_shift of runiLQR_MPC_GPU: final EE weights from knot N-1-shift[b] on  MPCHelpers.cuh:866,876 */
    double Q_EE1, Q_EE2, QF_EE1, QF_EE2;   /* _Q_EE1 (xyz) _Q_EE2 (rpy) and the final ones                                   */
    double R_EE, Q_xEE, QF_xEE, Q_xdEE, QF_xdEE;   /* control weight; nominal-state weights on q and qd (target: array "xTarget") */
    double ee_on_link_z;  /* EE_ON_LINK_Z: tool point on the last link's z axis (0.0635 = EE_TYPE 1, flange)   dynamics_arm.cuh:48-65 */
    int ee_initial_cost_fix; /* 0 (default) = the reference: an MPC solve with the EE cost reads its initial cost from d_JT[alphaIndex]
                              * (nisInitHelpers.cuh:392), i.e. the cost of ONE knot whenever the previous solve ended on a shortened step, and
                              * then rejects every iteration; 1 = always the whole trajectory's cost (d_JT[0]).  Not a reference behaviour. */
    int use_finite_diff;  /* USE_FINITE_DIFF (config.cuh:68): [A B] of the Euler step by central differences of the plant's `dynamics`, column by column
                           * (finiteDiffInner, DDPHelpers/nisInitHelpers.cuh:138-183) instead of the analytic gradient.  Euler only, joint-space cost. */
    double finite_diff_epsilon; /* FINITE_DIFF_EPSILON (config.cuh:69-71), default 0.00001 */
    int boundary_cost_to_go_only; /* 0 (default) = the reference: every backward pass leaves the cost-to-go P, p of EVERY knot in the device arrays (d_P, d_p;
                           * the MPC warm start shifts the whole arrays, MPCHelpers.cuh:602-655).  1 = throughput option of the matrix-core backward pass: only the
                           * slot in front of every block's first knot is written -- the only ones a later pass reads (the reference's d_Pp / d_pp boundary slots);
                           * the interior cost-to-go is not an output of runiLQR_GPU.  A handle that iterated with 1 refuses a warm-started pddp_mpc_solve
                           * (clear_vars = 0) until a solve has run with every slot kept. */
    int use_smooth_abs;   /* USE_SMOOTH_ABS (config.cuh:174-176; ee_cost = 1): the tool-point term c of a knot becomes sqrt(2 c + alpha^2) - alpha, its gradient is
                           * divided by sqrt(2 c + alpha^2); the Gauss-Newton Hessian is left as it is (eeCost / deeCost, plants/cost_arm.cuh:218-220,242-251) */
    double smooth_abs_alpha; /* SMOOTH_ABS_ALPHA, default 0.2 (cost_arm.cuh:116-118) */
    int use_limits;       /* USE_LIMITS_FLAG (config.cuh:171-173), KUKA arm with the joint-space cost: quadratic penalties 100 x 0.5 (|v| - limit)^2 beyond 0.8 x the
                           * iiwa's position / velocity / torque limits are added to the cost and to its GRADIENT -- not to H (costFunc / costGrad,
                           * plants/cost_arm.cuh:13-94,136-149,176-199).  With ee_cost = 1: the same penalties in the cost, the gradient AND the diagonal of H
                           * (:289-291,341-343,374-376).  Handles with use_limits / use_smooth_abs run the thread-lane or the wave-cooperative kernels (the lane-group
                           * family does not carry the variants). */
    int ee_type;          /* EE_TYPE (dynamics_arm.cuh:50-65): 0 no end effector, 1 flange (default), 2 flange + peg.  With wafr_urdf = 0 link 7's inertia is the base
                           * values x INERTIA_MODIFIER (1 / 3 / 5) and its mass 1.2 + WEIGHT_MODIFIER (0 / 0.03 / 0.5) (:338-347); the tool offset EE_ON_LINK_Z
                           * (0 / 0.0635 / 0.1524) is ee_on_link_z above -- pddp_default_config writes EE_TYPE 1's, the source-level facade sets both from the macro. */
    pddp_kernel_selection kernels;   /* all zero: the library chooses */
} pddp_config;

/* Reference defaults for a plant (the per-plant blocks of config.cuh:24-61 and the #ifndef defaults below them). */
int pddp_default_config(pddp_config* cfg, int plant);
int pddp_state_size(int plant);     /* STATE_SIZE   */
int pddp_control_size(int plant);   /* CONTROL_SIZE */
const char* pddp_last_error(void);

/* allocateMemory_GPU (nisInitHelpers.cuh:768-861): all device buffers for `batch` problems, alpha[i] =
 * ALPHA_BASE^i (:829), robot constants uploaded (:844-853).  freeMemory_GPU (:865-882). */
int pddp_create(const pddp_config* cfg, pddp_handle* out);
int pddp_destroy(pddp_handle h);

/* loadVarsGPU (nisInitHelpers.cuh:596-652) + initAlgGPU (:355-397).
 * x0 [batch][N][n], u0 [batch][N][m], xGoal [batch][n] on the HOST.  clear_vars: zero P,p,KT,d (the reference's
 * clearVarsFlag = 1); otherwise they keep the values of the previous solve (warm start).
 * After this call every input is resident in HBM; Jout[0] / alphaOut[0] are set. */
int pddp_load(pddp_handle h, const void* x0, const void* u0, const void* xGoal, int clear_vars, int ignore_first_defect);

/* The same with every option of the reference call: warm-start arrays, used when clear_vars == 0 (KT0 [batch][N][n*m],
 * P0 [batch][N][n*n], p0 [batch][N][n], d0 [batch][N][n]; P0/p0 also seed Pp/pp, nisInitHelpers.cuh:621-628; a NULL array
 * keeps the device values of the previous solve), and forward_rollout = the reference's forwardRolloutFlag (:642-648):
 * every shooting segment is first rolled out from x0[b*N/M] with u0 (and K from KT0), and the result -- with its boundary
 * defects -- becomes the initial trajectory; alphaOut[0] is then 0 instead of -1 (:363). */
int pddp_load_ex(pddp_handle h, const void* x0, const void* u0, const void* xGoal, const void* KT0, const void* P0, const void* p0,
                 const void* d0, int forward_rollout, int clear_vars, int ignore_first_defect);

/* The hot loop of runiLQR_GPU (DDPWrappers.cuh:52-114): `sweeps` x { backward pass, forward sweep+sim+cost,
 * line search + accept/reject, next-iteration setup }, enqueued on the solver's stream with NO host
 * synchronisation.  Problems that have met an exit condition idle through the remaining sweeps. */
int pddp_iterate(pddp_handle h, int sweeps);
int pddp_sync(pddp_handle h);
/* done[b]: 0 running, 1 TOL_COST exit, 2 MAX_ITER exit, 3 RHO_MAX exit; iters[b]: reference `iter` at exit. */
int pddp_status(pddp_handle h, int* done, int* iters);

/* storeVarsGPU (nisInitHelpers.cuh:741-750): solution to the host.  Any pointer may be NULL.
 * x [batch][N][n], u [batch][N][m], KT [batch][N][n*m], Jout/alphaOut [batch][max_iter+2], dmax [batch]. */
int pddp_store(pddp_handle h, void* x, void* u, void* KT, void* Jout, int* alphaOut, void* dmax);

/* runiLQR_GPU (DDPWrappers.cuh:10-138) for the whole batch: load, init, iterate until every problem exits, store.
 * times_ms[0] = total, [1] = init (load+init+store), like *tTime / *initTime. */
int pddp_solve(pddp_handle h, void* x0_inout, void* u0_inout, const void* xGoal, void* Jout, int* alphaOut,
               int clear_vars, int ignore_first_defect, double* times_ms);

/* The full reference call.  Warm-start arrays / forward_rollout as in pddp_load_ex.  phase_ms, when not NULL, is
 * [5][max_iter+2] doubles (bp, sweep+sim, line search+accept/reject, next-iteration setup, and -- a part of row 1 -- the linear
 * forward sweep's own kernel): the duration of each kernel of sweep i measured with HIP events on the solver's stream -- what
 * the reference's bpTime[], sweepTime[]+simTime[], nisTime[] and sweepTime[] report (DDPWrappers.cuh:54-105; row 4 is 0 on
 * kernel selections whose rollout kernel sweeps itself).  With phase_ms the sweeps are launched kernel by kernel instead of as a graph; in
 * both modes the loop never synchronises with the host except to poll the exit flags every `poll_every` sweeps. */
int pddp_solve_ex(pddp_handle h, void* x0_inout, void* u0_inout, const void* xGoal, const void* KT0, const void* P0, const void* p0,
                  const void* d0, void* Jout, int* alphaOut, int forward_rollout, int clear_vars, int ignore_first_defect,
                  int poll_every, double* times_ms, double* phase_ms, int* sweeps_out);
/* runiLQR_MPC_GPU (DDPHelpers/MPCHelpers.cuh:864-1045) for the batch (joint-space or end-effector cost).  The handle holds the previous solution (seed it
 * with pddp_solve).  Per problem: loadVarsGPU_MPC (:602-655) shifts it by shift[b] knots (x, d, P, p hold their last knot; u, KT are
 * zero-filled), or clears u, KT, P, p when clear_vars, and rolls the trajectory out open loop from the measured state xActual
 * (full_rollout = FULL_ROLLOUT, :37-39: the whole horizon; otherwise the first shooting segment plus the last shift[b] knots with
 * feedback); then the iLQR loop with this call's max_iter (<= config.max_iter) and, when > 0, a time budget in ms checked every
 * poll_every sweeps; then storeVarsGPU_MPC (:755-774): success[b] = an accepted iteration used a step-size index > 0 (sic, :986-991),
 * otherwise the solution falls back to the shifted previous one.  x [batch][N][n], u, KT, Jout, alphaOut as pddp_store. */
int pddp_mpc_solve(pddp_handle h, const void* xActual, const void* xGoal, const int* shift, int clear_vars, int full_rollout,
                   int ignore_first_defect, int max_iter, double time_budget_ms, int poll_every, void* x, void* u, void* KT, void* Jout,
                   int* alphaOut, int* success, int* iters);
/* End-effector cost weights for the following loads / solves (costParams of runiLQR_MPC_GPU, MPCHelpers.cuh:118-135); requires a
 * handle created with ee_cost = 1. */
int pddp_set_cost_ee(pddp_handle h, double Q_EE1, double Q_EE2, double QF_EE1, double QF_EE2, double R_EE, double Q_xEE, double QF_xEE,
                     double Q_xdEE, double QF_xdEE);
/* use_cost_shift of runiLQR_MPC_GPU (MPCHelpers.cuh:866,876) for the following pddp_mpc_solve calls: same as config.ee_cost_shift. */
int pddp_set_ee_cost_shift(pddp_handle h, int on);
/* New joint-space cost weights for the following loads / solves (the reference passes Q1, Q2, R, QF1, QF2 on every call,
 * DDPWrappers.cuh:17-21, MPCHelpers.cuh:862-866 through costParams).  Takes effect at the next pddp_load / pddp_solve / pddp_mpc_solve:
 * the cost gradient and Hessian of the current trajectory are rebuilt there.  Arm plant only (the other plants' weights are
 * constants of plants/cost_{pend,cart,quad}.cuh). */
int pddp_set_cost(pddp_handle h, double Q1, double Q2, double R, double QF1, double QF2);
/* ---- the lock-step experiment around the solver ------------------------- */
/* simulateForward<T, SUBSTEPS> (examples/WAFR_MPC_examples.cu:111-139): the simulated robot of the lock-step MPC experiment.  Starting
 * from xActual at plant time t0_us (the plan's t0), integrates the plant IN DOUBLE for elapsed_us in `substeps` steps under the trajectory
 * runner's control law getHardwareControls (MPCHelpers.cuh:819-858: zero-order hold on u and K, first-order hold on the nominal state,
 * evaluated in the plan's precision).  x [N][n], u [N][m], KT [N][n*m] = the plan (host, the handle's dtype); xActual_inout [n].
 * goal_xyz (3 values, arm only, may be NULL): *avg_err = (sum of |tool point - goal| over the substeps' start states and the final
 * state) / substeps, sic (:137-138).  *failed = 1 when the time leaves the plan (k >= N-2): state untouched, error 0 (:129-130). */
int pddp_simulate(pddp_handle h, const void* x, const void* u, const void* KT, double t0_us, double elapsed_us, int substeps,
                  const void* goal_xyz, void* xActual_inout, double* avg_err, int* failed);
/* compute_eePos_scratch (plants/dynamics_arm.cuh:1953-1960): tool point (x, y, z, roll, pitch, yaw) of `count` states [count][n] -> [count][6]. */
int pddp_ee_pos(pddp_handle h, int count, const void* x, void* eePos);
/* The HIP stream every kernel of this handle is enqueued on (a hipStream_t). */
int pddp_stream(pddp_handle h, void** hip_stream);

/* ---- measurement ---------------------------------------------------------------------------------- */
/* Runs `sweeps` sweeps timed with HIP events on the solver's stream.  ms_phase == NULL: one event pair around the sweeps as
 * pddp_iterate enqueues them, ms_total = elapsed.  Otherwise the sweeps are launched kernel by kernel with an event after
 * every launch: ms_phase[4] = summed durations of the four phases (backward pass, forward pass, line search, next-
 * iteration setup), ms_total = their sum. */
int pddp_time_sweeps(pddp_handle h, int sweeps, float* ms_total, float* ms_phase);
/* Per KERNEL: `sweeps` sweeps launched kernel by kernel on the solver's stream with a HIP event after every launch; ms6[k] = average duration of
 * slot k (0 backward pass, 1 linear sweep, 2 rollouts + cost + defect, 3 line search, 4 winner re-roll, 5 next-iteration setup); names (optional,
 * 6 x name_stride chars) receives the kernels' names for the handle's kernel selection -- empty where the selection has no separate kernel.
 * Measurement aid of bench.py (the reference prints per-phase times, DDPWrappers.cuh:54-105). */
int pddp_time_kernels(pddp_handle h, int sweeps, float* ms6, char* names, int name_stride);
/* Freeze / unfreeze the exit tests so a benchmark can time a fixed number of full-work sweeps. */
int pddp_set_benchmark_mode(pddp_handle h, int on);

/* Profiling aid: `reps` launches of k_hbm_calib_dword copying `bytes` (use > 256 MiB, the Infinity Cache size) with the
 * sweep kernels' access width, so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated on a known byte count. */
int pddp_hbm_calibration(int device, size_t bytes, int reps);

/* ---- teacher-forced phase hooks (tests) ------------------------------------------------------------- */
/* Named device arrays: xs us ds xb ucur dcur P p Pp pp AB H g KT du ApBK Bdu J dmax dJexp alpha xGoal xTarget costk Jout
 * (element type = dtype), err alphaOut tshift (int), state (see pddp_state below). */
int pddp_array_bytes(pddp_handle h, const char* name, size_t* bytes);
int pddp_set_array(pddp_handle h, const char* name, const void* host, size_t bytes);
int pddp_get_array(pddp_handle h, const char* name, void* host, size_t bytes);
/* Device address of a named array (the reference hands its callers the raw device buffers, nisInitHelpers.cuh:768-772);
 * lets the host layer run an RCCL collective on the cost table without staging it through the host. */
int pddp_array_ptr(pddp_handle h, const char* name, void** device_ptr, size_t* bytes);
/* Arrays the reference leaves behind as by-products of copies this design does not make, rebuilt on demand from the state the last sweep left:
 *   "ApBK", "Bdu"    A - B K and B du of every knot (computeFSVars, bpHelpers.cuh:281-312, M > 1): production sweeps compose the forward sweep's segment maps inside
 *                    the backward pass and never write them; pddp_get_array of either name rebuilds both first;
 *   "xs" "us" "ds"   the accepted trajectory in EVERY step size's slot (memcpyCurrAKern x 3, nisInitHelpers.cuh:24-32,270-272): here the slots keep the candidates
 *                    of the last line search.
 * pddp_refresh_reference_views writes both into the device arrays (the ones pddp_array_ptr names): what a caller that reads the raw device buffers after
 * runiLQR_GPU -- the source-level facade hands them out as d_ApBK, d_Bdu, d_x / h_d_x ... -- needs to see the reference's contents.  One small launch + a synchronisation.
 * VALID AFTER AN EXIT (every problem done: pddp_solve, or pddp_iterate until pddp_status says so): the reference's loop breaks BEFORE nextIterationSetupGPU
 * (DDPWrappers.cuh:104-113), so its d_ApBK is A - B K of the trajectory the last backward pass linearised about and its slots hold that exit's state.  On a handle
 * with problems still running, the last sweep's setup has already moved [A B] to the newly accepted trajectory while K / du are the previous backward pass's: the rebuilt
 * "ApBK" is then A_new - B_new K_old, a mixture the reference never holds -- call it between sweeps only for the slots view ("xs" "us" "ds"). */
int pddp_refresh_reference_views(pddp_handle h);

/* ---- multi-GPU (mode R): one process per GPU, rank g owns the problems {r : r % world == g} in its own handle; a sweep needs
 * no exchange.  The two exchanges a caller needs -- "has every problem on every rank exited?" and the cost table of all rollouts -- are RCCL
 * collectives issued on the solver's own stream (ncclAllReduce(max) of one int, ncclAllGather of [batch][2] costs), straight from HBM.
 * The reference has no multi-GPU path (no NCCL / MPI anywhere in its tree); this is the batch-axis shard BASELINE configs[3] asks for.
 * Rendezvous: rank 0 calls pddp_comm_unique_id and ships the PDDP_COMM_ID_BYTES bytes to the other ranks by whatever means the caller has
 * (a file, an environment variable, MPI, a TCP store); every rank then calls pddp_comm_init with the same bytes. */
#define PDDP_COMM_ID_BYTES 128
typedef struct pddp_comm* pddp_comm_handle;
int pddp_comm_unique_id(void* id /* PDDP_COMM_ID_BYTES */);
int pddp_comm_init(pddp_comm_handle* out, int rank, int world, const void* id, int device);
int pddp_comm_destroy(pddp_comm_handle c);
int pddp_comm_ranks(pddp_comm_handle c, int* rank, int* world);
/* *all_done = 1 when every problem of every rank's handle has met an exit condition (one all-reduce(max) over the ranks, after everything
 * already enqueued on the handle's stream) */
int pddp_comm_all_done(pddp_comm_handle c, pddp_handle h, int* all_done);
/* costs[world * batch][2] (double, host) = (J_initial, J at the problem's last iteration) of every problem of every rank in GLOBAL problem order
 * (problem g lives on rank g % world at local index g / world); every rank's handle must have the same batch */
int pddp_comm_allgather_costs(pddp_comm_handle c, pddp_handle h, double* costs);
/* value = max over the ranks of value (a host double; doubles as a barrier: it returns when every rank has entered) */
int pddp_comm_allreduce_max(pddp_comm_handle c, double* value);
/* The per-iteration cost table (north_star: "an RCCL all-reduce of the per-alpha cost over xGMI"; SURVEY.md 8(e) mode R): J[batch][A] of the handle's LAST line search
 * from every rank, on every rank, as [world * batch][A] doubles in global problem order.  Optional -- the ranks' solves are independent, nothing on the data path waits
 * for it -- and off the sweep's critical path: _begin enqueues the exchange on the communicator's own stream behind an event on the solver's stream (the line search of
 * the last enqueued sweep) and returns; the solver may iterate on meanwhile; _end waits for it and fills `table`.  One exchange in flight per communicator.  The gather
 * runs on a DUPLICATE of the communicator (ncclCommSplit at the first _begin, which is therefore collective: every rank makes its first _begin at the same point of its
 * loop), so pddp_comm_all_done / _allgather_costs issued between _begin and _end do not queue behind it. */
int pddp_comm_cost_table_begin(pddp_comm_handle c, pddp_handle h);
int pddp_comm_cost_table_end(pddp_comm_handle c, double* table /* [world * batch][A] */);
/* the configuration a handle was created with */
int pddp_get_config(pddp_handle h, pddp_config* out);

typedef struct pddp_state {      /* per problem, all scalars as double regardless of dtype */
    double rho, drho, prevJ, dJ, z;
    int iter, alphaIndex, ignore_defect, accepted, done, cur, cur2, bp_retries;
    int pw;   /* the backward pass writes array "P" (0) or "Pp" (1) and reads the boundary cost-to-go from the other one */
} pddp_state;
int pddp_get_state(pddp_handle h, pddp_state* out /* [batch] */);
int pddp_set_state(pddp_handle h, const pddp_state* in /* [batch] */);

#define PDDP_PHASE_BP        0   /* backPassKern                   bpHelpers.cuh:339-420            */
#define PDDP_PHASE_FP        1   /* forwardSweepKern+forwardSimKern+costKern+defectKern  fpHelpers.cuh */
#define PDDP_PHASE_LS        2   /* line search + acceptRejectTrajGPU                                */
#define PDDP_PHASE_NIS       3   /* nextIterationSetupGPU          nisInitHelpers.cuh:247-279       */
#define PDDP_PHASE_INIT_NIS  4   /* derivatives part of initAlgGPU nisInitHelpers.cuh:365-371       */
#define PDDP_PHASE_INIT_COST 5   /* cost part of initAlgGPU        nisInitHelpers.cuh:385-395       */
#define PDDP_PHASE_BP_COOP   6   /* the wave-cooperative backward pass (all plants); for the KUKA arm PDDP_PHASE_BP is the lane-group
                                    kernel and this one exists so that tests can require the two to agree bit for bit */
#define PDDP_PHASE_BP_FUSED  7   /* handles whose production sweep composes the forward sweep's per-segment maps inside the matrix-core backward pass
                                    (KUKA arm; the quadrotor's k_bp_mq; M > 1): that backward pass -- every output of PDDP_PHASE_BP except A - B K / B du, plus the maps */
#define PDDP_PHASE_SWEEP_FUSED 8 /* ... and the kernel that finishes forwardSweepKern from those maps: every candidate's segment start states -> xs (arm) /
                                    -> the state part of the candidates' records of the boundary knots, array "xw" [problem][knot][step size][n + m] (12-state plants) */
#define PDDP_PHASE_ROLLOUT   9   /* forwardSimKern + costKern + defectKern WITHOUT the sweep: every candidate's segments start from the states that stand in
                                    "xs" at their first knots (fpHelpers.cuh:225-301 teacher-forced from stored start states, e.g. the reference's own)      */
int pddp_run_phase(pddp_handle h, int phase);

/* Plant plug-in evaluations on the device, `count` independent (x,u) pairs:
 * what = 0 dynamics -> qdd[count][npos]                 (dynamics<T>,          plants/dynamics_*.cuh)
 *        1 dynamicsGradient -> dqdd[count][npos*(n+m)]  (dynamicsGradient<T>)
 *        2 _integrator -> xnext[count][n]               (utils/integrators.cuh)
 *        3 _integratorGradient -> AB[count][n*(n+m)]
 *        4 dynamics on lane groups (KUKA arm only; the forward pass's code path) -> qdd[count][npos]
 *        6 the same with packed 6x6 products (the variant the forward pass runs) -> qdd[count][npos]
 *        5 dynamicsGradient on lane groups (KUKA arm only; next-iteration setup's code path) -> dqdd[count][npos*(n+m)]
 *        7 dynamics, one thread per evaluation (KUKA arm with a built-in robot model; the thread-lane kernels' code path) -> qdd[count][npos]
 *        8 dynamicsGradient, one thread per evaluation (composite form, csrc/plant_arm_tl.hpp) -> dqdd[count][npos*(n+m)]
 *        9 tool point and its Jacobian, one thread per evaluation (compute_eePos, plants/dynamics_arm.cuh:1879-1925) -> [count][6 + 42]
 *
 * Kernel selection is automatic per handle (plant, element type, cost family, problems in flight).  pddp_config.kernels (pddp_kernel_selection above) pins a family per
 * phase for comparison tests and measurements -- it never changes WHAT is computed, only which kernel family computes it.  The library reads no
 * environment variable for it. */
int pddp_plant_eval(pddp_handle h, int what, int count, const void* x, const void* u, void* out);

#ifdef __cplusplus
}
#endif
#endif
