/* ORACLE -- test infrastructure, NOT product code.
 *
 * A plain-C, CPU restatement of the reference's parallel DDP / iLQR hot path
 * (plancherb1/parallel-DDP @ v1).  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it, and only as the checker / the reported CPU baseline.
 * The product path (parallel-ddp_amd/) never includes, links or calls anything in oracle/.
 *
 * Pinning status: no oracle/_ref build exists -- the reference cannot be compiled here without stand-in CUDA
 * headers (cuda.h / cuda_runtime.h / cublas_v2.h / cusolverDn.h are absent from this image).  Instead the reference's OWN STATEMENTS are
 * executed at fixture-generation time: tests/golden/refc2py.py preprocesses the reference sources where they lie and rewrites the requested
 * functions statement by statement into Python (host branches directly; `__CUDA_ARCH__` branches and __global__ kernels under a SIMT emulation
 * with the reference's launch geometry, __syncthreads() = barrier), tests/golden/make_phase_fixtures.py runs them in float64 on stored inputs
 * and commits numbers only (tests/golden/phase_fixtures.{npz,json}).  tests/test_phase_pins.py: this oracle's float64 instantiation equals
 * them to 1e-12 (measured: bit for bit) --
 *   PINNED, phase level, kernel AND host semantics: linearXfrmOrLoad, backprop, invHuu / invHuu_dim4 / computeKTdu_dim1 + invertMatrix,
 *   computeKTdu, computeCTG, computeFSVars, computeExpRed (backPassKern / backPassThreaded), forwardSweepInner, computeControlKT,
 *   forwardSimInner, _integrator / _integratorGradient of the three rules, costFunc / costGrad, costKern / costThreaded, defectKern /
 *   defectComp, the line search of forwardSimGPU;
 *   PINNED, plant level: dynamics<T> / dynamicsGradient<T> of the arm with everything they call (both robot models, gravity on / off), the
 *   tables of initI / initT, the closed-form plug-ins; additionally the independent URDF model (tests/test_urdf_pins.py), the reference's
 *   closed-form statements (tests/test_closed_form_pins.py) and the recorded figure-eight run (tests/test_fig8_pins.py);
 *   PINNED, end-effector cost family (ee_cost = 1): compute_eePos with its Jacobian, the in-sim cost accumulation of forwardSimKern,
 *   costGrad's gradient and Gauss-Newton Hessian, costKern<T,0/1>;
 *   PINNED, solver level: whole solves of runiLQR_GPU (host driver + every kernel, emulated end to end) -- step-size indices, rejections,
 *   exits, J, x, u, K -- against ora_run_ilqr_gpusem.
 *   PINNED, MPC wrapper (ora_gs_*): receding-horizon sequences of the reference's runiLQR_MPC_GPU with loadVarsGPU_MPC / storeVarsGPU_MPC
 *   executed end to end on one persistent GPUVars / trajVars pair (tests/test_mpc_pins.py: step-size indices, J, success bookkeeping,
 *   device-side x, u, K, d of every control cycle, fall-back cycles included).
 *   NOT PINNED by reference statements: runiLQR_CPU's thread spawning around the pinned host phases (a line-by-line restatement with
 *   citations).  The J / alpha traces in
 *   tests/golden/survey_kat.json (a build against stand-in CUDA headers) stay a quarantined regression check, not a pin.
 *
 * liboracle_fma.so is the same source compiled with contracted multiply-adds (how nvcc compiles the reference's device code): a member of
 * the float32 noise-floor ensemble of tests/test_fp32_bar.py, never a reference by itself.
 *
 * Two instantiations are exported: suffix _f32 (algType float, config.cuh:74) and _f64.
 */
#ifndef PDDP_ORACLE_H
#define PDDP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_MAX_ALPHA 64

typedef struct ora_cfg {
    int plant;          /* PLANT: 1 pendulum, 2 cart-pole, 3 quadrotor, 4 KUKA arm   config.cuh:21-61   */
    int N;              /* NUM_TIME_STEPS                                            config.cuh:133-135 */
    int M;              /* M_BLOCKS (= M_BLOCKS_B = M_BLOCKS_F)                      config.cuh:90-94   */
    int A;              /* NUM_ALPHA                                                 config.cuh:113-115 */
    int integrator;     /* INTEGRATOR 1 Euler, 2 midpoint, 3 RK3                     config.cuh:78-80   */
    int wafr_urdf;      /* USE_WAFR_URDF (arm)                                       config.cuh:182-184 */
    int mpc_mode;       /* MPC_MODE (arm gravity 0)                                  config.cuh:185-187 */
    int max_iter;       /* MAX_ITER                                                  config.cuh:83      */
    int ignore_max_rho_exit; /* IGNORE_MAX_ROX_EXIT                                  config.cuh:105-107 */
    /* CPU-baseline thread structure (config.cuh:146-161); <=0 means "derive from cores like the reference" */
    int cores;          /* value to use for CPU_CORES; <=0 -> sysconf                                    */
    int spawn_threads;  /* 1: create/join pthreads per phase like the reference; 0: run the same partition serially */
    int survey_int_minmax; /* PIN TEST ONLY: emulate the SURVEY harness artefact (integer min/max in the rho schedule) */
    int survey_double_trig; /* PIN TEST ONLY: second artefact of that build -- sin/cos(float) bound to the double libm functions */
    double total_time;  /* TOTAL_TIME                                                config.cuh:130-132 */
    double alpha_base;  /* ALPHA_BASE                                                config.cuh:110-112 */
    double rho_init;    /* RHO_INIT                                                  config.cuh:99-101  */
    double max_defect;  /* MAX_DEFECT_SIZE                                           config.cuh:124-126 */
    double tol_cost;    /* TOL_COST                                                  config.cuh:85-87   */
    double exp_red_min, exp_red_max;    /*                                           config.cuh:117-122 */
    double Q1, Q2, R, QF1, QF2;         /* arm joint-space cost weights  plants/cost_arm.cuh:97-103     */
    /* end-effector cost family (EE_COST 1 with USE_EE_VEL_COST 0, USE_SMOOTH_ABS 0, USE_LIMITS_FLAG 0; plants/cost_arm.cuh:104-115,206-389).
     * Pinned by the reference's own statements (tests/test_phase_pins.py: tool point, in-sim cost, gradient / Hessian, a whole solve). */
    int ee_cost;        /* EE_COST: xGoal = (x, y, z, roll, pitch, yaw) of the tool point          config.cuh:165-167 */
    int ee_cost_shift;  /* use_cost_shift of runiLQR_MPC_GPU (finalCostShift = shift)              MPCHelpers.cuh:866,876 */
    double Q_EE1, Q_EE2, QF_EE1, QF_EE2, R_EE, Q_xEE, QF_xEE, Q_xdEE, QF_xdEE;
    double ee_on_link_z;                /* EE_ON_LINK_Z (EE_TYPE 1: 0.0635)                         dynamics_arm.cuh:48-65 */
    double xTarget[14];                 /* nominal-state target (d_xTarget); zeros = the reference's xTarget == nullptr case */
    int ee_type;                        /* EE_TYPE 0 none / 1 flange (default) / 2 flange + peg: link-7 INERTIA_MODIFIER, WEIGHT_MODIFIER
                                           (default-URDF branch only)                              dynamics_arm.cuh:48-65,338-347 */
    int use_finite_diff;                /* USE_FINITE_DIFF: [A B] of the Euler step by central differences of `dynamics` (nisInitHelpers.cuh:138-166)  config.cuh:68 */
    double finite_diff_epsilon;         /* FINITE_DIFF_EPSILON                                                                                         config.cuh:69-71 */
    int use_smooth_abs;                 /* USE_SMOOTH_ABS (EE_COST 1): the tool-point term of a knot becomes sqrt(2 c + alpha^2) - alpha, its gradient c' / sqrt(2 c + alpha^2)   config.cuh:174-176, cost_arm.cuh:218-220,242-251 */
    double smooth_abs_alpha;            /* SMOOTH_ABS_ALPHA, default 0.2                                                                                                            cost_arm.cuh:116-118 */
    int use_limits;                     /* USE_LIMITS_FLAG, joint-space cost: quadratic penalties beyond 0.8 x the position / velocity / torque limits added to the cost and its GRADIENT
                                           (not to H -- costGrad, plants/cost_arm.cuh:13-94,136-149,176-199); with EE_COST 1: cost, gradient AND the diagonal of H (:289-291,341-343,374-376)                                                      config.cuh:171-173 */
} ora_cfg;

/* PLANT 5 -- a user plug-in (config.cuh:240-252: the reference compiles a cost file and a plant file into the solver).  The oracle restates the SOLVER; a plug-in is an
 * input to it, so plant 5 dispatches to callbacks a test registers (tests/plugin/plugin_shim.cpp compiles the user's own files for the host with the reference's
 * one-thread loop helpers).  Signatures mirror the plug-in functions: dynamics / dynamicsGradient (dqdd[col * npos + row]), costFunc / costGrad with the five
 * run-time weights of the current joint-space signature (plants/cost_arm.cuh:130,158).  n = 2 npos <= 14, m <= 7. */
#define ORA_PLUGIN_DECL(SUF, REAL)                                                                                                        \
    typedef struct ora_plugin_##SUF {                                                                                                     \
        int npos, m;                                                                                                                      \
        void (*dynamics)(REAL *qdd, const REAL *x, const REAL *u);                                                                        \
        void (*dynamics_gradient)(REAL *dqdd, REAL *qdd, const REAL *x, const REAL *u);                                                   \
        REAL (*cost_func)(const REAL *xk, const REAL *uk, const REAL *xg, int k, REAL Q1, REAL Q2, REAL R, REAL QF1, REAL QF2);           \
        void (*cost_grad)(REAL *Hk, REAL *gk, const REAL *xk, const REAL *uk, const REAL *xg, int k, int ld_H, REAL Q1, REAL Q2, REAL R,  \
                          REAL QF1, REAL QF2);                                                                                            \
    } ora_plugin_##SUF;                                                                                                                   \
    void ora_set_plugin_##SUF(const ora_plugin_##SUF *p);   /* NULL unregisters */
ORA_PLUGIN_DECL(f32, float)
ORA_PLUGIN_DECL(f64, double)

/* fill a config with the reference defaults for `plant` (config.cuh per-plant blocks) */
void ora_default_cfg(ora_cfg *c, int plant);
int  ora_state_size(int plant);
int  ora_control_size(int plant);

/* result of a full solve */
typedef struct ora_result {
    int iters;              /* value of `iter` at exit (DDPWrappers.cuh:24,515-516) */
    double t_total_ms, t_init_ms;
} ora_result;

#define ORA_DECL(SUF, REAL)                                                                                   \
    /* plant-level entry points (G1) */                                                                       \
    void ora_dynamics_##SUF(const ora_cfg *c, REAL *qdd, const REAL *x, const REAL *u);                       \
    void ora_dynamics_gradient_##SUF(const ora_cfg *c, REAL *dqdd, REAL *qdd, const REAL *x, const REAL *u);  \
    void ora_integrator_##SUF(const ora_cfg *c, REAL *xkp1, const REAL *x, const REAL *u);                    \
    void ora_integrator_gradient_##SUF(const ora_cfg *c, REAL *ABk, const REAL *x, const REAL *u);            \
    REAL ora_cost_func_##SUF(const ora_cfg *c, const REAL *xk, const REAL *uk, const REAL *xg, int k);        \
    void ora_cost_grad_##SUF(const ora_cfg *c, REAL *Hk, REAL *gk, const REAL *xk, const REAL *uk,            \
                             const REAL *xg, int k);                                                          \
    /* phase-level entry points (G2); sem_gpu selects kernel (1) or host-thread (0) semantics */              \
    int  ora_backward_pass_##SUF(const ora_cfg *c, int sem_gpu, const REAL *AB, REAL *P, REAL *p, REAL *Pp,   \
                                 REAL *pp, REAL *H, REAL *g, REAL *KT, REAL *du, const REAL *d, REAL *ApBK,   \
                                 REAL *Bdu, const REAL *x, const REAL *xp2, REAL rho, REAL *dJexp, int *err); \
    void ora_forward_sweep_##SUF(const ora_cfg *c, REAL *x, const REAL *ApBK, const REAL *Bdu, const REAL *d, \
                                 const REAL *xp, REAL alpha);                                                 \
    void ora_forward_sim_##SUF(const ora_cfg *c, REAL *x, REAL *u, const REAL *KT, const REAL *du, REAL *d,   \
                               REAL alpha, const REAL *xp);                                                   \
    REAL ora_total_cost_##SUF(const ora_cfg *c, int sem_gpu, const REAL *x, const REAL *u, const REAL *xg);   \
    REAL ora_max_defect_##SUF(const ora_cfg *c, int sem_gpu, const REAL *d);                                  \
    void ora_next_iteration_setup_##SUF(const ora_cfg *c, const REAL *x, const REAL *u, const REAL *xg,       \
                                        REAL *AB, REAL *H, REAL *g);                                          \
    int  ora_line_search_gpu_##SUF(const ora_cfg *c, const REAL *J, const REAL *dmax, const REAL *dJexp,      \
                                   REAL prevJ, int *ignore_defect, int *alphaIndex, REAL *dJ, REAL *z);       \
    /* solver-level (G3/G4): x0 [n*N], u0 [m*N] in/out; KT_out [n*m*N] optional */                            \
    int ora_run_ilqr_cpu_##SUF(const ora_cfg *c, REAL *x0, REAL *u0, const REAL *xGoal, REAL *Jout,           \
                               int *alphaOut, int rollout, int ignoreFirstDefect, REAL *KT_out,               \
                               ora_result *res);                                                              \
    int ora_run_ilqr_cpu2_##SUF(const ora_cfg *c, REAL *x0, REAL *u0, const REAL *xGoal, REAL *Jout,           \
                               int *alphaOut, int rollout, int ignoreFirstDefect, REAL *KT_out,               \
                               ora_result *res);                                                              \
    int ora_run_ilqr_gpusem_##SUF(const ora_cfg *c, REAL *x0, REAL *u0, const REAL *xGoal, REAL *Jout,        \
                                  int *alphaOut, int rollout, int ignoreFirstDefect, REAL *KT_out,            \
                                  ora_result *res);                                                           \
    /* end-effector kinematics and cost (arm, ee_cost = 1): eePos[6], deePos[7][6] (NULL to skip); cost of one knot; H_k, g_k */        \
    void ora_ee_pos_##SUF(const ora_cfg *c, const REAL *x, REAL *eePos, REAL *deePos);                        \
    REAL ora_ee_cost_##SUF(const ora_cfg *c, const REAL *xk, const REAL *uk, const REAL *goal, int k, int tshift); \
    void ora_ee_cost_grad_##SUF(const ora_cfg *c, REAL *Hk, REAL *gk, const REAL *xk, const REAL *uk, const REAL *goal, int k, int tshift); \
    /* forwardSimKern with EE_COST for one candidate: all M segments, JT[b] = the in-sim cost of segment b (d_JT[bInd + alphaInd * M_BLOCKS_F])      */ \
    void ora_forward_sim_ee_##SUF(const ora_cfg *c, REAL *x, REAL *u, const REAL *KT, const REAL *du, REAL *d, REAL alpha, const REAL *xp,          \
                                  const REAL *goal, int tshift, REAL *JT);                                                                          \
    /* lock-step experiment (examples/WAFR_MPC_examples.cu:111-139, MPCHelpers.cuh:819-858): the simulated robot -- plan x,u,KT of REAL, */ \
    /* plant in double; returns the average tracking error (0 and *failed = 1 when the time leaves the plan)                             */ \
    REAL ora_simulate_##SUF(const ora_cfg *c, const REAL *x, const REAL *u, const REAL *KT, double t0_us, double elapsed_us,          \
                            int substeps, const REAL *goal_xyz, REAL *xActual, int *failed);                                          \
    /* MPC wrapper, GPU semantics (DDPHelpers/MPCHelpers.cuh:602-655 load, :864-1016 loop, :755-774 store): a    */ \
    /* persistent state is seeded with a trajectory, then every solve shifts it by `shift` knots, rolls it out  */ \
    /* from the measured state and iterates; returns `iter`, *success = an accepted step with alpha index > 0   */ \
    void *ora_gs_create_##SUF(const ora_cfg *c);                                                              \
    void ora_gs_destroy_##SUF(void *h);                                                                       \
    void ora_gs_set_traj_##SUF(void *h, const REAL *x, const REAL *u);                                        \
    void ora_gs_get_traj_##SUF(void *h, REAL *x, REAL *u, REAL *KT, REAL *d);                                 \
    int ora_gs_mpc_solve_##SUF(void *h, const REAL *xActual, const REAL *xGoal, int shift, int clear_vars,    \
                               int full_rollout, int ignoreFirstDefect, int max_iter, REAL *Jout,            \
                               int *alphaOut, int *success);

/* the double-precision plant step the simulator uses from either instantiation (exported by ora_f64.c) */
void ora_internal_integrator_f64(const ora_cfg *c, double dt, double *xkp1, const double *x, const double *u);

ORA_DECL(f32, float)
ORA_DECL(f64, double)

#ifdef __cplusplus
}
#endif
#endif
