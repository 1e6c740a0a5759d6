// config.hpp -- the compile-time configuration surface of the reference (config.cuh) for the MI355X-native solver.
//
// A caller of the reference defines macros and then includes "config.cuh" (config.cuh:10-12); a caller of this
// package defines THE SAME macros and includes "hostapi/config.hpp".  Macro names, meanings and defaults follow
// config.cuh (line numbers below); the only differences are
//   * every macro is #ifndef-guarded: NUM_ALPHA, ALPHA_BASE, INTEGRATOR, RHO_INIT, MAX_DEFECT_SIZE (unconditional in the
//     per-plant blocks, config.cuh:24-58) and M_BLOCKS, MAX_ITER (:83,:90) can be overridden without editing the file;
//   * algType can be switched with -DPDDP_ALGTYPE_DOUBLE (the reference edits a typedef, :72-74);
//   * the macros configure a run-time record (pddp_config, include/pddp.h) instead of template code: the kernels live in
//     libpddp.so.
// Host-only C++11: compiles with g++ or hipcc; link with -lpddp.
#ifndef PDDP_HOSTAPI_CONFIG_HPP
#define PDDP_HOSTAPI_CONFIG_HPP

#include <sys/time.h>

#include "../../include/pddp.h"

#ifndef PLANT
#define PLANT 4                       // 1 pendulum, 2 cart-pole, 3 quadrotor, 4 KUKA iiwa14      config.cuh:21-23
#endif
#if PLANT == 1                        //                                                          config.cuh:24-28
#define NUM_POS 1
#define CONTROL_SIZE 1
#ifndef RHO_INIT
#define RHO_INIT 10.0
#endif
#elif PLANT == 2                      //                                                          config.cuh:29-34
#define NUM_POS 2
#define CONTROL_SIZE 1
#ifndef MAX_DEFECT_SIZE
#define MAX_DEFECT_SIZE 0.75
#endif
#ifndef RHO_INIT
#define RHO_INIT 10.0
#endif
#elif PLANT == 3                      //                                                          config.cuh:35-42
#define NUM_POS 6
#define CONTROL_SIZE 4
#ifndef ALPHA_BASE
#define ALPHA_BASE 0.5
#endif
#ifndef NUM_ALPHA
#define NUM_ALPHA 16
#endif
#ifndef RHO_INIT
#define RHO_INIT 1.0
#endif
#elif PLANT == 4                      //                                                          config.cuh:43-58
#define NUM_POS 7
#define CONTROL_SIZE 7
#ifndef TOTAL_TIME
#define TOTAL_TIME 0.5
#endif
#ifndef NUM_TIME_STEPS
#define NUM_TIME_STEPS 64
#endif
#ifndef ALPHA_BASE
#define ALPHA_BASE 0.5
#endif
#ifndef NUM_ALPHA
#define NUM_ALPHA 16
#endif
#ifndef RHO_INIT
#define RHO_INIT 12.5
#endif
#ifndef INTEGRATOR
#define INTEGRATOR 1                  // the arm's dynamics are only provided for Euler, as upstream (README.md:33)
#endif
#elif PLANT == 5                      // a user plant compiled into the library: make user PLANT_POLICY=<header>; link -lpddp_user (-lpddp_cpu_user)
#ifndef PDDP_USER_PLANT_NUM_POS       // the two sizes of the policy header (kUserPlantNPOS, kUserPlantNU), given to this translation unit as macros
#error "PLANT 5: define PDDP_USER_PLANT_NUM_POS and PDDP_USER_PLANT_CONTROL_SIZE (the policy header's kUserPlantNPOS / kUserPlantNU)"
#endif
#define NUM_POS PDDP_USER_PLANT_NUM_POS
#define CONTROL_SIZE PDDP_USER_PLANT_CONTROL_SIZE
#ifndef RHO_INIT
#define RHO_INIT 10.0
#endif
#else
#error "PLANT must be 1 (pendulum), 2 (cart-pole), 3 (quadrotor), 4 (KUKA iiwa14) or 5 (the user plant of a `make user` build)"
#endif
#define STATE_SIZE (2 * NUM_POS)

#ifdef PDDP_ALGTYPE_DOUBLE
typedef double algType;               //                                                          config.cuh:73
#else
typedef float algType;                //                                                          config.cuh:74
#endif

#ifndef INTEGRATOR
#define INTEGRATOR 3                  // 1 Euler, 2 midpoint, 3 RK3                               config.cuh:78-80
#endif
#ifndef MAX_ITER
#define MAX_ITER 100                  //                                                          config.cuh:83
#endif
#ifndef TOL_COST
#define TOL_COST 0.0001               //                                                          config.cuh:85-87
#endif
#ifndef M_BLOCKS
#define M_BLOCKS 4                    //                                                          config.cuh:90
#endif
#define M_BLOCKS_B M_BLOCKS           //                                                          config.cuh:91-94
#define M_BLOCKS_F M_BLOCKS
#define N_BLOCKS_B (NUM_TIME_STEPS / M_BLOCKS_B)
#define N_BLOCKS_F (NUM_TIME_STEPS / M_BLOCKS_F)
#ifndef RHO_INIT
#define RHO_INIT 1.0                  //                                                          config.cuh:99-101
#endif
#ifndef IGNORE_MAX_ROX_EXIT
#define IGNORE_MAX_ROX_EXIT 1         //                                                          config.cuh:105-107
#endif
#ifndef ALPHA_BASE
#define ALPHA_BASE 0.75               //                                                          config.cuh:110-112
#endif
#ifndef NUM_ALPHA
#define NUM_ALPHA 32                  //                                                          config.cuh:113-115
#endif
#ifndef EXP_RED_MIN
#define EXP_RED_MIN 0.05              //                                                          config.cuh:117-119
#endif
#ifndef EXP_RED_MAX
#define EXP_RED_MAX 1.25              //                                                          config.cuh:120-122
#endif
#ifndef USE_FINITE_DIFF
#define USE_FINITE_DIFF 0               // 1: [A B] by central differences of the dynamics (Euler)      config.cuh:68
#endif
#ifndef FINITE_DIFF_EPSILON
#define FINITE_DIFF_EPSILON 0.00001     //                                                          config.cuh:69-71
#endif
#ifndef MAX_DEFECT_SIZE
#define MAX_DEFECT_SIZE 1.0           //                                                          config.cuh:124-126
#endif
// The reference's FIXED switches (unconditional #defines of config.cuh, not overridable there without editing the file): the kernels are built for exactly these values.
// They are defined here under the same names so that caller code reading them compiles; a translation unit that pre-defines another value is refused instead of ignored.
#define PDDP_FIXED_SWITCH(name_is_ok, msg) static_assert(name_is_ok, msg)
#ifndef LINEAR_TRANSFORM_SWITCH
#define LINEAR_TRANSFORM_SWITCH 1     // boundary p = pp + Pp (x - xp2)                            config.cuh:81
#endif
#ifndef ALPHA_BEST_SWITCH
#define ALPHA_BEST_SWITCH 1           // best admissible step size, not the first                  config.cuh:82
#endif
#ifndef FORCE_PARALLEL
#define FORCE_PARALLEL 1              // blocks start from the previous pass's Pp / pp             config.cuh:95
#endif
#ifndef STATE_REG
#define STATE_REG 1                   // Tassa state regularisation                                config.cuh:98
#endif
#ifndef RHO_MAX
#define RHO_MAX 10000000.0            //                                                          config.cuh:102
#endif
#ifndef RHO_MIN
#define RHO_MIN 0.01                  //                                                          config.cuh:103
#endif
#ifndef RHO_FACTOR
#define RHO_FACTOR 1.25               //                                                          config.cuh:104
#endif
#ifndef USE_EXP_RED
#define USE_EXP_RED 1                 // EXP_RED_MIN < dJ / expected < EXP_RED_MAX                 config.cuh:116
#endif
#ifndef USE_MAX_DEFECT
#define USE_MAX_DEFECT 1              //                                                          config.cuh:123
#endif
#ifndef CONSTRAINTS_ON
#define CONSTRAINTS_ON 0              // (declared upstream, read by nothing)                      config.cuh:177-179
#endif
#ifndef DEBUG_SWITCH
#define DEBUG_SWITCH 0                //                                                          config.cuh:64
#endif
PDDP_FIXED_SWITCH(LINEAR_TRANSFORM_SWITCH == 1 && ALPHA_BEST_SWITCH == 1 && FORCE_PARALLEL == 1 && STATE_REG == 1 && USE_EXP_RED == 1 && USE_MAX_DEFECT == 1 && CONSTRAINTS_ON == 0,
                  "LINEAR_TRANSFORM_SWITCH / ALPHA_BEST_SWITCH / FORCE_PARALLEL / STATE_REG / USE_EXP_RED / USE_MAX_DEFECT are fixed at 1 and CONSTRAINTS_ON at 0, as config.cuh fixes them");
PDDP_FIXED_SWITCH(RHO_MAX == 10000000.0 && RHO_MIN == 0.01 && RHO_FACTOR == 1.25, "RHO_MAX 1e7, RHO_MIN 0.01 and RHO_FACTOR 1.25 are fixed, as config.cuh:102-104 fixes them");
#define onDefectBoundary(k) ((((k + 1) % N_BLOCKS_F) == 0) && (k < NUM_TIME_STEPS - 1))   //      config.cuh:127
#ifndef TOTAL_TIME
#define TOTAL_TIME 4.0                //                                                          config.cuh:130-132
#endif
#ifndef NUM_TIME_STEPS
#define NUM_TIME_STEPS 128            //                                                          config.cuh:133-135
#endif
#define TIME_STEP (TOTAL_TIME / (NUM_TIME_STEPS - 1))
#define NUM_STREAMS ((18 > 4 + NUM_ALPHA) ? 18 : (4 + NUM_ALPHA))   //                            config.cuh:144
#ifndef USE_WAFR_URDF
#define USE_WAFR_URDF 0               //                                                          config.cuh:182-184
#endif
#ifndef MPC_MODE
#define MPC_MODE 0                    //                                                          config.cuh:185-187
#endif
#ifndef EE_COST
#define EE_COST 0                     //                                                          config.cuh:165-167
#endif
// joint-space cost weights (plants/cost_arm.cuh:97-103); the pendulum / cart-pole / quadrotor weights are fixed inside
// the library exactly as plants/cost_{pend,cart,quad}.cuh fix them
#ifndef _Q1
#define _Q1 0.1
#endif
#ifndef _Q2
#define _Q2 0.001
#endif
#ifndef _R
#define _R 0.0001
#endif
#ifndef _QF1
#define _QF1 1000.0
#endif
#ifndef _QF2
#define _QF2 1000.0
#endif
// end-effector cost family (EE_COST 1; plants/cost_arm.cuh:104-115): xyz / rpy weights, control weight, nominal-state weights
#ifndef _Q_EE1
#define _Q_EE1 0.1
#endif
#ifndef _Q_EE2
#define _Q_EE2 0.0
#endif
#ifndef _R_EE
#define _R_EE 0.0001
#endif
#ifndef _QF_EE1
#define _QF_EE1 1000.0
#endif
#ifndef _QF_EE2
#define _QF_EE2 0.0
#endif
#ifndef _Q_xdEE
#define _Q_xdEE 0.1
#endif
#ifndef _QF_xdEE
#define _QF_xdEE 1000.0
#endif
#ifndef _Q_xEE
#define _Q_xEE 0.0
#endif
#ifndef _QF_xEE
#define _QF_xEE 0.0
#endif
// the end-effector VELOCITY cost (USE_EE_VEL_COST, upstream: "broken at this time", dynamics_arm.cuh:67-69) is not provided; its weights exist only so that call
// sites that pass them keep compiling.  USE_SMOOTH_ABS / USE_LIMITS_FLAG are (below).
#define _Q_EEV1 0.0
#define _Q_EEV2 0.0
#define _QF_EEV1 0.0
#define _QF_EEV2 0.0
#if defined(USE_EE_VEL_COST) && USE_EE_VEL_COST
#error "USE_EE_VEL_COST is not provided (upstream marks it broken, plants/dynamics_arm.cuh:67-69)"
#endif
#ifndef USE_SMOOTH_ABS
#define USE_SMOOTH_ABS 0           /* config.cuh:174-176: smooth-abs form of the tool-point term (EE_COST 1: pddp_config.use_smooth_abs) */
#endif
#ifndef SMOOTH_ABS_ALPHA
#define SMOOTH_ABS_ALPHA 0.2       /* plants/cost_arm.cuh:116-118 */
#endif
#if USE_SMOOTH_ABS && !EE_COST
#error "USE_SMOOTH_ABS belongs to the end-effector cost (EE_COST 1)"
#endif
#ifndef USE_LIMITS_FLAG
#define USE_LIMITS_FLAG 0          /* config.cuh:171-173: quadratic penalties beyond the position / velocity / torque limits (joint-space cost: pddp_config.use_limits) */
#endif
#if USE_LIMITS_FLAG && PLANT != 4
#error "USE_LIMITS_FLAG belongs to the KUKA arm's cost files (PLANT 4)"
#endif
#if EE_COST && PLANT != 4
#error "EE_COST belongs to the KUKA arm (PLANT 4)"
#endif
#ifndef EE_TYPE
#define EE_TYPE 1                     // flange, no end effector                                  config.cuh:47, dynamics_arm.cuh:50-52
#endif
#if EE_TYPE == 0
#define EE_ON_LINK_Z 0.0
#elif EE_TYPE == 1
#define EE_ON_LINK_Z 0.0635
#elif EE_TYPE == 2
#define EE_ON_LINK_Z 0.1524
#endif
#if EE_TYPE < 0 || EE_TYPE > 2
#error "EE_TYPE is 0 (no end effector), 1 (flange) or 2 (flange + peg) (dynamics_arm.cuh:50-65)"
#endif                                // (link 7's INERTIA_MODIFIER / WEIGHT_MODIFIER of the default URDF follow from pddp_config.ee_type, :53-65,338-347)

// matrix dimensions (config.cuh:195-236), column-major, leading dimension = rows
#define DIM_x_r STATE_SIZE
#define DIM_u_r CONTROL_SIZE
#define DIM_d_r STATE_SIZE
#define DIM_AB_r STATE_SIZE
#define DIM_AB_c (STATE_SIZE + CONTROL_SIZE)
#define DIM_A_r STATE_SIZE
#define DIM_A_c STATE_SIZE
#define DIM_H_r (STATE_SIZE + CONTROL_SIZE)
#define DIM_H_c (STATE_SIZE + CONTROL_SIZE)
#define DIM_g_r (STATE_SIZE + CONTROL_SIZE)
#define DIM_P_r STATE_SIZE
#define DIM_P_c STATE_SIZE
#define DIM_p_r STATE_SIZE
#define DIM_K_r CONTROL_SIZE
#define DIM_K_c STATE_SIZE
#define DIM_KT_r DIM_K_c
#define DIM_KT_c DIM_K_r
#define DIM_du_r CONTROL_SIZE
#define OFFSET_HXU (DIM_x_r * (DIM_x_r + DIM_u_r))
#define OFFSET_HUU (OFFSET_HXU + DIM_x_r)
#define OFFSET_HUX_GU DIM_x_r
#define OFFSET_B (DIM_AB_r * DIM_x_r)

#define get_time_us(time) (static_cast<double>(time.tv_sec * 1000000.0 + time.tv_usec))    //      config.cuh:137-141
#define get_time_ms(time) (get_time_us(time) / 1000.0)
#define time_delta_us(start, end) (static_cast<double>(get_time_us(end) - get_time_us(start)))
#define time_delta_ms(start, end) (time_delta_us(start, end) / 1000.0)

// A HIP stream handle (identical to hipStream_t when <hip/hip_runtime.h> is also included).
typedef struct ihipStream_t* pddpStream_t;

#include "DDPWrappers.hpp"
#if MPC_MODE
#include "MPCHelpers.hpp"             // config.cuh:264-270 includes MPCHelpers.cuh instead of DDPWrappers.cuh in MPC_MODE; here it adds to it
#include "exampleUtils.hpp"           // config.cuh:281: utils/exampleUtils.cuh (the lock-step experiment's helpers)
#include "LCMHelpers.hpp"             // config.cuh:272-279 (USE_LCM): message types, packing, trajectory runner -- without the LCM transport
#endif

#endif
