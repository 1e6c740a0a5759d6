// Example of a USER plant plugged into the library at build time (the reference's plug-in surface, config.cuh:240-282: a plant file supplies
// `dynamics`, `dynamicsGradient`, and a cost file supplies `costFunc` / `costGrad`, all compile-time; plants/dynamics_pend.cuh is the smallest one).
//
//     make -C parallel-ddp_amd user PLANT_POLICY=examples/plants/damped_pendulum.hpp     ->  lib/libpddp_user.so, lib/libpddp_cpu_user.so
//
// builds the whole library once more with this header compiled in as plant number 5 (pddp_config.plant = 5; PLANT 5 in hostapi/config.hpp).
// What a policy header provides, inside namespace pddp, is the closed-form plug-in shape of the reference's pendulum / cart-pole / quadrotor:
//   constexpr int kUserPlantNPOS, kUserPlantNU                  NUM_POS and CONTROL_SIZE (STATE_SIZE = 2 NUM_POS: x = [q; qd])
//   user_plant_dynamics<T>(qdd, x, u)                           `dynamics`: qdd[NUM_POS]                        (plants/dynamics_pend.cuh:30-38)
//   user_plant_gradient<T>(dqdd, qdd, x, u)                     `dynamicsGradient`: dqdd[col * NUM_POS + row], cols = q, qd, u     (:40-51)
//   user_plant_QR(i, N), user_plant_R(N), user_plant_QF(N)      the diagonal quadratic cost of plants/cost_pend.cuh:19-51: running weight of state /
//                                                               control index i, control weight, final state weight
// Integrators, the Riccati pass, the line search, the batch axis, the C ABI and the facade are the library's own: nothing else changes.
//
// This plant: a pendulum with viscous damping and a point mass on a rod,  m l^2 qdd = u - b qd - m g l sin(q).
// With the defaults below except DAMPING = 0 it is EXACTLY the library's built-in pendulum (plant 1; the tests use that as the pin).
#pragma once
#include "pddp_common.hpp"

#ifndef USER_PENDULUM_DAMPING
#define USER_PENDULUM_DAMPING 0.2
#endif
#ifndef USER_PENDULUM_MASS
#define USER_PENDULUM_MASS 1.0
#endif
#ifndef USER_PENDULUM_LENGTH
#define USER_PENDULUM_LENGTH 1.0
#endif

namespace pddp {

constexpr int kUserPlantNPOS = 1, kUserPlantNU = 1;

template <typename T> PDDP_HD void user_plant_dynamics(T* qdd, const T* x, const T* u) {
    const double ml2 = USER_PENDULUM_MASS * USER_PENDULUM_LENGTH * USER_PENDULUM_LENGTH;
    qdd[0] = (u[0] + (-9.81 * USER_PENDULUM_MASS * USER_PENDULUM_LENGTH) * tsin<T>(x[0]) - USER_PENDULUM_DAMPING * x[1]) / ml2;
}
template <typename T> PDDP_HD void user_plant_gradient(T* dqdd, T* qdd, const T* x, const T* u) {
    const double ml2 = USER_PENDULUM_MASS * USER_PENDULUM_LENGTH * USER_PENDULUM_LENGTH;
    user_plant_dynamics<T>(qdd, x, u);
    dqdd[0] = (-9.81 * USER_PENDULUM_MASS * USER_PENDULUM_LENGTH) * tcos<T>(x[0]) / ml2;      // d/dq
    dqdd[1] = -USER_PENDULUM_DAMPING / ml2;                                                     // d/dqd
    dqdd[2] = 1 / ml2;                                                                          // d/du
}
PDDP_HD double user_plant_QR(int i, int) { return i == 0 ? 1.0 : 0.1; }    // the pendulum's weights (plants/cost_pend.cuh:19-24)
PDDP_HD double user_plant_R(int) { return 0.1; }
PDDP_HD double user_plant_QF(int) { return 1000.0; }

}  // namespace pddp
