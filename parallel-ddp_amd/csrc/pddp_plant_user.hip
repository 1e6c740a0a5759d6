// libpddp_<tag>.so of a `make user` build: the handles of plant 5 -- a policy header (PLANT_POLICY) or a plant file + cost file in the reference's own form (PLANT_FILE /
// COST_FILE; csrc/ref_plugin.hpp) -- Solver<UserPlant<T>, integrator, T> with every kernel they launch.
#include "solver_impl.hpp"
#ifndef PDDP_USER_PLANT_HEADER
#error "pddp_plant_user.hip belongs to a `make user` build (PLANT_POLICY= or PLANT_FILE= / COST_FILE=)"
#endif

SolverBase* pddp_make_solver_user(const pddp_config& c) { return make_solver_of<UserPlant>(c); }
int pddp_user_plant_state_size() { return 2 * pddp::kUserPlantNPOS; }
int pddp_user_plant_control_size() { return pddp::kUserPlantNU; }

// a plant file + cost file in the reference's own form (make user PLANT_FILE=... COST_FILE=...): included LAST, so that what those files #define stays out of the library
#include "ref_plugin.hpp"
