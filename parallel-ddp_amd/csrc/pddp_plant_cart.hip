// libpddp.so: the handles of plant `cart` -- Solver<CartPlant<T>, integrator, T> for float / double and the three integrators, with every kernel they launch (solver_impl.hpp).
#include "solver_impl.hpp"

SolverBase* pddp_make_solver_cart(const pddp_config& c) { return make_solver_of<CartPlant>(c); }
