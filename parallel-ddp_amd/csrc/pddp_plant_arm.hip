// libpddp.so: the handles of the KUKA arm -- Solver<ArmPlant<T>, 1, T> (Euler only, as config.cuh:58) for float / double, with every kernel they launch (solver_impl.hpp).
#include "solver_impl.hpp"

SolverBase* pddp_make_solver_arm(const pddp_config& c) {
    if (c.integrator != 1) return nullptr;
    return c.dtype == 0 ? static_cast<SolverBase*>(new Solver<ArmPlant<float>, 1, float>()) : c.dtype == 1 ? static_cast<SolverBase*>(new Solver<ArmPlant<double>, 1, double>()) : nullptr;
}
