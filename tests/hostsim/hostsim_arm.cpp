// TEST TOOL -- NOT PRODUCT CODE.  libpddp_hostsim.so: the host emulation of the KUKA arm (hostsim_impl.hpp), one unit per element type.
#include "hostsim_impl.hpp"

#ifndef PDDP_HOSTSIM_ARM_HALF
#define PDDP_HOSTSIM_ARM_HALF 0        // 0: the float half + the factory, 1: the double half (two units: the arm is the largest instantiation by far)
#endif
Base* hostsim_make_arm_f64(const pddp_config& c);
#if PDDP_HOSTSIM_ARM_HALF == 0
Base* hostsim_make_arm(const pddp_config& c) {
    if (c.integrator != 1) return nullptr;                             // the arm is Euler-only, as config.cuh:58
    if (c.dtype == 0) { auto* s = new Sim<ArmPlant<float>, 1, float>(); s->cfg = c; s->init(); return s; }
    return c.dtype == 1 ? hostsim_make_arm_f64(c) : nullptr;
}
#else
Base* hostsim_make_arm_f64(const pddp_config& c) { auto* s = new Sim<ArmPlant<double>, 1, double>(); s->cfg = c; s->init(); return s; }
#endif
