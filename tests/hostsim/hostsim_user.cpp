// TEST TOOL -- NOT PRODUCT CODE.  libpddp_hostsim_<tag>.so of a `make user` build: the host emulation of plant 5 (policy header or reference-form plug-in).
#include "hostsim_impl.hpp"
#ifndef PDDP_HOSTSIM_HAS_USER_PLANT
#error "hostsim_user.cpp belongs to a `make user` build"
#endif

Base* hostsim_make_user(const pddp_config& c) { return mk_plant<UserPlant>(c); }
int hostsim_user_state_size() { return 2 * pddp::kUserPlantNPOS; }
int hostsim_user_control_size() { return pddp::kUserPlantNU; }
std::string hostsim_user_setup(const pddp_config& c) {
#ifdef PDDP_REF_PLANT_FILE
    return c.dtype == 0 ? ref_plugin_setup<float>(c.N) : ref_plugin_setup<double>(c.N);
#else
    (void)c; return std::string();
#endif
}

// a plant file + cost file in the reference's own form (make user PLANT_FILE=... COST_FILE=...): included LAST, so that what those files #define stays out of this file
#include "../../parallel-ddp_amd/csrc/ref_plugin.hpp"
