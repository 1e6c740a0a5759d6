// TEST TOOL -- NOT PRODUCT CODE.  libpddp_hostsim.so: the host emulation of plant `pend` (hostsim_impl.hpp).
#include "hostsim_impl.hpp"

Base* hostsim_make_pend(const pddp_config& c) { return mk_plant<PendPlant>(c); }
