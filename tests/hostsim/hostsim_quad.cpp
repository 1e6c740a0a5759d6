// TEST TOOL -- NOT PRODUCT CODE.  libpddp_hostsim.so: the host emulation of plant `quad` (hostsim_impl.hpp).
#include "hostsim_impl.hpp"

Base* hostsim_make_quad(const pddp_config& c) { return mk_plant<QuadPlant>(c); }
