/* ORACLE (test infrastructure only): precision-independent helpers. */
#include <string.h>
#include "oracle.h"

int ora_plugin_npos = 0, ora_plugin_m = 0;      /* dimensions of the registered plug-in (either precision registers both) */
int ora_state_size(int plant) { return plant == 1 ? 2 : plant == 2 ? 4 : plant == 3 ? 12 : plant == 5 ? 2 * ora_plugin_npos : 14; }
int ora_control_size(int plant) { return plant == 1 ? 1 : plant == 2 ? 1 : plant == 3 ? 4 : plant == 5 ? ora_plugin_m : 7; }

/* reference defaults: config.cuh:24-61 (per plant), :78-136 (algorithm), plants/cost_arm.cuh:97-103 */
void ora_default_cfg(ora_cfg *c, int plant) {
    memset(c, 0, sizeof(*c));
    c->plant = plant;
    c->N = plant == 4 ? 64 : 128;
    c->M = 4;
    c->A = (plant == 3 || plant == 4) ? 16 : 32;
    c->integrator = plant == 4 ? 1 : 3;
    c->wafr_urdf = 0; c->mpc_mode = 0;
    c->max_iter = 100;
    c->ignore_max_rho_exit = 1;
    c->cores = 0; c->spawn_threads = 1;
    c->total_time = plant == 4 ? 0.5 : 4.0;
    c->alpha_base = (plant == 3 || plant == 4) ? 0.5 : 0.75;
    c->rho_init = plant == 4 ? 12.5 : (plant == 3 ? 1.0 : 10.0);
    c->max_defect = plant == 2 ? 0.75 : 1.0;
    c->tol_cost = 0.0001;
    c->exp_red_min = 0.05; c->exp_red_max = 1.25;
    c->Q1 = 0.1; c->Q2 = 0.001; c->R = 0.0001; c->QF1 = 1000.0; c->QF2 = 1000.0;
    /* plants/cost_arm.cuh:104-115; EE_TYPE 1 (dynamics_arm.cuh:57-58) */
    c->Q_EE1 = 0.1; c->Q_EE2 = 0.0; c->QF_EE1 = 1000.0; c->QF_EE2 = 0.0; c->R_EE = 0.0001; c->Q_xEE = 0.0; c->QF_xEE = 0.0; c->Q_xdEE = 0.1; c->QF_xdEE = 1000.0;
    c->ee_on_link_z = 0.0635;
    c->ee_type = 1;
    c->use_finite_diff = 0; c->finite_diff_epsilon = 0.00001;
    c->use_smooth_abs = 0; c->smooth_abs_alpha = 0.2;
    c->use_limits = 0;
}
