/* ORACLE (test infrastructure only): algType = double instantiation (config.cuh:73). */
typedef double real;
#define ORA_SUF f64
#define RSIN sin
#define RCOS cos
#define RABS fabs
#define RATAN2 atan2
#define RSQRT sqrt
#include "ora_impl.h"

/* _integrator<double> with an explicit step (simulateForward integrates elapsed/SUBSTEPS, not TIME_STEP) */
void ora_internal_integrator_f64(const ora_cfg *c, double dt, double *xkp1, const double *x, const double *u) {
    FN(ctx) K; FN(ctx_init)(&K, c); K.dt = dt; FN(integrator)(&K, xkp1, x, u);
}
