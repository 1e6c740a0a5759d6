/* ORACLE (test infrastructure only): algType = double instantiation (config.cuh:73). */
typedef double real;
#define ORA_SUF f64
#define RSIN sin
#define RCOS cos
#define RABS fabs
#define RATAN2 atan2
#define RSQRT sqrt
#include "ora_impl.h"
