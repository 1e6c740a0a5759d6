/* ORACLE (test infrastructure only): algType = float instantiation (config.cuh:74). */
typedef float real;
#define ORA_SUF f32
#define RSIN sinf
#define RCOS cosf
#define RABS fabsf
#define RATAN2 atan2f
#define RSQRT sqrtf
#include "ora_impl.h"
