/* ORACLE (test infrastructure only): one precision instantiation = this header with `real` etc. defined. */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>
#include "oracle.h"
#include "iiwa14_model_data.h"
#define ORA_CAT3_(a, b, c) a##b##c
#define ORA_CAT3(a, b, c) ORA_CAT3_(a, b, c)
#define FN(name) ORA_CAT3(o_, name, ORA_SUF)
#define RMAX(a, b) ((a) > (b) ? (a) : (b))
#define RMIN(a, b) ((a) < (b) ? (a) : (b))
static void FN(gauss_jordan)(real *A, int DIM);
#include "ora_arm.inc"
#include "ora_plants.inc"
#include "ora_core.inc"
#include "ora_api.inc"
