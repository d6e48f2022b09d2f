/*
 * oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's FK / objective /
 * gradient path in plain C, built in fp32 (`*_f32`) and fp64 (`*_f64`) from oracle_impl.inc.
 *
 * Pinned against the reference: tests/test_oracle_golden.py checks every function against
 * golden vectors produced by running /root/reference itself (oracle/gen_golden.py).
 * Allowed callers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 * The product never loads this library.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp; no dependency on HIP or the reference)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/trk.h"

#define REAL float
#define FN(x) x##_f32
#define COS cosf
#define SIN sinf
#define SQRT sqrtf
#define FABS fabsf
#define FLOOR floorf
#define POW powf
#define ATAN2 atan2f
#define ASIN asinf
#include "oracle_impl.inc"
#undef REAL
#undef FN
#undef COS
#undef SIN
#undef SQRT
#undef FABS
#undef FLOOR
#undef POW
#undef ATAN2
#undef ASIN

#define REAL double
#define FN(x) x##_f64
#define COS cos
#define SIN sin
#define SQRT sqrt
#define FABS fabs
#define FLOOR floor
#define POW pow
#define ATAN2 atan2
#define ASIN asin
#include "oracle_impl.inc"

int orc_abi_version(void) { return TRK_ABI_VERSION; }

#ifdef _OPENMP
#include <omp.h>
int orc_max_threads(void) { return omp_get_max_threads(); }
void orc_set_threads(int n) { omp_set_num_threads(n); }
#else
int orc_max_threads(void) { return 1; }
void orc_set_threads(int n) { (void)n; }
#endif
