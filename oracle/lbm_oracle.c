/*
 * oracle/lbm_oracle.c -- TEST INFRASTRUCTURE ONLY (the CPU oracle).
 *
 * Plain-C serial restatement of the reference's hot path (timestep_new2 and
 * the functions around it, /root/reference/d2q9-bgk.c) in float and double.
 * See lbm_oracle_impl.h for the per-function reference citations.
 *
 * Parity status: PINNED.
 *   - double flavour (strict build: -O2 -ffp-contract=off, no fast-math)
 *     reproduces the reference's shipped golden files
 *     (tests/golden/{128x128,128x256}.{av_vels,final_state}.dat,
 *      {256x256,1024x1024}.av_vels.dat) at all 12 printed digits;
 *   - float flavour (same strict flags) matches, bit for bit, a strict-flag
 *     build of the reference's own source (oracle/_ref/libd2q9_ref_strict.so,
 *     built by oracle/Makefile from /root/reference where that exists).
 * Tests: tests/test_oracle_golden.py, tests/test_oracle_vs_reference.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  It is the checker, never the product.
 */
#include <math.h>
#include <string.h>

typedef struct {
  int nx;            /* cells in x */
  int ny;            /* cells in y */
  int maxIters;      /* time steps */
  int reynolds_dim;  /* dimension for the Reynolds number */
  double density;    /* density per link (parsed value; cast per flavour) */
  double accel;      /* density redistribution */
  double omega;      /* relaxation parameter */
} orc_param;

static inline float orc_sqrt_f32(float x) { return sqrtf(x); }
static inline double orc_sqrt_f64(double x) { return sqrt(x); }

#define ORC_REAL float
#define ORC_SUFFIX f32
#include "lbm_oracle_impl.h"
#undef ORC_REAL
#undef ORC_SUFFIX

#define ORC_REAL double
#define ORC_SUFFIX f64
#include "lbm_oracle_impl.h"
#undef ORC_REAL
#undef ORC_SUFFIX

/* Build-flavour tag so a test can assert which library it loaded. */
const char* orc_build_flavour(void)
{
#ifdef ORC_FAST_BUILD
  return "fast";   /* reference Makefile flags (-Ofast ...): CPU timing baseline only */
#else
  return "strict"; /* IEEE, no contraction: the parity oracle */
#endif
}
