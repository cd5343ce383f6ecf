// lbm_exact_math.hip.h -- correctly rounded 1/x and sqrt(x) in 3 and 6 instructions instead of the compiler's
// 10 and ~16 (the reference divides and takes square roots in IEEE float, /root/reference/d2q9-bgk.c:1018-1130,
// and the lattice must come out bit for bit).
//
// The compiler's expansions are safe for every input: scaling for denormal operands and results, special
// cases, two correction steps.  For the operands a lattice actually produces -- densities and squared speeds
// that are ordinary normal numbers, or exactly zero at rest -- one Newton / Markstein step on the hardware's
// 1-ulp v_rcp_f32 / v_rsq_f32 already lands on the correctly rounded result.  That is not an estimate:
// tools/exact_math_check.hip runs ALL 2^32 float bit patterns through the short sequences and through the
// IEEE expansions on the GPU and compares the results bit for bit (tests/test_gpu_parity.py runs it):
//   recip_short  differs from 1.0f / x only for biased exponents 0, 253, 254, 255 of x (zeros and denormals;
//                results that are denormal; infinities and NaNs)
//   root_short   differs from sqrtf(x) only for exponents 0 .. 24 (other than +0 itself) and 255
// so a range check on the operand decides, wave-uniformly, between the short sequence and the full one; the
// guards keep a margin (exponents 1 .. 252, and 32 .. 254 or +0).  A wave holding any operand outside its guard
// takes the IEEE path for that operation: same result either way, by the exhaustive comparison.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lbm {

constexpr int kRecipExpLo = 1, kRecipExpHi = 252;    // biased exponents of x for which recip_short(x) == 1.0f / x (all of them checked)
constexpr int kRootExpLo = 32, kRootExpHi = 254;     // ... root_short(x) == sqrtf(x), x > 0; and x == +0

__device__ __forceinline__ float recip_short(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}

__device__ __forceinline__ float root_short(float x) {
  const float r = __builtin_amdgcn_rsqf(__builtin_fmaxf(x, 0x1p-126f));   // (x = 0: a finite r, and g = d = 0 below)
  const float g = x * r, h = 0.5f * r;
  const float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}

// 1.0f / x, correctly rounded, for every x
__device__ __forceinline__ float recip_exact(float x) {
  const uint32_t u = __float_as_uint(x);
  const bool ok = (u - ((uint32_t)kRecipExpLo << 23)) < ((uint32_t)(kRecipExpHi + 1 - kRecipExpLo) << 23);   // (positive x)
  if (__builtin_expect(__all(ok) != 0, 1)) return recip_short(x);
  return 1.0f / x;
}

// sqrtf(x), correctly rounded, for every x
__device__ __forceinline__ float root_exact(float x) {
  const uint32_t u = __float_as_uint(x);
  const bool ok = (u == 0u) || ((u - ((uint32_t)kRootExpLo << 23)) < ((uint32_t)(kRootExpHi + 1 - kRootExpLo) << 23));
  if (__builtin_expect(__all(ok) != 0, 1)) return root_short(x);
  return sqrtf(x);
}

}  // namespace lbm
