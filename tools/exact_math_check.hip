// tools/exact_math_check.hip -- do the short reciprocal / square-root sequences of lbm_kernels.hip.h give
// the correctly rounded result?  Exhaustive: every one of the 2^32 float bit patterns goes through the
// candidate and through the compiler's IEEE expansion (1.0f / x, sqrtf(x): HIP's default is correctly
// rounded division and square root), results compared bit for bit (NaNs compare equal to NaNs).
// Prints, per candidate, the number of inputs that differ and where they lie (by biased exponent), so that
// the guard in front of the short sequence can be drawn around exactly the inputs it is proven for.
//   ./tools/exact_math_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#include "../advanced-hpc-lbm_amd/csrc/lbm_exact_math.hip.h"

using lbm::recip_short; using lbm::root_short; using lbm::recip_exact; using lbm::root_exact;

__device__ __forceinline__ bool same(float a, float b) {
  const uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
  const bool nx = (x & 0x7fffffffu) > 0x7f800000u, ny = (y & 0x7fffffffu) > 0x7f800000u;
  return (nx && ny) || x == y;
}

// which = 0: recip_short vs 1/x; 1: root_short vs sqrtf; 2, 3: the guarded recip_exact / root_exact the kernels call (must
// never differ).  hist[biased exponent of x][sign] counts mismatches.
template <int WHICH>
__global__ void sweep_all(unsigned long long* hist, unsigned long long* first_bad) {
  const uint64_t n = 1ull << 32;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)i);
    float got, want;
    if (WHICH == 0) { got = recip_short(x); want = 1.0f / x; }
    else if (WHICH == 1) { got = root_short(x); want = sqrtf(x); }
    else if (WHICH == 2) { got = recip_exact(x); want = 1.0f / x; }
    else { got = root_exact(x); want = sqrtf(x); }
    if (!same(got, want)) {
      const uint32_t e = ((uint32_t)i >> 23) & 0xffu, s = (uint32_t)i >> 31;
      atomicAdd(&hist[e * 2 + s], 1ull);
      atomicMin(first_bad, (unsigned long long)i);
    }
  }
}

int main() {
  unsigned long long *hist, *bad;
  CK(hipMalloc(&hist, 512 * 8)); CK(hipMalloc(&bad, 8));
  int rc = 0;
  const char* names[4] = {"recip_short", "root_short", "recip_exact", "root_exact"};
  for (int which = 0; which < 4; ++which) {
    CK(hipMemset(hist, 0, 512 * 8));
    CK(hipMemset(bad, 0xff, 8));
    if (which == 0) hipLaunchKernelGGL(sweep_all<0>, dim3(4096), dim3(256), 0, 0, hist, bad);
    if (which == 1) hipLaunchKernelGGL(sweep_all<1>, dim3(4096), dim3(256), 0, 0, hist, bad);
    if (which == 2) hipLaunchKernelGGL(sweep_all<2>, dim3(4096), dim3(256), 0, 0, hist, bad);
    if (which == 3) hipLaunchKernelGGL(sweep_all<3>, dim3(4096), dim3(256), 0, 0, hist, bad);
    CK(hipDeviceSynchronize());
    unsigned long long h[512], b;
    CK(hipMemcpy(h, hist, sizeof h, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
    unsigned long long total = 0, guarded = 0;
    for (int e = 0; e < 256; ++e)
      for (int s = 0; s < 2; ++s) {
        total += h[e * 2 + s];
        // inputs the guard of the short sequence lets through: see lbm_exact_math.hip.h
        const bool in_guard = which >= 2 ? true
                            : which == 0 ? (e >= lbm::kRecipExpLo && e <= lbm::kRecipExpHi) : (s == 0 && e >= lbm::kRootExpLo && e <= lbm::kRootExpHi);
        if (in_guard) guarded += h[e * 2 + s];
      }
    printf("%s: %llu of 2^32 inputs differ from the IEEE result, %llu of them %s\n",
           names[which], total, guarded, which >= 2 ? "count (guarded function: none may)" : "inside the guard");
    if (total) {
      // biased exponents (of x > 0, then of x < 0) with at least one differing input, as ranges
      printf("  first differing input: 0x%08llx\n", b);
      for (int sgn = 0; sgn < 2; ++sgn) {
        printf("  x %s 0, exponents with differing inputs:", sgn ? "<" : ">");
        for (int e = 0; e < 256; ++e) {
          if (!h[e * 2 + sgn]) continue;
          int e1 = e;
          while (e1 + 1 < 256 && h[(e1 + 1) * 2 + sgn]) ++e1;
          if (e1 > e) printf(" %d-%d", e, e1); else printf(" %d", e);
          e = e1;
        }
        printf("\n");
      }
    }
    if (guarded) rc = 1;
  }
  printf(rc ? "FAILED: a guarded input gives a different result\n"
            : "OK: inside their guards both short sequences are correctly rounded for every input, and the guarded functions for all 2^32\n");
  return rc;
}
