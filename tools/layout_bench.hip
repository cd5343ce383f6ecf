// tools/layout_bench.hip -- what the HBM system gives the marching kernel's access pattern under
// different lattice layouts (development tool; no arithmetic, results are not lattices).
//
// One block per (strip, chunk) like lbm_march: per iteration the four waves of group 0 fetch one
// 256-column row of the nine planes by LDS-DMA (1 KiB per plane), and group 3 stores 224 columns of
// the nine planes of an earlier row from LDS (896 B per plane); groups 1, 2 only meet the barrier.
//   layout 0  plane-major       addr(k, y, x) = k * plane + y * pitch + x            (what the library uses)
//   layout 1  row-interleaved   addr(k, y, x) = (y * 9 + k) * pitch + x
//   layout 2  strip-major       addr(k, y, x) = ((s * ny + y) * 9 + k) * 224 + x - 224 s,  s = x / 224
//                               (a block's own columns of a row are 8064 contiguous bytes, rows follow on)
// plus a plain float4 copy of the same number of bytes.
//   ./tools/layout_bench [n=8192] [rows_per_chunk=241] [reps=5]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args {
  const float* src; float* dst;
  long plane; int pitch, nx, ny, H, nstrips;
};

constexpr int W = 256, HALO = 16, WOUT = 224, RING = 6;

template <int LAYOUT>
__device__ __forceinline__ long addr(const Args& a, int k, int y, int x) {
  if (LAYOUT == 0) return (long)k * a.plane + (long)y * a.pitch + x;
  if (LAYOUT == 1) return ((long)y * 9 + k) * a.pitch + x;
  const int s = x / WOUT;
  return (((long)s * a.ny + y) * 9 + k) * WOUT + (x - s * WOUT);
}

template <int LAYOUT>
__global__ __launch_bounds__(1024) void stream_march(const Args a) {
  __shared__ __attribute__((aligned(16))) float lds[9 * RING * W + 64];
  const int tid = threadIdx.x, t = tid & 255, lane = tid & 63;
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = gridDim.x;
  int b;
  { const int x = blockIdx.x & 7, i = blockIdx.x >> 3, q = nb >> 3, r = nb & 7; b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i; }
  const int chunk = b / a.nstrips, strip = b - chunk * a.nstrips;
  const int X0 = strip * WOUT, Y0 = chunk * a.H;
  const int wx = min(WOUT, a.nx - X0), hy = min(a.H, a.ny - Y0);
  const unsigned lds0 = (unsigned)(size_t)lds;
  int gx = X0 - HALO + 4 * lane;
  gx += (gx < 0) ? a.nx : 0; gx -= (gx >= a.nx) ? a.nx : 0;
  const int niter = hy + 5;
  // fetch wave w: planes 3w .. 3w+2 of row Y0 + j, 5 iterations before the store group reads it (per-lane source address)
  auto fetch2 = [&](int j) {
    int y = Y0 + j; y -= (y >= a.ny) ? a.ny : 0;
    const int slot = j % RING;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int k = 3 * wave + i;
      const float* p = a.src + addr<LAYOUT>(a, k, y, gx);
      unsigned keep; const unsigned la = lds0 + 4u * (unsigned)((k * RING + slot) * W);
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(p), "s"(la) : "memory");
    }
  };
  if (wave < 3) { for (int j = 0; j < 5; ++j) fetch2(j); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int j = 0; j < niter; ++j) {
    if (wave < 3) fetch2(j + 5);
    if (grp == 3 && j < hy) {
      const int slot = j % RING;
      const bool own = (t >= HALO) && (t < HALO + wx);
      if (own) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const float v = lds[(k * RING + slot) * W + t];
          __builtin_nontemporal_store(v, a.dst + addr<LAYOUT>(a, k, Y0 + j, X0 + t - HALO));
        }
      }
    }
    if (wave < 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ s, float4* __restrict__ d, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = s[i];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192;
  const int H = argc > 2 ? atoi(argv[2]) : 241;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const int pitch = (n + 63) / 64 * 64;
  const int nstrips = (n + WOUT - 1) / WOUT;
  const long plane = (long)n * pitch + 5184;
  // one allocation large enough for every layout (strip-major pads the last strip)
  const size_t floats = std::max((size_t)9 * plane, (size_t)nstrips * n * 9 * WOUT) + 4096;
  float *src, *dst;
  CK(hipMalloc(&src, floats * 4)); CK(hipMalloc(&dst, floats * 4));
  CK(hipMemset(src, 0, floats * 4)); CK(hipMemset(dst, 0, floats * 4));
  Args a{src, dst, plane, pitch, n, n, H, nstrips};
  const int nchunks = (n + H - 1) / H, nb = nstrips * nchunks;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double rd = 36.0 * 258 / 224 * n * n, wr = 36.0 * n * n;   // bytes moved per pass (reads incl. halo columns)
  printf("# n=%d rows/chunk=%d blocks=%d; per pass: %.1f MB read (nominal, halo columns included) + %.1f MB written\n", n, H, nb, rd / 1e6, wr / 1e6);
  for (int layout = 0; layout < 3; ++layout) {
    std::vector<float> ms;
    for (int r = 0; r < reps + 1; ++r) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 4; ++it) {
        if (layout == 0) hipLaunchKernelGGL(stream_march<0>, dim3(nb), dim3(1024), 0, 0, a);
        if (layout == 1) hipLaunchKernelGGL(stream_march<1>, dim3(nb), dim3(1024), 0, 0, a);
        if (layout == 2) hipLaunchKernelGGL(stream_march<2>, dim3(nb), dim3(1024), 0, 0, a);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (r) ms.push_back(t / 4);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    printf("layout %d (%s): %.1f us per pass, %.0f GB/s nominal\n", layout,
           layout == 0 ? "plane-major" : layout == 1 ? "row-interleaved" : "strip-major", med * 1e3, (rd + wr) / (med * 1e-3) / 1e9);
  }
  {
    const long nv = (long)9 * n * pitch / 4;
    std::vector<float> ms;
    for (int r = 0; r < reps + 1; ++r) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(copy4, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, 0, (const float4*)src, (float4*)dst, nv);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (r) ms.push_back(t / 4);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    printf("float4 copy of 9 planes: %.1f us per pass, %.0f GB/s\n", med * 1e3, 2.0 * nv * 16 / (med * 1e-3) / 1e9);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
