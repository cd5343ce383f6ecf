// tools/kbench.hip -- kernel-level A/B bench for the sweep kernel (development tool).
//
// Times variants of lbm::lbm_sweep on a synthetic lattice in ONE process, interleaved
// rounds (cdna_hip_programming.md §5.4 rule 24), with HIP events on the launch stream.
//   ./tools/kbench [n=8192] [steps=60] [rounds=5] [plane_pad_bytes=0]
// Prints, per variant: median and min us/step, MLUPS, GB/s at 72 B/LUP, fraction of 8 TB/s.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "../advanced-hpc-lbm_amd/csrc/lbm_kernels.hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Lat {
  int nx, ny, pitch; long plane;
  float* lat[2]; uint8_t* blocked; float* partials[2];
};

// ceiling reference: straight copy of the 9 planes, 16 B per lane, no stencil, no math
__global__ __launch_bounds__(256) void copy9(const float* __restrict__ src, float* __restrict__ dst, long plane, long nvec) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const lbm::f4a v = *reinterpret_cast<const lbm::f4a*>(src + k * plane + 4 * i);
    *reinterpret_cast<lbm::f4a*>(dst + k * plane + 4 * i) = v;
  }
}
static void launch_copy9(const Lat& L, int cur, int q, hipStream_t st, bool) {
  const long nvec = (long)L.ny * L.pitch / 4;
  hipLaunchKernelGGL(copy9, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, L.lat[cur], L.lat[cur ^ 1], L.plane, nvec);
}

template <int V, int MODE>
static void launch(const Lat& L, int cur, int q, hipStream_t st, bool accel) {
  lbm::SweepArgs a{};
  a.src = L.lat[cur]; a.dst = L.lat[cur ^ 1];
  a.plane = L.plane; a.pitch = L.pitch; a.nx = L.nx; a.nyl = L.ny;
  a.y_begin = 0; a.y_count = L.ny; a.y_stride = 1;
  const long top = (long)(L.ny - 1) * L.pitch;
  a.south2 = a.src + 2 * L.plane + top; a.south5 = a.src + 5 * L.plane + top; a.south6 = a.src + 6 * L.plane + top;
  a.north4 = a.src + 4 * L.plane; a.north7 = a.src + 7 * L.plane; a.north8 = a.src + 8 * L.plane;
  a.blocked = L.blocked; a.omega = 1.85f;
  a.accel_row = accel ? L.ny - 2 : -1; a.a1 = 0.1f * 0.01f / 9.f; a.a2 = 0.1f * 0.01f / 36.f;
  a.partials = L.partials[q];
  const long threads = (long)L.ny * (L.nx / V);
  const int grid = (int)((threads + lbm::kBlock - 1) / lbm::kBlock);
  a.prev_partials = L.partials[q ^ 1]; a.prev_count = grid; a.prev_sum = (double*)(L.partials[0] + 0) ;
  a.prev_partials = nullptr;  // the fold is negligible; keep the A/B about the sweep itself
  hipLaunchKernelGGL((lbm::lbm_sweep<V, MODE>), dim3(grid), dim3(lbm::kBlock), 0, st, a);
}

template <int TX, int TY, int MODE, int NT = 256>
static void launch2(const Lat& L, int cur, int q, hipStream_t st, bool accel) {
  lbm::Sweep2Args a{};
  a.src = L.lat[cur]; a.dst = L.lat[cur ^ 1];
  a.plane = L.plane; a.pitch = L.pitch; a.nx = L.nx; a.ny = L.ny;
  a.blocked = L.blocked; a.omega = 1.85f;
  a.accel_row = L.ny - 2; a.accel_out = accel ? 1 : 0; a.a1 = 0.1f * 0.01f / 9.f; a.a2 = 0.1f * 0.01f / 36.f;
  a.partials1 = L.partials[q]; a.partials2 = L.partials[q] + (long)L.nx * L.ny / 512;
  a.prev1 = a.prev2 = nullptr;
  a.by_begin = 0; a.by_count = L.ny / TY; a.by_stride = 1;
  const int grid = (L.nx / TX) * (L.ny / TY);
  hipLaunchKernelGGL((lbm::lbm_sweep2<TX, TY, MODE, lbm::kSweep2Plain, NT>), dim3(grid), dim3(NT), 0, st, a);
}

// state after `pairs` x 2 steps with the two-step kernel vs 2 x pairs single steps: must agree
template <int TX, int TY, int MODE>
static void selfcheck(Lat& L, hipStream_t st, int pairs) {
  const size_t nb = sizeof(float) * 9 * L.plane;
  std::vector<float> init(9 * L.plane), r1(9 * L.plane), r2(9 * L.plane);
  CK(hipMemcpy(init.data(), L.lat[0], nb, hipMemcpyDeviceToHost));
  int cur = 0;
  // the accelerate phase of the first step, as lbm_run's prologue does it
  hipLaunchKernelGGL(lbm::lbm_accelerate_row, dim3((L.nx + 255) / 256), dim3(256), 0, st, L.lat[0], L.plane, L.pitch,
                     L.nx, L.ny - 2, L.blocked, 0.1f * 0.01f / 9.f, 0.1f * 0.01f / 36.f);
  for (int t = 0; t < 2 * pairs; ++t) { launch<4, MODE>(L, cur, t & 1, st, t != 2 * pairs - 1); cur ^= 1; }
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(r1.data(), L.lat[cur], nb, hipMemcpyDeviceToHost));
  CK(hipMemcpy(L.lat[0], init.data(), nb, hipMemcpyHostToDevice));
  cur = 0;
  hipLaunchKernelGGL(lbm::lbm_accelerate_row, dim3((L.nx + 255) / 256), dim3(256), 0, st, L.lat[0], L.plane, L.pitch,
                     L.nx, L.ny - 2, L.blocked, 0.1f * 0.01f / 9.f, 0.1f * 0.01f / 36.f);
  for (int t = 0; t < pairs; ++t) { launch2<TX, TY, MODE>(L, cur, t & 1, st, t != pairs - 1); cur ^= 1; }
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(r2.data(), L.lat[cur], nb, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0; long ndiff = 0;
  for (int k = 0; k < 9; ++k)
    for (int y = 0; y < L.ny; ++y)
      for (int x = 0; x < L.nx; ++x) {
        const size_t i = (size_t)k * L.plane + (size_t)y * L.pitch + x;
        const double d = fabs((double)r1[i] - (double)r2[i]);
        if (d > maxd) maxd = d;
        if (fabs(r1[i]) > maxv) maxv = fabs(r1[i]);
        if (r1[i] != r2[i]) ++ndiff;
      }
  printf("# selfcheck sweep2<%d,%d,%d> vs 2x sweep<4>: %d pairs, max |diff| %.3e (max |f| %.3e), %ld values differ\n",
         TX, TY, MODE, pairs, maxd, maxv, ndiff);
  CK(hipMemcpy(L.lat[0], init.data(), nb, hipMemcpyHostToDevice));
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192;
  const int steps = argc > 2 ? atoi(argv[2]) : 60;
  const int rounds = argc > 3 ? atoi(argv[3]) : 5;
  const long pad = argc > 4 ? atol(argv[4]) : 0;
  Lat L;
  L.nx = n; L.ny = n; L.pitch = (n + 63) / 64 * 64;
  L.plane = (long)L.ny * L.pitch + pad / 4;
  for (int i = 0; i < 2; ++i) CK(hipMalloc((void**)&L.lat[i], sizeof(float) * 9 * L.plane));
  CK(hipMalloc((void**)&L.blocked, (size_t)L.plane));
  for (int i = 0; i < 2; ++i) CK(hipMalloc((void**)&L.partials[i], sizeof(float) * ((long)n * n / 256 + 16)));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  {  // box + wall obstacles, rest equilibrium
    std::vector<uint8_t> ob((size_t)L.plane, 0);
    for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x)
      if (y == 0 || y == n - 1 || x == 0 || x == n - 1 || x == (341 * n) / 1024) ob[(size_t)y * L.pitch + x] = 1;
    CK(hipMemcpy(L.blocked, ob.data(), ob.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(lbm::lbm_fill_equilibrium, dim3((unsigned)((L.plane + 255) / 256)), dim3(256), 0, st,
                       L.lat[0], L.plane, 0.1f * 4.f / 9.f, 0.1f / 9.f, 0.1f / 36.f);
    CK(hipStreamSynchronize(st));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Var { const char* name; void (*fn)(const Lat&, int, int, hipStream_t, bool); int steps_per_launch = 1; };
  if (n <= 2048) {
    selfcheck<64, 16, lbm::kFastMath>(L, st, 3);
    selfcheck<128, 8, lbm::kFastMath>(L, st, 5);
    selfcheck<32, 32, 0>(L, st, 4);
  }
  using namespace lbm;
  const Var vars[] = {
      {"copy9", launch_copy9},
      {"V4", launch<4, 0>}, {"V4 fast", launch<4, kFastMath>}, {"V4 fast nts", launch<4, kFastMath | kNtStore>},
      {"V4 fast ntl nts", launch<4, kFastMath | kNtLoad | kNtStore>},
      {"V2", launch<2, 0>}, {"V2 fast", launch<2, kFastMath>}, {"V2 fast nts", launch<2, kFastMath | kNtStore>},
      {"V2 fast ntl nts", launch<2, kFastMath | kNtLoad | kNtStore>},
      {"V1 fast", launch<1, kFastMath>},
      {"T2 64x16 fast", launch2<64, 16, kFastMath>, 2}, {"T2 64x16 fast nt", launch2<64, 16, kFastMath | kNtLoad | kNtStore>, 2},
      {"T2 64x16 fast nts", launch2<64, 16, kFastMath | kNtStore>, 2},
      {"T2 64x16 fast 512t", launch2<64, 16, kFastMath, 512>, 2}, {"T2 64x16 fast 1024t", launch2<64, 16, kFastMath, 1024>, 2},
      {"T2 64x16 nts 512t", launch2<64, 16, kFastMath | kNtStore, 512>, 2}, {"T2 64x16 nts 1024t", launch2<64, 16, kFastMath | kNtStore, 1024>, 2},
      {"T2 128x8 fast 512t", launch2<128, 8, kFastMath, 512>, 2},
      {"T2 128x8 fast", launch2<128, 8, kFastMath>, 2}, {"T2 128x8 fast nt", launch2<128, 8, kFastMath | kNtLoad | kNtStore>, 2},
      {"T2 32x32 fast", launch2<32, 32, kFastMath>, 2}, {"T2 32x32 fast nt", launch2<32, 32, kFastMath | kNtLoad | kNtStore>, 2},
      {"T2 256x4 fast", launch2<256, 4, kFastMath>, 2},
      {"V4 nomath", launch<4, kBenchNoMath>}, {"V4 fast aligned", launch<4, kFastMath | kBenchAlignedOnly>},
      {"V4 nomath aligned", launch<4, kBenchNoMath | kBenchAlignedOnly>},
      {"V2 nomath", launch<2, kBenchNoMath>}, {"V2 fast aligned", launch<2, kFastMath | kBenchAlignedOnly>},
  };
  const int nv = sizeof(vars) / sizeof(vars[0]);
  std::vector<std::vector<double>> us(nv);
  int cur = 0;
  for (int r = 0; r < rounds + 1; ++r) {
    for (int v = 0; v < nv; ++v) {
      CK(hipEventRecord(e0, st));
      for (int t = 0; t < steps; ++t) { vars[v].fn(L, cur, t & 1, st, true); cur ^= 1; }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) us[v].push_back(ms * 1e3 / steps / vars[v].steps_per_launch);
    }
  }
  printf("# n=%d steps=%d rounds=%d plane_pad=%ld B\n", n, steps, rounds, pad);
  for (int v = 0; v < nv; ++v) {
    std::sort(us[v].begin(), us[v].end());
    const double med = us[v][us[v].size() / 2], mn = us[v][0];
    const double mlups = (double)n * n / med;
    printf("%-20s median %9.2f us  min %9.2f us  %9.0f MLUPS  %7.0f GB/s  frac %.3f\n", vars[v].name, med, mn,
           mlups, mlups * 72 / 1e3, mlups * 72 / 8e6);
  }
  return 0;
}
