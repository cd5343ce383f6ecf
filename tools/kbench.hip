// tools/kbench.hip -- kernel-level A/B bench for the sweep kernel (development tool).
//
// Times variants of lbm::lbm_sweep on a synthetic lattice in ONE process, interleaved
// rounds (cdna_hip_programming.md §5.4 rule 24), with HIP events on the launch stream.
//   ./tools/kbench [n=8192] [steps=60] [rounds=5] [plane_pad_bytes=0]
// Prints, per variant: median and min us/step, MLUPS, GB/s at 72 B/LUP, fraction of 8 TB/s.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
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
  struct Var { const char* name; void (*fn)(const Lat&, int, int, hipStream_t, bool); };
  using namespace lbm;
  const Var vars[] = {
      {"copy9", launch_copy9},
      {"V4", launch<4, 0>}, {"V4 fast", launch<4, kFastMath>}, {"V4 fast nts", launch<4, kFastMath | kNtStore>},
      {"V4 fast ntl", launch<4, kFastMath | kNtLoad>}, {"V4 fast ntl nts", launch<4, kFastMath | kNtLoad | kNtStore>},
      {"V2", launch<2, 0>}, {"V2 fast", launch<2, kFastMath>}, {"V2 fast nts", launch<2, kFastMath | kNtStore>},
      {"V2 fast ntl", launch<2, kFastMath | kNtLoad>}, {"V2 fast ntl nts", launch<2, kFastMath | kNtLoad | kNtStore>},
      {"V1 fast", launch<1, kFastMath>},
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
      if (r > 0) us[v].push_back(ms * 1e3 / steps);
    }
  }
  printf("# n=%d steps=%d rounds=%d plane_pad=%ld B\n", n, steps, rounds, pad);
  for (int v = 0; v < nv; ++v) {
    std::sort(us[v].begin(), us[v].end());
    const double med = us[v][us[v].size() / 2], mn = us[v][0];
    const double mlups = (double)n * n / med;
    printf("%-18s median %9.2f us  min %9.2f us  %9.0f MLUPS  %7.0f GB/s  frac %.3f\n", vars[v].name, med, mn,
           mlups, mlups * 72 / 1e3, mlups * 72 / 8e6);
  }
  return 0;
}
