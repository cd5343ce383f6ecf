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

// EXPERIMENT (not in the library): three steps per pass.  Bit-identical to three single steps, but
// the extra level makes the kernel VALU-bound with 16 waves per CU: measured 8192^2 469 us/step
// against 478-498 for the two-step kernel on the same box, 1024^2 7.6 against 7.6-8.4 -- not worth
// a third halo protocol.  Kept here so the number can be re-measured.
namespace lbm {
// ---------------------------------------------------------------------------
// Three time steps in one pass (slab alone on its GPU, periodic wrap).  Same idea as lbm_sweep2
// with one more level: phase A computes step t+1 on the tile plus a two-cell ring
// ((TX+4) x (TY+4)) from HBM into LDS level 1, phase B step t+2 on the tile plus a one-cell ring
// from level 1 into LDS level 2, phase C step t+3 on the tile from level 2 to HBM.  Both LDS
// levels use the consumer-coordinate layout, so they hold (TX+2)(TY+2) and TX TY floats per
// plane: 79.6 KB for a 64 x 16 tile -> two blocks of 512 threads per CU.  HBM bytes per THREE
// updates: 36 written + 36 (TX+4)(TY+4)/(TX TY) = 47.8 read -> 27.9 B per lattice update.
struct Sweep3Args {
  const float* src;
  float* dst;
  long plane;
  int pitch, nx, ny;
  const uint8_t* blocked;
  lbm::Relax omega;
  int accel_row;               // row ny-2, or kNoRow
  int accel_out;               // accelerate the outputs too (0 when the run ends with this launch)
  float a1, a2;
  float* partials[3];          // per block: speed sums of steps t+1, t+2, t+3
  const float* prev[3];        // previous launch's partials, folded by block 0 (or nullptr)
  int prev_count;
  double* prev_sum;            // prev_sum[0..2]
};

template <int TX, int TY, int MODE, int NT>
__global__ __launch_bounds__(NT) void lbm_sweep3(const Sweep3Args a) {
  constexpr int V = TX * TY / NT;
  static_assert(V * NT == TX * TY && (V == 4 || V == 2 || V == 1), "phase C: V = 4, 2 or 1 cells per thread");
  constexpr bool FAST = (MODE & kFastMath) != 0;
  constexpr bool NTL = (MODE & kNtLoad) != 0, NTS = (MODE & kNtStore) != 0;
  constexpr int NW = NT / 64;
  constexpr int AW = TX + 4, AH = TY + 4;          // step t+1 region
  constexpr int BW = TX + 2, BH = TY + 2;          // step t+2 region
  constexpr int NA = (AW * AH + NT - 1) / NT, NB = (BW * BH + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float l1[9][BH][BW];   // values of step t+1, at their consumer (B-region) coordinates
  __shared__ __attribute__((aligned(16))) float l2[9][TY][TX];   // values of step t+2, at their consumer (tile) coordinates
  __shared__ float red_f[3][NW];
  __shared__ double red_d[NW];

  if (blockIdx.x == 0 && a.prev[0] != nullptr) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double sj = 0.0;
      for (int i = threadIdx.x; i < a.prev_count; i += NT) sj += (double)a.prev[j][i];
      sj = block_sum<double, NW>(sj, red_d);
      if (threadIdx.x == 0) a.prev_sum[j] = sj;
      __syncthreads();
    }
  }

  const int ntx = a.nx / TX;
  const int nblk = gridDim.x;
  int b = blockIdx.x;
  if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);   // XCD-contiguous, row-major
  const int by = b / ntx, bx = b - by * ntx;
  const int X0 = bx * TX, Y0 = by * TY;
  const long P = a.plane;
  const float* s = a.src;

  // ---- phase A: step t+1 on (TX+4) x (TY+4), HBM -> l1
  float sum1 = 0.f;
  {
    float q[NA][9];
    bool blk[NA];
    int acc[NA];
#pragma unroll
    for (int m = 0; m < NA; ++m) {
      const int idx = threadIdx.x + m * NT;
      if (idx < AW * AH) {
        const int cy = idx / AW, cx = idx - cy * AW;
        int gx = X0 - 2 + cx; gx += (gx < 0) ? a.nx : 0; gx -= (gx >= a.nx) ? a.nx : 0;
        int gy = Y0 - 2 + cy; gy += (gy < 0) ? a.ny : 0; gy -= (gy >= a.ny) ? a.ny : 0;
        const int xw = gx ? gx - 1 : a.nx - 1, xe = (gx + 1 == a.nx) ? 0 : gx + 1;
        const int ys = gy ? gy - 1 : a.ny - 1, yn = (gy + 1 == a.ny) ? 0 : gy + 1;
        const long rc = (long)gy * a.pitch, rs = (long)ys * a.pitch, rn = (long)yn * a.pitch;
        q[m][0] = ldg<NTL>(s + rc + gx);
        q[m][1] = ldg<NTL>(s + P + rc + xw);
        q[m][2] = ldg<NTL>(s + 2 * P + rs + gx);
        q[m][3] = ldg<NTL>(s + 3 * P + rc + xe);
        q[m][4] = ldg<NTL>(s + 4 * P + rn + gx);
        q[m][5] = ldg<NTL>(s + 5 * P + rs + xw);
        q[m][6] = ldg<NTL>(s + 6 * P + rs + xe);
        q[m][7] = ldg<NTL>(s + 7 * P + rn + xe);
        q[m][8] = ldg<NTL>(s + 8 * P + rn + xw);
        blk[m] = a.blocked[rc + gx] != 0;
        acc[m] = (gy == a.accel_row);
      }
    }
#pragma unroll
    for (int m = 0; m < NA; ++m) {
      const int idx = threadIdx.x + m * NT;
      if (idx < AW * AH) {
        const int cy = idx / AW, cx = idx - cy * AW;
        const float sp = collide_cell<FAST>(q[m], blk[m], a.omega);
        if (acc[m]) accelerate_cell(q[m], blk[m], a.a1, a.a2);
        sum1 += ((cx >= 2) && (cx < TX + 2) && (cy >= 2) && (cy < TY + 2)) ? sp : 0.f;
        const int x0 = cx - 1, y0 = cy - 1;              // B-region coordinates of this cell
        const bool xc = (x0 >= 0) && (x0 < BW), xe = (x0 + 1 >= 0) && (x0 + 1 < BW), xw = (x0 - 1 >= 0) && (x0 - 1 < BW);
        const bool yc = (y0 >= 0) && (y0 < BH), yn = (y0 + 1 >= 0) && (y0 + 1 < BH), ys = (y0 - 1 >= 0) && (y0 - 1 < BH);
        if (xc && yc) l1[0][y0][x0] = q[m][0];
        if (xe && yc) l1[1][y0][x0 + 1] = q[m][1];
        if (xc && yn) l1[2][y0 + 1][x0] = q[m][2];
        if (xw && yc) l1[3][y0][x0 - 1] = q[m][3];
        if (xc && ys) l1[4][y0 - 1][x0] = q[m][4];
        if (xe && yn) l1[5][y0 + 1][x0 + 1] = q[m][5];
        if (xw && yn) l1[6][y0 + 1][x0 - 1] = q[m][6];
        if (xw && ys) l1[7][y0 - 1][x0 - 1] = q[m][7];
        if (xe && ys) l1[8][y0 - 1][x0 + 1] = q[m][8];
      }
    }
  }
  __syncthreads();

  // ---- phase B: step t+2 on (TX+2) x (TY+2), l1 -> l2
  float sum2 = 0.f;
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    const int idx = threadIdx.x + m * NT;
    if (idx < BW * BH) {
      const int cy = idx / BW, cx = idx - cy * BW;
      int gx = X0 - 1 + cx; gx += (gx < 0) ? a.nx : 0; gx -= (gx >= a.nx) ? a.nx : 0;
      int gy = Y0 - 1 + cy; gy += (gy < 0) ? a.ny : 0; gy -= (gy >= a.ny) ? a.ny : 0;
      float p[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) p[k] = l1[k][cy][cx];
      const bool bl = a.blocked[(long)gy * a.pitch + gx] != 0;
      const float sp = collide_cell<FAST>(p, bl, a.omega);
      if (gy == a.accel_row) accelerate_cell(p, bl, a.a1, a.a2);
      sum2 += ((cx >= 1) && (cx <= TX) && (cy >= 1) && (cy <= TY)) ? sp : 0.f;
      const int x0 = cx - 1, y0 = cy - 1;                // tile coordinates of this cell
      const bool xc = (x0 >= 0) && (x0 < TX), xe = (x0 + 1 >= 0) && (x0 + 1 < TX), xw = (x0 - 1 >= 0) && (x0 - 1 < TX);
      const bool yc = (y0 >= 0) && (y0 < TY), yn = (y0 + 1 >= 0) && (y0 + 1 < TY), ys = (y0 - 1 >= 0) && (y0 - 1 < TY);
      if (xc && yc) l2[0][y0][x0] = p[0];
      if (xe && yc) l2[1][y0][x0 + 1] = p[1];
      if (xc && yn) l2[2][y0 + 1][x0] = p[2];
      if (xw && yc) l2[3][y0][x0 - 1] = p[3];
      if (xc && ys) l2[4][y0 - 1][x0] = p[4];
      if (xe && yn) l2[5][y0 + 1][x0 + 1] = p[5];
      if (xw && yn) l2[6][y0 + 1][x0 - 1] = p[6];
      if (xw && ys) l2[7][y0 - 1][x0 - 1] = p[7];
      if (xe && ys) l2[8][y0 - 1][x0 + 1] = p[8];
    }
  }
  __syncthreads();

  // ---- phase C: step t+3 on the tile, l2 -> HBM
  using RB = Row<V, false, NTS>;
  using RH = Row<V, false, false>;
  const int tx = threadIdx.x % (TX / V), ty = threadIdx.x / (TX / V);
  const int x = V * tx;
  const int gy = Y0 + ty;
  const long rrow = (long)gy * a.pitch;
  float o[9][V];
#pragma unroll
  for (int k = 0; k < 9; ++k) RH::ld(&l2[k][ty][0], x, o[k]);
  bool ob[V];
  if constexpr (V == 4) {
    const uint32_t mb = *reinterpret_cast<const uint32_t*>(a.blocked + rrow + X0 + x);
    ob[0] = (mb & 0xffu) != 0; ob[1] = (mb & 0xff00u) != 0; ob[2] = (mb & 0xff0000u) != 0; ob[3] = (mb & 0xff000000u) != 0;
  } else if constexpr (V == 2) {
    const uint16_t mb = *reinterpret_cast<const uint16_t*>(a.blocked + rrow + X0 + x);
    ob[0] = (mb & 0xffu) != 0; ob[1] = (mb & 0xff00u) != 0;
  } else {
    ob[0] = a.blocked[rrow + X0 + x] != 0;
  }
  const bool do_accel = a.accel_out && (gy == a.accel_row);
  float sum3 = 0.f;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    float p[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) p[k] = o[k][v];
    sum3 += collide_cell<FAST>(p, ob[v], a.omega);
    if (do_accel) accelerate_cell(p, ob[v], a.a1, a.a2);
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k][v] = p[k];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) RB::st(a.dst + k * P + rrow, X0 + x, o[k]);

  const float b1 = block_sum<float, NW>(sum1, red_f[0]);
  const float b2 = block_sum<float, NW>(sum2, red_f[1]);
  const float b3 = block_sum<float, NW>(sum3, red_f[2]);
  if (threadIdx.x == 0) { a.partials[0][blockIdx.x] = b1; a.partials[1][blockIdx.x] = b2; a.partials[2][blockIdx.x] = b3; }
}

}  // namespace lbm

template <int TX, int TY, int MODE, int NT>
static void launch3(const Lat& L, int cur, int q, hipStream_t st, bool accel) {
  lbm::Sweep3Args a{};
  a.src = L.lat[cur]; a.dst = L.lat[cur ^ 1];
  a.plane = L.plane; a.pitch = L.pitch; a.nx = L.nx; a.ny = L.ny;
  a.blocked = L.blocked; a.omega = 1.85f;
  a.accel_row = L.ny - 2; a.accel_out = accel ? 1 : 0; a.a1 = 0.1f * 0.01f / 9.f; a.a2 = 0.1f * 0.01f / 36.f;
  const long third = (long)L.nx * L.ny / 1024 + 4;
  for (int j = 0; j < 3; ++j) { a.partials[j] = L.partials[q] + j * third; a.prev[j] = nullptr; }
  const int grid = (L.nx / TX) * (L.ny / TY);
  hipLaunchKernelGGL((lbm::lbm_sweep3<TX, TY, MODE, NT>), dim3(grid), dim3(NT), 0, st, a);
}

// state after `triples` x 3 steps with the three-step kernel vs 3 x triples single steps
template <int TX, int TY, int MODE, int NT>
static void selfcheck3(Lat& L, hipStream_t st, int triples) {
  const size_t nb = sizeof(float) * 9 * L.plane;
  std::vector<float> init(9 * L.plane), r1(9 * L.plane), r2(9 * L.plane);
  CK(hipMemcpy(init.data(), L.lat[0], nb, hipMemcpyDeviceToHost));
  int cur = 0;
  hipLaunchKernelGGL(lbm::lbm_accelerate_row, dim3((L.nx + 255) / 256), dim3(256), 0, st, L.lat[0], L.plane, L.pitch,
                     L.nx, L.ny - 2, L.blocked, 0.1f * 0.01f / 9.f, 0.1f * 0.01f / 36.f);
  for (int t = 0; t < 3 * triples; ++t) { launch<4, MODE>(L, cur, t & 1, st, t != 3 * triples - 1); cur ^= 1; }
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(r1.data(), L.lat[cur], nb, hipMemcpyDeviceToHost));
  CK(hipMemcpy(L.lat[0], init.data(), nb, hipMemcpyHostToDevice));
  cur = 0;
  hipLaunchKernelGGL(lbm::lbm_accelerate_row, dim3((L.nx + 255) / 256), dim3(256), 0, st, L.lat[0], L.plane, L.pitch,
                     L.nx, L.ny - 2, L.blocked, 0.1f * 0.01f / 9.f, 0.1f * 0.01f / 36.f);
  for (int t = 0; t < triples; ++t) { launch3<TX, TY, MODE, NT>(L, cur, t & 1, st, t != triples - 1); cur ^= 1; }
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(r2.data(), L.lat[cur], nb, hipMemcpyDeviceToHost));
  double maxd = 0; long ndiff = 0;
  for (int k = 0; k < 9; ++k)
    for (int y = 0; y < L.ny; ++y)
      for (int x = 0; x < L.nx; ++x) {
        const size_t i = (size_t)k * L.plane + (size_t)y * L.pitch + x;
        const double d = fabs((double)r1[i] - (double)r2[i]);
        if (d > maxd) maxd = d;
        if (r1[i] != r2[i]) ++ndiff;
      }
  printf("# selfcheck sweep3<%d,%d,%d,%d> vs 3x sweep<4>: %d triples, max |diff| %.3e, %ld values differ\n", TX, TY, MODE, NT,
         triples, maxd, ndiff);
  CK(hipMemcpy(L.lat[0], init.data(), nb, hipMemcpyHostToDevice));
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
    selfcheck3<64, 16, lbm::kFastMath, 512>(L, st, 3);
    selfcheck3<64, 16, 0, 1024>(L, st, 2);
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
      {"T3 64x16 fast 512t", launch3<64, 16, kFastMath, 512>, 3}, {"T3 64x16 fast 1024t", launch3<64, 16, kFastMath, 1024>, 3},
      {"T3 64x16 nts 512t", launch3<64, 16, kFastMath | kNtStore, 512>, 3}, {"T3 64x16 fast 256t", launch3<64, 16, kFastMath, 256>, 3},
      {"T3 32x32 fast 512t", launch3<32, 32, kFastMath, 512>, 3}, {"T3 128x8 fast 512t", launch3<128, 8, kFastMath, 512>, 3},
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
