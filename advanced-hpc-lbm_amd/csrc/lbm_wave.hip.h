// lbm_wave.hip.h -- K time steps per pass with the time skew held in REGISTERS: every wavefront marches
// up its own 64-column strip of the lattice on its own, no LDS, no barrier, no other wave to wait for
// (reference step: /root/reference/d2q9-bgk.c:228-1813; per-cell arithmetic = collide_cell /
// accelerate_cell, so the lattice is bit-identical to K single steps).
//
// A lane owns one column.  In iteration j the wave
//     loads row  S0 + j          of the source lattice (nine aligned 256-byte row segments, prefetched one
//                                iteration ahead), which plays "level 0", and then for l = 1 .. K
//     computes step t+l on row  S0 + j - l  from what level l-1 produced in this iteration (the row
//                                above: planes 4,7,8), in the previous one (the same row: planes 0,1,3) and
//                                the one before (the row below: planes 2,5,6),
//     and stores level K's row to the destination lattice.
// What a level needs from its producer's older rows is nine registers per lane (three planes one row back,
// three planes one and two rows back); the neighbouring COLUMNS are the neighbouring lanes, one whole-wave
// DPP shift per diagonal / east / west plane (v_mov_b32_dpp wave_shr:1 / wave_shl:1).  A level is valid one
// lane less far out than its producer on either side, so a wave delivers 64 - 2K columns; rows cost 2K
// extra iterations per chunk.  HBM traffic per K updates: 36 B x 64/(64-2K) read + 36 B written = 19.3 B
// per lattice update at K = 4, 13.7 at K = 6, 10.5 at K = 8.  Everything a wave touches while it
// marches is its registers and the two lattices: waves drift apart freely, which is what overlaps one
// wave's loads with another's arithmetic (lbm_march, the LDS form of the same idea, spends a third of
// its time in the barrier that keeps its sixteen waves in step).
#pragma once
#include <type_traits>
#include "lbm_kernels.hip.h"

namespace lbm {

struct WaveArgs {
  const float* src;            // lattice at step t (accelerate phase of step t+1 already applied)
  float* dst;                  // lattice at step t+K
  long plane;
  int pitch, nx, ny;
  const uint8_t* blocked;
  float omega;
  int accel_row;               // global row ny-2
  int accel_out;               // apply the accelerate phase of step t+K+1 to the outputs
  float a1, a2;
  int H;                       // rows per chunk
  int nwc, nchunks;            // wave columns (of 64 - 2K output columns) x chunks = waves with work
  float* partials;             // [K][gridDim.x]: per block, speed sums of steps t+1 .. t+K
  const float* prev;           // the previous launch's partials, folded by block 0 (or nullptr)
  int prev_count;              // its block count
  double* prev_sum;            // K doubles
  // ---- a slab with neighbours (SLAB kernels; the same protocol as lbm_march's, lbm_march.hip.h): `ny` rows are this
  // slab's, the K rows below row 0 and above row ny-1 are read straight from the neighbours' lattices; the host orders
  // the launches (launch n+1 of a slab starts after launch n of both neighbours).
  const float* src_s; const float* src_n;
  long plane_s, plane_n;
  int ny_s, ny_n;
  const uint8_t* blocked_s; const uint8_t* blocked_n;
  int acc_rows[3];             // this slab's row indices of the lattice's accelerate row and of its periodic images
};

constexpr int kWaveBlock = 256;   // four independent waves per block (they only meet for the final sums)

// lane i <- lane i-1 (the column to the west) / lane i+1 (east); the wave's outermost lanes get 0,
// which only ever reaches columns the wave does not deliver
__device__ __forceinline__ float from_west(float v) {   // wave_shr:1, bound_ctrl: no source lane -> 0, no `old` register to set up
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_east(float v) {   // wave_shl:1
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xf, 0xf, true));
}

// waves per SIMD the register allocation must leave room for: the loop is a chain of K dependent cell
// updates per iteration, so it is other waves, not instruction-level parallelism, that keep a SIMD busy
constexpr int wave_min_occupancy(int K) { return K <= 6 ? 4 : 3; }

template <int K, int MODE, bool SLAB = false>
__global__ __launch_bounds__(kWaveBlock) __attribute__((amdgpu_waves_per_eu(wave_min_occupancy(K))))
void lbm_wave(const WaveArgs a) {
  constexpr bool FAST = (MODE & kFastMath) != 0, NTS = (MODE & kNtStore) != 0, NTL = (MODE & kNtLoad) != 0;
  constexpr int VW = 64 - 2 * K;
  static_assert(K >= 1 && K <= 12, "a wave must keep some columns");
  __shared__ double red_d[kWaveBlock / 64];
  __shared__ float red_f[kWaveBlock / 64][K];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

  // block 0 folds the previous launch's per-block sums (K steps) in double, fixed order
  if (blockIdx.x == 0 && a.prev != nullptr) {
    for (int l = 0; l < K; ++l) {
      double s = 0.0;
      for (int i = tid; i < a.prev_count; i += kWaveBlock) s += (double)a.prev[(long)l * a.prev_count + i];
      s = block_sum<double, kWaveBlock / 64>(s, red_d);
      if (tid == 0) a.prev_sum[l] = s;
      __syncthreads();
    }
  }

  // wave -> (wave column, chunk); the four waves of a block are neighbours in x (they share halo columns
  // through L1 / L2); blocks get an XCD-aware order so that neighbouring blocks share an L2 too
  const int nb = gridDim.x;
  int b;
  {
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3, q = nb >> 3, r = nb & 7;
    b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int g = b * (kWaveBlock / 64) + w;
  float sum[K];
#pragma unroll
  for (int l = 0; l < K; ++l) sum[l] = 0.f;

  if (g < a.nwc * a.nchunks) {
    const int chunk = g / a.nwc, wc = g - chunk * a.nwc;
    const int X0 = wc * VW, Y0 = chunk * a.H;
    const int wx = min(VW, a.nx - X0), hy = min(a.H, a.ny - Y0);
    const int S0 = Y0 - K;                      // first source row
    const int niter = hy + 2 * K;
    int gx = X0 - K + lane;                     // this lane's column (periodic)
    gx += (gx < 0) ? a.nx : 0; gx -= (gx >= a.nx) ? a.nx : 0;
    const bool out_ok = (lane >= K) && (lane < K + wx);
    // row walk of the loads: byte offset of (row, column) inside a plane / inside the obstacle map
    int gy = S0 % a.ny; gy += (gy < 0) ? a.ny : 0;
    unsigned ld_off = ((unsigned)gy * (unsigned)a.pitch + (unsigned)gx) * 4u;
    const unsigned ld_step = (unsigned)a.pitch * 4u, ld_back = (unsigned)(a.ny - 1) * (unsigned)a.pitch * 4u;
    // ... of the stores (level K's first row is Y0; its columns never wrap)
    unsigned st_off = ((unsigned)Y0 * (unsigned)a.pitch + (unsigned)(X0 - K + lane)) * 4u;
    // iterations (minus the level) in which a level's row is the accelerate row: the chunk plus its fill
    // rows may pass it twice
    int jacc = (a.accel_row - S0) % a.ny; jacc += (jacc < 0) ? a.ny : 0;
    int jacc2 = jacc + a.ny, jacc3 = -1;
    if constexpr (SLAB) {                       // nothing wraps in y: the three images of the accelerate row, as they are
      jacc = a.acc_rows[0] - S0; jacc2 = a.acc_rows[1] - S0; jacc3 = a.acc_rows[2] - S0;
    }
    int srow = S0;                              // SLAB: the source row about to be loaded, in this slab's numbering

    auto load_row = [&](float (&f)[9], int& blk) {
      if constexpr (SLAB) {
        // rows below 0 live at the top of the southern neighbour's lattice, rows from ny up at the bottom of the northern one's
        const float* base = a.src; long pl = a.plane; const uint8_t* bl = a.blocked; int r = srow;
        if (srow < 0) { base = a.src_s; pl = a.plane_s; bl = a.blocked_s; r = srow + a.ny_s; }
        else if (srow >= a.ny) { base = a.src_n; pl = a.plane_n; bl = a.blocked_n; r = srow - a.ny; }
        const unsigned off = ((unsigned)r * (unsigned)a.pitch + (unsigned)gx) * 4u;
#pragma unroll
        for (int k = 0; k < 9; ++k)
          f[k] = ldg<NTL>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + k * pl) + off));
        blk = bl[off >> 2];
        ++srow;
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k)
          f[k] = ldg<NTL>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.src + k * a.plane) + ld_off));
        blk = a.blocked[ld_off >> 2];
        ++gy;
        if (gy == a.ny) { gy = 0; ld_off -= ld_back; } else { ld_off += ld_step; }
      }
    };

    {
    // interface l (between level l and level l+1): planes 0,1,3 of the producer's previous row, planes 2,5,6
    // of its previous two rows
    float Bp[K][3], C1[K][3], C2[K][3];
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
      for (int i = 0; i < 3; ++i) { Bp[l][i] = 1.f; C1[l][i] = 1.f; C2[l][i] = 1.f; }
    unsigned mreg = 0u;                          // bit l = obstacle flag of the row level l works on
    float nxt[9]; int nblk;
    load_row(nxt, nblk);
    // One iteration.  STEADY: past the 2K fill iterations of the chunk every level has its history, the "is this level
    // running yet" tests are gone and with them the register copies their merge points force (a quarter of the loop).
    auto iteration = [&](auto steady_c, int j) {
      constexpr bool STEADY = decltype(steady_c)::value;
      float cur[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) cur[k] = nxt[k];
      mreg = (mreg << 1) | (nblk != 0 ? 1u : 0u);
      if (j + 1 < niter) load_row(nxt, nblk);    // next iteration's source row, in flight behind this one's arithmetic
#pragma unroll
      for (int l = 1; l <= K; ++l) {
        float p[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) p[k] = cur[k];
        if (STEADY || j >= 2 * l) {
          // pull (d2q9-bgk.c:2139-2147): row above = cur, same row = Bp, row below = C2 of interface l-1
          p[0] = Bp[l - 1][0];
          p[1] = from_west(Bp[l - 1][1]);
          p[3] = from_east(Bp[l - 1][2]);
          p[2] = C2[l - 1][0];
          p[5] = from_west(C2[l - 1][1]);
          p[6] = from_east(C2[l - 1][2]);
          p[4] = cur[4];
          p[7] = from_east(cur[7]);
          p[8] = from_west(cur[8]);
          const bool blk = ((mreg >> l) & 1u) != 0u;
          const float sp = collide_cell<FAST>(p, blk, a.omega);
          const int jl = j - l;
          if ((jl == jacc || jl == jacc2 || (SLAB && jl == jacc3)) && (l < K || a.accel_out != 0)) accelerate_cell(p, blk, a.a1, a.a2);
          sum[l - 1] += (out_ok && jl >= K && jl < K + hy) ? sp : 0.f;
        }
        if (STEADY || j >= 2 * (l - 1)) {        // the producer's row of this iteration becomes history for the next two
          C2[l - 1][0] = C1[l - 1][0]; C2[l - 1][1] = C1[l - 1][1]; C2[l - 1][2] = C1[l - 1][2];
          C1[l - 1][0] = cur[2]; C1[l - 1][1] = cur[5]; C1[l - 1][2] = cur[6];
          Bp[l - 1][0] = cur[0]; Bp[l - 1][1] = cur[1]; Bp[l - 1][2] = cur[3];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) cur[k] = p[k];
      }
      if (STEADY || j >= 2 * K) {                // level K's row S0 + j - K = Y0 + (j - 2K)
        if (out_ok) {
#pragma unroll
          for (int k = 0; k < 9; ++k)
            stg<NTS>(reinterpret_cast<float*>(reinterpret_cast<char*>(a.dst + k * a.plane) + st_off), cur[k]);
        }
        st_off += ld_step;
      }
    };
    int j = 0;
    for (; j < 2 * K && j < niter; ++j) iteration(std::false_type{}, j);
    for (; j < niter; ++j) iteration(std::true_type{}, j);
    }
  }

  // per level: block sum of the speeds -> partials[level][block]
#pragma unroll
  for (int l = 0; l < K; ++l) {
    const float s = wave_sum(sum[l]);
    if (lane == 0) red_f[w][l] = s;
  }
  __syncthreads();
  if (tid < K) a.partials[(long)tid * nb + blockIdx.x] = red_f[0][tid] + red_f[1][tid] + red_f[2][tid] + red_f[3][tid];
}

}  // namespace lbm
