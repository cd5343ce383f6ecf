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
  Relax omega;
  int accel_row;               // global row ny-2
  int accel_out;               // apply the accelerate phase of step t+K+1 to the outputs
  float a1, a2;
  int H;                       // rows per chunk
  int y_begin, y_end;          // rows this launch covers, in chunks of H from y_begin ([0, ny): the whole lattice / slab)
  int nchunks_a;               // chunks of that range; chunks beyond it (up to nchunks) cover a SECOND range of rows --
  int yb_begin, yb_end;        // the two edge chunks of a slab, bottom and top, in one launch (nchunks_a = nchunks: no second range)
  int nwc, nchunks;            // wave columns (of 64 C - 2K output columns) x chunks = waves with work
  float* partials;             // [K][pstride]: per block, speed sums of steps t+1 .. t+K; this launch's blocks start at pbase
  int pstride, pbase;          // (one launch per group: pstride = gridDim.x, pbase = 0; a group of several launches -- the
                               // edge chunks before the ghost-row exchange, the interior chunks beside it -- shares one array)
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
// updates per iteration, so it is other waves, not instruction-level parallelism, that keep a SIMD busy.
// Two columns per lane (C = 2) double the registers AND the independent work per wave: two waves per SIMD then
// issue what four did (a SIMD needs two waves to issue every cycle pair, MI355X_MICROARCH.md, wave scheduling).
constexpr int wave_min_occupancy(int K, int C = 1) { return C == 2 ? 2 : (K <= 6 ? 4 : 3); }

// C = columns per lane.  C = 1: the wave covers 64 columns and delivers 64 - 2K.  C = 2: a lane owns columns
// 2 lane, 2 lane + 1 of a 128-column strip (one aligned 8-byte access per plane and row), the wave delivers 128 - 2K
// -- 112 of 128 at K = 8 instead of 48 of 64 -- and half of the east / west neighbours are the lane's own other
// column: six DPP shifts per PAIR of cells instead of twelve.  Needs K and nx even (a lane's pair never straddles the
// periodic wrap, and is delivered or dropped as a whole).
template <int K, int MODE, bool SLAB = false, int C = 1>
__global__ __launch_bounds__(kWaveBlock) __attribute__((amdgpu_waves_per_eu(wave_min_occupancy(K, C))))
void lbm_wave(const WaveArgs a) {
  constexpr bool FAST = (MODE & kFastMath) != 0, NTS = (MODE & kNtStore) != 0, NTL = (MODE & kNtLoad) != 0;
  constexpr int VW = 64 * C - 2 * K;
  static_assert(K >= 1 && K <= 12, "a wave must keep some columns");
  static_assert(C == 1 || (C == 2 && K % 2 == 0), "one or two columns per lane; pairs need an even K");
  using fC = std::conditional_t<C == 1, float, f2a>;     // a lane's columns of one plane and row: one aligned access
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
    // rows [y_begin, y_end) of the lattice / slab in chunks of H rows (y_begin = 0, y_end = ny: all of it)
    const bool second = chunk >= a.nchunks_a;
    const int X0 = wc * VW, Y0 = second ? a.yb_begin + (chunk - a.nchunks_a) * a.H : a.y_begin + chunk * a.H;
    const int wx = min(VW, a.nx - X0), hy = min(a.H, (second ? a.yb_end : a.y_end) - Y0);
    const int S0 = Y0 - K;                      // first source row
    const int niter = hy + 2 * K;
    int gx = X0 - K + C * lane;                 // this lane's (first) column (periodic)
    gx += (gx < 0) ? a.nx : 0; gx -= (gx >= a.nx) ? a.nx : 0;
    const bool out_ok = (C * lane >= K) && (C * lane < K + wx);
    // row walk of the loads: byte offset of (row, column) inside a plane / inside the obstacle map
    int gy = S0 % a.ny; gy += (gy < 0) ? a.ny : 0;
    unsigned ld_off = ((unsigned)gy * (unsigned)a.pitch + (unsigned)gx) * 4u;
    const unsigned ld_step = (unsigned)a.pitch * 4u, ld_back = (unsigned)(a.ny - 1) * (unsigned)a.pitch * 4u;
    // ... of the stores (level K's first row is Y0; its columns never wrap)
    unsigned st_off = ((unsigned)Y0 * (unsigned)a.pitch + (unsigned)(X0 - K + C * lane)) * 4u;
    // iterations (minus the level) in which a level's row is the accelerate row: the chunk plus its fill
    // rows may pass it twice (never three times: the host keeps ny >= 2K)
    int jacc = (a.accel_row - S0) % a.ny; jacc += (jacc < 0) ? a.ny : 0;
    int jacc2 = jacc + a.ny, jacc3 = -1;
    if constexpr (SLAB) {                       // nothing wraps in y: the three images of the accelerate row, as they are
      jacc = a.acc_rows[0] - S0; jacc2 = a.acc_rows[1] - S0; jacc3 = a.acc_rows[2] - S0;
    }
    int srow = S0;                              // SLAB: the source row about to be loaded, in this slab's numbering

    // The source lattice is loaded plane by plane WHERE LEVEL 1 FIRST NEEDS IT, not row by row: for iteration j
    // planes 4,7,8 of row S0+j (pulled from the row above), planes 0,1,3 of row S0+j-1 (the row itself) and planes 2,5,6 of
    // row S0+j-2 (the row below) -- every plane of every row still exactly once, and level 1 needs no history registers
    // (nine fewer per column).  In the first two iterations the two older rows lie outside the chunk's rows: the addresses
    // are valid (periodic wrap; SLAB: clamped) and nothing uses the values (level 1 starts in iteration 2).
    auto load_row = [&](float (&f)[C][9], unsigned& blk) {
      const float *baseA = a.src, *baseB = a.src, *baseC = a.src;
      long plA = a.plane, plB = a.plane, plC = a.plane;
      const uint8_t* bl = a.blocked;
      unsigned offA, offB, offC;
      if constexpr (SLAB) {
        // rows below 0 live at the top of the southern neighbour's lattice (or of the ghost band that mirrors it), rows
        // from ny up at the bottom of the northern one's
        auto where = [&](int r, const float*& base, long& pl, const uint8_t** blp) -> unsigned {
          if (r < 0) { base = a.src_s; pl = a.plane_s; if (blp) *blp = a.blocked_s; r = max(r + a.ny_s, 0); }
          else if (r >= a.ny) { base = a.src_n; pl = a.plane_n; if (blp) *blp = a.blocked_n; r = r - a.ny; }
          return ((unsigned)r * (unsigned)a.pitch + (unsigned)gx) * 4u;
        };
        offA = where(srow, baseA, plA, &bl);
        offB = where(srow - 1, baseB, plB, nullptr);
        offC = where(srow - 2, baseC, plC, nullptr);
        ++srow;
      } else {
        offA = ld_off;
        offB = (gy == 0) ? ld_off + ld_back : ld_off - ld_step;
        offC = (gy == 0) ? ld_off + ld_back - ld_step : (gy == 1) ? ld_off - ld_step + ld_back : ld_off - 2u * ld_step;
      }
      auto ld = [&](int k, const float* base, long pl, unsigned off) {
        const fC v = ldg<NTL>(reinterpret_cast<const fC*>(reinterpret_cast<const char*>(base + k * pl) + off));
        if constexpr (C == 1) { f[0][k] = v; } else { f[0][k] = v.x; f[1][k] = v.y; }
      };
      ld(4, baseA, plA, offA); ld(7, baseA, plA, offA); ld(8, baseA, plA, offA);
      ld(0, baseB, plB, offB); ld(1, baseB, plB, offB); ld(3, baseB, plB, offB);
      ld(2, baseC, plC, offC); ld(5, baseC, plC, offC); ld(6, baseC, plC, offC);
      if constexpr (C == 1) blk = bl[offA >> 2];
      else blk = *reinterpret_cast<const uint16_t*>(bl + (offA >> 2));     // byte c = column c
      if constexpr (!SLAB) {
        ++gy;
        if (gy == a.ny) { gy = 0; ld_off -= ld_back; } else { ld_off += ld_step; }
      }
    };

    {
    // interface l (between level l and level l+1, l >= 1; level 1 pulls from the loads themselves): planes 0,1,3 of
    // the producer's previous row, planes 2,5,6 of its previous two rows  (index 0 of the arrays is never used)
    float Bp[K][C][3], C1[K][C][3], C2[K][C][3];
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i) { Bp[l][c][i] = 1.f; C1[l][c][i] = 1.f; C2[l][c][i] = 1.f; }
    unsigned mreg[C];                            // bit l = obstacle flag of the row level l works on, per column
#pragma unroll
    for (int c = 0; c < C; ++c) mreg[c] = 0u;
    float nxt[C][9]; unsigned nblk;
    load_row(nxt, nblk);
    // One iteration.  STEADY: past the 2K fill iterations of the chunk every level has its history, the "is this level
    // running yet" tests are gone and with them the register copies their merge points force (a quarter of the loop).
    auto iteration = [&](auto steady_c, int j) {
      constexpr bool STEADY = decltype(steady_c)::value;
      float cur[C][9];
#pragma unroll
      for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int k = 0; k < 9; ++k) cur[c][k] = nxt[c][k];
        mreg[c] = (mreg[c] << 1) | (((nblk >> (8 * c)) & 0xffu) != 0u ? 1u : 0u);
      }
      if (j + 1 < niter) load_row(nxt, nblk);    // next iteration's source row, in flight behind this one's arithmetic
#pragma unroll
      for (int l = 1; l <= K; ++l) {
        float p[C][9];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int k = 0; k < 9; ++k) p[c][k] = cur[c][k];
        if (STEADY || j >= 2 * l) {
          // pull (d2q9-bgk.c:2139-2147): row above = cur, same row = Bp, row below = C2 of interface l-1.
          // The column to the west of column c is the lane's own column c-1, or (c = 0) the last column of the lane
          // to the west: one whole-wave DPP shift; likewise to the east.
          const bool acc = ((j - l == jacc) || (j - l == jacc2) || (SLAB && j - l == jacc3)) && (l < K || a.accel_out != 0);
          const bool own_row = (j - l >= K) && (j - l < K + hy);
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const int cw = (c == 0) ? C - 1 : c - 1, ce = (c == C - 1) ? 0 : c + 1;
            // (level 1: all nine come straight from this iteration's loads, each plane from the row it is pulled from)
            const float s0 = (l == 1) ? cur[c][0] : Bp[l - 1][c][0];
            const float s1 = (l == 1) ? cur[cw][1] : Bp[l - 1][cw][1];
            const float s3 = (l == 1) ? cur[ce][3] : Bp[l - 1][ce][2];
            const float s2 = (l == 1) ? cur[c][2] : C2[l - 1][c][0];
            const float s5 = (l == 1) ? cur[cw][5] : C2[l - 1][cw][1];
            const float s6 = (l == 1) ? cur[ce][6] : C2[l - 1][ce][2];
            p[c][0] = s0;
            p[c][1] = (c == 0) ? from_west(s1) : s1;
            p[c][3] = (c == C - 1) ? from_east(s3) : s3;
            p[c][2] = s2;
            p[c][5] = (c == 0) ? from_west(s5) : s5;
            p[c][6] = (c == C - 1) ? from_east(s6) : s6;
            p[c][4] = cur[c][4];
            p[c][7] = (c == C - 1) ? from_east(cur[ce][7]) : cur[ce][7];
            p[c][8] = (c == 0) ? from_west(cur[cw][8]) : cur[cw][8];
          }
          if constexpr (C == 1) {
            const bool blk = ((mreg[0] >> l) & 1u) != 0u;
            const float sp = collide_cell<FAST>(p[0], blk, a.omega);
            if (acc) accelerate_cell(p[0], blk, a.a1, a.a2);
            sum[l - 1] += (out_ok && own_row) ? sp : 0.f;
          } else {
            bool blk[C];
            float sp[C];
#pragma unroll
            for (int c = 0; c < C; ++c) blk[c] = ((mreg[c] >> l) & 1u) != 0u;
            collide_cells<FAST, C>(p, blk, a.omega, sp);      // the lane's cells statement by statement: independent chains
#pragma unroll
            for (int c = 0; c < C; ++c)
              if (acc) accelerate_cell(p[c], blk[c], a.a1, a.a2);
            // (the K sums stay in registers: as lane-private LDS words -- read, add, write per level -- they cost the two-column
            // kernel 12 % at K = 8 and 18 % at K = 6, more than the eight dwords the K = 8 loop spills without them)
#pragma unroll
            for (int c = 0; c < C; ++c) sum[l - 1] += (out_ok && own_row) ? sp[c] : 0.f;
          }
        }
        if (l >= 2 && (STEADY || j >= 2 * (l - 1))) {   // the producer's row of this iteration becomes history for the next two
#pragma unroll
          for (int c = 0; c < C; ++c) {
            C2[l - 1][c][0] = C1[l - 1][c][0]; C2[l - 1][c][1] = C1[l - 1][c][1]; C2[l - 1][c][2] = C1[l - 1][c][2];
            C1[l - 1][c][0] = cur[c][2]; C1[l - 1][c][1] = cur[c][5]; C1[l - 1][c][2] = cur[c][6];
            Bp[l - 1][c][0] = cur[c][0]; Bp[l - 1][c][1] = cur[c][1]; Bp[l - 1][c][2] = cur[c][3];
          }
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int k = 0; k < 9; ++k) cur[c][k] = p[c][k];
      }
      if (STEADY || j >= 2 * K) {                // level K's row S0 + j - K = Y0 + (j - 2K)
        if (out_ok) {
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            fC v;
            if constexpr (C == 1) { v = cur[0][k]; } else { v.x = cur[0][k]; v.y = cur[1][k]; }
            stg<NTS>(reinterpret_cast<fC*>(reinterpret_cast<char*>(a.dst + k * a.plane) + st_off), v);
          }
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
  if (tid < K) a.partials[(long)tid * a.pstride + a.pbase + blockIdx.x] = red_f[0][tid] + red_f[1][tid] + red_f[2][tid] + red_f[3][tid];
}

}  // namespace lbm
