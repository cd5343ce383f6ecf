// lbm_march.hip.h -- K time steps per pass over a lattice streamed from HBM: row-marching temporal
// blocking (the k > 2 form of §8(f)2; reference step: /root/reference/d2q9-bgk.c:228-1813, per-cell
// arithmetic = collide_cell / accelerate_cell, so the lattice is bit-identical to K single steps).
//
// A block owns a column strip of the lattice -- 224 output columns plus 16 halo columns either
// side = 256 columns, one thread per column -- and marches up a chunk of H rows.  Its K groups of
// 256 threads are the K time levels of a software pipeline over rows: in iteration j
//     level 1 computes step t+1 on row  Yb + j          from the SOURCE lattice,
//     level s computes step t+s on row  Yb + j - 2(s-1)  from level s-1's rows (in LDS),
//     level K stores step t+K of its row to the DESTINATION lattice,
// with ONE workgroup barrier per iteration.  Each level pulls from a three-row window of its
// input, so it trails its producer by two rows; a level's row is valid one column less far out
// than its input on either side and the chunk is one row longer at either end per level, which is
// what the halo columns and the 3(K-1) extra iterations pay for.  There is no redundancy in y
// beyond those fill rows and 256/224 in x: HBM traffic is 36 B x 258/224 read + 36 B written per
// K updates = 19.4 B per lattice update for K = 4 (72 for the one-step kernel, 38.9 for lbm_sweep2).
//
// All global READS are LDS-DMA (global_load_lds: no VGPR, no VALU).  The four waves of level 1 are
// the fetchers: wave 0 moves planes 4,7,8 (pulled from the row above: needed first), wave 1 planes
// 0,1,3, wave 2 planes 2,5,6, wave 3 the row's 256 obstacle bytes -- one 1-KiB row per instruction,
// addressed by a scalar plane base plus a per-lane byte offset (the periodic wrap in x lives in that
// offset, the wrap in y in a row counter; nothing else in the kernel wraps).  Every group is fetched
// five iterations before its first use and the DMAs stay in flight across the barriers: raw
// s_barrier with lgkmcnt(0) only, and a counted vmcnt in the fetching wave (cdna_hip_programming.md
// §5, Pipelining across barriers).  Rows live in ring buffers: six rows per source plane, and per
// level 2 / 3 / 4 rows for the planes pulled from the row above / the same row / the row below
// (they die in that order), which is what lets the source ring, three level rings and the obstacle
// ring share 160 KiB.  Every ring size divides 12 and the iteration loop is unrolled by 12, so all
// ring slots are immediates in the LDS instructions: the loop body has no address arithmetic.
#pragma once
#include <type_traits>
#include "lbm_kernels.hip.h"

namespace lbm {

struct MarchArgs {
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
  int nstrips, nchunks;        // gridDim.x = nstrips * nchunks
  float* partials;             // [K][gridDim.x]: per block, speed sums of steps t+1 .. t+K
  const float* prev;           // the previous launch's partials, folded by block 0 (or nullptr)
  int prev_count;              // its block count
  double* prev_sum;            // K doubles
  // ---- a slab with neighbours (SLAB kernels): `ny` rows are this slab's; the K rows below row 0 and above row
  // ny-1 are READ STRAIGHT FROM THE NEIGHBOURS' LATTICES (same process: peer access; other process: hipIpc mapping;
  // over xGMI when the neighbour is another GPU) -- no halo buffers, no packing, no copies.  The host orders the
  // launches: launch n+1 of a slab starts after launch n of both neighbours has finished (that covers the rows it
  // reads and the rows of ITS source lattice the neighbours were still reading).
  const float* src_s; const float* src_n;            // the neighbours' lattices at step t
  long plane_s, plane_n;                              // their plane strides
  int ny_s, ny_n;                                     // their row counts (row -1 here = row ny_s-1 of the southern one)
  const uint8_t* blocked_s; const uint8_t* blocked_n;
  int acc_rows[3];             // this slab's row indices of the lattice's accelerate row and of its periodic images
                               // (any of them may lie in the K rows outside the slab; kNoRow where there is none)
};

template <int K>
struct MarchCfg {
  static constexpr int W = 256;        // columns (= threads) per level
  static constexpr int HALO = 16;      // halo columns either side (>= K, and keeps every row segment 64-B aligned)
  static constexpr int WOUT = W - 2 * HALO;
  static constexpr int U = 12;         // unroll of the iteration loop = common multiple of all ring sizes
  static constexpr int S0 = 6;         // source ring, rows per plane: a row is fetched 5 iterations before its only use
  static constexpr int SLA = 2, SLB = 3, SLC = 4;   // level rings: planes 4,7,8 / 0,1,3 / 2,5,6
  static constexpr int SM = 12;        // obstacle ring (fetched 4 iterations before level 1 reads it, last read by level K
                                       // 2(K-1) iterations later), stored TWICE, 12 rows apart, so that a level's slot
                                       // (iteration + a per-level constant) needs no modulo: constant in a register, iteration an immediate
  static constexpr int kPad = 64;      // column -1 of the first row stays inside the array
  static constexpr int r0A = kPad, r0B = r0A + 3 * S0 * W, r0C = r0B + 3 * S0 * W;
  static constexpr int lvl0 = r0C + 3 * S0 * W;
  static constexpr int lvl_floats = 3 * (SLA + SLB + SLC) * W;
  static constexpr int lA = 0, lB = 3 * SLA * W, lC = lB + 3 * SLB * W;   // inside one level ring
  static constexpr int mask0 = lvl0 + (K - 1) * lvl_floats;             // 2 SM rows of 256 bytes
  static constexpr int red0 = mask0 + 2 * SM * 64;                       // reductions (64 floats, 8-B aligned)
  static constexpr int total = red0 + 64 + 64;                           // + column 256 of the last row
  static_assert(HALO >= K, "a level loses one valid column per side");
  static_assert(U % S0 == 0 && U % SLA == 0 && U % SLB == 0 && U % SLC == 0 && U == SM, "ring slots must be immediates");
  static_assert(6 + 2 * (K - 1) <= SM, "obstacle ring lifetime");
  static_assert(total * 4 <= 160 * 1024, "LDS");
  static_assert((r0A % 4) == 0 && (mask0 % 4) == 0 && (red0 % 2) == 0, "DMA rows 16-B aligned");
};

// One row segment global -> LDS, no registers: 64 lanes x 16 B (a 256-float row) or x 4 B (256
// obstacle bytes) land at lds_byte_addr + lane * size.  base is wave-uniform, voff the lane's byte offset.
__device__ __forceinline__ void march_dma16(unsigned voff, const void* base, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void march_dma4(unsigned voff, const void* base, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_byte_addr) : "memory");
}

__device__ __forceinline__ int march_wrap(int r, int n) { r %= n; return r < 0 ? r + n : r; }

// A pointer the compiler cannot prove wave-uniform, made so (it IS uniform: it depends on block and iteration only);
// the LDS-DMA wants its base in scalar registers.
template <typename T>
__device__ __forceinline__ const T* march_uniform(const T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const T*)(((unsigned long long)hi << 32) | lo);
}

// What a block knows about its place in the lattice.
struct MarchGeom {
  int X0, Y0, wx, hy;          // output columns [X0, X0+wx), rows [Y0, Y0+hy)
  int Yb;                      // first row of level 1 = Y0 - (K-1)
  int niter;                   // hy + 3 (K-1)
};

constexpr int kMarchFirst = 0, kMarchMiddle = 1, kMarchLast = 2;

// The loop of one level.  ROLE: first = level 1 (pulls from the source ring, its four waves also fetch),
// middle = levels 2..K-1, last = level K (stores to the destination lattice).  One copy of the loop per
// role keeps each wave's instruction stream free of the other roles' branches and addresses.
template <int K, int MODE, int ROLE, bool SLAB>
__device__ __forceinline__ float march_level(const MarchArgs& a, const MarchGeom& g, float* lds, int t, int lane, int lvl, int wave) {
  using C = MarchCfg<K>;
  constexpr bool FAST = (MODE & kFastMath) != 0, NTS = (MODE & kNtStore) != 0;
  constexpr int W = C::W, S0 = C::S0;

  // ---- the fetchers (ROLE first, waves 0..3): in iteration j wave w moves, of source row rho (lattice
  // row Yb-1+rho), plane group w; lane L moves strip columns 4L .. 4L+3 (periodic in x; nx % 4 == 0)
  //   wave 0: planes 4,7,8 of rho = j+7 (level 1 pulls them in iteration rho-2)
  //   wave 1: planes 0,1,3 of rho = j+6 (pulled in iteration rho-1)
  //   wave 2: planes 2,5,6 of rho = j+5 (pulled in iteration rho)
  //   wave 3: the obstacle bytes of rho = j+6, twice (read by level s in iteration rho-1+2(s-1))
  // so every row is fetched five iterations (obstacles: four) before its first use and four fetches per
  // wave stay in flight across the barriers.  Fetches run on past the chunk's last row (a few rows more
  // than any level pulls): the counted wait below then needs no end game.
  const int f_ahead = (wave == 0) ? 7 : (wave == 2) ? 5 : 6;
  const int pk0 = (wave == 0) ? 4 : (wave == 1) ? 0 : 2, pk1 = (wave == 0) ? 7 : (wave == 1) ? 1 : 5,
            pk2 = (wave == 0) ? 8 : (wave == 1) ? 3 : 6;
  const float* fb0 = a.src + pk0 * a.plane;
  const float* fb1 = a.src + pk1 * a.plane;
  const float* fb2 = a.src + pk2 * a.plane;
  const unsigned funit = (wave == 3) ? 1u : 4u;                  // bytes per cell of what this wave fetches
  int gxl = g.X0 - C::HALO + 4 * lane;
  gxl += (gxl < 0) ? a.nx : 0; gxl -= (gxl >= a.nx) ? a.nx : 0;
  // lone lattice: the row counter wraps inside [0, ny); slab: it runs from -K on and rows outside [0, ny) are the neighbours'
  int fgy = SLAB ? g.Yb - 1 : march_wrap(g.Yb - 1, a.ny);        // row of the next source row to fetch
  unsigned fvoff = SLAB ? 0u : ((unsigned)fgy * (unsigned)a.pitch + (unsigned)gxl) * funit;
  const unsigned fstep = (unsigned)a.pitch * funit, fback = (unsigned)(a.ny - 1) * (unsigned)a.pitch * funit;
  const unsigned lds0 = (unsigned)(size_t)lds;                    // LDS byte address of the array (low half of the flat address)
  const unsigned fring = lds0 + 4u * (unsigned)((wave == 0) ? C::r0A : (wave == 1) ? C::r0B : C::r0C);
  auto fetch = [&](int slot6, int slot12) {   // the next row in sequence, into ring slot rho % 6 (rho % 12)
    const float *b0 = fb0, *b1 = fb1, *b2 = fb2;
    const uint8_t* bm = a.blocked;
    if (SLAB) {
      // whose row is it?  (rows fetched beyond the K a level ever pulls -- the prefetch runs on past the chunk -- are clamped)
      int row = fgy;
      if (fgy < 0) {
        row = max(a.ny_s + fgy, 0);
        b0 = a.src_s + pk0 * a.plane_s; b1 = a.src_s + pk1 * a.plane_s; b2 = a.src_s + pk2 * a.plane_s; bm = a.blocked_s;
      } else if (fgy >= a.ny) {
        row = min(fgy - a.ny, a.ny_n - 1);
        b0 = a.src_n + pk0 * a.plane_n; b1 = a.src_n + pk1 * a.plane_n; b2 = a.src_n + pk2 * a.plane_n; bm = a.blocked_n;
      }
      fvoff = ((unsigned)__builtin_amdgcn_readfirstlane(row) * (unsigned)a.pitch + (unsigned)gxl) * funit;
      b0 = march_uniform(b0); b1 = march_uniform(b1); b2 = march_uniform(b2); bm = march_uniform(bm);
    }
    if (wave == 3) {
      march_dma4(fvoff, bm, lds0 + 4u * (unsigned)(C::mask0 + slot12 * 64));
      march_dma4(fvoff, bm, lds0 + 4u * (unsigned)(C::mask0 + (slot12 + C::SM) * 64));
    } else {
      march_dma16(fvoff, b0, fring + 4u * (unsigned)((0 * S0 + slot6) * W));
      march_dma16(fvoff, b1, fring + 4u * (unsigned)((1 * S0 + slot6) * W));
      march_dma16(fvoff, b2, fring + 4u * (unsigned)((2 * S0 + slot6) * W));
    }
    ++fgy;
    if (!SLAB) { if (fgy == a.ny) { fgy = 0; fvoff -= fback; } else { fvoff += fstep; } }
  };
  auto fetch_wait = [&]() {   // all but this wave's four newest row fetches have landed
    if (wave == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  };
  if (ROLE == kMarchFirst) {
#pragma unroll
    for (int rho = 0; rho < 7; ++rho)
      if (rho < f_ahead) fetch(rho % S0, rho % C::SM);
    fetch_wait();
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- per-thread constants of the loop
  const int j_first = 3 * lvl, j_last = g.hy + 2 * K - 3 + lvl;      // this level's iterations
  const int j_own0 = (K - 1) + 2 * lvl, j_own1 = j_own0 + g.hy - 1;  // ... whose row belongs to this chunk (speed sum)
  // iterations in which this level's row is the accelerate row (the chunk plus its fill rows may pass it twice)
  const bool do_acc = (ROLE != kMarchLast) || (a.accel_out != 0);
  // (row r is computed in iteration r - Yb + 2 lvl)
  const int j_acc = !do_acc ? -1 : SLAB ? a.acc_rows[0] - g.Yb + 2 * lvl : march_wrap(a.accel_row - g.Yb, a.ny) + 2 * lvl;
  const int j_acc2 = !do_acc ? -1 : SLAB ? a.acc_rows[1] - g.Yb + 2 * lvl : j_acc + a.ny;
  const int j_acc3 = (do_acc && SLAB) ? a.acc_rows[2] - g.Yb + 2 * lvl : -1;
  const bool col_own = (t >= C::HALO) && (t < C::HALO + g.wx);
  const float* in0 = lds + t;                                           // level 1 pulls from the source ring
  const float* inL = lds + C::lvl0 + (lvl - 1) * C::lvl_floats + t;     // levels 2..K from the ring below
  float* outL = lds + C::lvl0 + lvl * C::lvl_floats + t;                // levels 1..K-1 write their own ring
  // obstacle byte of this level's row in iteration j: source row rho = j - 2 lvl + 1 -> slot (jj + c) with c in a register
  const uint8_t* mk = reinterpret_cast<const uint8_t*>(lds + C::mask0) + ((1 - 2 * lvl + 2 * C::SM) % C::SM) * 256 + t;
  // level K: byte offset of (row, column) inside a plane of the destination lattice; advances by one row per iteration
  unsigned st_off = ((unsigned)g.Y0 * (unsigned)a.pitch + (unsigned)(g.X0 + t - C::HALO)) * 4u;
  const unsigned st_step = (unsigned)a.pitch * 4u;
  float sum = 0.f;

  // one iteration; jj = j % 12 is a compile-time constant, which makes every ring slot an immediate
  auto iter = [&](auto jjc, const int j) {
    constexpr int jj = decltype(jjc)::value;
    if (ROLE == kMarchFirst) {
      const int s6 = (wave == 0) ? (jj + 7) % S0 : (wave == 2) ? (jj + 5) % S0 : (jj + 6) % S0;
      fetch(s6, (jj + 6) % C::SM);
    }
    if (j >= j_first && j <= j_last) {
      float p[9];
      if (ROLE == kMarchFirst) {
        // source rows rho = j (planes 2,5,6: row below), j+1 (0,1,3), j+2 (4,7,8: row above); d2q9-bgk.c:2139-2147
        constexpr int qa = (jj + 2) % S0, qb = (jj + 1) % S0, qc = jj % S0;
        p[0] = in0[C::r0B + (0 * S0 + qb) * W];
        p[1] = in0[C::r0B + (1 * S0 + qb) * W - 1];
        p[3] = in0[C::r0B + (2 * S0 + qb) * W + 1];
        p[2] = in0[C::r0C + (0 * S0 + qc) * W];
        p[5] = in0[C::r0C + (1 * S0 + qc) * W - 1];
        p[6] = in0[C::r0C + (2 * S0 + qc) * W + 1];
        p[4] = in0[C::r0A + (0 * S0 + qa) * W];
        p[7] = in0[C::r0A + (1 * S0 + qa) * W + 1];
        p[8] = in0[C::r0A + (2 * S0 + qa) * W - 1];
      } else {
        // the producer wrote its row of iteration i into slot i % size; this level wants the rows of
        // iterations j-1 (4,7,8), j-2 (0,1,3), j-3 (2,5,6) = slot (j+1) % size in all three rings
        constexpr int qa = (jj + 1) % C::SLA, qb = (jj + 1) % C::SLB, qc = (jj + 1) % C::SLC;
        p[0] = inL[C::lB + (0 * C::SLB + qb) * W];
        p[1] = inL[C::lB + (1 * C::SLB + qb) * W - 1];
        p[3] = inL[C::lB + (2 * C::SLB + qb) * W + 1];
        p[2] = inL[C::lC + (0 * C::SLC + qc) * W];
        p[5] = inL[C::lC + (1 * C::SLC + qc) * W - 1];
        p[6] = inL[C::lC + (2 * C::SLC + qc) * W + 1];
        p[4] = inL[C::lA + (0 * C::SLA + qa) * W];
        p[7] = inL[C::lA + (1 * C::SLA + qa) * W + 1];
        p[8] = inL[C::lA + (2 * C::SLA + qa) * W - 1];
      }
      const bool blk = mk[jj * 256] != 0;
      const float sp = collide_cell<FAST>(p, blk, a.omega);
      if (j == j_acc || j == j_acc2 || (SLAB && j == j_acc3)) accelerate_cell(p, blk, a.a1, a.a2);
      sum += (col_own && j >= j_own0 && j <= j_own1) ? sp : 0.f;
      if (ROLE != kMarchLast) {
        constexpr int wa = jj % C::SLA, wb = jj % C::SLB, wc = jj % C::SLC;
        outL[C::lB + (0 * C::SLB + wb) * W] = p[0];
        outL[C::lB + (1 * C::SLB + wb) * W] = p[1];
        outL[C::lB + (2 * C::SLB + wb) * W] = p[3];
        outL[C::lC + (0 * C::SLC + wc) * W] = p[2];
        outL[C::lC + (1 * C::SLC + wc) * W] = p[5];
        outL[C::lC + (2 * C::SLC + wc) * W] = p[6];
        outL[C::lA + (0 * C::SLA + wa) * W] = p[4];
        outL[C::lA + (1 * C::SLA + wa) * W] = p[7];
        outL[C::lA + (2 * C::SLA + wa) * W] = p[8];
      } else {
        if (col_own) {
#pragma unroll
          for (int k = 0; k < 9; ++k)
            stg<NTS>(reinterpret_cast<float*>(reinterpret_cast<char*>(a.dst + k * a.plane) + st_off), p[k]);
        }
        st_off += st_step;
      }
    }
    if (ROLE == kMarchFirst) fetch_wait();             // the rows iteration j+1 pulls have landed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  static_assert(C::U == 12, "twelve explicit calls below");
  const int niter = g.niter;
  for (int j0 = 0; j0 < niter; j0 += 12) {
    iter(std::integral_constant<int, 0>{}, j0);
    if (j0 + 1 >= niter) break;
    iter(std::integral_constant<int, 1>{}, j0 + 1);
    if (j0 + 2 >= niter) break;
    iter(std::integral_constant<int, 2>{}, j0 + 2);
    if (j0 + 3 >= niter) break;
    iter(std::integral_constant<int, 3>{}, j0 + 3);
    if (j0 + 4 >= niter) break;
    iter(std::integral_constant<int, 4>{}, j0 + 4);
    if (j0 + 5 >= niter) break;
    iter(std::integral_constant<int, 5>{}, j0 + 5);
    if (j0 + 6 >= niter) break;
    iter(std::integral_constant<int, 6>{}, j0 + 6);
    if (j0 + 7 >= niter) break;
    iter(std::integral_constant<int, 7>{}, j0 + 7);
    if (j0 + 8 >= niter) break;
    iter(std::integral_constant<int, 8>{}, j0 + 8);
    if (j0 + 9 >= niter) break;
    iter(std::integral_constant<int, 9>{}, j0 + 9);
    if (j0 + 10 >= niter) break;
    iter(std::integral_constant<int, 10>{}, j0 + 10);
    if (j0 + 11 >= niter) break;
    iter(std::integral_constant<int, 11>{}, j0 + 11);
  }
  return sum;
}

template <int K, int MODE, bool SLAB = false>
__global__ __launch_bounds__(K * 256) void lbm_march(const MarchArgs a) {
  using C = MarchCfg<K>;
  constexpr int NW = K * 4;
  // ONE shared array (cdna_hip_programming.md §5, trap 4a)
  __shared__ __attribute__((aligned(16))) float lds[C::total];
  float* red_f = lds + C::red0;
  double* red_d = reinterpret_cast<double*>(lds + C::red0);

  const int tid = threadIdx.x, t = tid & 255, lane = tid & 63;
  const int lvl = __builtin_amdgcn_readfirstlane(tid >> 8);      // level lvl+1
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // block 0 folds the previous launch's per-block sums (K steps) in double, fixed order
  if (blockIdx.x == 0 && a.prev != nullptr) {
    for (int l = 0; l < K; ++l) {
      double s = 0.0;
      for (int i = tid; i < a.prev_count; i += K * 256) s += (double)a.prev[(long)l * a.prev_count + i];
      s = block_sum<double, NW>(s, red_d);
      if (tid == 0) a.prev_sum[l] = s;
      __syncthreads();
    }
  }

  // block -> (strip, chunk): XCD-aware (block ids are dealt round-robin over the 8 XCDs; give each a
  // contiguous run, so strips that share halo columns mostly share an L2)
  const int nb = gridDim.x;
  int b;
  {
    // bijective for any grid size: XCD x (= blockIdx % 8) gets q+1 items if x < r, else q  (q = nb / 8, r = nb % 8)
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3, q = nb >> 3, r = nb & 7;
    b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int chunk = b / a.nstrips, strip = b - chunk * a.nstrips;
  MarchGeom g;
  g.X0 = strip * C::WOUT; g.Y0 = chunk * a.H;
  g.wx = min(C::WOUT, a.nx - g.X0); g.hy = min(a.H, a.ny - g.Y0);
  g.Yb = g.Y0 - (K - 1);
  g.niter = g.hy + 3 * (K - 1);

  float sum;
  if (lvl == 0) sum = march_level<K, MODE, kMarchFirst, SLAB>(a, g, lds, t, lane, lvl, wave);
  else if (lvl == K - 1) sum = march_level<K, MODE, kMarchLast, SLAB>(a, g, lds, t, lane, lvl, wave);
  else sum = march_level<K, MODE, kMarchMiddle, SLAB>(a, g, lds, t, lane, lvl, wave);

  // per level: block sum of the speeds -> partials[level][block]
  sum = wave_sum(sum);
  if (lane == 0) red_f[wave] = sum;
  __syncthreads();
  if (t == 0) a.partials[(long)lvl * nb + blockIdx.x] = red_f[4 * lvl] + red_f[4 * lvl + 1] + red_f[4 * lvl + 2] + red_f[4 * lvl + 3];
}

}  // namespace lbm
