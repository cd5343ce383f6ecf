// lbm_resident.hip.h -- the whole step loop in ONE launch for lattices that fit on the chip.
//
// MI355X has 256 CUs x 160 KiB of LDS = 40 MiB: a 1024 x 1024 lattice (36 MiB of distributions)
// fits.  lbm_resident gives every CU one tile of the lattice, keeps the tile's nine planes in that
// CU's LDS for ALL the steps of a run, and trades only the populations that cross a tile border with
// the eight neighbouring tiles, through mailboxes in global memory (L2 / Infinity Cache).  Between
// the first load and the last store of a run no lattice byte touches HBM, and there is no kernel
// boundary, no grid-wide barrier and no host in the loop: a tile waits for exactly the eight tiles
// it depends on.
//
// One step of one tile (reference step: /root/reference/d2q9-bgk.c:228-1813; the per-cell
// arithmetic is collide_cell / accelerate_cell of lbm_kernels.hip.h, so the lattice is
// bit-identical to lbm_sweep's):
//   pull     each thread reads the nine pulled values of its V cells from LDS at ITS OWN
//            coordinates (aligned ds_read_b128 for V = 4) -- populations are stored at the
//            coordinates of the cell that will pull them ("consumer coordinates");
//            threads on the tile border take the populations that come from another tile out of
//            the tile's mailbox: 8-byte {value, tag} granules, polled until the tag says
//            "state s-1" (MI355X_MICROARCH.md, Valid forms, R2: the data is the flag; one aligned
//            8-byte sc1 store per granule, sc1 loads, no fence on either side)
//   barrier  (every pull done before LDS is overwritten)
//   collide  / bounce back / accelerate-at-write-time / speed sum, in registers
//   push     each population goes to its consumer's coordinates: the x-shift is one lane shuffle
//            per diagonal/E/W plane (so that the LDS store is an aligned vector again), the y-shift
//            is the row the vector is stored to; what leaves the tile goes to the neighbour's
//            mailbox as granules tagged "state s"
//   barrier  (every push landed before the next pull)
// State 0 is loaded from the source lattice in HBM (accelerate phase of the first step applied in
// registers) and pushed like any other state; the last step stores to the destination lattice
// instead of pushing.
//
// Mailbox of a tile: per parity (state & 1), per plane k = 1..8, a ROW part (tx granules: the
// populations entering through the tile's south or north edge row, indexed by consumer x) and a
// COLUMN part (ty granules: those entering through the west or east edge column on any other row,
// indexed by consumer y).  Every granule has exactly one producer cell; a corner population
// arrives from the diagonal tile in the row part.  Two parities suffice: a tile pushes state s+2
// only after pulling state s+1 from ALL eight neighbours, and a neighbour pushed its state s+1
// only after it had pulled this tile's state s (the thread that produces a granule and the thread
// that consumes the opposite granule own the same cell).
//
// Every wait is bounded (wall clock) and watches a global abort word: a tile that gives up raises
// it, every other tile leaves its loop at the next poll, nobody stores to the destination lattice
// ... except tiles that had already finished; the host treats the run as not done (the SOURCE
// lattice is untouched) and repeats it with the streaming kernels.
#pragma once
#include "lbm_kernels.hip.h"

namespace lbm {

constexpr int kResidentMaxCells = 4096;                       // per tile: 9 x 4096 floats = 144 KiB of LDS
constexpr int kResidentLdsFloats = 9 * kResidentMaxCells + 64;   // + per-wave sums (2 x 16), abort word
constexpr long long kResidentTimeoutTicks = 100000000LL;       // 1 s of the 100 MHz wall clock
// timing experiments only (wrong results): LBM_RESIDENT_DEBUG=1 never waits for a tag, 2 also sends nothing
constexpr int kResDebugNoWait = 64, kResDebugNoSend = 128;

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct ResidentArgs {
  const float* src;            // source lattice (state 0), plane k at src + k*plane
  float* dst;                  // destination lattice (state nsteps)
  long plane;
  int pitch, nx, ny;
  const uint8_t* blocked;
  float omega;
  int accel_row;               // global row ny-2
  float a1, a2;
  int tx, ty;                  // tile size (tx = V * 2^m <= 64 V lanes; tx*ty <= kResidentMaxCells)
  int ntx, nty;                // tiles per lattice row / column (gridDim.x = ntx * nty)
  int nsteps;
  uint32_t tag0;               // tag of state 0 of this run; state s carries tag0 + s
  unsigned long long* mail;    // [tile][parity 2][plane 8][tx + ty] granules
  float* partials;             // [nsteps][ntiles]: per-tile speed sum of every step
  uint32_t* abort_word;        // global: non-zero = a tile gave up
};

// Granule index inside the mailbox array: tile `tile`, parity, plane k = 1..8, idx = consumer x (row
// part) or tx + consumer y (column part).  32-bit on purpose (the array is a few MB): the address
// is a scalar base plus a 32-bit offset.
__device__ __forceinline__ unsigned resident_goff(const ResidentArgs& a, int tile, int parity, int k, int idx) {
  return (unsigned)(((tile * 2 + parity) * 8 + (k - 1)) * (a.tx + a.ty) + idx);
}
__device__ __forceinline__ gu64* resident_granule(const ResidentArgs& a, int tile, int parity, int k, int idx) {
  return (gu64*)a.mail + resident_goff(a, tile, parity, k, idx);
}

template <bool SEND = true>
__device__ __forceinline__ void resident_send(gu64* g, uint32_t tag, float v) {
  if constexpr (SEND) __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ONE aligned 8-byte sc1 store
}

// direction vectors of the nine populations (d2q9-bgk.c:7-13): 1 E, 2 N, 3 W, 4 S, 5 NE, 6 NW, 7 SW, 8 SE
__device__ __forceinline__ constexpr int res_cx(int k) { return (k == 1 || k == 5 || k == 8) ? 1 : (k == 3 || k == 6 || k == 7) ? -1 : 0; }
__device__ __forceinline__ constexpr int res_cy(int k) { return (k == 2 || k == 5 || k == 6) ? 1 : (k == 4 || k == 7 || k == 8) ? -1 : 0; }

struct ResidentThread {
  int x0, y;                   // tile coordinates of the thread's V cells (x0 .. x0+V-1, y)
  int bx, by;                  // the tile
  int nbr[3][3];               // tile index of the neighbour at (dy + 1, dx + 1), periodic; [1][1] = this tile
  bool live;                   // the thread owns cells (blocks are padded to whole waves)
  bool on_w, on_e, on_s, on_n; // its cells touch the tile's west / east column, south / north row
};

// Push state `o` (own coordinates) to consumer coordinates: LDS inside the tile, granules outside.
template <int V, bool SEND = true>
__device__ __forceinline__ void resident_push(const ResidentArgs& a, const ResidentThread& t, float* lds,
                                              const float (&o)[9][V], uint32_t tag, int parity) {
  using RL = Row<V, false, false>;
  const int cells = a.tx * a.ty;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int cx = res_cx(k), cy = res_cy(k);
    float vec[V];
    if (cx == 0) {
#pragma unroll
      for (int j = 0; j < V; ++j) vec[j] = o[k][j];
    } else if (cx > 0) {            // consumer x0+j pulls from source x0+j-1
      const float left = __shfl_up(o[k][V - 1], 1, 64);
      vec[0] = left;
#pragma unroll
      for (int j = 1; j < V; ++j) vec[j] = o[k][j - 1];
    } else {                        // consumer x0+j pulls from source x0+j+1
      const float right = __shfl_down(o[k][0], 1, 64);
#pragma unroll
      for (int j = 0; j + 1 < V; ++j) vec[j] = o[k][j + 1];
      vec[V - 1] = right;
    }
    if (!t.live) continue;
    const int Y = t.y + cy;
    const bool out_y = (cy > 0 && t.on_n) || (cy < 0 && t.on_s);
    if (!out_y) {
      RL::st(lds + k * cells + Y * a.tx, t.x0, vec);     // (the entry-column element is rewritten by the pull)
      if (cx > 0 && t.on_e) resident_send<SEND>(resident_granule(a, t.nbr[1][2], parity, k, a.tx + Y), tag, o[k][V - 1]);
      if (cx < 0 && t.on_w) resident_send<SEND>(resident_granule(a, t.nbr[1][0], parity, k, a.tx + Y), tag, o[k][0]);
    } else {
      // leaves through the north (cy > 0) or south edge: row part of the tile above / below;
      // the element whose source lies in the tile to the west / east is that tile's to send
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const bool foreign = (cx > 0 && t.on_w && j == 0) || (cx < 0 && t.on_e && j == V - 1);
        if (!foreign) resident_send<SEND>(resident_granule(a, t.nbr[1 + cy][1], parity, k, t.x0 + j), tag, vec[j]);
      }
      if (cx > 0 && t.on_e) resident_send<SEND>(resident_granule(a, t.nbr[1 + cy][2], parity, k, 0), tag, o[k][V - 1]);
      if (cx < 0 && t.on_w) resident_send<SEND>(resident_granule(a, t.nbr[1 + cy][0], parity, k, a.tx - 1), tag, o[k][0]);
    }
  }
}

// One granule into `dst`; clears ok unless its tag is `want`.
__device__ __forceinline__ void resident_recv(const gu64* g, uint32_t want, float& dst, bool& ok) {
  const unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 load
  ok = ok && ((uint32_t)(x >> 32) == want);
  dst = __uint_as_float((uint32_t)x);
}

// Pull: the nine values of the thread's cells out of LDS, the border ones out of the mailbox
// (tag `want`).  Returns false if the wait was given up.
template <int V, bool WAIT = true>
__device__ __forceinline__ bool resident_pull(const ResidentArgs& a, const ResidentThread& t, const float* lds,
                                              float (&o)[9][V], uint32_t want, int parity) {
  using RL = Row<V, false, false>;
  const int cells = a.tx * a.ty;
  if (t.live) {
#pragma unroll
    for (int k = 0; k < 9; ++k) RL::ld(lds + k * cells + t.y * a.tx, t.x0, o[k]);
  }
  if (!(t.live && (t.on_w || t.on_e || t.on_s || t.on_n))) return true;
  const int ks[3] = {2, 5, 6}, kn[3] = {4, 7, 8}, kw[3] = {1, 5, 8}, ke[3] = {3, 6, 7};
  const int box = t.nbr[1][1];
  const int irow = t.x0, icol = a.tx + t.y;                      // the thread's granule indices
  const long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
    if (t.on_s) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < V; ++j) resident_recv(resident_granule(a, box, parity, ks[i], irow + j), want, o[ks[i]][j], ok);
    }
    if (t.on_n) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < V; ++j) resident_recv(resident_granule(a, box, parity, kn[i], irow + j), want, o[kn[i]][j], ok);
    }
    if (t.on_w) {     // plane 5 on the south row and plane 8 on the north row came with the row parts
      resident_recv(resident_granule(a, box, parity, kw[0], icol), want, o[1][0], ok);
      if (!t.on_s) resident_recv(resident_granule(a, box, parity, kw[1], icol), want, o[5][0], ok);
      if (!t.on_n) resident_recv(resident_granule(a, box, parity, kw[2], icol), want, o[8][0], ok);
    }
    if (t.on_e) {
      resident_recv(resident_granule(a, box, parity, ke[0], icol), want, o[3][V - 1], ok);
      if (!t.on_s) resident_recv(resident_granule(a, box, parity, ke[1], icol), want, o[6][V - 1], ok);
      if (!t.on_n) resident_recv(resident_granule(a, box, parity, ke[2], icol), want, o[7][V - 1], ok);
    }
    if (ok || !WAIT) return true;
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > kResidentTimeoutTicks ||
        __hip_atomic_load((gu32*)a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      __hip_atomic_store((gu32*)a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}

// gridDim.x = ntx * nty tiles, every one resident at once (host: at most one per CU);
// blockDim.x = (tx / V) * ty rounded up to whole waves.
template <int V, int MODE>
__global__ __launch_bounds__(1024) void lbm_resident(const ResidentArgs a) {
  constexpr bool FAST = (MODE & kFastMath) != 0;
  using RG = Row<V, false, false>;
  // ONE shared array (a second __shared__ object can de-pipeline the loop: cdna_hip_programming.md §5 trap 4a)
  __shared__ __attribute__((aligned(16))) float lds[kResidentLdsFloats];
  float* red = lds + 9 * kResidentMaxCells;                  // [2][16] per-wave speed sums
  uint32_t* lds_abort = reinterpret_cast<uint32_t*>(lds + 9 * kResidentMaxCells + 32);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
  const int nt = gridDim.x;
  // XCD-aware placement (speed only): block ids are dealt round-robin over the 8 XCDs; give each
  // XCD a contiguous run of tiles so that most neighbours share an L2
  int tile = blockIdx.x;
  if ((nt & 7) == 0) tile = (tile & 7) * (nt >> 3) + (tile >> 3);
  ResidentThread t;
  t.by = tile / a.ntx; t.bx = tile - t.by * a.ntx;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      int nx_ = t.bx + dx, ny_ = t.by + dy;
      nx_ += (nx_ < 0) ? a.ntx : 0; nx_ -= (nx_ >= a.ntx) ? a.ntx : 0;
      ny_ += (ny_ < 0) ? a.nty : 0; ny_ -= (ny_ >= a.nty) ? a.nty : 0;
      t.nbr[dy + 1][dx + 1] = ny_ * a.ntx + nx_;
    }
  const int lanes_x = a.tx / V;
  const int ry = tid / lanes_x;
  t.live = ry < a.ty;
  t.x0 = (tid - ry * lanes_x) * V;
  t.y = t.live ? ry : 0;
  t.on_w = t.x0 == 0; t.on_e = t.x0 + V == a.tx; t.on_s = t.y == 0; t.on_n = t.y == a.ty - 1;
  const int gx0 = t.bx * a.tx + t.x0, gy = t.by * a.ty + t.y;
  const long grow = (long)gy * a.pitch;
  if (tid == 0) *lds_abort = 0u;

  // ---- state 0: own cells from the source lattice; accelerate phase of the first step
  float o[9][V];
  bool blk[V];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int v = 0; v < V; ++v) o[k][v] = 1.f;
#pragma unroll
  for (int v = 0; v < V; ++v) blk[v] = true;
  if (t.live) {
#pragma unroll
    for (int k = 0; k < 9; ++k) RG::ld(a.src + k * a.plane + grow, gx0, o[k]);
    if constexpr (V == 4) {
      const uint32_t m = *reinterpret_cast<const uint32_t*>(a.blocked + grow + gx0);
      blk[0] = (m & 0xffu) != 0; blk[1] = (m & 0xff00u) != 0; blk[2] = (m & 0xff0000u) != 0; blk[3] = (m & 0xff000000u) != 0;
    } else if constexpr (V == 2) {
      const uint16_t m = *reinterpret_cast<const uint16_t*>(a.blocked + grow + gx0);
      blk[0] = (m & 0xffu) != 0; blk[1] = (m & 0xff00u) != 0;
    } else {
      blk[0] = a.blocked[grow + gx0] != 0;
    }
  }
  const bool accel_here = t.live && (gy == a.accel_row);
  if (accel_here) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      float p[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) p[k] = o[k][v];
      accelerate_cell(p, blk[v], a.a1, a.a2);
#pragma unroll
      for (int k = 0; k < 9; ++k) o[k][v] = p[k];
    }
  }
  resident_push<V, (MODE & kResDebugNoSend) == 0>(a, t, lds, o, a.tag0, 0);

  // ---- the step loop
  bool aborted = false;
  for (int s = 1; s <= a.nsteps; ++s) {
    __syncthreads();                                   // pushes of state s-1 are in LDS
    if (s > 1 && tid < 64) {                           // speed sum of step s-1 (per-wave sums -> tile sum)
      float v = (lane < nw) ? red[((s - 1) & 1) * 16 + lane] : 0.f;
      v = wave_sum(v);
      if (tid == 0) a.partials[(long)(s - 2) * nt + tile] = v;
    }
    // (the parity goes through an empty asm so that the mailbox addresses of BOTH parities are not
    // hoisted out of the loop and kept in registers: that spilled a hundred VGPRs)
    int par_in = (s - 1) & 1, par_out = s & 1;
    asm volatile("" : "+s"(par_in), "+s"(par_out));
    if (!resident_pull<V, (MODE & kResDebugNoWait) == 0>(a, t, lds, o, a.tag0 + (uint32_t)(s - 1), par_in)) *lds_abort = 1u;
    __syncthreads();                                   // every pull done before LDS is overwritten
    if (*lds_abort != 0u) { aborted = true; break; }   // (uniform: read after the barrier)
    const bool last = (s == a.nsteps);
    float sp = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      float p[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) p[k] = o[k][v];
      const float c = collide_cell<FAST>(p, blk[v], a.omega);
      sp += t.live ? c : 0.f;
      if (accel_here && !last) accelerate_cell(p, blk[v], a.a1, a.a2);
#pragma unroll
      for (int k = 0; k < 9; ++k) o[k][v] = p[k];
    }
    sp = wave_sum(sp);
    if (lane == 0) red[(s & 1) * 16 + wave] = sp;
    if (!last) {
      resident_push<V, (MODE & kResDebugNoSend) == 0>(a, t, lds, o, a.tag0 + (uint32_t)s, par_out);
    } else if (t.live) {
#pragma unroll
      for (int k = 0; k < 9; ++k) RG::st(a.dst + k * a.plane + grow, gx0, o[k]);
    }
  }
  __syncthreads();
  if (!aborted && tid < 64) {
    float v = (lane < nw) ? red[(a.nsteps & 1) * 16 + lane] : 0.f;
    v = wave_sum(v);
    if (tid == 0) a.partials[(long)(a.nsteps - 1) * nt + tile] = v;
  }
}

// Per-step tile sums -> per-step lattice sums (double, fixed order): one wave per step.  `sums` and
// `abort_out` may be pinned host memory; the abort word the tiles watched travels with the sums.
__global__ __launch_bounds__(kBlock) void lbm_fold_steps(const float* partials, int ntiles, int nsteps, double* sums,
                                                         const uint32_t* abort_word, uint32_t* abort_out) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && abort_out != nullptr) *abort_out = *abort_word;
  const int step = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (step >= nsteps) return;
  const int lane = threadIdx.x & 63;
  const float* p = partials + (long)step * ntiles;
  double s = 0.0;
  for (int i = lane; i < ntiles; i += 64) s += (double)p[i];
  s = wave_sum(s);
  if (lane == 0) sums[step] = s;
}

}  // namespace lbm
