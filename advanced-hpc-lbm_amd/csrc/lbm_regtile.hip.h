// lbm_regtile.hip.h -- the whole step loop in ONE launch with the lattice resident in REGISTERS
// (the first form of this engine kept the tiles in LDS and spent its time on 770 eight-byte stores per tile and
// step behind __syncthreads' vmcnt(0); it was removed in round 3).
//
// One block per CU holds a tile of 64 columns x (NW x R) rows: wave w owns rows w R .. w R + R - 1, a lane
// owns one column of them, 9 R registers.  One step of one wave (reference step:
// /root/reference/d2q9-bgk.c:228-1813; per-cell arithmetic = collide_cell / accelerate_cell, bit-identical
// to lbm_sweep):
//   east / west neighbours   the neighbouring lanes: one whole-wave DPP shift per plane (wave_shr / wave_shl)
//   north / south neighbours  the lane's own other rows; for the band's first and last row the neighbouring
//                             wave's edge row, which it wrote to LDS at the end of the previous step
//                             (3 planes x 64 floats each way, double-buffered by step parity:
//                             ONE raw s_barrier per step, lgkmcnt only)
//   across the tile border    16-byte {three populations, tag} granules in the neighbours' mailboxes in
//                             global memory (MI355X_MICROARCH.md, Valid forms, R2: the data is the flag --
//                             one aligned sc1 store, sc1 loads, no fence), polled until the tag says
//                             "state s".  A ROW of the tile's east / west column is one granule (lane 63 /
//                             lane 0 stores the three populations that leave the tile there; lanes 0,1,2 /
//                             63,62,61 fetch the granules of rows rho, rho-1, rho+1 and two DPP row shifts
//                             bring the diagonal ones to the edge lane); a COLUMN of the tile's bottom /
//                             top row is one granule per lane (waves 0 / NW-1).  One store and one load
//                             instruction per row -- a memory instruction costs its issue slot whether one
//                             lane is active or sixty-four -- through a buffer descriptor (32-bit offsets);
//                             nothing in the loop waits for a store (no vmcnt wait, no __syncthreads).
// A hand-off through memory takes about 2000 cycles (a write-through store, then an sc1 load that misses L2 by
// design), a fifth of a step of a full tile, and the first form of this loop -- mail sent after the last row, polled
// before the first -- waited for all of it.  Now the mail travels ROW BY ROW: a row's mail is fetched while the row
// before it is computed, and a row's granules leave right after the wait for the next row's mail (not before it: a
// store behind a branch would make that wait a wait for the store's acknowledgement, see the note at ew_voff).  What
// makes that hide most of the hand-off is the order of the rows: even waves sweep their rows upwards, odd waves
// downwards (when the tile has an even number of waves), so that rows which are neighbours in space -- across waves
// and across tiles -- are at most one row apart in time and every granule is two row-times old when it is asked for;
// and s_setprio falling with every finished row, so that the four waves of a SIMD take turns row by row instead of
// oldest first.  One-row waves (small tiles) have nothing to pipeline: they poll behind the barrier, three loads in flight.
// The update is in place, with three saved registers (the old planes 2,5,6 -- going down: 4,7,8 -- of the
// row just overwritten).  Mailbox of a tile, per parity: Sin / Nin [64 columns] (from the tile below /
// above), Win / Ein [TY + 2 rows] (rows -1 .. TY: the two extra rows are the diagonal tiles' corner rows).
// Two parities suffice: a tile sends state s+1 of a row only after pulling state s of the neighbouring rows from ALL the
// tiles that row touches, and a neighbour sent ITS state s only after it had pulled this tile's state s-1 -- the thread that consumes a granule owns
// the cell that produces the opposite one (a row's granule feeds rows rho-1, rho, rho+1 of the neighbour,
// and the row is not recomputed before all three have been).
// Every wait is bounded and watches a global abort word; a run that gives up leaves the source lattice
// untouched and the host repeats it with the streaming kernels.
#pragma once
#include <type_traits>
#include "lbm_kernels.hip.h"

namespace lbm {

constexpr long long kResidentTimeoutTicks = 100000000LL;       // a wait for mail gives up after 1 s of the 100 MHz wall clock
// timing experiments only (wrong results): LBM_RESIDENT_DEBUG=1 never waits for a tag, 2 also sends nothing
constexpr int kResDebugNoWait = 64, kResDebugNoSend = 128;
constexpr int kRegAsync = 4096;   // lbm_regtile: mail loads / stores of the loop as inline asm with counted s_waitcnt vmcnt(N)
constexpr int kRegSlab = 8192;    // lbm_regtile: the lattice is a slab with neighbours -- the tile rows below its first and above its
                                  // last belong to OTHER slabs (same tiling), whose mailboxes live in their own mail areas
typedef __attribute__((address_space(1))) unsigned int gu32;

// Per-step sums of a whole-run launch: partials[step][tile] -> sums[step] (double, fixed order), one wave per step; also
// hands the kernel's abort word to the host.
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

// one process per GPU: the abort word of a run as one more double behind the per-step sums (the all-reduce that ends the
// run then tells every rank whether ANY rank gave up)
__global__ void lbm_abort_to_sum(const uint32_t* abort_word, double* out) {
  if (threadIdx.x == 0) *out = (*abort_word != 0u) ? 1.0 : 0.0;
}

struct RegTileArgs {
  const float* src; float* dst; long plane; int pitch, nx, ny;
  const uint8_t* blocked;
  Relax omega; int accel_row; float a1, a2;
  int ty;                      // rows per tile = waves per block x R
  int ntx, nty;                // tiles per lattice row (nx / 64) / column (ny / ty); gridDim.x = ntx * nty
  int nsteps;
  uint32_t tag0;               // tag of state 0 of this run; state s carries tag0 + s
  void* mail;                  // [tile][parity][regtile_box(ty) bytes]
  unsigned mail_bytes;
  float* partials;             // [nsteps][ntiles]
  uint32_t* abort_word;
  int fault;                   // test hook: tile 0 never starts (its neighbours time out, the host falls back)
  unsigned long long* stats;   // development: [0] += waits that found their mail missing, [1] += extra fetches (or nullptr)
  // ---- a slab with neighbours (MODE & kRegSlab; launched through lbm_regtile_slabs): `ny` rows and `nty` tile rows are this
  // slab's; the granules that leave through its bottom / top edge go into the mail area of the slab to the south / north
  // (same process: its pointer, with peer access when it lives on another GPU; other process: a hipIpc mapping), whose
  // tile rows are numbered 0 .. nty_s-1 / nty_n-1 with the same 64-column x ty-row tiles.  What ARRIVES is always in `mail`.
  void* mail_s; void* mail_n;
  unsigned mail_bytes_s, mail_bytes_n;
  int nty_s, nty_n;
};

// LDS bytes of a block of nw waves with r rows per wave (see the kernel)
__host__ __device__ constexpr int regtile_lds_bytes(int nw, int r) { return 4 * (nw * (2 * 6 * 64 + (r == 4 ? r * 3 * 64 : 0)) + 2 * 16 + 16); }

// bytes of one mailbox (one tile, one parity): Sin[64], Nin[64], Win[ty+2], Ein[ty+2] granules of 16 bytes
__host__ __device__ constexpr int regtile_box(int ty) { return 16 * (2 * 64 + 2 * (ty + 2)); }

typedef unsigned int rt_u4 __attribute__((ext_vector_type(4)));

// lane i <- lane i-1 (wave_shr:1) / lane i+1 (wave_shl:1); the lane with no source keeps `edge`
__device__ __forceinline__ float rt_west(float edge, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float rt_east(float edge, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
// lane i <- lane i+N / i-N inside its row of 16 lanes (row_shl:N / row_shr:N)
template <int N> __device__ __forceinline__ float rt_row_shl(unsigned v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xf, 0xf, false));
}
template <int N> __device__ __forceinline__ float rt_row_shr(unsigned v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, false));
}

// blockDim.x = 64 NW; R rows per wave.
template <int R, int MODE>
__device__ __forceinline__ void regtile_body(const RegTileArgs& a) {
  constexpr bool FAST = (MODE & kFastMath) != 0;
  constexpr bool SLAB = (MODE & kRegSlab) != 0;
  // timing experiments only (wrong results), LBM_RESIDENT_DEBUG: 1 = one pass over the inbox, no waiting; 2 = also no
  // stores to other tiles; 3 = also no inbox loads at all; 4 = like 1, the stores issued but dropped by an empty buffer
  // descriptor (what the instructions cost without their memory traffic); 5 = like 1, stores without sc1
  constexpr bool DBG_NOWAIT = (MODE & kResDebugNoWait) != 0, DBG_NOSEND = (MODE & kResDebugNoSend) != 0, DBG_NOLOAD = (MODE & 256) != 0;
  constexpr bool DBG_DROP = (MODE & 512) != 0, DBG_PLAIN = (MODE & 1024) != 0;
  constexpr bool TRACE = (MODE & 2048) != 0;     // development: time stamps of one tile's waves (LBM_REGTILE_TRACE)
  // The mail of the loop issued and waited for BY HAND (R > 1): see "the asynchronous loop" below
  constexpr bool ASYNC = (MODE & kRegAsync) != 0 && R > 1;
  static_assert(R == 1 || R == 2 || R == 4, "rows per wave");
  // LDS (dynamic, regtile_lds_bytes(nw, R)): edge rows between the waves of the tile [parity][wave][6][64]; the
  // populations no other row ever pulls from -- planes 0, 1, 3 of every row, [wave][R][3][64]: a row's own update is
  // the only reader and writer, so they wait in LDS instead of twelve registers the loop does not have; per-wave
  // speed sums [parity][16]; abort word
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nw_ = (int)(blockDim.x >> 6);
  float* red = lds + nw_ * (2 * 6 * 64 + (R == 4 ? R * 3 * 64 : 0));
  uint32_t* lds_abort = reinterpret_cast<uint32_t*>(red + 32);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)(blockDim.x >> 6);
  const int nt = gridDim.x;
  int tile = blockIdx.x;
  if ((nt & 7) == 0) tile = (tile & 7) * (nt >> 3) + (tile >> 3);      // neighbouring tiles mostly share an XCD (speed only)
  const int by = tile / a.ntx, bx = tile - by * a.ntx;
  const int TY = a.ty;
  const unsigned BOX = (unsigned)regtile_box(TY);
  // (a slab: the row below tile row 0 is the LAST tile row of the slab to the south, numbered in that slab's mail area; the
  // row above the last is row 0 of the slab to the north -- see `south_out` / `north_out` for where such granules are sent)
  auto tile_of = [&](int dx, int dy) {
    int x = bx + dx, y = by + dy;
    x += (x < 0) ? a.ntx : 0; x -= (x >= a.ntx) ? a.ntx : 0;
    y += (y < 0) ? (SLAB ? a.nty_s : a.nty) : 0; y -= (y >= a.nty) ? a.nty : 0;
    return y * a.ntx + x;
  };
  // mailbox sections (byte offsets inside a box); the whole mail area is a few MB: 32-bit offsets behind one descriptor
  const unsigned oS = 0u, oN = 1024u, oW = 2048u, oE = 2048u + 16u * (unsigned)(TY + 2);
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.mail, 0, (int)a.mail_bytes, 0x00020000);
  const auto rsrc_st = DBG_DROP ? __builtin_amdgcn_make_buffer_rsrc(a.mail, 0, 0, 0x00020000) : rsrc;
  // where the granules that leave through the tile's bottom / top edge (and the two corners on that side) are stored: this
  // slab's mail area, or the neighbouring slab's for the slab's first / last tile row (wave-uniform, fixed for the run)
  const bool south_out = SLAB && by == 0, north_out = SLAB && by == a.nty - 1;
  const auto rsrc_s = south_out ? __builtin_amdgcn_make_buffer_rsrc(a.mail_s, 0, (int)a.mail_bytes_s, 0x00020000) : rsrc_st;
  const auto rsrc_n = north_out ? __builtin_amdgcn_make_buffer_rsrc(a.mail_n, 0, (int)a.mail_bytes_n, 0x00020000) : rsrc_st;
  using ToOwn = std::integral_constant<int, 0>;
  using ToSouth = std::integral_constant<int, 1>;
  using ToNorth = std::integral_constant<int, 2>;
  auto box = [&](int t) { return (unsigned)(t * 2) * BOX; };            // parity 0; parity 1: + BOX
  // (per-lane offset in a register, wave-uniform offset in the instruction's scalar operand: one register per KIND of access)
  // (granules for another slab: sc0 sc1 = system scope, the neighbour may live on another GPU)
  auto send = [&](auto to, unsigned voff, unsigned soff, float v0, float v1, float v2, uint32_t tag) {
    constexpr int TO = decltype(to)::value;
    rt_u4 g;
    g.x = __float_as_uint(v0); g.y = __float_as_uint(v1); g.z = __float_as_uint(v2); g.w = tag;
    if (DBG_NOSEND) return;
    if constexpr (!SLAB || TO == 0) __builtin_amdgcn_raw_buffer_store_b128(g, rsrc_st, voff, soff, DBG_PLAIN ? 0 : 16);      // aux 16 = sc1
    else __builtin_amdgcn_raw_buffer_store_b128(g, TO == 1 ? rsrc_s : rsrc_n, voff, soff, 17);
  };
  auto load = [&](unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 16); };
  const int tS = tile_of(0, -1), tN = tile_of(0, 1), tW = tile_of(-1, 0), tE = tile_of(1, 0);
  const int tSW = tile_of(-1, -1), tSE = tile_of(1, -1), tNW = tile_of(-1, 1), tNE = tile_of(1, 1);
  const bool first = (w == 0), last = (w == nw - 1);
  const bool down = ((nw & 1) == 0) && ((w & 1) != 0);      // this wave sweeps its rows downwards
  const int rho0 = w * R;                                   // tile row of this wave's first row
  const int gx = bx * 64 + lane, gy0 = by * TY + rho0;
  if (tid == 0) *lds_abort = 0u;
  // planes 0, 1, 3 of row r: in LDS where the registers are short (four rows per wave), in f otherwise
  constexpr bool OWN_LDS = (R == 4);
  float* own = lds + nw * (2 * 6 * 64) + w * (R * 3 * 64) + lane;   // own[(r * 3 + j) * 64]: plane {0,1,3}[j] of row r

  // east / west mail: lane 63 stores its row into the west inbox of the tile to the east, lane 0 into the east inbox of
  // the tile to the west (inbox row index = tile row + 1); lanes 0,1,2 fetch rows rho, rho-1, rho+1 of this tile's west
  // inbox, lanes 63,62,61 of its east inbox.  Offsets for row 0 of the wave, parity 0.
  const bool edge_lane = (lane == 0) || (lane == 63);
  const bool courier = (lane < 3) || (lane > 60);
  // The mail LOADS of the loop are issued by every lane of every wave, unconditionally: the lanes (and waves) a load
  // does not concern carry an offset beyond the end of the buffer, and the descriptor's range check answers them
  // with zeros.  In THIS (compiler-scheduled) loop the STORES stay behind their branches (lanes 0 and 63, the tile's first
  // and last wave): one build of it with every store masked the same way hung on multi-wave tiles and the next build,
  // different only by diagnostic code, passed -- a property of one instruction schedule, not of masked stores (the
  // asynchronous loop below uses nothing else; tools/oob_store_order probes the hardware).  A branch with a
  // memory operation in it makes the compiler wait for ALL outstanding memory operations (s_waitcnt vmcnt(0)) the
  // next time a loaded value is used behind it, acknowledgements of the sc1 stores included (~1500 cycles) -- so
  // a row's granules are sent only once the NEXT row's mail has been waited for (do_row): no load is ever in flight
  // across the stores, and the stores have a whole row's arithmetic to complete in.
  constexpr unsigned OOB = 0x80000000u;
  const unsigned lane16 = 16u * (unsigned)lane;
  const unsigned ew_voff = ((lane == 0) ? box(tW) + oE : box(tE) + oW) + 16u * (unsigned)(rho0 + 1);   // (lanes 0 and 63)
  unsigned rv_voff = OOB;
  if (courier) {
    const int d = (lane < 3) ? lane : 63 - lane;            // 0: row rho, 1: rho-1, 2: rho+1
    rv_voff = ((lane < 3) ? oW : oE) + 16u * (unsigned)(rho0 + 1 + (d == 1 ? -1 : d == 2 ? 1 : 0));   // (+ box(tile): scalar)
  }
  const unsigned first_voff = first ? lane16 : OOB, last_voff = last ? lane16 : OOB;     // the tile's bottom / top row
  const unsigned cs_voff = ((lane == 0) ? box(tSW) + oE : box(tSE) + oW) + 16u * (unsigned)(TY + 1);    // its corners (lanes 0 and 63)
  const unsigned cn_voff = (lane == 0) ? box(tNW) + oE : box(tNE) + oW;
  const unsigned mybox = __builtin_amdgcn_readfirstlane(box(tile));

  // ---- state 0
  float f[R][9];               // (planes 0, 1, 3 live in `own`: those elements are never used)
  bool blk[R];

  // what a row hands to the neighbouring tiles, from the state in f (pb = parity x BOX)
  auto send_row = [&](auto rc, uint32_t tag, unsigned pb, const float (&q)[9]) {   // q = the row's nine populations
    constexpr int r = decltype(rc)::value;
    const float e1 = q[1], e5 = q[5], e8 = q[8], w3 = q[3], w6 = q[6], w7 = q[7];
    const bool wl = lane == 0;
    const float v0 = wl ? w3 : e1, v1 = wl ? w6 : e5, v2 = wl ? w7 : e8;
    if (r == 0 && first)             // the tile's bottom row enters the tile below through ITS north inbox
      send(ToSouth{}, lane16, box(tS) + oN + pb, q[4], q[7], q[8], tag);
    if (r == R - 1 && last)
      send(ToNorth{}, lane16, box(tN) + oS + pb, q[2], q[5], q[6], tag);
    if (edge_lane) send(ToOwn{}, ew_voff, pb + 16u * r, v0, v1, v2, tag);
    // the tile's corners: the same granule is row -1 / row TY of the diagonal tile's inbox
    if (r == R - 1 && last && edge_lane) send(ToNorth{}, cn_voff, pb, v0, v1, v2, tag);
    if (r == 0 && first && edge_lane) send(ToSouth{}, cs_voff, pb, v0, v1, v2, tag);
  };
  // the edge rows of the state in f for the neighbouring waves: bottom row's planes 4,7,8, top row's 2,5,6
  auto publish_lds = [&](int parity) {
    float* me = lds + ((parity * nw + w) * 6) * 64 + lane;
    me[0 * 64] = f[0][4]; me[1 * 64] = f[0][7]; me[2 * 64] = f[0][8];
    me[3 * 64] = f[R - 1][2]; me[4 * 64] = f[R - 1][5]; me[5 * 64] = f[R - 1][6];
  };

  // the mail of one row: the couriers' granules (g); for the tile's bottom row the row below it (e), for its top row
  // the row above (e -- or x where one row is both: R = 1 in a tile of one wave)
  struct Mail { rt_u4 g, e, x; };
  auto fetch = [&](auto rc, unsigned pb, Mail& m) {
    constexpr int r = decltype(rc)::value;
    if (DBG_NOLOAD) return;
    m.g = load(rv_voff, mybox + pb + 16u * r);
    if (r == 0) m.e = load(first_voff, mybox + oS + pb);
    if (r == R - 1) {
      if constexpr (R == 1) m.x = load(last_voff, mybox + oN + pb);
      else m.e = load(last_voff, mybox + oN + pb);
    }
  };
  auto arrived = [&](auto rc, const Mail& m, uint32_t want) {
    constexpr int r = decltype(rc)::value;
    bool ok = true;
    if (courier) ok = m.g.w == want;
    if ((r == 0 && first) || (R > 1 && r == R - 1 && last)) ok = ok && m.e.w == want;
    if (R == 1 && last) ok = ok && m.x.w == want;
    return __all(ok) != 0;
  };
  unsigned nmiss = 0u, nspin = 0u;
  // development trace (a.stats != nullptr): shader-clock stamps of the waves of one tile over four steps,
  // stats[2 + ((wave * 4 + step - s0) * 16 + slot)]
  const bool tracing = TRACE && a.stats != nullptr && tile == (int)a.stats[2];
  const int trace_s0 = (TRACE && a.stats != nullptr) ? (int)a.stats[3] : 0;
  auto stamp = [&](int s, int slot) {
    if constexpr (TRACE) {
      if (tracing && s >= trace_s0 && s < trace_s0 + 4 && lane == 0)
        a.stats[4 + ((w * 4 + (s - trace_s0)) * 16 + slot)] = __builtin_amdgcn_s_memtime();
    }
  };
  // Wait until the mail of a row is there.  With several rows per wave the fetch that was started a row earlier
  // usually has it, and if not it is fetched again.  A one-row wave (the small decks: the step IS the hand-off) polls
  // with THREE loads in flight, a hundred cycles apart, each tested as it returns: the hand-off is seen one load
  // latency (~2000 cycles through memory) after the granule became visible, not up to two.
  auto await = [&](auto rc, unsigned pb, uint32_t want, Mail& m) {
    if (DBG_NOLOAD || DBG_NOWAIT || arrived(rc, m, want)) return;
    const long long t0 = wall_clock64();
    ++nmiss;
    for (;;) {
      if constexpr (R == 1) {
        Mail p1, p2;
        fetch(rc, pb, m);
        __builtin_amdgcn_s_sleep(2);
        fetch(rc, pb, p1);
        __builtin_amdgcn_s_sleep(2);
        fetch(rc, pb, p2);
        nspin += 3u;
        if (arrived(rc, m, want)) return;
        if (arrived(rc, p1, want)) { m = p1; return; }
        if (arrived(rc, p2, want)) { m = p2; return; }
      } else {
        __builtin_amdgcn_s_sleep(1);
        ++nspin;
        fetch(rc, pb, m);
        if (arrived(rc, m, want)) return;
      }
      if (wall_clock64() - t0 > kResidentTimeoutTicks ||
          __hip_atomic_load((gu32*)a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        // development (LBM_REGTILE_STATS): the FIRST wave that gives up (not one that follows the abort word) says where it
        // stood -- stats[1028 + 0..]: 1 + tile, wave, row, tag wanted, then per lane the tags it holds (couriers' granule; the
        // row below / above the tile)
        if (a.stats != nullptr && wall_clock64() - t0 > kResidentTimeoutTicks) {
          unsigned long long* d = a.stats + 1028;
          unsigned long long claimed = 0ull;
          if (lane == 0) claimed = atomicCAS(d, 0ull, (unsigned long long)(tile + 1));
          claimed = __shfl(claimed, 0, 64);
          if (claimed == 0ull) {
            if (lane == 0) { d[1] = (unsigned long long)w; d[2] = (unsigned long long)decltype(rc)::value; d[3] = want; }
            d[4 + lane] = ((unsigned long long)m.g.w << 32) | (unsigned long long)m.e.w;
          }
        }
        __hip_atomic_store((gu32*)a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lds_abort = 1u;      // (this wave carries on with what it has; everybody leaves together behind the next barrier)
        return;
      }
    }
  };
  auto blank = [&](Mail& m) { m.g = rt_u4{0u, 0u, 0u, 0u}; m.e = m.g; m.x = m.g; };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1 % R>;
  using I2 = std::integral_constant<int, 2 % R>;
  using I3 = std::integral_constant<int, 3 % R>;
  using ILast = std::integral_constant<int, R - 1>;

  if (a.fault != 0 && tile == 0) return;
  // ---- state 0: loaded, and sent row by row like every other state
  // (every row's nine loads are issued before any row is used: with a row's sends between them the loads of the next row
  // wait behind the compiler's drain for this row's -- R round trips to memory instead of one, 2-3 us each, paid per RUN)
  float q0[R][9];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long o = (long)(gy0 + r) * a.pitch + gx;
#pragma unroll
    for (int k = 0; k < 9; ++k) q0[r][k] = __builtin_nontemporal_load(&a.src[k * a.plane + o]);
    blk[r] = a.blocked[o] != 0;
  }
  auto first_state = [&](auto rc) {
    constexpr int r = decltype(rc)::value;
    float q[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) q[k] = q0[r][k];
    if (gy0 + r == a.accel_row) accelerate_cell(q, blk[r], a.a1, a.a2);   // accelerate phase of the first step
    f[r][2] = q[2]; f[r][4] = q[4]; f[r][5] = q[5]; f[r][6] = q[6]; f[r][7] = q[7]; f[r][8] = q[8];
    if constexpr (OWN_LDS) { own[(r * 3 + 0) * 64] = q[0]; own[(r * 3 + 1) * 64] = q[1]; own[(r * 3 + 2) * 64] = q[3]; }
    else { f[r][0] = q[0]; f[r][1] = q[1]; f[r][3] = q[3]; }
    send_row(rc, a.tag0, 0u, q);
  };
  first_state(I0{});
  if constexpr (R > 1) first_state(I1{});
  if constexpr (R > 2) { first_state(I2{}); first_state(I3{}); }
  publish_lds(0);

  // ============================================================================================================
  // The asynchronous loop (R > 1, MODE & kRegAsync).  The loop above this point's successor -- mail fetched one row
  // ahead, a row's granules sent only behind the wait for the next row's mail -- spends a step on four load latencies:
  // a hand-off takes ~2000 cycles, a row's arithmetic ~1000, and hipcc turns every wait for a load behind a branch with
  // a store in it into s_waitcnt vmcnt(0), store acknowledgements (~1500 cycles) included.  Here the loop's mail traffic
  // is inline asm the compiler neither counts nor waits for (cdna_hip_programming.md 5.7, form (ii)):
  //   * every wave issues THE SAME vector-memory operations per row, unconditionally -- lanes and waves an operation does
  //     not concern carry an offset beyond the descriptor's range (loads answer zeros, stores are dropped; such operations
  //     keep their place in the wave's vmcnt queue: tools/oob_store_order, measured) -- so the number of operations
  //     younger than a given fetch is a compile-time constant and the wait for it is a COUNTED s_waitcnt vmcnt(N);
  //   * a row's granules leave right behind its arithmetic, its mail is requested D = R/2 rows ahead: the granule is
  //     R - D row-times old when it is asked for and has D row-times to arrive;
  //   * per row i (in the wave's sweep order):  wait(mail i) -> check tags (else: fetch again, drain, bounded) -> unpack
  //     -> fetch(mail i + D) -> arithmetic -> stores of row i.  Operations younger than fetch(i) at wait(i):
  //     the loads of fetch(i+1 .. i+D-1) and the stores of rows i-D .. i-1:  N(i) = sum L(i+k), k < D, + sum S(i-k), k <= D,
  //     with L = 2 loads for a band's first / last row (couriers + the row below / above the tile), 1 otherwise, and
  //     S = 3 stores for those rows (east/west granule, north/south granules, corner granule), 1 otherwise.
  //     The first step waits with vmcnt(0) (its queue holds the prologue, not the steady pattern); the last step still
  //     fetches "the next step's" mail (unused), so that its own waits see the steady pattern; unknown extra operations
  //     (the step sums' store of wave 0) only make a counted wait stricter.  A miss drains the queue (vmcnt(0)), after
  //     which counted waits are trivially satisfied until the pattern has refilled.
  if constexpr (ASYNC) {
    // Where the mail lands: in ordinary registers the loads name as asm outputs ("=v") and a wait statement takes as
    // read-write operands ("+v") before their first use -- which pins the ORDER of issue, wait and use, not the registers:
    // hipcc may still copy such a value while it is in flight (it did: with the two sweep directions as branches inside ONE
    // step loop it merged their slots with v_mov at the join, of registers whose load had not landed).  So each direction
    // gets a step loop of its own (no join inside the loop), and the build is AUDITED: tools/audit_regtile_isa.py walks the
    // ISA from every asm load to the wait that retires it and fails on any instruction in between that touches the
    // destination registers (run by tests/test_abi.py).  (Accumulator registers named in the asm text would be out of the
    // compiler's reach, but a kernel that uses them gets the register file split 64 / 64.)
    // rows a row's mail is requested ahead.  R / 2 looked right on paper (the granule R/2 row-times old when asked for, R/2
    // row-times to arrive) and lost: a third of the early requests found yesterday's tag (1.28 misses per wave and step at
    // R = 4 against 0.26 one row ahead -- neighbouring tiles are not in lockstep to a row) and a miss costs a drain and a
    // second round trip: 1024x1024 5.24 us per step against 4.15 (both in one call, profiles/r03_regtile_async.log)
    constexpr int D = 1;
    struct Slot { rt_u4 g, e; };
    const unsigned boxS = __builtin_amdgcn_readfirstlane(box(tS) + oN), boxN = __builtin_amdgcn_readfirstlane(box(tN) + oS);
    const unsigned ew_m = edge_lane ? ew_voff : OOB;                       // east / west granule: lanes 0 and 63
    const unsigned cs_m = (first && edge_lane) ? cs_voff : OOB, cn_m = (last && edge_lane) ? cn_voff : OOB;   // corners
    auto aload = [&](rt_u4& dst, unsigned voff, unsigned soff) {
      const auto rs = rsrc;
      // (no s_nop in front: descriptor and scalar offset come from scalar-ALU code, never fresh from a vector instruction --
      // tools/audit_regtile_isa.py checks the five instructions in front of every one of these for a VALU write to them)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    auto astore = [&](auto to, unsigned voff, unsigned soff, float v0, float v1, float v2, uint32_t tag) {
      constexpr int TO = decltype(to)::value;
      rt_u4 g;
      g.x = __float_as_uint(v0); g.y = __float_as_uint(v1); g.z = __float_as_uint(v2); g.w = tag;
      if constexpr (!SLAB || TO == 0) {
        const auto rs = rsrc;
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" :: "v"(g), "v"(voff), "s"(rs), "s"(soff) : "memory");
      } else {
        const auto rs = (TO == 1) ? rsrc_s : rsrc_n;
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc0 sc1\n\ts_nop 1" :: "v"(g), "v"(voff), "s"(rs), "s"(soff) : "memory");
      }
    };
    auto afetch = [&](auto rc, unsigned pbx, Slot& m) {
      constexpr int r = decltype(rc)::value;
      aload(m.g, rv_voff, mybox + pbx + 16u * r);
      if constexpr (r == 0) aload(m.e, first_voff, mybox + oS + pbx);
      else if constexpr (r == R - 1) aload(m.e, last_voff, mybox + oN + pbx);
    };
    // all but the N youngest operations are done: the slot's registers hold its granules ("retire" marks the statement for the audit)
    auto aretire = [&](auto rc, auto nc, Slot& m) {
      constexpr int r = decltype(rc)::value, N = decltype(nc)::value;
      if constexpr (r == 0 || r == R - 1) asm volatile("s_waitcnt vmcnt(%2) ; retire %0 %1" : "+v"(m.g), "+v"(m.e) : "n"(N));
      else asm volatile("s_waitcnt vmcnt(%1) ; retire %0" : "+v"(m.g) : "n"(N));
    };
    auto aarrived = [&](auto rc, const Slot& m, uint32_t want) {
      constexpr int r = decltype(rc)::value;
      bool ok = true;
      if (courier) ok = m.g.w == want;
      if ((r == 0 && first) || (r == R - 1 && last)) ok = ok && m.e.w == want;
      return __all(ok) != 0;
    };
    auto aslow = [&](auto rc, unsigned pbx, uint32_t want, Slot& m) {       // the mail was not there: fetch again, drained, bounded
      const long long t0 = wall_clock64();
      ++nmiss;
      for (;;) {
        __builtin_amdgcn_s_sleep(1);
        ++nspin;
        afetch(rc, pbx, m);
        aretire(rc, std::integral_constant<int, 0>{}, m);
        if (aarrived(rc, m, want)) return;
        if (wall_clock64() - t0 > kResidentTimeoutTicks ||
            __hip_atomic_load((gu32*)a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          __hip_atomic_store((gu32*)a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *lds_abort = 1u;
          return;
        }
      }
    };
    bool aborted = false;
    // the whole step loop for one sweep direction
    auto run = [&](auto up_c) {
      constexpr bool UP = decltype(up_c)::value;
      Slot slot[D];
#pragma unroll
      for (int d = 0; d < D; ++d) { slot[d].g = rt_u4{0u, 0u, 0u, 0u}; slot[d].e = slot[d].g; }
      // prologue: the mail of the first D rows of step 1 (state 0, parity 0)
      {
        unsigned pb0 = 0u;
        asm volatile("" : "+s"(pb0));
        afetch(std::integral_constant<int, UP ? 0 : R - 1>{}, pb0, slot[0]);
        if constexpr (D > 1) afetch(std::integral_constant<int, UP ? 1 : R - 2>{}, pb0, slot[1]);
      }
      for (int s = 1; s <= a.nsteps; ++s) {
        int par = (s - 1) & 1;
        asm volatile("" : "+s"(par));
        const unsigned pb = (unsigned)par * BOX, pbn = BOX - pb;
        const uint32_t want = a.tag0 + (uint32_t)(s - 1), tagn = want + 1u;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave's edge rows of state s-1 are in LDS
        if (*lds_abort != 0u) { aborted = true; break; }
        stamp(s, 0);
        if (s > 1 && tid < 64) {                       // speed sum of step s-1 (DPP adds: the LDS-permute form cost wave 0 ~1300 cycles at
          float v = (lane < nw) ? red[par * 16 + lane] : 0.f;    // the head of EVERY step, and the tile's barrier waits for its slowest wave)
          v = wave_sum_dpp(v);
          if (tid == 0) a.partials[(long)(s - 2) * nt + tile] = v;
        }
        const bool laststep = (s == a.nsteps), firststep = (s == 1);
        float sp = 0.f;
        float sv[3] = {0.f, 0.f, 0.f};               // going up: old planes 2,5,6 of the row just overwritten; going down: 4,7,8
        auto one = [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int r = UP ? i : R - 1 - i;
          using RC = std::integral_constant<int, r>;
          Slot& m = slot[i % D];
          __builtin_amdgcn_s_setprio(3 - (i * 4) / R);
          stamp(s, 1 + 3 * i);
          // ---- wait for this row's mail: all but the N(i) youngest operations are done
          constexpr int N = [] { int n = 0; for (int k = 1; k < D; ++k) { int j = (((i + k) % R) + R) % R; n += (j == 0 || j == R - 1) ? 2 : 1; }
                                  for (int k = 1; k <= D; ++k) { int j = (((i - k) % R) + R) % R; n += (j == 0 || j == R - 1) ? 3 : 1; } return n; }();
          // (ONE statement names the slot's registers on every path: two of them, one per branch, and hipcc gives each its own
          // registers and copies the in-flight slot from one set to the other at the loop's back edge)
          if (firststep) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          aretire(RC{}, std::integral_constant<int, N>{}, m);
          if (!aarrived(RC{}, m, want)) aslow(RC{}, pb, want, m);
          stamp(s, 2 + 3 * i);
          // ---- unpack: couriers -> edge lanes; the row below / above the tile
          const float m0 = __uint_as_float(m.g.x);
          const float m1w = rt_row_shl<1>(m.g.y), m1e = rt_row_shr<1>(m.g.y);
          const float m2w = rt_row_shl<2>(m.g.z), m2e = rt_row_shr<2>(m.g.z);
          float lo[3], hi[3], keep[3];
          if constexpr (UP) {
            constexpr int ra = (r < R - 1) ? r + 1 : r;
            lo[0] = sv[0]; lo[1] = sv[1]; lo[2] = sv[2];
            hi[0] = f[ra][4]; hi[1] = f[ra][7]; hi[2] = f[ra][8];
            keep[0] = f[r][2]; keep[1] = f[r][5]; keep[2] = f[r][6];
          } else {
            constexpr int rb = (r > 0) ? r - 1 : r;
            hi[0] = sv[0]; hi[1] = sv[1]; hi[2] = sv[2];
            lo[0] = f[rb][2]; lo[1] = f[rb][5]; lo[2] = f[rb][6];
            keep[0] = f[r][4]; keep[1] = f[r][7]; keep[2] = f[r][8];
          }
          if (r == 0) {
            if (first) { lo[0] = __uint_as_float(m.e.x); lo[1] = __uint_as_float(m.e.y); lo[2] = __uint_as_float(m.e.z); }
            else {
              const float* q = lds + ((par * nw + (w - 1)) * 6 + 3) * 64 + lane;
              lo[0] = q[0]; lo[1] = q[64]; lo[2] = q[128];
            }
          }
          if (r == R - 1) {
            if (last) { hi[0] = __uint_as_float(m.e.x); hi[1] = __uint_as_float(m.e.y); hi[2] = __uint_as_float(m.e.z); }
            else {
              const float* q = lds + ((par * nw + (w + 1)) * 6) * 64 + lane;
              hi[0] = q[0]; hi[1] = q[64]; hi[2] = q[128];
            }
          }
          // ---- the mail of the row D positions on (this step's, or the next step's first rows: always issued, see above)
          {
            constexpr int j = i + D;
            constexpr int rj = UP ? (j % R) : R - 1 - (j % R);
            if constexpr (j < R) afetch(std::integral_constant<int, rj>{}, pb, m);
            else afetch(std::integral_constant<int, rj>{}, pbn, m);
          }
          // ---- the row
          float p[9];
          float c0, c1, c3;
          if constexpr (OWN_LDS) { c0 = own[(r * 3 + 0) * 64]; c1 = own[(r * 3 + 1) * 64]; c3 = own[(r * 3 + 2) * 64]; }
          else { c0 = f[r][0]; c1 = f[r][1]; c3 = f[r][3]; }
          p[0] = c0;
          p[1] = rt_west(m0, c1);
          p[3] = rt_east(m0, c3);
          p[2] = lo[0];
          p[5] = rt_west(m1w, lo[1]);
          p[6] = rt_east(m1e, lo[2]);
          p[4] = hi[0];
          p[7] = rt_east(m2e, hi[1]);
          p[8] = rt_west(m2w, hi[2]);
          sp += collide_cell<FAST, true>(p, blk[r], a.omega);
          if (gy0 + r == a.accel_row && !laststep) accelerate_cell(p, blk[r], a.a1, a.a2);
          f[r][2] = p[2]; f[r][4] = p[4]; f[r][5] = p[5]; f[r][6] = p[6]; f[r][7] = p[7]; f[r][8] = p[8];
          if constexpr (OWN_LDS) { own[(r * 3 + 0) * 64] = p[0]; own[(r * 3 + 1) * 64] = p[1]; own[(r * 3 + 2) * 64] = p[3]; }
          else { f[r][0] = p[0]; f[r][1] = p[1]; f[r][3] = p[3]; }
          sv[0] = keep[0]; sv[1] = keep[1]; sv[2] = keep[2];
          // ---- its granules, at once (S(i) stores, the same in every wave)
          {
            const bool wl = lane == 0;
            const float v0 = wl ? p[3] : p[1], v1 = wl ? p[6] : p[5], v2 = wl ? p[7] : p[8];
            astore(ToOwn{}, ew_m, pbn + 16u * r, v0, v1, v2, tagn);
            if constexpr (r == 0) {
              astore(ToSouth{}, first_voff, boxS + pbn, p[4], p[7], p[8], tagn);
              astore(ToSouth{}, cs_m, pbn, v0, v1, v2, tagn);
            }
            if constexpr (r == R - 1) {
              astore(ToNorth{}, last_voff, boxN + pbn, p[2], p[5], p[6], tagn);
              astore(ToNorth{}, cn_m, pbn, v0, v1, v2, tagn);
            }
          }
          __builtin_amdgcn_sched_barrier(0);           // one row at a time
          stamp(s, 3 + 3 * i);
        };
        one(std::integral_constant<int, 0>{});
        one(std::integral_constant<int, 1>{});
        if constexpr (R > 2) { one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{}); }
        if (!laststep) publish_lds(s & 1);
        stamp(s, 13);
        sp = wave_sum_dpp(sp);
        if (lane == 0) red[(s & 1) * 16 + w] = sp;
      }
      // nothing of the loop's mail is in flight beyond this point (the slots' last fetches are never used: retire them)
#pragma unroll
      for (int d = 0; d < D; ++d) asm volatile("s_waitcnt vmcnt(0) ; retire %0 %1" : "+v"(slot[d].g), "+v"(slot[d].e));
    };
    if (!down) run(std::true_type{}); else run(std::false_type{});
    __syncthreads();
    if (a.stats != nullptr && lane == 0) { atomicAdd(a.stats, (unsigned long long)nmiss); atomicAdd(a.stats + 1, (unsigned long long)nspin); }
    if (aborted || *lds_abort != 0u) return;          // (the host repeats the run from the untouched source lattice)
    {
      int gyq = gy0, gxq = gx;
      asm volatile("" : "+v"(gyq), "+v"(gxq));
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const long o = (long)(gyq + r) * a.pitch + gxq;
        if constexpr (OWN_LDS) { __builtin_nontemporal_store(own[(r * 3 + 0) * 64], &a.dst[o]); __builtin_nontemporal_store(own[(r * 3 + 1) * 64], &a.dst[a.plane + o]); __builtin_nontemporal_store(own[(r * 3 + 2) * 64], &a.dst[3 * a.plane + o]); }
        else { __builtin_nontemporal_store(f[r][0], &a.dst[o]); __builtin_nontemporal_store(f[r][1], &a.dst[a.plane + o]); __builtin_nontemporal_store(f[r][3], &a.dst[3 * a.plane + o]); }
#pragma unroll
        for (int k = 2; k < 9; ++k)
          if (k != 3) __builtin_nontemporal_store(f[r][k], &a.dst[k * a.plane + o]);
      }
    }
    if (tid < 64) {
      float v = (lane < nw) ? red[(a.nsteps & 1) * 16 + lane] : 0.f;
      v = wave_sum(v);
      if (tid == 0) a.partials[(long)(a.nsteps - 1) * nt + tile] = v;
    }
    return;
  }

  Mail pre;                                        // the mail of a step's first row: in flight across the barrier
  blank(pre);
  {
    unsigned pb0 = 0u;
    asm volatile("" : "+s"(pb0));
    if constexpr (R > 1) { if (!down) fetch(I0{}, pb0, pre); else fetch(ILast{}, pb0, pre); }
  }

  bool aborted = false;
  for (int s = 1; s <= a.nsteps; ++s) {
    int par = (s - 1) & 1;                         // parity of the state being pulled
    asm volatile("" : "+s"(par));                  // (keeps both parities' addresses from being hoisted into registers)
    const unsigned pb = (unsigned)par * BOX, pbn = BOX - pb;
    const uint32_t want = a.tag0 + (uint32_t)(s - 1), tagn = want + 1u;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave's edge rows of state s-1 are in LDS
    if (*lds_abort != 0u) { aborted = true; break; }                  // (set before the barrier: every wave leaves here together)
    stamp(s, 0);
    if (s > 1 && tid < 64) {                       // speed sum of step s-1
      float v = (lane < nw) ? red[par * 16 + lane] : 0.f;   // (written with parity (s-1)&1 at the end of step s-1)
      v = wave_sum_dpp(v);
      if (tid == 0) a.partials[(long)(s - 2) * nt + tile] = v;
    }
    const bool laststep = (s == a.nsteps);
    float sp = 0.f;

    // one row: wait for its mail, start the fetch of the next row's, update the row in place, send its granules.
    //   lo = planes 2,5,6 of the row below, hi = planes 4,7,8 of the row above (old values); the band's first
    //   row takes lo, its last row hi, from the neighbouring wave (LDS) or the neighbouring tile (mail)
    auto send_stored = [&](auto rc) {             // the granules of a row that is finished: its populations are in f and own
      constexpr int r = decltype(rc)::value;
      float q[9];
      q[0] = 0.f;
      if constexpr (OWN_LDS) { q[1] = own[(r * 3 + 1) * 64]; q[3] = own[(r * 3 + 2) * 64]; } else { q[1] = f[r][1]; q[3] = f[r][3]; }
      q[2] = f[r][2]; q[4] = f[r][4]; q[5] = f[r][5]; q[6] = f[r][6]; q[7] = f[r][7]; q[8] = f[r][8];
      send_row(rc, tagn, pbn, q);
    };
    auto do_row = [&](auto rc, auto next_c, auto prev_c, Mail& m, Mail& mnext, float (&lo)[3], float (&hi)[3]) {
      constexpr int r = decltype(rc)::value, rnext = decltype(next_c)::value, rprev = decltype(prev_c)::value;
      const int ord = down ? R - 1 - r : r;
      stamp(s, 1 + 3 * ord);
      await(rc, pb, want, m);
      stamp(s, 2 + 3 * ord);
      if constexpr (rprev >= 0) { if (!laststep) send_stored(prev_c); }   // (behind the wait: see the note on the stores above)
      if (r == 0) {
        if (first) { lo[0] = __uint_as_float(m.e.x); lo[1] = __uint_as_float(m.e.y); lo[2] = __uint_as_float(m.e.z); }
        else {
          const float* q = lds + ((par * nw + (w - 1)) * 6 + 3) * 64 + lane;
          lo[0] = q[0]; lo[1] = q[64]; lo[2] = q[128];
        }
      }
      if (r == R - 1) {
        if (last) {
          const rt_u4 n = (R == 1) ? m.x : m.e;
          hi[0] = __uint_as_float(n.x); hi[1] = __uint_as_float(n.y); hi[2] = __uint_as_float(n.z);
        } else {
          const float* q = lds + ((par * nw + (w + 1)) * 6) * 64 + lane;
          hi[0] = q[0]; hi[1] = q[64]; hi[2] = q[128];
        }
      }
      // couriers -> edge lanes: lane 0 / 63 holds row rho itself, the diagonal rows come one and two lanes over
      const float m0 = __uint_as_float(m.g.x);
      const float m1w = rt_row_shl<1>(m.g.y), m1e = rt_row_shr<1>(m.g.y);
      const float m2w = rt_row_shl<2>(m.g.z), m2e = rt_row_shr<2>(m.g.z);
      if constexpr (rnext >= 0) fetch(next_c, pb, mnext);   // the next row's mail, in flight behind this row's arithmetic
      float p[9];
      float c0, c1, c3;
      if constexpr (OWN_LDS) { c0 = own[(r * 3 + 0) * 64]; c1 = own[(r * 3 + 1) * 64]; c3 = own[(r * 3 + 2) * 64]; }
      else { c0 = f[r][0]; c1 = f[r][1]; c3 = f[r][3]; }
      p[0] = c0;
      p[1] = rt_west(m0, c1);
      p[3] = rt_east(m0, c3);
      p[2] = lo[0];
      p[5] = rt_west(m1w, lo[1]);
      p[6] = rt_east(m1e, lo[2]);
      p[4] = hi[0];
      p[7] = rt_east(m2e, hi[1]);
      p[8] = rt_west(m2w, hi[2]);
      sp += collide_cell<FAST, true>(p, blk[r], a.omega);
      if (gy0 + r == a.accel_row && !laststep) accelerate_cell(p, blk[r], a.a1, a.a2);
      f[r][2] = p[2]; f[r][4] = p[4]; f[r][5] = p[5]; f[r][6] = p[6]; f[r][7] = p[7]; f[r][8] = p[8];
      if constexpr (OWN_LDS) { own[(r * 3 + 0) * 64] = p[0]; own[(r * 3 + 1) * 64] = p[1]; own[(r * 3 + 2) * 64] = p[3]; }
      else { f[r][0] = p[0]; f[r][1] = p[1]; f[r][3] = p[3]; }
      __builtin_amdgcn_sched_barrier(0);             // one row at a time: interleaving the rows costs more registers than the tile has to spare
      if constexpr (rnext < 0) { if (!laststep) send_row(rc, tagn, pbn, p); }   // the wave's last row of the step sends at once
      stamp(s, 3 + 3 * ord);
    };
    // the wave's rows in order i = 0 .. R-1: row i going up, row R-1-i going down
    auto sweep = [&](auto up_c) {
      constexpr bool UP = decltype(up_c)::value;
      float sv[3] = {0.f, 0.f, 0.f};                 // going up: old planes 2,5,6 of the row just overwritten; going down: 4,7,8
      auto one = [&](auto ic, Mail& m, Mail& mnext) {
        constexpr int i = decltype(ic)::value;
        constexpr int r = UP ? i : R - 1 - i;
        // the SIMD issues by priority, then age: left alone it runs its oldest wave through all its rows before the
        // next one starts, and what the row-by-row mail counts on -- neighbouring rows at most a row apart in TIME --
        // is gone.  A wave's priority falls with every row it finishes: the four waves of a SIMD take turns, row by row.
        __builtin_amdgcn_s_setprio(3 - i);
        constexpr int rn = (i + 1 < R) ? (UP ? r + 1 : r - 1) : -1;
        constexpr int rp = (i > 0) ? (UP ? r - 1 : r + 1) : -1;
        float lo[3], hi[3], keep[3];
        if constexpr (UP) {
          constexpr int ra = (r < R - 1) ? r + 1 : r;
          lo[0] = sv[0]; lo[1] = sv[1]; lo[2] = sv[2];
          hi[0] = f[ra][4]; hi[1] = f[ra][7]; hi[2] = f[ra][8];       // (the band's last row: replaced in do_row)
          keep[0] = f[r][2]; keep[1] = f[r][5]; keep[2] = f[r][6];
        } else {
          constexpr int rb = (r > 0) ? r - 1 : r;
          hi[0] = sv[0]; hi[1] = sv[1]; hi[2] = sv[2];
          lo[0] = f[rb][2]; lo[1] = f[rb][5]; lo[2] = f[rb][6];       // (the band's first row: replaced in do_row)
          keep[0] = f[r][4]; keep[1] = f[r][7]; keep[2] = f[r][8];
        }
        do_row(std::integral_constant<int, r>{}, std::integral_constant<int, rn>{}, std::integral_constant<int, rp>{}, m, mnext, lo, hi);
        sv[0] = keep[0]; sv[1] = keep[1]; sv[2] = keep[2];
      };
      Mail ma, mb;
      if (DBG_NOLOAD) { blank(ma); blank(mb); }
      one(I0{}, pre, ma);
      if constexpr (R > 1) one(I1{}, ma, mb);
      if constexpr (R > 2) { one(I2{}, mb, ma); one(I3{}, ma, mb); }
      // the first row of the next step: its mail may be on its way already
      // (a one-row wave has only just sent: what it would fetch cannot be there yet, it polls behind the barrier)
      if constexpr (R > 1) { if (!laststep) fetch(std::integral_constant<int, UP ? 0 : R - 1>{}, pbn, pre); }
    };
    if (!down) sweep(std::true_type{}); else sweep(std::false_type{});
    if (!laststep) publish_lds(s & 1);
    stamp(s, 13);
    sp = wave_sum_dpp(sp);
    if (lane == 0) red[(s & 1) * 16 + w] = sp;
  }
  __syncthreads();
  if (a.stats != nullptr && lane == 0) { atomicAdd(a.stats, (unsigned long long)nmiss); atomicAdd(a.stats + 1, (unsigned long long)nspin); }
  if (aborted || *lds_abort != 0u) return;          // (the host repeats the run from the untouched source lattice)
  {
    int gyq = gy0, gxq = gx;                         // (addresses worked out again here: kept from the prologue they would live in scratch)
    asm volatile("" : "+v"(gyq), "+v"(gxq));
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long o = (long)(gyq + r) * a.pitch + gxq;
      if constexpr (OWN_LDS) { __builtin_nontemporal_store(own[(r * 3 + 0) * 64], &a.dst[o]); __builtin_nontemporal_store(own[(r * 3 + 1) * 64], &a.dst[a.plane + o]); __builtin_nontemporal_store(own[(r * 3 + 2) * 64], &a.dst[3 * a.plane + o]); }
      else { __builtin_nontemporal_store(f[r][0], &a.dst[o]); __builtin_nontemporal_store(f[r][1], &a.dst[a.plane + o]); __builtin_nontemporal_store(f[r][3], &a.dst[3 * a.plane + o]); }
#pragma unroll
      for (int k = 2; k < 9; ++k)
        if (k != 3) __builtin_nontemporal_store(f[r][k], &a.dst[k * a.plane + o]);
    }
  }
  if (tid < 64) {
    float v = (lane < nw) ? red[(a.nsteps & 1) * 16 + lane] : 0.f;
    v = wave_sum(v);
    if (tid == 0) a.partials[(long)(a.nsteps - 1) * nt + tile] = v;
  }
}

// A lattice alone on its GPU: gridDim.x = its tiles.
template <int R, int MODE>
__global__ __launch_bounds__(1024) void lbm_regtile(const RegTileArgs a) {
  regtile_body<R, MODE>(a);
}

// The slabs of ONE device in ONE launch (MODE & kRegSlab): gridDim.x = tiles per slab, gridDim.y = slabs of this device,
// table[blockIdx.y] = that slab's arguments.  One launch, because tiles of different slabs wait for each other exactly as
// tiles of one slab do: they must all be resident at once, which separate launches on separate streams do not promise.
// (The table is read through the constant address space: scalar loads, the arguments live in SGPRs as kernel arguments do.)
template <int R, int MODE>
__global__ __launch_bounds__(1024) void lbm_regtile_slabs(const RegTileArgs* table) {
  static_assert((MODE & kRegSlab) != 0, "slab flavour");
  typedef const __attribute__((address_space(4))) unsigned* cwords_t;
  static_assert(sizeof(RegTileArgs) % 4 == 0, "copied word by word");
  const cwords_t src = (cwords_t)(unsigned long long)(table + blockIdx.y);
  RegTileArgs a;
  unsigned* dst = reinterpret_cast<unsigned*>(&a);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(RegTileArgs) / 4); ++i) dst[i] = src[i];
  regtile_body<R, MODE>(a);
}

}  // namespace lbm
