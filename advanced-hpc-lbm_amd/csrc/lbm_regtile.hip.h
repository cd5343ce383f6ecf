// lbm_regtile.hip.h -- the whole step loop in ONE launch with the lattice resident in REGISTERS
// (second form of the resident engine; the first, lbm_resident.hip.h, keeps tiles in LDS and spends its
// time on 770 eight-byte stores per tile and step behind __syncthreads' vmcnt(0)).
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
//   across the tile border    lane 0 / lane 63 (west / east column) and wave 0 / wave NW-1 (south / north
//                             row) take the three populations that enter the tile out of the tile's
//                             mailbox in global memory: 8-byte {value, tag} granules (MI355X_MICROARCH.md,
//                             Valid forms, R2: the data is the flag; one aligned sc1 store, sc1 loads, no
//                             fence), polled until the tag says "state s"; and the same threads store
//                             their new edge values into the neighbours' mailboxes, fire and forget --
//                             nothing in the loop waits for a store (no vmcnt wait, no __syncthreads).
// The update is in place, row by row upwards, with three saved registers (the old planes 2,5,6 of the row
// just overwritten).  Mailbox of a tile, per parity: Sin / Nin [3 planes][64 columns] (from the tile
// below / above), Win / Ein [3 planes][TY + 2 rows] (rows -1 .. TY: the two extra rows are the corner
// populations, written by the diagonal tiles).  Two parities suffice for the reason given in
// lbm_resident.hip.h: the thread that consumes a granule owns the cell that produces the opposite one.
// Every wait is bounded and watches a global abort word; a run that gives up leaves the source lattice
// untouched and the host repeats it with the streaming kernels.
#pragma once
#include <type_traits>
#include "lbm_kernels.hip.h"
#include "lbm_resident.hip.h"   // gu64 / gu32, kResidentTimeoutTicks, lbm_fold_steps

namespace lbm {

struct RegTileArgs {
  const float* src; float* dst; long plane; int pitch, nx, ny;
  const uint8_t* blocked;
  float omega; int accel_row; float a1, a2;
  int ty;                      // rows per tile = waves per block x R
  int ntx, nty;                // tiles per lattice row (nx / 64) / column (ny / ty); gridDim.x = ntx * nty
  int nsteps;
  uint32_t tag0;               // tag of state 0 of this run; state s carries tag0 + s
  unsigned long long* mail;    // [tile][parity][regtile_box(ty)] granules
  float* partials;             // [nsteps][ntiles]
  uint32_t* abort_word;
};

// granules of one mailbox (one tile, one parity): Sin[3][64], Nin[3][64], Win[3][ty+2], Ein[3][ty+2]
__host__ __device__ constexpr int regtile_box(int ty) { return 2 * 3 * 64 + 2 * 3 * (ty + 2); }

__device__ __forceinline__ float rt_west(float v) {   // lane i <- lane i-1
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float rt_east(float v) {   // lane i <- lane i+1
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xf, 0xf, true));
}

// v (all lanes) with lane `LANE` replaced by the wave-uniform value s (this toolchain has the readlane builtin only)
template <int LANE>
__device__ __forceinline__ int rt_writelane(int v, int s) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(LANE));
  return v;
}

// blockDim.x = 64 NW; R rows per wave.
template <int R, int MODE>
__global__ __launch_bounds__(1024) void lbm_regtile(const RegTileArgs a) {
  constexpr bool FAST = (MODE & kFastMath) != 0;
  // timing experiments only (wrong results), LBM_RESIDENT_DEBUG: 1 = one pass over the inbox, no waiting; 2 = also no
  // stores to other tiles; 3 = also no inbox loads at all
  constexpr bool DBG_NOWAIT = (MODE & kResDebugNoWait) != 0, DBG_NOSEND = (MODE & kResDebugNoSend) != 0, DBG_NOLOAD = (MODE & 256) != 0;
  // LDS: edge rows between the waves of the tile [parity][wave][6][64] + per-wave speed sums [parity][16] + abort word
  __shared__ __attribute__((aligned(16))) float lds[2 * 16 * 6 * 64 + 2 * 16 + 16];
  float* red = lds + 2 * 16 * 6 * 64;
  uint32_t* lds_abort = reinterpret_cast<uint32_t*>(red + 32);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)(blockDim.x >> 6);
  const int nt = gridDim.x;
  int tile = blockIdx.x;
  if ((nt & 7) == 0) tile = (tile & 7) * (nt >> 3) + (tile >> 3);      // neighbouring tiles mostly share an XCD (speed only)
  const int by = tile / a.ntx, bx = tile - by * a.ntx;
  const int TY = a.ty, BOX = regtile_box(TY);
  auto tile_of = [&](int dx, int dy) {
    int x = bx + dx, y = by + dy;
    x += (x < 0) ? a.ntx : 0; x -= (x >= a.ntx) ? a.ntx : 0;
    y += (y < 0) ? a.nty : 0; y -= (y >= a.nty) ? a.nty : 0;
    return y * a.ntx + x;
  };
  // mailbox sections (granule offsets inside a box)
  const int oS = 0, oN = 3 * 64, oW = 6 * 64, oE = 6 * 64 + 3 * (TY + 2);
  auto box = [&](int t, int parity) { return (gu64*)a.mail + (unsigned)((t * 2 + parity) * BOX); };   // (a few MB: 32-bit offsets)
  auto send = [&](gu64* g, uint32_t tag, float v) {
    if (!DBG_NOSEND) __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  const int tS = tile_of(0, -1), tN = tile_of(0, 1), tW = tile_of(-1, 0), tE = tile_of(1, 0);
  const int tSW = tile_of(-1, -1), tSE = tile_of(1, -1), tNW = tile_of(-1, 1), tNE = tile_of(1, 1);
  const bool first = (w == 0), last = (w == nw - 1);
  const int rho0 = w * R;                                   // tile row of this wave's first row
  const int gx = bx * 64 + lane, gy0 = by * TY + rho0;
  if (tid == 0) *lds_abort = 0u;

  // ---- east / west mail in ONE store and ONE load instruction per wave and step (a memory instruction costs
  // its issue slot whether one lane is active or sixty-four, and it was ~390 one-lane instructions per tile and
  // step that made the first version of this loop memory-instruction bound).  Lane i of the wave is a courier:
  //   i in [0, 3R)       slot i / R, row i % R of the EAST column (lane 63's planes 1,5,8) -> west inbox of the tile to the east
  //   i in [3R, 6R)      the same of the WEST column (lane 0's planes 3,6,7)              -> east inbox of the tile to the west
  //   6R .. 6R+3         the four corner populations (edge waves only)
  // and on the way in lane i fetches the granule that lane 0 (i < 3R) or lane 63 needs for the same slot / row.
  constexpr int NM = 3 * R;
  unsigned send_off = 0u, recv_off = 0u;       // granule index inside a.mail for parity 0 (parity 1: + BOX)
  bool send_on = false, recv_on = false;
  {
    const int i = lane % NM, side = lane / NM;              // side 0: east column, 1: west column
    const int slot = i / R, r = i % R;
    if (lane < 2 * NM) {
      send_on = recv_on = true;
      send_off = (unsigned)(((side == 0 ? tE : tW) * 2) * BOX + (side == 0 ? oW : oE) + slot * (TY + 2) + (rho0 + 1) + r);
      // what the edge lane of row r needs: slot 0 of row rho, slot 1 of row rho-1, slot 2 of row rho+1 (inbox row index = row + 1)
      recv_off = (unsigned)((tile * 2) * BOX + (side == 0 ? oW : oE) + slot * (TY + 2) + (rho0 + 1) + r + (slot == 1 ? -1 : slot == 2 ? 1 : 0));
    } else if (lane < 2 * NM + 4) {
      const int c = lane - 2 * NM;                          // 0: NE (plane 5 of the top row), 1: NW (6), 2: SE (8 of the bottom row), 3: SW (7)
      send_on = (c < 2) ? last : first;
      const int t = (c == 0) ? tNE : (c == 1) ? tNW : (c == 2) ? tSE : tSW;
      const int sec = (c == 0 || c == 2) ? oW : oE;
      send_off = (unsigned)((t * 2) * BOX + sec + (c < 2 ? (TY + 2) + 0 : 2 * (TY + 2) + (TY + 1)));
    }
  }
  static_assert(2 * NM + 4 <= 64, "couriers must fit a wave");

  // ---- state 0
  float f[R][9];
  bool blk[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long o = (long)(gy0 + r) * a.pitch + gx;
#pragma unroll
    for (int k = 0; k < 9; ++k) f[r][k] = a.src[k * a.plane + o];
    blk[r] = a.blocked[o] != 0;
    if (gy0 + r == a.accel_row) accelerate_cell(f[r], blk[r], a.a1, a.a2);   // accelerate phase of the first step
  }

  // publish the edge values of the state in f: LDS rows for the neighbouring waves, granules for the
  // neighbouring tiles
  auto publish = [&](uint32_t tag, int parity) {
    if (first) {            // the tile's bottom row enters the tile below through ITS north inbox
      gu64* b = box(tS, parity) + oN + lane;
      send(b, tag, f[0][4]); send(b + 64, tag, f[0][7]); send(b + 128, tag, f[0][8]);
    }
    if (last) {
      gu64* b = box(tN, parity) + oS + lane;
      send(b, tag, f[R - 1][2]); send(b + 64, tag, f[R - 1][5]); send(b + 128, tag, f[R - 1][6]);
    }
    // east / west columns and corners: lane transposition (readlane from the edge lane, writelane into the courier lane), one store
    int sv = 0;
    auto put = [&](auto lane_c, float v, int from) {
      sv = rt_writelane<decltype(lane_c)::value>(sv, __builtin_amdgcn_readlane(__float_as_int(v), from));
    };
    auto put_rows = [&](auto slot_c) {
      constexpr int slot = decltype(slot_c)::value;
      constexpr int ke = (slot == 0) ? 1 : (slot == 1) ? 5 : 8, kw = (slot == 0) ? 3 : (slot == 1) ? 6 : 7;
      put(std::integral_constant<int, slot * R + 0>{}, f[0][ke], 63);
      put(std::integral_constant<int, NM + slot * R + 0>{}, f[0][kw], 0);
      if constexpr (R > 1) { put(std::integral_constant<int, slot * R + 1>{}, f[1 % R][ke], 63); put(std::integral_constant<int, NM + slot * R + 1>{}, f[1 % R][kw], 0); }
      if constexpr (R > 2) {
        put(std::integral_constant<int, slot * R + 2>{}, f[2 % R][ke], 63); put(std::integral_constant<int, NM + slot * R + 2>{}, f[2 % R][kw], 0);
        put(std::integral_constant<int, slot * R + 3>{}, f[3 % R][ke], 63); put(std::integral_constant<int, NM + slot * R + 3>{}, f[3 % R][kw], 0);
      }
    };
    static_assert(R == 1 || R == 2 || R == 4, "rows per wave");
    put_rows(std::integral_constant<int, 0>{});
    put_rows(std::integral_constant<int, 1>{});
    put_rows(std::integral_constant<int, 2>{});
    put(std::integral_constant<int, 2 * NM + 0>{}, f[R - 1][5], 63);
    put(std::integral_constant<int, 2 * NM + 1>{}, f[R - 1][6], 0);
    put(std::integral_constant<int, 2 * NM + 2>{}, f[0][8], 63);
    put(std::integral_constant<int, 2 * NM + 3>{}, f[0][7], 0);
    if (send_on) send((gu64*)a.mail + (send_off + (unsigned)(parity * BOX)), tag, __int_as_float(sv));
    float* me = lds + ((parity * 16 + w) * 6) * 64 + lane;
    // bottom row's planes 4,7,8 for the wave below; top row's 2,5,6 for the wave above
    me[0 * 64] = f[0][4]; me[1 * 64] = f[0][7]; me[2 * 64] = f[0][8];
    me[3 * 64] = f[R - 1][2]; me[4 * 64] = f[R - 1][5]; me[5 * 64] = f[R - 1][6];
  };
  publish(a.tag0, 0);

  bool aborted = false;
  for (int s = 1; s <= a.nsteps; ++s) {
    int par = (s - 1) & 1;                         // parity of the state being pulled
    asm volatile("" : "+s"(par));                  // (keeps both parities' addresses from being hoisted into registers)
    const uint32_t want = a.tag0 + (uint32_t)(s - 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave's edge rows of state s-1 are in LDS
    if (*lds_abort != 0u) { aborted = true; break; }                  // (set before the barrier: every wave leaves here together)
    if (s > 1 && tid < 64) {                       // speed sum of step s-1
      float v = (lane < nw) ? red[par * 16 + lane] : 0.f;   // (written with parity (s-1)&1 at the end of step s-1)
      v = wave_sum(v);
      if (tid == 0) a.partials[(long)(s - 2) * nt + tile] = v;
    }
    // ---- what enters the band: the row below (planes 2,5,6) and the row above (4,7,8)
    float S[3], N[3];
    bool ok = true;
    const gu64* mybox = box(tile, par);
    if (!first) {
      const float* q = lds + ((par * 16 + (w - 1)) * 6 + 3) * 64 + lane;
      S[0] = q[0]; S[1] = q[64]; S[2] = q[128];
    }
    if (!last) {
      const float* q = lds + ((par * 16 + (w + 1)) * 6) * 64 + lane;
      N[0] = q[0]; N[1] = q[64]; N[2] = q[128];
    }
    // ---- granules: south / north rows (edge waves), west / east columns (lanes 0 / 63), polled until all carry the tag
    int rv = 0;                                    // the courier lanes' granule values
    if (!DBG_NOLOAD) {
      const long long t0 = wall_clock64();
      for (;;) {
        ok = true;
        auto recv = [&](const gu64* g, float& dst) {
          const unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && ((uint32_t)(x >> 32) == want);
          dst = __uint_as_float((uint32_t)x);
        };
        if (first) { recv(mybox + oS + lane, S[0]); recv(mybox + oS + 64 + lane, S[1]); recv(mybox + oS + 128 + lane, S[2]); }
        if (last) { recv(mybox + oN + lane, N[0]); recv(mybox + oN + 64 + lane, N[1]); recv(mybox + oN + 128 + lane, N[2]); }
        if (recv_on) { float v; recv((const gu64*)a.mail + (recv_off + (unsigned)(par * BOX)), v); rv = __float_as_int(v); }
        if (__all(ok) || DBG_NOWAIT) break;
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > kResidentTimeoutTicks ||
            __hip_atomic_load((gu32*)a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          __hip_atomic_store((gu32*)a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *lds_abort = 1u;
          break;
        }
      }
    }
    // (a wave that gave up carries on with what it has; everybody leaves together behind the next barrier)
    // ---- the step, in place, row by row upwards
    const bool laststep = (s == a.nsteps);
    // lane 0's population from the west tile sits in courier lane i (slot * R + row), lane 63's from the east tile in lane NM + i:
    // the west tile's values belong to the EAST-column couriers' counterparts -- careful: couriers [0, NM) FETCH from this tile's
    // WEST inbox (recv_off, side 0) and couriers [NM, 2 NM) from its EAST inbox
    auto edge_w = [&](float v, int i) { return __int_as_float(rt_writelane<0>(__float_as_int(v), __builtin_amdgcn_readlane(rv, i))); };
    auto edge_e = [&](float v, int i) { return __int_as_float(rt_writelane<63>(__float_as_int(v), __builtin_amdgcn_readlane(rv, NM + i))); };
    float b2 = S[0], b5 = S[1], b6 = S[2];            // planes 2,5,6 of the row below the one being updated (old values)
    float sp = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float p[9];
      const int ra = (r < R - 1) ? r + 1 : r;          // (compile-time after unrolling)
      const float u4 = f[ra][4], u7 = f[ra][7], u8 = f[ra][8];
      const float a4 = (r < R - 1) ? u4 : N[0], a7 = (r < R - 1) ? u7 : N[1], a8 = (r < R - 1) ? u8 : N[2];
      p[0] = f[r][0];
      p[1] = edge_w(rt_west(f[r][1]), 0 * R + r);
      p[3] = edge_e(rt_east(f[r][3]), 0 * R + r);
      p[2] = b2;
      p[5] = edge_w(rt_west(b5), 1 * R + r);
      p[6] = edge_e(rt_east(b6), 1 * R + r);
      p[4] = a4;
      p[7] = edge_e(rt_east(a7), 2 * R + r);
      p[8] = edge_w(rt_west(a8), 2 * R + r);
      b2 = f[r][2]; b5 = f[r][5]; b6 = f[r][6];       // saved before the row is overwritten
      sp += collide_cell<FAST>(p, blk[r], a.omega);
      if (gy0 + r == a.accel_row && !laststep) accelerate_cell(p, blk[r], a.a1, a.a2);
#pragma unroll
      for (int k = 0; k < 9; ++k) f[r][k] = p[k];
      __builtin_amdgcn_sched_barrier(0);             // one row at a time: interleaving the rows costs more registers than the tile has to spare
    }
    if (!laststep) {                                   // mail first: it has the longest way to go
      int parn = s & 1;
      asm volatile("" : "+s"(parn));
      publish(a.tag0 + (uint32_t)s, parn);
    }
    sp = wave_sum(sp);
    if (lane == 0) red[(s & 1) * 16 + w] = sp;
    if (laststep) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const long o = (long)(gy0 + r) * a.pitch + gx;
#pragma unroll
        for (int k = 0; k < 9; ++k) a.dst[k * a.plane + o] = f[r][k];
      }
    }
  }
  __syncthreads();
  if (!aborted && *lds_abort == 0u && tid < 64) {
    float v = (lane < nw) ? red[(a.nsteps & 1) * 16 + lane] : 0.f;
    v = wave_sum(v);
    if (tid == 0) a.partials[(long)(a.nsteps - 1) * nt + tile] = v;
  }
}

}  // namespace lbm
