// lbm_kernels.hip.h -- hand-written HIP kernels for the D2Q9-BGK time step on
// CDNA4 (gfx950, wave64).  Device code only; the C ABI lives in lbm_api.hip.
//
// What one launch of lbm_sweep does for every cell of the rows it covers
// (the reference's fused step, /root/reference/d2q9-bgk.c:228-1813, in one
// pull-scheme pass over SoA planes):
//   pull-stream  (gather map d2q9-bgk.c:2139-2147)
//   bounce-back  on blocked cells (d2q9-bgk.c:971-981)
//   BGK collide  on fluid cells   (d2q9-bgk.c:982-1100)
//   |u'| of the stored values, summed per block for av_vels (d2q9-bgk.c:1103-1130)
//   accelerate   (d2q9-bgk.c:230-260) -- applied AT WRITE TIME to row ny-2 of
//                the lattice being written, i.e. the accelerate phase of the
//                NEXT step, in the same float operations (SURVEY.md §7 hard
//                part 1b); a one-row prologue kernel covers the first step and
//                the last step of a run leaves the lattice unbiased.
//
// Data layout in HBM (per slab of nyl rows): 9 planes of nyl x pitch floats
// (+ padding on the plane stride), plane k = distribution k, x fastest; one
// byte per cell for the blocked map.
//
// Kernels in this file:
//   lbm_sweep<V, MODE>          one step per pass; a thread owns V consecutive cells of a row
//   lbm_sweep2<TX, TY, MODE, KIND, NT, PARTIAL>
//                               two steps per pass through LDS (default where it applies);
//                               KIND = plain / slab edge / whole slab with in-kernel
//                               peer-to-peer halo hand-off
//   lbm_p2p_push                halo wait / push outside the fused launch (peer-to-peer)
//   small ones: accelerate row, halo packing, partial-sum folds, layout conversion, derived fields
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lbm_exact_math.hip.h"

namespace lbm {

constexpr int kBlock = 256;  // 4 waves of 64

// The relaxation parameter and what the collision needs of it, worked out ONCE, on the host, in float (a GPU of this
// generation has no scalar float unit: derived inside a kernel these wave-uniform numbers would sit in vector registers).
// Every kernel's argument block carries one (`omega`; assigning a float converts), so every kernel collides with the same
// three numbers.
struct Relax {
  float omega;                 // d2q9-bgk.c: params.omega
  float omc;                   // 1 - omega
  float w1;                    // omega x 1/9: the axis directions' weight; the rest direction's is 4 x, the diagonals' 1/4 x (exact)
  Relax() = default;
  __host__ __device__ Relax(float om) : omega(om), omc(1.f - om), w1(om * (1.f / 9.f)) {}
};

struct SweepArgs {
  const float* src;            // source lattice: plane k at src + k*plane
  float* dst;                  // destination lattice
  long plane;                  // floats per plane (nyl * pitch)
  int pitch;                   // floats per row
  int nx;                      // cells per row
  int nyl;                     // rows in this slab
  // rows covered by this launch: y = y_begin + i*y_stride, i in [0, y_count)
  int y_begin, y_count, y_stride;
  // "row -1": planes 2,5,6 of the slab to the south (or own top row if alone)
  const float* south2; const float* south5; const float* south6;
  // "row nyl": planes 4,7,8 of the slab to the north (or own row 0 if alone)
  const float* north4; const float* north7; const float* north8;
  const uint8_t* blocked;      // nyl * pitch bytes, 1 = obstacle
  Relax omega;                 // params.omega and what the collision derives from it
  int accel_row;               // local row that receives the next step's accelerate, or -1
  float a1, a2;                // density*accel/9, density*accel/36
  float* partials;             // one float per block of this launch: sum of |u'|
  float* send_south;           // 3*nx floats (planes 4,7,8 of row 0) or nullptr
  float* send_north;           // 3*nx floats (planes 2,5,6 of row nyl-1) or nullptr
  // fold-in of the previous step's block partials (block 0 only)
  const float* prev_partials;  // or nullptr
  int prev_count;
  double* prev_sum;            // where the previous step's slab sum goes
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// The same sum by DPP adds alone (no LDS permutes, no lgkmcnt waits): row_shr 1, 2, 4, 8 gather every row of 16 lanes in
// its lane 15, row_bcast:15 / row_bcast:31 carry the row sums on; the total is in LANE 63 and returned wave-uniformly.
// A chain of six dependent v_add_f32 against six ds_bpermute round trips (~700 cycles): for sums taken inside a loop.
// (The order of the additions differs from wave_sum's: results agree to rounding, not bit for bit.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define LBM_DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, true))
  LBM_DPP_ADD(0x111, 0xf);   // row_shr:1
  LBM_DPP_ADD(0x112, 0xf);   // row_shr:2
  LBM_DPP_ADD(0x114, 0xf);   // row_shr:4
  LBM_DPP_ADD(0x118, 0xf);   // row_shr:8   -> lane 15 of every row: the row's sum
  LBM_DPP_ADD(0x142, 0xa);   // row_bcast:15, rows 1 and 3 -> lane 31: rows 0+1, lane 63: rows 2+3
  LBM_DPP_ADD(0x143, 0xc);   // row_bcast:31, rows 2 and 3 -> lane 63: everything
#undef LBM_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Block-wide sum; result valid in thread 0.  `red` holds one slot per wave (NW waves per block).
template <typename T, int NW = kBlock / 64>
__device__ __forceinline__ T block_sum(T v, T* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  T r = T(0);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < NW; ++w) r += red[w];
  }
  return r;
}

// Kernel mode bits (template parameter MODE of lbm_sweep).
constexpr int kFastMath = 1;   // v_rcp_f32 / v_sqrt_f32 (1 ulp) instead of the IEEE sequences
constexpr int kNtStore = 2;    // nontemporal stores of the destination lattice
constexpr int kNtLoad = 4;     // nontemporal loads of the source lattice
constexpr int kBenchNoMath = 8;        // tools/kbench only: pull + store, no collision (wrong results)
constexpr int kBenchAlignedOnly = 16;  // tools/kbench only: unshifted loads everywhere (wrong results)
constexpr int kSpeedFromStored = 32;   // the step's average speed re-summed from the post-collision populations, the
                                       // reference's own form (d2q9-bgk.c:1103-1130); lbm_sweep only (option
                                       // kernel_variant bit 3): the form every kernel used until round 2, kept so that
                                       // a discrepancy in av_vels can be bisected against it

// (not FAST: correctly rounded, by the short sequences of lbm_exact_math.hip.h wherever they are proven)
template <bool FAST> __device__ __forceinline__ float recip(float x) {
  if constexpr (FAST) return __builtin_amdgcn_rcpf(x); else return recip_exact(x);
}
template <bool FAST> __device__ __forceinline__ float root(float x) {
  if constexpr (FAST) return __builtin_amdgcn_sqrtf(x); else return root_exact(x);
}

// One cell: p[] holds the nine pulled values on entry and the nine values to
// store on exit.  Returns the cell's contribution to the step's speed sum.
// The arithmetic is SURVEY.md Appendix A (= d2q9-bgk.c:982-1130) with c_sq = 1/3 folded into the constants
// (1/c_sq = 3, 1/(2 c_sq^2) = 4.5, 1/(2 c_sq) = 1.5), ARRANGED FOR THE INSTRUCTION COUNT -- every kernel of this library
// that keeps more than one step on the chip is bound by the vector instructions of this function:
//   * one reciprocal of the density for both velocity components (the reference's own -Ofast build does the same);
//   * the three-population sums E, W, N, S serve the density AND the momenta (14 adds for what took 18);
//   * relaxation folded into the equilibrium: t_k = (1 - omega) p_k + (omega w_k rho) (base + u_k (3 + 4.5 u_k)) -- one
//     multiply-add behind the equilibrium instead of a subtraction and a multiply-add, the omega w_1 coming ready-made
//     (Relax) and w_0 = 4 w_1, w_5..8 = w_1 / 4 being exact in binary.
// 70 vector instructions per cell with the ten bounce-back selects (round 2's form: 83).  (Letting opposite directions share
// base + 4.5 u_k^2 would take three more off, and cost the two-column lbm_wave 45 spilled registers: measured, not used.)
// Mathematically the same numbers as before; the rounding differs in the last bits (tests/test_gpu_parity.py holds one step
// to 4e-6 per element of the reference's known answers, as before).  Every multiply-add is spelled out (fmaf) and
// contraction is switched off for the function, so the sequence of float operations per cell is fixed by this text and not
// by what the optimiser happens to fuse in a given instantiation: all kernels built from it (1, 2 or 4 cells per thread,
// any number of steps per pass, any decomposition) produce bit-identical lattices.
template <bool FAST, bool SPARSE = false, bool STORED = false>
__device__ __forceinline__ float collide_cell(float (&p)[9], bool is_blocked, const Relax& om) {
#pragma clang fp contract(off)
  const float E = (p[1] + p[5]) + p[8];
  const float W = (p[3] + p[6]) + p[7];
  const float N = (p[2] + p[5]) + p[6];
  const float S = (p[4] + p[7]) + p[8];
  float rho = p[0] + E;
  rho += W; rho += p[2]; rho += p[4];
  const float inv = recip<FAST>(rho);
  const float ux = (E - W) * inv;
  const float uy = (N - S) * inv;
  const float usq = __builtin_fmaf(ux, ux, uy * uy);
  const float base = __builtin_fmaf(-1.5f, usq, 1.f);          // 1 - u_sq / (2 c_sq)
  const float r1 = om.w1 * rho, r0 = 4.f * r1, r2 = 0.25f * r1;   // omega w_k rho
  const float upp = ux + uy, upm = ux - uy;
  // omega d_k = omega w_k rho (1 + u_k/c_sq + u_k^2/(2 c_sq^2) - u_sq/(2 c_sq)) = (omega w_k rho) (base + u_k (3 + 4.5 u_k))
  float d[9];
  d[0] = r0 * base;
  d[1] = r1 * __builtin_fmaf(ux, __builtin_fmaf(4.5f, ux, 3.f), base);
  d[2] = r1 * __builtin_fmaf(uy, __builtin_fmaf(4.5f, uy, 3.f), base);
  d[3] = r1 * __builtin_fmaf(-ux, __builtin_fmaf(-4.5f, ux, 3.f), base);
  d[4] = r1 * __builtin_fmaf(-uy, __builtin_fmaf(-4.5f, uy, 3.f), base);
  d[5] = r2 * __builtin_fmaf(upp, __builtin_fmaf(4.5f, upp, 3.f), base);
  d[6] = r2 * __builtin_fmaf(-upm, __builtin_fmaf(-4.5f, upm, 3.f), base);
  d[7] = r2 * __builtin_fmaf(-upp, __builtin_fmaf(-4.5f, upp, 3.f), base);
  d[8] = r2 * __builtin_fmaf(upm, __builtin_fmaf(4.5f, upm, 3.f), base);
  float t[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) t[k] = __builtin_fmaf(om.omc, p[k], d[k]);   // (1 - omega) p + omega d
  // The cell's speed for the step's average (d2q9-bgk.c:1783-1811 takes it from the post-collision populations).
  // BGK relaxation conserves the cell's mass and momentum -- sum t_k = rho and sum t_k c_k = rho u, the equilibrium
  // having been built from exactly these -- so the post-collision velocity IS (ux, uy), up to the rounding of the
  // float sums (a few 1e-9 absolute per cell, random in sign: parts in 1e8 of a step's average, against the 1 % of
  // the reference's checker and the 2e-6 the tests hold it to).  Recomputing it from t[] cost 24 instructions.
  // STORED: the reference's form -- density and velocity re-summed from the values about to be stored.
  float speed;
  if constexpr (STORED) {
    float rho2 = t[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) rho2 += t[k];
    const float inv2 = recip<FAST>(rho2);
    const float vx = (t[1] + t[5] + t[8] - (t[3] + t[6] + t[7])) * inv2;
    const float vy = (t[2] + t[5] + t[6] - (t[4] + t[7] + t[8])) * inv2;
    speed = root<FAST>(__builtin_fmaf(vx, vx, vy * vy));
  } else {
    speed = root<FAST>(usq);
  }
  // blocked cell: mirrored pulled values instead, no contribution.  SPARSE: most wavefronts hold no blocked cell at all
  // (0.5 % of the shipped 1024^2 deck is blocked) and skip the ten selects behind one wave-uniform branch.
  if constexpr (SPARSE) {
    if (__builtin_expect(__any(is_blocked) == 0, 1)) {
#pragma unroll
      for (int k = 0; k < 9; ++k) p[k] = t[k];
      return speed;
    }
  }
  const float b1 = p[3], b2 = p[4], b3 = p[1], b4 = p[2], b5 = p[7], b6 = p[8], b7 = p[5], b8 = p[6];
  p[0] = is_blocked ? p[0] : t[0];
  p[1] = is_blocked ? b1 : t[1];
  p[2] = is_blocked ? b2 : t[2];
  p[3] = is_blocked ? b3 : t[3];
  p[4] = is_blocked ? b4 : t[4];
  p[5] = is_blocked ? b5 : t[5];
  p[6] = is_blocked ? b6 : t[6];
  p[7] = is_blocked ? b7 : t[7];
  p[8] = is_blocked ? b8 : t[8];
  return is_blocked ? 0.f : speed;
}

// The same collision for C cells of one lane at once, statement by statement across the cells: each cell sees exactly
// collide_cell's sequence of float operations (the lattices stay bit-identical), but consecutive instructions belong to
// DIFFERENT cells, so that a wave has C independent chains to issue from (lbm_wave with two columns per lane runs two
// waves per SIMD: there it is instruction-level parallelism, not other waves, that has to keep the SIMD's issue slots full).
template <bool FAST, int C>
__device__ __forceinline__ void collide_cells(float (&p)[C][9], const bool (&is_blocked)[C], const Relax& om, float (&speed)[C]) {
#pragma clang fp contract(off)
  float E[C], W[C], N[C], S[C], rho[C], inv[C], ux[C], uy[C], usq[C], base[C], r0[C], r1[C], r2[C], upp[C], upm[C], d[C][9], t[C][9];
#define LBM_EACH for (int c = 0; c < C; ++c)
#pragma unroll
  LBM_EACH { E[c] = (p[c][1] + p[c][5]) + p[c][8]; W[c] = (p[c][3] + p[c][6]) + p[c][7]; }
#pragma unroll
  LBM_EACH { N[c] = (p[c][2] + p[c][5]) + p[c][6]; S[c] = (p[c][4] + p[c][7]) + p[c][8]; }
#pragma unroll
  LBM_EACH rho[c] = p[c][0] + E[c];
#pragma unroll
  LBM_EACH rho[c] += W[c];
#pragma unroll
  LBM_EACH rho[c] += p[c][2];
#pragma unroll
  LBM_EACH rho[c] += p[c][4];
#pragma unroll
  LBM_EACH inv[c] = recip<FAST>(rho[c]);
#pragma unroll
  LBM_EACH ux[c] = (E[c] - W[c]) * inv[c];
#pragma unroll
  LBM_EACH uy[c] = (N[c] - S[c]) * inv[c];
#pragma unroll
  LBM_EACH usq[c] = __builtin_fmaf(ux[c], ux[c], uy[c] * uy[c]);
#pragma unroll
  LBM_EACH base[c] = __builtin_fmaf(-1.5f, usq[c], 1.f);
#pragma unroll
  LBM_EACH { r1[c] = om.w1 * rho[c]; r0[c] = 4.f * r1[c]; r2[c] = 0.25f * r1[c]; upp[c] = ux[c] + uy[c]; upm[c] = ux[c] - uy[c]; }
#pragma unroll
  LBM_EACH d[c][0] = r0[c] * base[c];
#pragma unroll
  LBM_EACH d[c][1] = r1[c] * __builtin_fmaf(ux[c], __builtin_fmaf(4.5f, ux[c], 3.f), base[c]);
#pragma unroll
  LBM_EACH d[c][2] = r1[c] * __builtin_fmaf(uy[c], __builtin_fmaf(4.5f, uy[c], 3.f), base[c]);
#pragma unroll
  LBM_EACH d[c][3] = r1[c] * __builtin_fmaf(-ux[c], __builtin_fmaf(-4.5f, ux[c], 3.f), base[c]);
#pragma unroll
  LBM_EACH d[c][4] = r1[c] * __builtin_fmaf(-uy[c], __builtin_fmaf(-4.5f, uy[c], 3.f), base[c]);
#pragma unroll
  LBM_EACH d[c][5] = r2[c] * __builtin_fmaf(upp[c], __builtin_fmaf(4.5f, upp[c], 3.f), base[c]);
#pragma unroll
  LBM_EACH d[c][6] = r2[c] * __builtin_fmaf(-upm[c], __builtin_fmaf(-4.5f, upm[c], 3.f), base[c]);
#pragma unroll
  LBM_EACH d[c][7] = r2[c] * __builtin_fmaf(-upp[c], __builtin_fmaf(-4.5f, upp[c], 3.f), base[c]);
#pragma unroll
  LBM_EACH d[c][8] = r2[c] * __builtin_fmaf(upm[c], __builtin_fmaf(4.5f, upm[c], 3.f), base[c]);
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    LBM_EACH t[c][k] = __builtin_fmaf(om.omc, p[c][k], d[c][k]);
  }
#pragma unroll
  LBM_EACH speed[c] = is_blocked[c] ? 0.f : root<FAST>(usq[c]);
#pragma unroll
  LBM_EACH {
    const float b1 = p[c][3], b2 = p[c][4], b3 = p[c][1], b4 = p[c][2], b5 = p[c][7], b6 = p[c][8], b7 = p[c][5], b8 = p[c][6];
    p[c][0] = is_blocked[c] ? p[c][0] : t[c][0];
    p[c][1] = is_blocked[c] ? b1 : t[c][1];
    p[c][2] = is_blocked[c] ? b2 : t[c][2];
    p[c][3] = is_blocked[c] ? b3 : t[c][3];
    p[c][4] = is_blocked[c] ? b4 : t[c][4];
    p[c][5] = is_blocked[c] ? b5 : t[c][5];
    p[c][6] = is_blocked[c] ? b6 : t[c][6];
    p[c][7] = is_blocked[c] ? b7 : t[c][7];
    p[c][8] = is_blocked[c] ? b8 : t[c][8];
  }
#undef LBM_EACH
}

// The accelerate phase for one cell of row ny-2 (d2q9-bgk.c:246-258).
__device__ __forceinline__ void accelerate_cell(float (&p)[9], bool is_blocked, float a1, float a2) {
#pragma clang fp contract(off)
  if (!is_blocked && (p[3] - a1) > 0.f && (p[6] - a2) > 0.f && (p[7] - a2) > 0.f) {
    p[1] += a1; p[5] += a2; p[8] += a2;
    p[3] -= a1; p[6] -= a2; p[7] -= a2;
  }
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-B access, 4-B aligned
typedef float f4a __attribute__((ext_vector_type(4), aligned(16)));
typedef float f2a __attribute__((ext_vector_type(2), aligned(8)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

template <bool NT, typename T> __device__ __forceinline__ T ldg(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT, typename T> __device__ __forceinline__ void stg(T* p, const T& v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// V consecutive values of a row, shifted by one cell to the west / east, with
// the periodic wrap of the reference (x_w = ii ? ii-1 : nx-1, x_e = (ii+1)%nx;
// d2q9-bgk.c:2133-2135).  NTL / NTS: nontemporal loads / stores.
template <int V, bool NTL, bool NTS> struct Row;

template <bool NTL, bool NTS> struct Row<1, NTL, NTS> {
  static __device__ __forceinline__ void ld(const float* r, int x0, float (&o)[1]) { o[0] = ldg<NTL>(r + x0); }
  static __device__ __forceinline__ void ld_w(const float* r, int x0, int nx, float (&o)[1]) {
    o[0] = ldg<NTL>(r + (x0 ? x0 - 1 : nx - 1));
  }
  static __device__ __forceinline__ void ld_e(const float* r, int x0, int nx, float (&o)[1]) {
    o[0] = ldg<NTL>(r + (x0 + 1 == nx ? 0 : x0 + 1));
  }
  static __device__ __forceinline__ void st(float* r, int x0, const float (&v)[1]) { stg<NTS>(r + x0, v[0]); }
};

template <bool NTL, bool NTS> struct Row<2, NTL, NTS> {
  static __device__ __forceinline__ void ld(const float* r, int x0, float (&o)[2]) {
    const f2a v = ldg<NTL>(reinterpret_cast<const f2a*>(r + x0));
    o[0] = v.x; o[1] = v.y;
  }
  static __device__ __forceinline__ void ld_w(const float* r, int x0, int nx, float (&o)[2]) {
    if (x0 > 0) {
      const f2u v = ldg<NTL>(reinterpret_cast<const f2u*>(r + x0 - 1));
      o[0] = v.x; o[1] = v.y;
    } else {
      o[0] = r[nx - 1]; o[1] = r[0];
    }
  }
  static __device__ __forceinline__ void ld_e(const float* r, int x0, int nx, float (&o)[2]) {
    if (x0 + 2 < nx) {
      const f2u v = ldg<NTL>(reinterpret_cast<const f2u*>(r + x0 + 1));
      o[0] = v.x; o[1] = v.y;
    } else {
      o[0] = r[x0 + 1]; o[1] = r[0];
    }
  }
  static __device__ __forceinline__ void st(float* r, int x0, const float (&v)[2]) {
    f2a o; o.x = v[0]; o.y = v[1];
    stg<NTS>(reinterpret_cast<f2a*>(r + x0), o);
  }
};

template <bool NTL, bool NTS> struct Row<4, NTL, NTS> {
  static __device__ __forceinline__ void ld(const float* r, int x0, float (&o)[4]) {
    const f4a v = ldg<NTL>(reinterpret_cast<const f4a*>(r + x0));
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void ld_w(const float* r, int x0, int nx, float (&o)[4]) {
    if (x0 > 0) {
      const f4u v = ldg<NTL>(reinterpret_cast<const f4u*>(r + x0 - 1));
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
      o[0] = r[nx - 1]; o[1] = r[0]; o[2] = r[1]; o[3] = r[2];
    }
  }
  static __device__ __forceinline__ void ld_e(const float* r, int x0, int nx, float (&o)[4]) {
    if (x0 + 4 < nx) {
      const f4u v = ldg<NTL>(reinterpret_cast<const f4u*>(r + x0 + 1));
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
      o[0] = r[x0 + 1]; o[1] = r[x0 + 2]; o[2] = r[x0 + 3]; o[3] = r[0];
    }
  }
  static __device__ __forceinline__ void st(float* r, int x0, const float (&v)[4]) {
    f4a o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
    stg<NTS>(reinterpret_cast<f4a*>(r + x0), o);
  }
};

// The fused step.  Requires nx % V == 0 (host picks V); any nyl >= 1.
template <int V, int MODE = 0>
__global__ __launch_bounds__(kBlock) void lbm_sweep(const SweepArgs a) {
  constexpr bool FAST = (MODE & kFastMath) != 0;
  using R = Row<V, (MODE & kNtLoad) != 0, (MODE & kNtStore) != 0>;
  using RS = Row<V, false, false>;   // halo send buffers: plain stores
  __shared__ float red_f[kBlock / 64];
  __shared__ double red_d[kBlock / 64];

  // Block 0 first folds the previous step's per-block partials into one
  // double per step: deterministic (fixed order), no atomics, no extra launch.
  if (blockIdx.x == 0 && a.prev_partials != nullptr) {
    double s = 0.0;
    for (int i = threadIdx.x; i < a.prev_count; i += kBlock) s += (double)a.prev_partials[i];
    s = block_sum<double>(s, red_d);
    if (threadIdx.x == 0) *a.prev_sum = s;
  }

  const int nxv = a.nx / V;
  const long gid = (long)blockIdx.x * kBlock + threadIdx.x;
  const int ri = (int)(gid / nxv);
  const int x0 = (int)(gid - (long)ri * nxv) * V;
  float local = 0.f;

  if (ri < a.y_count) {
    const int y = a.y_begin + ri * a.y_stride;
    const long rc = (long)y * a.pitch;
    const float* s = a.src;
    const long P = a.plane;
    // centre row
    const float* r0 = s + rc;
    const float* r1 = s + P + rc;
    const float* r3 = s + 3 * P + rc;
    // row to the south (y-1) and to the north (y+1), halo rows at the slab edges
    const bool at_s = (y == 0), at_n = (y == a.nyl - 1);
    const float* r2 = at_s ? a.south2 : s + 2 * P + rc - a.pitch;
    const float* r5 = at_s ? a.south5 : s + 5 * P + rc - a.pitch;
    const float* r6 = at_s ? a.south6 : s + 6 * P + rc - a.pitch;
    const float* r4 = at_n ? a.north4 : s + 4 * P + rc + a.pitch;
    const float* r7 = at_n ? a.north7 : s + 7 * P + rc + a.pitch;
    const float* r8 = at_n ? a.north8 : s + 8 * P + rc + a.pitch;

    float q[9][V];
    if constexpr ((MODE & kBenchAlignedOnly) != 0) {
      R::ld(r0, x0, q[0]); R::ld(r1, x0, q[1]); R::ld(r2, x0, q[2]); R::ld(r3, x0, q[3]); R::ld(r4, x0, q[4]);
      R::ld(r5, x0, q[5]); R::ld(r6, x0, q[6]); R::ld(r7, x0, q[7]); R::ld(r8, x0, q[8]);
    } else {
      R::ld(r0, x0, q[0]);
      R::ld_w(r1, x0, a.nx, q[1]);
      R::ld(r2, x0, q[2]);
      R::ld_e(r3, x0, a.nx, q[3]);
      R::ld(r4, x0, q[4]);
      R::ld_w(r5, x0, a.nx, q[5]);
      R::ld_e(r6, x0, a.nx, q[6]);
      R::ld_e(r7, x0, a.nx, q[7]);
      R::ld_w(r8, x0, a.nx, q[8]);
    }

    bool blk[V];
    if constexpr (V == 4) {
      const uint32_t m = *reinterpret_cast<const uint32_t*>(a.blocked + rc + x0);
      blk[0] = m & 0xffu; blk[1] = m & 0xff00u; blk[2] = m & 0xff0000u; blk[3] = m & 0xff000000u;
    } else if constexpr (V == 2) {
      const uint16_t m = *reinterpret_cast<const uint16_t*>(a.blocked + rc + x0);
      blk[0] = m & 0xffu; blk[1] = m & 0xff00u;
    } else {
      blk[0] = a.blocked[rc + x0] != 0;
    }

    const bool do_accel = (y == a.accel_row);
#pragma unroll
    for (int v = 0; v < V; ++v) {
      float p[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) p[k] = q[k][v];
      if constexpr ((MODE & kBenchNoMath) == 0) {
        local += collide_cell<FAST, false, (MODE & kSpeedFromStored) != 0>(p, blk[v], a.omega);
        if (do_accel) accelerate_cell(p, blk[v], a.a1, a.a2);
      } else {
        local += blk[v] ? 0.f : p[0];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) q[k][v] = p[k];
    }

    float* d = a.dst + rc;
#pragma unroll
    for (int k = 0; k < 9; ++k) R::st(d + k * P, x0, q[k]);

    // packed halo rows for the neighbouring slabs (multi-slab runs only)
    if (at_s && a.send_south != nullptr) {
      RS::st(a.send_south, x0, q[4]);
      RS::st(a.send_south + a.nx, x0, q[7]);
      RS::st(a.send_south + 2 * a.nx, x0, q[8]);
    }
    if (at_n && a.send_north != nullptr) {
      RS::st(a.send_north, x0, q[2]);
      RS::st(a.send_north + a.nx, x0, q[5]);
      RS::st(a.send_north + 2 * a.nx, x0, q[6]);
    }
  }

  const float bs = block_sum<float>(local, red_f);
  if (threadIdx.x == 0) a.partials[blockIdx.x] = bs;
}

// ---------------------------------------------------------------------------
// Two time steps in one pass (temporal blocking through LDS).
//
// A block owns a TX x TY tile of the lattice at step t+2.  Phase A computes step
// t+1 on the tile plus a one-cell ring ((TX+2) x (TY+2) cells, the ring is
// recomputed by the neighbouring tiles too) straight from the source lattice in
// HBM and parks the nine planes in LDS; phase B pulls step t+2 for the tile out
// of LDS and writes it to the destination lattice.  HBM traffic per TWO lattice
// updates: 36 B written + 36 B x (TX+2)(TY+2)/(TX TY) read (1.16x for 64 x 16)
// instead of 144 B, so the sweep runs above the single-step 72 B/LUP roofline.
//
// Each value is stored in LDS at the coordinates of the tile cell that will pull it (plane k of a
// region cell P belongs to P + c_k), so LDS holds exactly TX x TY values per plane and every
// phase-B read is one aligned ds_read_b128 per plane.
// Per-cell arithmetic is collide_cell / accelerate_cell, exactly as in
// lbm_sweep: the accelerate phase of step t+2 is applied to the step-t+1 values
// of row ny-2 as they go into LDS (ring cells included), that of step t+3 to the
// outputs unless the run ends there.  Speed sums: step t+1 counts the tile's own
// cells only (not the ring), step t+2 the tile.
// Halo buffers of a slab (both the one-step and the two-step path use this layout;
// the one-step path moves slots 3..5 only).  Nine rows of nx floats:
//   to/from the SOUTH neighbour:  slots 0..5 = planes 0,1,3,4,7,8 of the sender's row 0
//                                 slots 6..8 = planes 4,7,8       of the sender's row 1
//     (arrives as the receiver's rows nyl and nyl+1)
//   to/from the NORTH neighbour:  slots 0..5 = planes 0,1,3,2,5,6 of the sender's row nyl-1
//                                 slots 6..8 = planes 2,5,6       of the sender's row nyl-2
//     (arrives as the receiver's rows -1 and -2)
// Two fused steps need the centre planes 0,1,3 of the adjacent row (its cells are recomputed
// as the ring), the three planes that stream across the boundary from it, and those three
// planes of the row behind it: 9 nx floats per direction per PAIR of steps instead of
// 2 x 3 nx -- half as many messages.
constexpr int kHaloSlots = 9;
constexpr int kNoRow = -100;   // "no accelerate row on this slab"

// In-kernel halo hand-off between neighbouring slabs (peer-to-peer mode).  Every launch group has a
// sequence number seq (the same on every slab).  A launch reads the halos its neighbours pushed
// with seq-1 (buffer parity (seq-1)&1) and pushes its own new edge rows with seq (parity seq&1)
// straight into the neighbours' halo buffers, which are uncached device memory mapped on both
// sides (same process: peer access; other process: hipIpc).  Protocol per edge block:
//   wait (lane 0 polls, system scope) until my flag from that side >= seq-1, THEN compute and
//   write remotely, system-scope fence, bump a local completion counter; the block that completes
//   the side's tile row stores seq to the neighbour's flag.
// Waiting before writing is what makes two buffers enough: the neighbour's flag seq-1 is only
// raised after all its edge blocks of launch seq-1 -- the last readers of the buffer this launch
// overwrites -- have finished.  Every spin is bounded (wall clock); a timeout raises *err.
// A halo wait gives up after this long (wall clock, 100 MHz ticks): 4 s.  It is also the largest skew
// between two neighbouring ranks' lbm_run calls that the peer-to-peer transport tolerates.
constexpr long long kP2PTimeoutTicks = 400000000LL;

struct P2PSync {
  const uint32_t* flag_s;      // written by the south neighbour: "pushed seq X into your ghost_s"
  const uint32_t* flag_n;      // written by the north neighbour
  uint32_t* rem_flag_s;        // the south neighbour's flag_n (I am its north side)
  uint32_t* rem_flag_n;        // the north neighbour's flag_s
  uint32_t* cnt_s;             // local completion counters, monotonic over the life of the context
  uint32_t* cnt_n;
  uint32_t cnt_target_s, cnt_target_n;   // counter value once the last edge block of this launch arrives
  uint32_t seq;
  uint32_t* err;
};

__device__ __forceinline__ void p2p_wait(const uint32_t* flag, uint32_t want, uint32_t* err) {
  // *err is sticky: once any wait of this slab has timed out, every later wait -- in this launch and in
  // all the launches already queued behind it -- returns at once, so the queue drains in microseconds
  // instead of holding the GPU for 4 s per launch and side.
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    const long long t0 = wall_clock64();   // 100 MHz
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > kP2PTimeoutTicks ||
          __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the neighbour is gone
        break;
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);   // system scope: invalidates this CU's L1 ...
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... and the barrier behind us waits for it
}

// Producer side, in this order (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms):
// every storing wave drains its stores (p2p_drain), the workgroup's barrier, then ONE lane:
// system-scope release, drain, count the block done; the block that completes the tile row
// raises the neighbour's flag.
__device__ __forceinline__ void p2p_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The halo stores themselves are system-scope write-through stores (p2p_store) into uncached
// memory, so no L2 write-back fence is needed here -- a system-scope release would write back the
// whole XCD L2, which is full of freshly written lattice lines (measured: +11 % on 8192^2).
__device__ __forceinline__ void p2p_signal(uint32_t* cnt, uint32_t target, uint32_t* rem_flag, uint32_t seq) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old + 1u == target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the returned add orders behind every block's drain
    __hip_atomic_store(rem_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// One float into a neighbour's halo block: system scope (sc0 sc1), written through, never left in L2.
__device__ __forceinline__ void p2p_store(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// kinds of two-step launch
constexpr int kSweep2Plain = 0;   // tile rows that never look outside the slab / slab alone (periodic wrap)
constexpr int kSweep2Edge = 1;    // edge tile rows of a slab with neighbours (halos by RCCL or peer copies)
constexpr int kSweep2P2P = 2;     // whole slab in one launch: edge tile rows first (in-kernel hand-off), then interior

struct Sweep2Args {
  const float* src;
  float* dst;
  long plane;
  int pitch, nx, ny;           // ny = rows of this slab (the whole lattice if alone)
  const uint8_t* blocked;
  Relax omega;                 // params.omega and what the collision derives from it
  int accel_row;               // local row of global row ny-2, or kNoRow
  int accel_out;               // apply the accelerate phase to the outputs (0 on the last pair)
  float a1, a2;
  float* partials1;            // per block: speed sum of step t+1
  float* partials2;            // per block: speed sum of step t+2
  const float* prev1;          // previous pair's partials, folded by block 0 (or nullptr)
  const float* prev2;
  int prev_count;
  double* prev_sum;            // prev_sum[0], prev_sum[1]
  // tile rows covered by this launch: by = by_begin + i*by_stride, i in [0, by_count)
  int by_begin, by_count, by_stride;
  // EDGE launches only (slab with neighbours): received halos, blocked map of rows -1 / nyl,
  // outgoing halos
  const float* ghost_s; const float* ghost_n;
  const uint8_t* blocked_gs; const uint8_t* blocked_gn;
  float* send_s; float* send_n;   // packed halos out: local send buffers, or (peer-to-peer) the
                                  // neighbours' own halo buffers, written directly over xGMI
  // peer-to-peer launches only (kSweep2P2P): in-kernel hand-off, see P2PSync
  P2PSync sync;
};

// Order of the tiles inside one XCD's contiguous run: G tile rows at a time, column by column.
// G > 1 would let vertically adjacent tiles (which share two ring rows of TX cells) run back to
// back and find those rows in the XCD's L2 -- measured, it loses more in DRAM locality than it
// saves in re-reads (8192^2, 64x16 tiles: G = 1 508 us, G = 2 513 us, G = 4 590 us per step),
// so tiles run row-major.
constexpr int kTileGroup = 1;
template <int G>
__device__ __forceinline__ void tile_order(int idx, int ntx, int nrows, int& row, int& col) {
  const int full = (nrows / G) * G * ntx;       // tiles in complete groups of G rows
  if (idx < full) {
    const int g = idx / (G * ntx), rem = idx - g * (G * ntx);
    col = rem / G;
    row = g * G + (rem - col * G);
  } else {                                     // the last, incomplete group: row-major
    const int rem = idx - full;
    row = (nrows / G) * G + rem / ntx;
    col = rem - (rem / ntx) * ntx;
  }
}

// Phase A gather of lbm_sweep2: the nine pulled values, blocked flag and region coordinates of the
// cells this thread computes for step t+1.  E = the tile row borders a neighbouring slab.
template <int TX, int TY, bool NTL, bool E, int NA, int NT>
__device__ __forceinline__ void sweep2_gather(const Sweep2Args& a, int X0, int Y0, float (&q)[NA][9], bool (&blk)[NA],
                                              int (&cxs)[NA], int (&cys)[NA]) {
  constexpr int IW = TX + 2, IH = TY + 2;
  const long P = a.plane;
  const float* s = a.src;
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    const int idx = threadIdx.x + m * NT;
    const int cy = idx / IW, cx = idx - cy * IW;
    cxs[m] = cx; cys[m] = cy;
    if (idx < IW * IH) {
      int gx = X0 - 1 + cx; gx += (gx < 0) ? a.nx : 0; gx -= (gx >= a.nx) ? a.nx : 0;
      const int xw = gx ? gx - 1 : a.nx - 1, xe = (gx + 1 == a.nx) ? 0 : gx + 1;
      int gy = Y0 - 1 + cy;
      const float *c0, *c1, *c3, *s2, *s5, *s6, *n4, *n7, *n8;   // row bases of the nine pulls
      if constexpr (!E) {
        gy += (gy < 0) ? a.ny : 0; gy -= (gy >= a.ny) ? a.ny : 0;
        const int ys = gy ? gy - 1 : a.ny - 1, yn = (gy + 1 == a.ny) ? 0 : gy + 1;
        const long rc = (long)gy * a.pitch, rs = (long)ys * a.pitch, rn = (long)yn * a.pitch;
        c0 = s + rc; c1 = s + P + rc; c3 = s + 3 * P + rc;
        s2 = s + 2 * P + rs; s5 = s + 5 * P + rs; s6 = s + 6 * P + rs;
        n4 = s + 4 * P + rn; n7 = s + 7 * P + rn; n8 = s + 8 * P + rn;
        blk[m] = a.blocked[rc + gx] != 0;
      } else {
        // gy in [-1, ny]; its south row in [-2, ny-1], its north row in [0, ny+1]
        const int nxl = a.nx;
        const long rc = (long)gy * a.pitch;
        if (gy < 0) {            // ring row -1: centre planes from the southern halo
          c0 = a.ghost_s; c1 = a.ghost_s + nxl; c3 = a.ghost_s + 2 * nxl;
          blk[m] = a.blocked_gs[gx] != 0;
        } else if (gy >= a.ny) { // ring row ny
          c0 = a.ghost_n; c1 = a.ghost_n + nxl; c3 = a.ghost_n + 2 * nxl;
          blk[m] = a.blocked_gn[gx] != 0;
        } else {
          c0 = s + rc; c1 = s + P + rc; c3 = s + 3 * P + rc;
          blk[m] = a.blocked[rc + gx] != 0;
        }
        if (gy <= 0) {           // south row is -2 (gy = -1) or -1 (gy = 0)
          const float* g = a.ghost_s + (gy < 0 ? 6 : 3) * nxl;
          s2 = g; s5 = g + nxl; s6 = g + 2 * nxl;
        } else {
          const long rs = rc - a.pitch;
          s2 = s + 2 * P + rs; s5 = s + 5 * P + rs; s6 = s + 6 * P + rs;
        }
        if (gy >= a.ny - 1) {    // north row is ny (gy = ny-1) or ny+1 (gy = ny)
          const float* g = a.ghost_n + (gy >= a.ny ? 6 : 3) * nxl;
          n4 = g; n7 = g + nxl; n8 = g + 2 * nxl;
        } else {
          const long rn = rc + a.pitch;
          n4 = s + 4 * P + rn; n7 = s + 7 * P + rn; n8 = s + 8 * P + rn;
        }
      }
      q[m][0] = ldg<NTL>(c0 + gx);
      q[m][1] = ldg<NTL>(c1 + xw);
      q[m][2] = ldg<NTL>(s2 + gx);
      q[m][3] = ldg<NTL>(c3 + xe);
      q[m][4] = ldg<NTL>(n4 + gx);
      q[m][5] = ldg<NTL>(s5 + xw);
      q[m][6] = ldg<NTL>(s6 + xe);
      q[m][7] = ldg<NTL>(n7 + xe);
      q[m][8] = ldg<NTL>(n8 + xw);
      cys[m] = cy | (gy == a.accel_row ? 0x10000 : 0);
    }
  }
}

// EDGE = false: the tile rows covered never look outside rows [0, ny) of this slab, or the slab
// is alone and wraps periodically in y.  EDGE = true: first / last tile row of a slab with
// neighbours; rows -2, -1, ny, ny+1 come from the halo buffers and the new edge rows are also
// packed for the neighbours.
//
// NT = threads per block.  256 (4 cells per thread in phase B, ~5 in phase A) keeps four blocks on a
// CU and is what a lattice that fills the chip wants.  A slab so small that each CU holds one block
// (1024 x 128 on one of 8 GPUs: 128 tiles for 256 CUs) is bound by that single block's load ->
// collide -> LDS -> collide -> store chain; 512 or 1024 threads per tile cut the serial work per
// thread to 2 or 1 cells per phase and give the CU's SIMDs 2 or 4 waves each to interleave.
// PARTIAL = the lattice does not tile exactly (lone slab only): east / north tiles are cut off.
template <int TX, int TY, int MODE, int KIND = kSweep2Plain, int NT = kBlock, bool PARTIAL = false>
__global__ __launch_bounds__(NT) void lbm_sweep2(const Sweep2Args a) {
  constexpr bool EDGE = (KIND == kSweep2Edge);
  constexpr int V = TX * TY / NT;                  // cells per thread in phase B
  static_assert(V * NT == TX * TY && (V == 4 || V == 2 || V == 1), "phase B: V = 4, 2 or 1 cells per thread");
  static_assert(TX % V == 0 && NT % 64 == 0, "whole waves, whole vectors per row");
  static_assert(TY >= 2, "rows 0,1 (and ny-2, ny-1) must sit in one tile row");
  constexpr bool FAST = (MODE & kFastMath) != 0;
  constexpr bool NTL = (MODE & kNtLoad) != 0, NTS = (MODE & kNtStore) != 0;
  constexpr int NW = NT / 64;
  constexpr int IW = TX + 2, IH = TY + 2;          // step-t+1 region: tile + ring
  constexpr int NA = (IW * IH + NT - 1) / NT;
  // LDS holds, per plane k, exactly the TX x TY values phase B will pull: the value f_k of region
  // cell P is wanted by the tile cell P + c_k only, so it is stored at THAT cell's coordinates (and
  // dropped if P + c_k falls outside the tile -- most components of the ring cells).  36 KB for a
  // 64 x 16 tile -> 4 blocks per CU, and every phase-B pull is an aligned ds_read_b128 at the
  // thread's own coordinates.
  __shared__ __attribute__((aligned(16))) float lds[9][TY][TX];
  __shared__ float red_f[NW];
  __shared__ float red_g[NW];
  __shared__ double red_d[NW];

  if (blockIdx.x == 0 && a.prev1 != nullptr) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < a.prev_count; i += NT) { s1 += (double)a.prev1[i]; s2 += (double)a.prev2[i]; }
    s1 = block_sum<double, NW>(s1, red_d);
    __syncthreads();
    s2 = block_sum<double, NW>(s2, red_d);
    if (threadIdx.x == 0) { a.prev_sum[0] = s1; a.prev_sum[1] = s2; }
    __syncthreads();
  }

  // XCD-aware tile order: block ids are dealt round-robin over the 8 XCDs, so give each XCD a
  // contiguous run of tiles -- x-neighbours, which share ring columns, then share an L2.
  const int ntx = (a.nx + TX - 1) / TX;   // a slab alone may end in partial tiles (slabs with neighbours tile exactly)
  int by, bx;
  bool edge = EDGE;            // this block's tile row borders a neighbouring slab
  if constexpr (KIND == kSweep2P2P) {
    // block ids [0, ntx): tile row 0; [ntx, 2 ntx): last tile row (if there is a second one);
    // the rest: interior tile rows.  Edge tiles get the lowest ids so they are dispatched first.
    const int nty = a.ny / TY;
    const int nedge = (nty >= 2 ? 2 : 1) * ntx;
    int b = blockIdx.x;
    if (b < nedge) {
      edge = true;
      by = (b < ntx) ? 0 : nty - 1;
      bx = (b < ntx) ? b : b - ntx;
    } else {
      b -= nedge;
      const int nint = (int)gridDim.x - nedge;
      if ((nint & 7) == 0) b = (b & 7) * (nint >> 3) + (b >> 3);
      int byi;
      tile_order<kTileGroup>(b, ntx, nty - 2, byi, bx);
      by = 1 + byi;
    }
    if (edge) {
      // wait for the halos of launch seq-1 before touching anything remote (see P2PSync)
      if (threadIdx.x == 0) {
        const uint32_t want = a.sync.seq - 1u;
        if (by == 0) p2p_wait(a.sync.flag_s, want, a.sync.err);
        if (by == nty - 1) p2p_wait(a.sync.flag_n, want, a.sync.err);
      }
      __syncthreads();
    }
  } else {
    const int nblk = gridDim.x;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);
    int byi;
    tile_order<kTileGroup>(b, ntx, a.by_count, byi, bx);
    by = a.by_begin + byi * a.by_stride;
  }
  const int X0 = bx * TX, Y0 = by * TY;
  const long P = a.plane;
  // Cells of this tile that exist in the lattice.  In a partial tile the region cells beyond the
  // ring (which sits right after the last real column / row and wraps to column / row 0) are
  // computed from wrapped, valid data and ignored.  Needs nx >= TX and ny >= TY (one wrap suffices).
  const int wx = (PARTIAL && a.nx - X0 < TX) ? a.nx - X0 : TX;
  const int hy = (PARTIAL && a.ny - Y0 < TY) ? a.ny - Y0 : TY;

  // ---- phase A: step t+1 on the (TX+2) x (TY+2) region -> LDS
  float q[NA][9];
  bool blk[NA];
  int cxs[NA], cys[NA];
  // The gather is instantiated twice with the edge decision as a compile-time constant, chosen
  // once per block: with a run-time test inside the cell loop the loads of different cells are
  // separated by branches and no longer issue back to back (8192^2: 502 -> 592 us per step).
  if (edge) sweep2_gather<TX, TY, NTL, true, NA, NT>(a, X0, Y0, q, blk, cxs, cys);
  else sweep2_gather<TX, TY, NTL, false, NA, NT>(a, X0, Y0, q, blk, cxs, cys);
  float sum1 = 0.f;
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    const int idx = threadIdx.x + m * NT;
    if (idx < IW * IH) {
      const int cx = cxs[m], cy = cys[m] & 0xffff;
      float sp = collide_cell<FAST>(q[m], blk[m], a.omega);
      if (cys[m] & 0x10000) accelerate_cell(q[m], blk[m], a.a1, a.a2);
      const bool own = (cx >= 1) && (cx <= wx) && (cy >= 1) && (cy <= hy);
      sum1 += own ? sp : 0.f;
      // tile coordinates of this region cell, and which neighbours exist inside the tile
      const int x0 = cx - 1, y0 = cy - 1;
      const bool xc = (x0 >= 0) && (x0 < TX), xe = (x0 + 1 >= 0) && (x0 + 1 < TX), xw = (x0 - 1 >= 0) && (x0 - 1 < TX);
      const bool yc = (y0 >= 0) && (y0 < TY), yn = (y0 + 1 >= 0) && (y0 + 1 < TY), ys = (y0 - 1 >= 0) && (y0 - 1 < TY);
      if (xc && yc) lds[0][y0][x0] = q[m][0];
      if (xe && yc) lds[1][y0][x0 + 1] = q[m][1];          // 1 = E: wanted by the cell to the east
      if (xc && yn) lds[2][y0 + 1][x0] = q[m][2];          // 2 = N
      if (xw && yc) lds[3][y0][x0 - 1] = q[m][3];          // 3 = W
      if (xc && ys) lds[4][y0 - 1][x0] = q[m][4];          // 4 = S
      if (xe && yn) lds[5][y0 + 1][x0 + 1] = q[m][5];      // 5 = NE
      if (xw && yn) lds[6][y0 + 1][x0 - 1] = q[m][6];      // 6 = NW
      if (xw && ys) lds[7][y0 - 1][x0 - 1] = q[m][7];      // 7 = SW
      if (xe && ys) lds[8][y0 - 1][x0 + 1] = q[m][8];      // 8 = SE
    }
  }
  __syncthreads();

  // ---- phase B: step t+2 on the tile, pulled from LDS; V consecutive cells per thread
  using RB = Row<V, false, NTS>;
  using RH = Row<V, false, false>;
  const int tx = threadIdx.x % (TX / V), ty = threadIdx.x / (TX / V);
  const int x = V * tx;
  const int gy = Y0 + ty;
  const bool live = !PARTIAL || ((x < wx) && (ty < hy));   // (nx % V == 0: a vector is inside or outside as a whole)
  const long rrow = live ? (long)gy * a.pitch : 0;
  float o[9][V];
#pragma unroll
  for (int k = 0; k < 9; ++k) RH::ld(&lds[k][ty][0], x, o[k]);   // aligned ds_read_b128 / b64 / b32
  bool ob[V];
  if (!live) {
#pragma unroll
    for (int v = 0; v < V; ++v) ob[v] = true;
  } else if constexpr (V == 4) {
    const uint32_t mb = *reinterpret_cast<const uint32_t*>(a.blocked + rrow + X0 + x);
    ob[0] = (mb & 0xffu) != 0; ob[1] = (mb & 0xff00u) != 0; ob[2] = (mb & 0xff0000u) != 0; ob[3] = (mb & 0xff000000u) != 0;
  } else if constexpr (V == 2) {
    const uint16_t mb = *reinterpret_cast<const uint16_t*>(a.blocked + rrow + X0 + x);
    ob[0] = (mb & 0xffu) != 0; ob[1] = (mb & 0xff00u) != 0;
  } else {
    ob[0] = a.blocked[rrow + X0 + x] != 0;
  }
  const bool do_accel = a.accel_out && (gy == a.accel_row);
  float sum2 = 0.f;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    float p[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) p[k] = o[k][v];
    sum2 += collide_cell<FAST>(p, ob[v], a.omega);
    if (do_accel) accelerate_cell(p, ob[v], a.a1, a.a2);
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k][v] = p[k];
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < 9; ++k) RB::st(a.dst + k * P + rrow, X0 + x, o[k]);
  }
  if (edge) {
    // pack the new edge rows for the neighbours (layout: kHaloSlots comment above)
    auto put = [&](float* buf, int slot, int k) {
      float* row = buf + (long)slot * a.nx;
      if constexpr (KIND == kSweep2P2P) {
#pragma unroll
        for (int v = 0; v < V; ++v) p2p_store(row + X0 + x + v, o[k][v]);
      } else {
        RH::st(row, X0 + x, o[k]);
      }
    };
    if (gy == 0) { put(a.send_s, 0, 0); put(a.send_s, 1, 1); put(a.send_s, 2, 3); put(a.send_s, 3, 4); put(a.send_s, 4, 7); put(a.send_s, 5, 8); }
    if (gy == 1) { put(a.send_s, 6, 4); put(a.send_s, 7, 7); put(a.send_s, 8, 8); }
    if (gy == a.ny - 1) { put(a.send_n, 0, 0); put(a.send_n, 1, 1); put(a.send_n, 2, 3); put(a.send_n, 3, 2); put(a.send_n, 4, 5); put(a.send_n, 5, 6); }
    if (gy == a.ny - 2) { put(a.send_n, 6, 2); put(a.send_n, 7, 5); put(a.send_n, 8, 6); }
  }

  const float b1 = block_sum<float, NW>(sum1, red_f);   // (contains a __syncthreads: all remote stores issued)
  const float b2 = block_sum<float, NW>(sum2, red_g);
  if (threadIdx.x == 0) { a.partials1[blockIdx.x] = b1; a.partials2[blockIdx.x] = b2; }
  if constexpr (KIND == kSweep2P2P) {
    if (edge) {
      // every wave's remote stores must have left before the block counts itself done
      p2p_drain();
      __syncthreads();
      if (threadIdx.x == 0) {
        const int nty = a.ny / TY;
        if (by == 0) p2p_signal(a.sync.cnt_s, a.sync.cnt_target_s, a.sync.rem_flag_s, a.sync.seq);
        if (by == nty - 1) p2p_signal(a.sync.cnt_n, a.sync.cnt_target_n, a.sync.rem_flag_n, a.sync.seq);
      }
    }
  }
}

// Peer-to-peer mode, outside the fused two-step launch: waits for the neighbours' halos of launch
// seq-1 (both sides), then -- if lat != nullptr -- packs the nine halo slots of a resident lattice
// straight into the neighbours' buffers and raises their flags with seq.  One block per 256 columns.
__global__ __launch_bounds__(kBlock) void lbm_p2p_push(const float* lat, long plane, int pitch, int nx, int nyl,
                                                       float* rem_s, float* rem_n, P2PSync sync, int do_push) {
  if (threadIdx.x == 0) {
    p2p_wait(sync.flag_s, sync.seq - 1u, sync.err);
    p2p_wait(sync.flag_n, sync.seq - 1u, sync.err);
  }
  __syncthreads();
  if (!do_push) return;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < nx) {
    const long r0 = x, r1 = (long)pitch + x;
    const long t1 = (long)(nyl - 1) * pitch + x, t2 = (long)(nyl - 2) * pitch + x;
    const int ks[6] = {0, 1, 3, 4, 7, 8}, kn[6] = {0, 1, 3, 2, 5, 6};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      p2p_store(rem_s + (long)i * nx + x, lat[ks[i] * plane + r0]);
      p2p_store(rem_n + (long)i * nx + x, lat[kn[i] * plane + t1]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      p2p_store(rem_s + (long)(6 + i) * nx + x, lat[ks[3 + i] * plane + r1]);
      p2p_store(rem_n + (long)(6 + i) * nx + x, lat[kn[3 + i] * plane + t2]);
    }
  }
  p2p_drain();
  __syncthreads();
  if (threadIdx.x == 0) {
    // every block serves both neighbours: it counts itself done on both sides
    p2p_signal(sync.cnt_s, sync.cnt_target_s, sync.rem_flag_s, sync.seq);
    p2p_signal(sync.cnt_n, sync.cnt_target_n, sync.rem_flag_n, sync.seq);
  }
}

// Peer-to-peer mode, marching launches: "launch group seq of this slab has finished" to both neighbours.  Runs on the
// slab's stream right behind the launch it speaks for (the kernel boundary has released that launch's stores).
__global__ void lbm_p2p_raise(uint32_t* rem_flag_s, uint32_t* rem_flag_n, uint32_t seq) {
  if (threadIdx.x == 0) {
    __hip_atomic_store(rem_flag_s, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(rem_flag_n, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Packs all nine halo slots of a resident lattice (start of a run).  Needs nyl >= 2.
__global__ void lbm_pack_halos9(const float* lat, long plane, int pitch, int nx, int nyl,
                                float* out_s, float* out_n) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nx) return;
  const long r0 = x, r1 = (long)pitch + x;
  const long t1 = (long)(nyl - 1) * pitch + x, t2 = (long)(nyl - 2) * pitch + x;
  const int ks[6] = {0, 1, 3, 4, 7, 8}, kn[6] = {0, 1, 3, 2, 5, 6};
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    out_s[(long)i * nx + x] = lat[ks[i] * plane + r0];
    out_n[(long)i * nx + x] = lat[kn[i] * plane + t1];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    out_s[(long)(6 + i) * nx + x] = lat[ks[3 + i] * plane + r1];
    out_n[(long)(6 + i) * nx + x] = lat[kn[3 + i] * plane + t2];
  }
}

// Ghost bands of the marching kernels under the RCCL transport: the K bottom rows of all nine planes packed for the
// southern neighbour (they become its northern band), the K top rows for the northern one.  out_* = [9][K][pitch].
__global__ void lbm_pack_band_rows(const float* lat, long plane, int pitch, int nyl, int K, float* out_s, float* out_n) {
  const long n = (long)K * pitch;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long top = (long)(nyl - K) * pitch;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    out_s[k * n + i] = lat[k * plane + i];
    out_n[k * n + i] = lat[k * plane + top + i];
  }
}

// Folds block partials into slab sums (after the last step of a run): block b folds the `count`
// floats at partials + b*stride into out[b] -- both steps of a final pair in one launch.
__global__ __launch_bounds__(kBlock) void lbm_fold_partials(const float* partials, int count, double* out, int stride) {
  __shared__ double red_d[kBlock / 64];
  const float* p = partials + (long)blockIdx.x * stride;
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) s += (double)p[i];
  s = block_sum<double>(s, red_d);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// Accelerate phase on one row of a resident lattice (first step of a run).
__global__ void lbm_accelerate_row(float* lat, long plane, int pitch, int nx, int row,
                                   const uint8_t* blocked, float a1, float a2) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nx) return;
  const long c = (long)row * pitch + x;
  if (blocked[c]) return;
  const float f3 = lat[3 * plane + c], f6 = lat[6 * plane + c], f7 = lat[7 * plane + c];
  if ((f3 - a1) > 0.f && (f6 - a2) > 0.f && (f7 - a2) > 0.f) {
    lat[1 * plane + c] += a1; lat[5 * plane + c] += a2; lat[8 * plane + c] += a2;
    lat[3 * plane + c] = f3 - a1; lat[6 * plane + c] = f6 - a2; lat[7 * plane + c] = f7 - a2;
  }
}

// Packs the halo rows of a resident lattice (start of a run: nothing has been
// sent yet).  out_s = planes 4,7,8 of row 0; out_n = planes 2,5,6 of row nyl-1.
__global__ void lbm_pack_halos(const float* lat, long plane, int pitch, int nx, int nyl,
                               float* out_s, float* out_n) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nx) return;
  const long top = (long)(nyl - 1) * pitch + x;
  out_s[x] = lat[4 * plane + x]; out_s[nx + x] = lat[7 * plane + x]; out_s[2 * nx + x] = lat[8 * plane + x];
  out_n[x] = lat[2 * plane + top]; out_n[nx + x] = lat[5 * plane + top]; out_n[2 * nx + x] = lat[6 * plane + top];
}

// Host layout <-> device layout.  aos = t_speed[rows*nx] (9 floats per cell).
__global__ void lbm_aos_to_soa(const float* aos, float* lat, long plane, int pitch, int nx, long ncell) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const long y = c / nx; const int x = (int)(c - y * nx);
  const long o = y * pitch + x;
#pragma unroll
  for (int k = 0; k < 9; ++k) lat[k * plane + o] = aos[9 * c + k];
}
__global__ void lbm_soa_to_aos(const float* lat, float* aos, long plane, int pitch, int nx, long ncell) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const long y = c / nx; const int x = (int)(c - y * nx);
  const long o = y * pitch + x;
#pragma unroll
  for (int k = 0; k < 9; ++k) aos[9 * c + k] = lat[k * plane + o];
}
__global__ void lbm_fill_equilibrium(float* lat, long plane, float w0, float w1, float w2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= plane) return;
  lat[i] = w0;
#pragma unroll
  for (int k = 1; k < 5; ++k) lat[k * plane + i] = w1;
#pragma unroll
  for (int k = 5; k < 9; ++k) lat[k * plane + i] = w2;
}
__global__ void lbm_pack_blocked(const int* obst, uint8_t* blocked, int pitch, int nx, long ncell) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const long y = c / nx; const int x = (int)(c - y * nx);
  blocked[y * pitch + x] = obst[c] ? 1 : 0;
}

// Derived fields of write_values() (d2q9-bgk.c:2935-2976) and the speed sum of
// av_velocity() (d2q9-bgk.c:2665-2714) in one pass.  out4 may be nullptr.
__global__ __launch_bounds__(kBlock) void lbm_derive(const float* lat, long plane, int pitch, int nx,
                                                     long ncell, const uint8_t* blocked, float density,
                                                     float* out4, float* partials, double* mass_partials) {
  __shared__ float red_f[kBlock / 64];
  __shared__ double red_d[kBlock / 64];
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float sp = 0.f;
  double mass = 0.0;
  if (c < ncell) {
    const long y = c / nx; const int x = (int)(c - y * nx);
    const long o = y * pitch + x;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = lat[k * plane + o];
    float rho = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) rho += f[k];
    mass = (double)rho;
    float ux = 0.f, uy = 0.f, u = 0.f, pr = density * (1.f / 3.f);
    if (!blocked[o]) {
      ux = (f[1] + f[5] + f[8] - (f[3] + f[6] + f[7])) / rho;
      uy = (f[2] + f[5] + f[6] - (f[4] + f[7] + f[8])) / rho;
      u = sqrtf(ux * ux + uy * uy);
      pr = rho * (1.f / 3.f);
      sp = u;
    }
    if (out4 != nullptr) {
      f4a v; v.x = ux; v.y = uy; v.z = u; v.w = pr;
      *reinterpret_cast<f4a*>(out4 + 4 * c) = v;
    }
  }
  const float bs = block_sum<float>(sp, red_f);
  const double bm = block_sum<double>(mass, red_d);
  if (threadIdx.x == 0) { partials[blockIdx.x] = bs; mass_partials[blockIdx.x] = bm; }
}

__global__ __launch_bounds__(kBlock) void lbm_fold_double(const double* in, int count, double* out) {
  __shared__ double red_d[kBlock / 64];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) s += in[i];
  s = block_sum<double>(s, red_d);
  if (threadIdx.x == 0) *out = s;
}

}  // namespace lbm
