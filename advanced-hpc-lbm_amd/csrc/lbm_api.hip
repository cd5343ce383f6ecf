// lbm_api.hip -- the C ABI of include/lbm_mi355x.h on top of the HIP kernels.
//
// Host-side structure (MI355X-first, nothing here mirrors the reference's
// serial layout):
//   * a context owns one or more ROW SLABS; each slab lives on one GPU with
//     both lattices resident in HBM as 9 SoA planes, a byte mask and nine-slot
//     halo buffers per launch parity;
//   * a launch group advances the lattice by two steps (lbm_sweep2, wherever
//     the lattice tiles) or by one (lbm_sweep);
//   * slabs with neighbours trade halos once per launch group, by one of three
//     transports: RCCL send/recv on a second stream between an edge launch and
//     the interior launch it overlaps (events join the streams); peer copies
//     inside one process; or peer-to-peer -- the edge tiles of ONE launch per
//     group store straight into the neighbour's halo block over xGMI and hand
//     off through flags polled in-kernel (run_p2p: no events, no host-side
//     exchange, no collective in the loop);
//   * no host synchronisation inside the step loop; per-step speed sums stay
//     on the device (one double per step and slab, folded from per-block
//     partials by block 0 of the NEXT launch) and are reduced once at the end
//     of the run.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lbm_mi355x.h"
#include "lbm_kernels.hip.h"
#include "lbm_march.hip.h"
#include "lbm_wave.hip.h"
#include "lbm_regtile.hip.h"

// ----------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIPC(call)                                                                         \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(LBM_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

extern "C" const char* lbm_last_error(void) { return g_err; }

// ----------------------------------------------------------------- RCCL (loaded on demand)
// Minimal declarations of the RCCL entry points used (rccl.h: ncclGetUniqueId,
// ncclCommInitRank, ncclCommInitAll, ncclSend/ncclRecv, ncclGroupStart/End,
// ncclAllReduce, ncclCommDestroy).  dlopen by SONAME so that inside a process
// that already carries RCCL (PyTorch) the same instance is shared.
namespace rccl {
typedef struct ncclComm* comm_t;
struct unique_id { char internal[128]; };
enum { kInt8 = 0, kFloat32 = 7, kFloat64 = 8, kSum = 0 };
static int (*GetUniqueId)(unique_id*);
static int (*CommInitRank)(comm_t*, int, unique_id, int);
static int (*CommInitAll)(comm_t*, int, const int*);
static int (*CommDestroy)(comm_t);
static int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t);
static int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t);
static int (*GroupStart)();
static int (*GroupEnd)();
static int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
static int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t);
static const char* (*GetErrorString)(int);
static void* handle = nullptr;

static std::mutex load_mutex;

static int load() {
  std::lock_guard<std::mutex> guard(load_mutex);
  if (handle) return LBM_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (handle) break;
  }
  if (!handle) return fail(LBM_ERCCL, "cannot load librccl: %s", dlerror());
#define SYM(var, name)                                                       \
  *(void**)(&var) = dlsym(handle, name);                                     \
  if (!var) return fail(LBM_ERCCL, "librccl lacks symbol %s", name);
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(AllReduce, "ncclAllReduce");
  SYM(AllGather, "ncclAllGather");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  return LBM_OK;
}
}  // namespace rccl

#define NCCLC(call)                                                                        \
  do {                                                                                     \
    int r_ = (call);                                                                       \
    if (r_ != 0)                                                                           \
      return fail(LBM_ERCCL, "%s failed: %s (%s:%d)", #call, rccl::GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

// ----------------------------------------------------------------- context
namespace {

struct Slab {
  int dev = 0;
  int row0 = 0, nyl = 0;       // global rows [row0, row0+nyl)
  int pitch = 0;               // floats per row
  long plane = 0;              // floats per plane
  float* lat[2] = {nullptr, nullptr};
  uint8_t* blocked = nullptr;
  uint8_t* blocked_gs = nullptr;   // blocked map of the row below the slab (global row0-1) ...
  uint8_t* blocked_gn = nullptr;   // ... and of the row above it (two-step kernel ring rows)
  // Halo buffers, nine slots of nx floats each (layout: lbm_kernels.hip.h, kHaloSlots), one pair
  // per launch parity.  One-step launches move slots 3..5 only.
  float* ghost_s[2] = {nullptr, nullptr};  // received from the south neighbour (its top rows)
  float* ghost_n[2] = {nullptr, nullptr};  // received from the north neighbour (its bottom rows)
  float* send_s[2] = {nullptr, nullptr};   // own rows 0, 1 packed for the south neighbour (RCCL / copy transports)
  float* send_n[2] = {nullptr, nullptr};   // own rows nyl-1, nyl-2 packed for the north neighbour
  float* partials[2] = {nullptr, nullptr};
  int partial_cap = 0;
  double* sums = nullptr;      // one double per step of the current run
  double* sums_host = nullptr; // pinned host copy of the same
  bool sums_direct = false;    // the kernels write the sums straight into sums_host (sums aliases it): no copy at the end of a run
  int sums_cap = 0;
  uint32_t* err_host = nullptr;  // pinned: copy of the peer-to-peer error word, fetched with the sums
  double* scratch_d = nullptr; // small double scratch (derive / reductions)
  int scratch_cap = 0;
  // sc: interior launches (and everything outside the step loop); se: edge launches, higher
  // priority, concurrent with the interior launch of the same step; sx: halo exchange
  hipStream_t sc = nullptr, se = nullptr, sx = nullptr;
  hipEvent_t ev_bnd[2] = {nullptr, nullptr}, ev_recv[2] = {nullptr, nullptr}, ev_int[2] = {nullptr, nullptr};
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
  hipEvent_t ev_march[2] = {nullptr, nullptr};   // "marching launch n of this slab has finished" (by launch parity)
  int accel_row = -1;          // local index of global row ny-2, or -1
  // Ghost bands (marching kernels under the RCCL transport): the K rows below / above the slab as the neighbours hold
  // them, received once per K steps.  band_*[i] accompanies lat[i]: [9 planes][K rows][pitch]; the obstacle bytes of
  // kBandRows rows either side are uploaded at create (row i of band_blk_s = global row row0 - kBandRows + i).
  float* band_s[2] = {nullptr, nullptr};
  float* band_n[2] = {nullptr, nullptr};
  float* band_send_s = nullptr;    // own bottom / top K rows, packed for the neighbours
  float* band_send_n = nullptr;
  uint8_t* band_blk_s = nullptr;
  uint8_t* band_blk_n = nullptr;
  int band_K = 0;              // K the float bands are sized for (0: none yet)
  rccl::comm_t comm = nullptr;
  // peer-to-peer halos (LBM_EXCHANGE_P2P): one uncached block holds ghost_s[2], ghost_n[2] and the
  // two flags the neighbours raise; the neighbours' blocks are mapped here (peer access or hipIpc)
  char* comm_block = nullptr;
  size_t halo_bytes = 0;       // one nine-slot halo buffer, rounded up to 256 B
  char* peer_s = nullptr;      // south / north neighbour's comm block as seen from this device
  char* peer_n = nullptr;
  bool peer_s_ipc = false, peer_n_ipc = false;
  uint32_t* counters = nullptr;  // device: [0] cnt_s, [16] cnt_n, [32] err (separate 64-B lines)
  // peer-to-peer marching launches read the neighbours' lattices in place: [0] = southern, [1] = northern neighbour
  const float* nb_lat[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [side][lattice 0 / 1]
  const uint8_t* nb_blocked[2] = {nullptr, nullptr};
  long nb_plane[2] = {0, 0};
  int nb_nyl[2] = {0, 0};
  void* nb_ipc[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};   // hipIpc mappings to close (rank mode)
  uint32_t cnt_s_total = 0, cnt_n_total = 0;
  // register tiles across slabs (lbm_regtile_slabs): this slab's mailboxes (uncached: the neighbours may store into them
  // over xGMI), its per-step tile sums, and the neighbours' mail areas as this device sees them ([0] south, [1] north)
  char* tmail = nullptr;
  size_t tmail_bytes = 0;
  float* rpartials = nullptr;
  long rpartials_cap = 0;          // in steps
  char* tmail_nb[2] = {nullptr, nullptr};
  size_t tmail_nb_bytes[2] = {0, 0};
  bool tmail_nb_ipc[2] = {false, false};
  uint32_t* rabort = nullptr;      // abort word of this slab's device group (owned by the group's first slab)
  hipEvent_t ev_rt = nullptr;      // "the group's launch is over" (recorded on the first slab's stream)
};

}  // namespace

struct lbm_ctx {
  lbm_param p;
  std::vector<Slab> slabs;     // slabs owned by THIS process
  int exchange = LBM_EXCHANGE_AUTO;  // resolved: 0 = none
  int cur = 0;                 // which lattice holds the current state
  bool rank_mode = false;
  int rank = 0, nranks = 1;    // position in the global ring (rank mode); else 0 / nslabs
  long tot_fluid = 0;          // non-blocked cells of the GLOBAL lattice
  int V = 1;                   // cells per thread
  long variant = 0;
  int time_block = 1;          // 2: fuse pairs of steps through LDS (lbm_sweep2) where eligible;
                               // 4: four steps per pass, row-marching (lbm_march), where eligible, then 2, then 1
  int t2_threads = 256;        // threads per tile of the two-step kernel (256 / 512 / 1024)
  int march_rows = 0;          // rows per chunk of the marching kernel (lbm_march, time_block = 4); 0 = not chosen yet
  int march_kernel = -1;       // which marching kernel runs time_block >= 4: 0 = lbm_march (one block per strip, LDS rings,
                               // K = 4, widths that are multiples of 4 from 256 up), 1 = lbm_wave (one wave per strip,
                               // register pipeline, K = 4 / 6 / 8, any width from 64 up), -1 = lbm_march where it can run
                               // (269 GLUPS at 8192^2 against 262 for lbm_wave<8>), lbm_wave elsewhere
  int wave_rows = 0;           // rows per chunk of lbm_wave; 0 = not chosen yet
  int wave_cols = 1;           // columns per lane of lbm_wave: 1 (a wave delivers 64 - 2K columns) or 2 (128 - 2K; K = 8; even widths from 128)
  int wave_capacity = 0;       // waves of lbm_wave<time_block> the device holds at once (occupancy query)
  uint32_t seq = 0;            // peer-to-peer: sequence number of the last launch group (same on all slabs)
  bool p2p_connected = false;
  bool no_comm = false;        // rank mode without RCCL: results are this rank's contribution
  bool p2p_failed = false;     // a peer-to-peer halo wait timed out: the lattice is no longer defined
  int march_slabs = -1;        // slabs of one process march too (neighbour rows read in place): -1 not decided, 0 no, 1 yes
  // engine: which kernel family lbm_run uses for a lattice alone on its GPU
  //   0 auto = the register-resident kernel (lbm_regtile) where the lattice tiles onto the CUs (the four
  //   shipped decks: 1.5-2x the streaming kernels), the streaming kernels elsewhere,
  //   1 streaming only (lbm_sweep2 / lbm_sweep / the marching kernels),
  //   3 resident in registers (lbm_regtile) or fail   (2 was the LDS-resident engine, removed in round 3)
  int engine = 0;
  int engine_last = 0;         // what the last lbm_run used: 1 streaming, 3 resident in registers
  // register-tile engine (engine 3): 64 x ty tiles, nw waves of r rows (ty == 0: none); bpc = blocks of this tiling a CU
  // takes by the occupancy query (0 = not asked yet, -1 = the query failed or the grid does not fit)
  struct { int ty = 0, r = 0, nw = 0, ntx = 0, nty = 0, bpc = 0; } tplan;
  unsigned long long* tmail = nullptr;   // its mailboxes
  float* rpartials = nullptr;  // [steps][tiles]
  long rpartials_cap = 0;      // in steps
  int rpartials_tiles = 0;     // tiles per step it was sized for
  uint32_t* rabort = nullptr;  // device abort word of the resident kernel
  int regtile_async = 1;       // lbm_regtile, R > 1: the loop's mail issued and waited for by hand (counted vmcnt), granules sent at once
                               // (0: the round-2 loop, compiler-scheduled loads and stores; R = 1 always runs that one)
  uint32_t rtag = 1;           // next unused mailbox tag (0 = never written); never goes back except when the mailboxes are cleared
  bool resident_broken = false;   // the resident kernel cannot run here (not every tile resident, set-up failed, or a run
                                  // gave up): stay with the streaming kernels
  char resident_why[160] = "";    // ... and why (lbm_last_error does not carry it: the run itself succeeds)
  // register tiles across slabs: the same 64 x ty tiling on every slab (ty == 0: none); nty = tile rows per slab
  struct { int ty = 0, r = 0, nw = 0, ntx = 0, nty = 0, bpc = 0; } splan;
  lbm::RegTileArgs* rtable = nullptr;   // pinned, device-mapped: one entry per local slab, grouped by device
  lbm::RegTileArgs* rtable_dev = nullptr;
  int ncu = 0;                 // CUs of slab 0's device
  double gpu_ms = 0.0, wall_ms = 0.0;
};

// tile of the two-step kernel
constexpr int kT2X = 64, kT2Y = 16;
// rows of obstacle bytes kept either side of a slab for the ghost bands of the marching kernels (the largest K)
constexpr int kBandRows = 8;

namespace {

int pick_vector_width(int nx) {
  const char* e = getenv("LBM_VECTOR_WIDTH");
  int want = e ? atoi(e) : 4;
  if (want >= 4 && nx % 4 == 0 && nx >= 8) return 4;
  if (want >= 2 && nx % 2 == 0 && nx >= 4) return 2;
  return 1;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Halo buffers of a slab for the context's exchange mode (callable again after a fallback from
// peer-to-peer to RCCL: slab_free_halos first).
void slab_free_halos(Slab& s) {
  (void)hipSetDevice(s.dev);
  for (int i = 0; i < 2; ++i) {
    if (s.ghost_s[i] && !s.comm_block) (void)hipFree(s.ghost_s[i]);
    if (s.ghost_n[i] && !s.comm_block) (void)hipFree(s.ghost_n[i]);
    if (s.send_s[i]) (void)hipFree(s.send_s[i]);
    if (s.send_n[i]) (void)hipFree(s.send_n[i]);
    s.ghost_s[i] = s.ghost_n[i] = s.send_s[i] = s.send_n[i] = nullptr;
  }
  if (s.blocked_gs) (void)hipFree(s.blocked_gs);
  if (s.blocked_gn) (void)hipFree(s.blocked_gn);
  s.blocked_gs = s.blocked_gn = nullptr;
  for (int i = 0; i < 2; ++i) {
    if (s.band_s[i]) (void)hipFree(s.band_s[i]);
    if (s.band_n[i]) (void)hipFree(s.band_n[i]);
    s.band_s[i] = s.band_n[i] = nullptr;
  }
  if (s.band_send_s) (void)hipFree(s.band_send_s);
  if (s.band_send_n) (void)hipFree(s.band_send_n);
  if (s.band_blk_s) (void)hipFree(s.band_blk_s);
  if (s.band_blk_n) (void)hipFree(s.band_blk_n);
  s.band_send_s = s.band_send_n = nullptr; s.band_blk_s = s.band_blk_n = nullptr; s.band_K = 0;
  for (int side = 0; side < 2; ++side)
    for (int i = 0; i < 3; ++i) {
      if (s.nb_ipc[side][i] && (side == 0 || s.nb_ipc[1][i] != s.nb_ipc[0][i])) (void)hipIpcCloseMemHandle(s.nb_ipc[side][i]);
    }
  for (int side = 0; side < 2; ++side) {
    for (int i = 0; i < 3; ++i) s.nb_ipc[side][i] = nullptr;
    s.nb_lat[side][0] = s.nb_lat[side][1] = nullptr; s.nb_blocked[side] = nullptr;
  }
  if (s.peer_s_ipc && s.peer_s) (void)hipIpcCloseMemHandle(s.peer_s);
  if (s.peer_n_ipc && s.peer_n && s.peer_n != s.peer_s) (void)hipIpcCloseMemHandle(s.peer_n);
  s.peer_s = s.peer_n = nullptr; s.peer_s_ipc = s.peer_n_ipc = false;
  if (s.comm_block) (void)hipFree(s.comm_block);
  if (s.counters) (void)hipFree(s.counters);
  s.comm_block = nullptr; s.counters = nullptr;
}

int slab_alloc_halos(lbm_ctx* c, Slab& s) {
  HIPC(hipSetDevice(s.dev));
  const int nx = c->p.nx;
  if (c->exchange == LBM_EXCHANGE_P2P) {
    HIPC(hipMalloc((void**)&s.blocked_gs, (size_t)nx));
    HIPC(hipMalloc((void**)&s.blocked_gn, (size_t)nx));
    s.halo_bytes = (sizeof(float) * lbm::kHaloSlots * (size_t)nx + 255) / 256 * 256;
    const size_t total = 4 * s.halo_bytes + 512 + 256;   // halos, two flag lines, the hipIpc handles of both lattices and the obstacle map
    // uncached: the neighbours write it over xGMI behind this GPU's L2.  Fine-grained memory is NOT
    // accepted as a substitute (the local L2 may keep ghost rows a neighbour has since rewritten):
    // without uncached memory peer-to-peer halos count as unavailable and the caller uses RCCL.
    hipError_t e = hipExtMallocWithFlags((void**)&s.comm_block, total, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); s.comm_block = nullptr; return fail(LBM_EHIP, "cannot allocate uncached halo memory: %s", hipGetErrorString(e)); }
    HIPC(hipMemset(s.comm_block, 0, total));
    HIPC(hipMalloc((void**)&s.counters, 64 * sizeof(uint32_t)));
    HIPC(hipMemset(s.counters, 0, 64 * sizeof(uint32_t)));
    HIPC(hipDeviceSynchronize());
    for (int i = 0; i < 2; ++i) {
      s.ghost_s[i] = (float*)(s.comm_block + (size_t)i * s.halo_bytes);
      s.ghost_n[i] = (float*)(s.comm_block + (size_t)(2 + i) * s.halo_bytes);
    }
    if (c->rank_mode && c->nranks > 1) {
      // a neighbouring PROCESS reads this slab's rows in place (marching launches): it finds the handles here
      hipIpcMemHandle_t h[3];
      HIPC(hipIpcGetMemHandle(&h[0], s.lat[0]));
      HIPC(hipIpcGetMemHandle(&h[1], s.lat[1]));
      HIPC(hipIpcGetMemHandle(&h[2], s.blocked));
      static_assert(sizeof(hipIpcMemHandle_t) == 64, "three handles in 192 bytes");
      HIPC(hipMemcpy(s.comm_block + 4 * s.halo_bytes + 512, h, sizeof(h), hipMemcpyHostToDevice));
    }
  } else {
    const size_t hb = sizeof(float) * lbm::kHaloSlots * (size_t)nx;
    HIPC(hipMalloc((void**)&s.blocked_gs, (size_t)nx));
    HIPC(hipMalloc((void**)&s.blocked_gn, (size_t)nx));
    HIPC(hipMalloc((void**)&s.band_blk_s, (size_t)kBandRows * s.pitch));
    HIPC(hipMalloc((void**)&s.band_blk_n, (size_t)kBandRows * s.pitch));
    for (int i = 0; i < 2; ++i) {
      HIPC(hipMalloc((void**)&s.ghost_s[i], hb));
      HIPC(hipMalloc((void**)&s.ghost_n[i], hb));
      HIPC(hipMalloc((void**)&s.send_s[i], hb));
      HIPC(hipMalloc((void**)&s.send_n[i], hb));
    }
  }
  return LBM_OK;
}

int slab_alloc(lbm_ctx* c, Slab& s, bool exchanging) {
  HIPC(hipSetDevice(s.dev));
  const int nx = c->p.nx;
  s.pitch = (nx + 63) / 64 * 64;
  // Plane stride: rows + 20.25 KiB.  A power-of-two stride (8192^2: exactly 256 MiB) puts the
  // same cell of all nine planes on the same HBM channel; a few KiB of padding spread the 18
  // concurrent streams (kbench: plain 9-plane copy 4.9 -> 5.4 TB/s, sweep 5.35 -> 5.6 TB/s with
  // 4 KiB; a sweep over pads on two boxes put 5 x 4 KiB + 256 B at or next to the best for the
  // copy, the one-step and the two-step kernels, 2-8 % ahead of 4 KiB).
  s.plane = (long)s.nyl * s.pitch + 5184;
  const size_t lat_bytes = sizeof(float) * 9 * (size_t)s.plane;
  for (int i = 0; i < 2; ++i) HIPC(hipMalloc((void**)&s.lat[i], lat_bytes));
  HIPC(hipMalloc((void**)&s.blocked, (size_t)s.plane));
  // one partial per block; worst case V = 1, one launch covering all rows (+2 for split launches)
  s.partial_cap = std::max({cdiv((long)s.nyl * nx, lbm::kBlock), 2 * cdiv(nx, kT2X) * cdiv(s.nyl, kT2Y),
                            4 * cdiv(nx, 224) * s.nyl,
                            8 * cdiv((long)cdiv(nx, 48) * cdiv(s.nyl, 16), 4)}) + 8;   // (last two: the marching kernels with small chunks)
  for (int i = 0; i < 2; ++i) HIPC(hipMalloc((void**)&s.partials[i], sizeof(float) * s.partial_cap));
  s.scratch_cap = s.partial_cap;
  HIPC(hipMalloc((void**)&s.scratch_d, sizeof(double) * (s.scratch_cap + 8)));
  if (exchanging) { int rc = slab_alloc_halos(c, s); if (rc) return rc; }
  HIPC(hipStreamCreateWithFlags(&s.sc, hipStreamNonBlocking));
  {
    int lo = 0, hi = 0;  // numerically lower = higher priority
    HIPC(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPC(hipStreamCreateWithPriority(&s.se, hipStreamNonBlocking, hi));
    HIPC(hipStreamCreateWithPriority(&s.sx, hipStreamNonBlocking, hi));
  }
  // everything that touches slab memory is ordered on s.sc (the streams are
  // non-blocking: a null-stream memset would race with the first kernels)
  HIPC(hipMemsetAsync(s.blocked, 0, (size_t)s.plane, s.sc));
  for (int i = 0; i < 2; ++i) {
    HIPC(hipEventCreateWithFlags(&s.ev_bnd[i], hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&s.ev_recv[i], hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&s.ev_int[i], hipEventDisableTiming));
  }
  HIPC(hipEventCreate(&s.ev_t0));
  HIPC(hipEventCreate(&s.ev_t1));
  for (int i = 0; i < 2; ++i) HIPC(hipEventCreateWithFlags(&s.ev_march[i], hipEventDisableTiming));
  return LBM_OK;
}

// Blocked maps of the two rows just outside the slab (periodic in the global lattice): the ring
// rows of the two-step kernel.
int upload_ghost_masks(lbm_ctx* c, Slab& s, const int* obstacles) {
  HIPC(hipSetDevice(s.dev));
  const int nx = c->p.nx, ny = c->p.ny;
  const int rs = (s.row0 + ny - 1) % ny, rn = (s.row0 + s.nyl) % ny;
  std::vector<uint8_t> gs(nx), gn(nx);
  for (int x = 0; x < nx; ++x) { gs[x] = obstacles[(long)rs * nx + x] ? 1 : 0; gn[x] = obstacles[(long)rn * nx + x] ? 1 : 0; }
  HIPC(hipMemcpyAsync(s.blocked_gs, gs.data(), nx, hipMemcpyHostToDevice, s.sc));
  HIPC(hipMemcpyAsync(s.blocked_gn, gn.data(), nx, hipMemcpyHostToDevice, s.sc));
  std::vector<uint8_t> bs, bn;
  if (s.band_blk_s) {
    // the kBandRows rows below and above the slab (periodic in the global lattice), for the ghost bands
    bs.assign((size_t)kBandRows * s.pitch, 0); bn.assign((size_t)kBandRows * s.pitch, 0);
    for (int i = 0; i < kBandRows; ++i) {
      const int gs_row = ((s.row0 - kBandRows + i) % ny + ny) % ny, gn_row = (s.row0 + s.nyl + i) % ny;
      for (int x = 0; x < nx; ++x) {
        bs[(size_t)i * s.pitch + x] = obstacles[(long)gs_row * nx + x] ? 1 : 0;
        bn[(size_t)i * s.pitch + x] = obstacles[(long)gn_row * nx + x] ? 1 : 0;
      }
    }
    HIPC(hipMemcpyAsync(s.band_blk_s, bs.data(), bs.size(), hipMemcpyHostToDevice, s.sc));
    HIPC(hipMemcpyAsync(s.band_blk_n, bn.data(), bn.size(), hipMemcpyHostToDevice, s.sc));
  }
  HIPC(hipStreamSynchronize(s.sc));
  return LBM_OK;
}

// A device staging buffer that is released on every path out of its scope.
struct DeviceTemp {
  void* p = nullptr;
  ~DeviceTemp() { if (p) (void)hipFree(p); }
};

// Uploads the slab's rows of the global host arrays and converts to the device layout.
int slab_upload(lbm_ctx* c, Slab& s, const int* obstacles, const float* cells) {
  HIPC(hipSetDevice(s.dev));
  const int nx = c->p.nx;
  const long ncell = (long)s.nyl * nx;
  const int grid = cdiv(ncell, 256);
  {
    DeviceTemp t;
    HIPC(hipMalloc(&t.p, sizeof(int) * ncell));
    int* d_ob = (int*)t.p;
    HIPC(hipMemcpyAsync(d_ob, obstacles + (long)s.row0 * nx, sizeof(int) * ncell, hipMemcpyHostToDevice, s.sc));
    hipLaunchKernelGGL(lbm::lbm_pack_blocked, dim3(grid), dim3(256), 0, s.sc, d_ob, s.blocked, s.pitch, nx, ncell);
    HIPC(hipGetLastError());
    if (s.blocked_gs) { int rc2 = upload_ghost_masks(c, s, obstacles); if (rc2) return rc2; }
    HIPC(hipStreamSynchronize(s.sc));
  }
  if (cells != nullptr) {
    DeviceTemp t;
    HIPC(hipMalloc(&t.p, sizeof(float) * 9 * ncell));
    float* d_aos = (float*)t.p;
    HIPC(hipMemcpyAsync(d_aos, cells + 9L * s.row0 * nx, sizeof(float) * 9 * ncell, hipMemcpyHostToDevice, s.sc));
    hipLaunchKernelGGL(lbm::lbm_aos_to_soa, dim3(grid), dim3(256), 0, s.sc, d_aos, s.lat[c->cur], s.plane, s.pitch, nx, ncell);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(s.sc));
  } else {
    // rest equilibrium, every cell (d2q9-bgk.c:2802-2823); padding columns get it too
    const float w0 = c->p.density * 4.f / 9.f, w1 = c->p.density / 9.f, w2 = c->p.density / 36.f;
    hipLaunchKernelGGL(lbm::lbm_fill_equilibrium, dim3(cdiv(s.plane, 256)), dim3(256), 0, s.sc, s.lat[c->cur], s.plane, w0, w1, w2);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(s.sc));
  }
  // the other lattice is fully overwritten by the first step (padding columns never read)
  return LBM_OK;
}

// Default kernel flavour for a context (overridable: LBM_VECTOR_WIDTH / LBM_KERNEL_VARIANT
// env, or lbm_set_option).  Measured on MI355X (tools/kbench, profiles/):
//   * lattices that stay resident in the 256 MiB Infinity Cache (both lattices of 1024^2
//     are 75 MB): 4 cells per thread, default cache policy -- nontemporal accesses bypass
//     the cache the next step would hit (12.9 us vs 15-19 us per step);
//   * lattices streamed from HBM (8192^2: 4.8 GB): 2 cells per thread (8 waves per SIMD)
//     with nontemporal loads and stores (862 us vs 887-940 us per step).
// v_rcp_f32 / v_sqrt_f32 (1 ulp) replace the IEEE divide and sqrt sequences in both.
void pick_defaults(lbm_ctx* c) {
  double bytes = 0;
  for (auto& s : c->slabs) bytes += 2.0 * 9 * sizeof(float) * (double)s.plane;
  const bool cache_resident = bytes / (double)c->slabs.size() <= 160.0 * 1024 * 1024;
  const int nx = c->p.nx;
  if (cache_resident) {
    c->V = pick_vector_width(nx);
    c->variant = lbm::kFastMath;
  } else {
    c->V = (nx % 2 == 0 && nx >= 4) ? 2 : 1;
    c->variant = lbm::kFastMath | lbm::kNtLoad | lbm::kNtStore;
  }
  // Two steps per pass through LDS: 1.4x (1024^2) to 1.65x (8192^2) over the single-step
  // sweep (kbench); nontemporal stores only pay off when the lattice streams from HBM.
  c->time_block = 2;
  // Threads per 64 x 16 tile of the two-step kernel (tools/t2_threads_check.py,
  // tools/strong_scaling_proxy.py): 512 (2 cells per thread and phase, 62 VGPRs, 32 waves per CU)
  // beats 256 on every lattice alone on a GPU -- 1024^2 8.7 -> 7.6 us/step, 128^2/256^2 4.1 -> 3.3,
  // 8192^2 equal -- and on slabs; a slab with at most one tile per CU (1024 x 128 on one of 8
  // GPUs) is bound by a single block's serial chain and does best with 1024 (4.6 -> 4.3 us/step).
  c->t2_threads = 512;
  if (c->exchange != 0) {
    long tiles = 0;
    for (auto& s : c->slabs) tiles = std::max(tiles, (long)(nx / kT2X) * (s.nyl / kT2Y));
    if (tiles <= 256) c->t2_threads = 1024;
  }
  const char* e;
  if ((e = getenv("LBM_VECTOR_WIDTH"))) c->V = pick_vector_width(nx);
  if ((e = getenv("LBM_KERNEL_VARIANT"))) c->variant = atol(e) & 15;
  if ((e = getenv("LBM_TIME_BLOCK"))) { const int v = atoi(e); c->time_block = (v == 8 || v == 6 || v == 4 || v == 2) ? v : 1; }
  if ((e = getenv("LBM_MARCH_KERNEL"))) c->march_kernel = atoi(e) == 0 ? 0 : atoi(e) == 1 ? 1 : -1;
  if ((e = getenv("LBM_WAVE_ROWS")) && atoi(e) > 0) c->wave_rows = std::min(atoi(e), c->p.ny);
  if ((e = getenv("LBM_WAVE_COLS"))) c->wave_cols = atoi(e) == 2 ? 2 : 1;
  if ((e = getenv("LBM_MARCH_ROWS")) && atoi(e) > 0) c->march_rows = std::min(atoi(e), c->p.ny);
  if ((e = getenv("LBM_T2_THREADS"))) { const int t = atoi(e); if (t == 256 || t == 512 || t == 1024) c->t2_threads = t; }
}

// The two-step kernel covers whole 64 x 16 tiles.  With neighbours every slab must tile too,
// and every rank must come to the same answer (the halo message size depends on it).
bool t2_eligible(const lbm_ctx* c) {
  if (c->time_block < 2 || (c->variant & 8)) return false;   // (variant bit 3: the reference's speed sum, one-step kernel only)
  // a slab alone: any lattice of at least one tile (partial tiles at the east / north end)
  if (c->exchange == 0) return c->slabs.size() == 1 && c->p.nx >= kT2X && c->p.ny >= kT2Y;
  return c->p.nx % kT2X == 0 && c->p.ny % (c->nranks * kT2Y) == 0;
}

// Steps per pass of the marching kernel (lbm_march.hip.h).
constexpr int kMarchK = 4;
// lbm_wave: columns per lane actually used (two need an even width of at least one 128-column strip and K = 8),
// and the columns a wave delivers
inline int wave_C(const lbm_ctx* c, int K) { return (c->wave_cols == 2 && K == 8 && c->p.nx % 2 == 0 && c->p.nx >= 128) ? 2 : 1; }
inline int wave_out_cols(const lbm_ctx* c, int K) { return 64 * wave_C(c, K) - 2 * K; }

// The marching kernel runs on a lattice alone on its GPU (periodic wrap inside the kernel); its row
// fetches are 16-byte LDS-DMA pieces, so columns must come in fours, and a strip is 256 columns wide.
inline bool march_block_ok(const lbm_ctx* c) {   // lbm_march's own requirements
  return c->time_block == kMarchK && c->p.nx % 4 == 0 && c->p.nx >= lbm::MarchCfg<kMarchK>::W && c->p.ny >= 2 * kMarchK;
}
inline bool use_wave_kernel(const lbm_ctx* c) { return c->march_kernel == 1 || (c->march_kernel < 0 && !march_block_ok(c)); }
bool march_eligible(const lbm_ctx* c) {
  if (c->time_block < 4 || c->exchange != 0 || c->slabs.size() != 1 || (c->variant & 8)) return false;
  if ((double)c->p.ny * c->slabs[0].pitch * 4.0 >= 4.0e9) return false;   // 32-bit byte offsets inside a plane
  // lbm_wave: any width of at least one wave; ny >= 2K because the kernel applies the accelerate phase at two
  // periodic images of row ny-2 per chunk (lbm_wave.hip.h: jacc, jacc2) and a chunk plus its 2K fill rows spans
  // up to ny + 2K rows: on a shorter lattice a third image would fall among them
  if (use_wave_kernel(c)) return c->p.nx >= 64 && c->p.ny >= 2 * c->time_block;
  return march_block_ok(c);
}

// Useful rows over rows of time for chunks of h rows: `rounds` rounds of full-height blocks on every CU,
// each paying its fill iterations.
double march_efficiency(const lbm_ctx* c, int h) {
  const int ns = cdiv(c->p.nx, lbm::MarchCfg<kMarchK>::WOUT), ncu = std::max(c->ncu, 1), ny = c->p.ny;
  const long blocks = (long)ns * cdiv(ny, h);
  const long rounds = (blocks + ncu - 1) / ncu;
  return (double)ny * ns / ((double)rounds * ncu * (h + 3 * (kMarchK - 1)));
}

int march_pick_rows(const lbm_ctx* c) {
  const int ny = c->p.ny;
  int best_h = std::min(ny, 256);
  double best = -1.0;
  for (int h = std::min(ny, 32); h <= std::min(ny, 1024); ++h) {
    const double eff = march_efficiency(c, h);
    if (eff > best + 1e-9) { best = eff; best_h = h; }
  }
  return best_h;
}

// Tiles of a lone slab (partial ones included).
inline int t2_tiles(const lbm_ctx* c, int nyl) { return cdiv(c->p.nx, kT2X) * cdiv(nyl, kT2Y); }

template <int MODE, int KIND, int NT>
void launch_sweep2_mkn(const lbm::Sweep2Args& a, int grid, hipStream_t st) {
  if constexpr (KIND == lbm::kSweep2Plain) {
    if (a.nx % kT2X != 0 || a.ny % kT2Y != 0) {   // lone slab that does not tile exactly
      hipLaunchKernelGGL((lbm::lbm_sweep2<kT2X, kT2Y, MODE, KIND, NT, true>), dim3(grid), dim3(NT), 0, st, a);
      return;
    }
  }
  hipLaunchKernelGGL((lbm::lbm_sweep2<kT2X, kT2Y, MODE, KIND, NT, false>), dim3(grid), dim3(NT), 0, st, a);
}

template <int MODE, int KIND>
void launch_sweep2_mk(const lbm_ctx* c, const lbm::Sweep2Args& a, int grid, hipStream_t st) {
  int nt = c->t2_threads;                      // threads per tile: see lbm_sweep2
  if (a.nx % (1024 / nt) != 0) nt = 1024;      // phase B moves 1024/nt cells per thread as one vector
  switch (nt) {
    case 1024: launch_sweep2_mkn<MODE, KIND, 1024>(a, grid, st); break;
    case 512: launch_sweep2_mkn<MODE, KIND, 512>(a, grid, st); break;
    default: launch_sweep2_mkn<MODE, KIND, 256>(a, grid, st); break;
  }
}

template <int KIND>
void launch_sweep2_k(const lbm_ctx* c, const lbm::Sweep2Args& a, int grid, hipStream_t st) {
  // cache-resident lattices: default policy; streamed lattices: nontemporal stores (kbench)
  switch ((int)(c->variant & (lbm::kFastMath | lbm::kNtStore))) {
    case 0: launch_sweep2_mk<0, KIND>(c, a, grid, st); break;
    case 1: launch_sweep2_mk<1, KIND>(c, a, grid, st); break;
    case 2: launch_sweep2_mk<2, KIND>(c, a, grid, st); break;
    default: launch_sweep2_mk<3, KIND>(c, a, grid, st); break;
  }
}

void launch_sweep2(const lbm_ctx* c, const lbm::Sweep2Args& a, int grid, hipStream_t st, bool edge) {
  if (edge) launch_sweep2_k<lbm::kSweep2Edge>(c, a, grid, st);
  else launch_sweep2_k<lbm::kSweep2Plain>(c, a, grid, st);
}

void slab_free(Slab& s) {
  slab_free_halos(s);
  (void)hipSetDevice(s.dev);
  for (int i = 0; i < 2; ++i) {
    if (s.lat[i]) (void)hipFree(s.lat[i]);
    if (s.partials[i]) (void)hipFree(s.partials[i]);
    if (s.ev_bnd[i]) (void)hipEventDestroy(s.ev_bnd[i]);
    if (s.ev_recv[i]) (void)hipEventDestroy(s.ev_recv[i]);
    if (s.ev_int[i]) (void)hipEventDestroy(s.ev_int[i]);
  }
  if (s.blocked) (void)hipFree(s.blocked);
  if (s.sums && !s.sums_direct) (void)hipFree(s.sums);
  if (s.sums_host) (void)hipHostFree(s.sums_host);
  if (s.err_host) (void)hipHostFree(s.err_host);
  if (s.scratch_d) (void)hipFree(s.scratch_d);
  for (int i = 0; i < 2; ++i) if (s.ev_march[i]) (void)hipEventDestroy(s.ev_march[i]);
  if (s.ev_t0) (void)hipEventDestroy(s.ev_t0);
  if (s.ev_t1) (void)hipEventDestroy(s.ev_t1);
  if (s.comm) (void)rccl::CommDestroy(s.comm);
  if (s.sc) (void)hipStreamDestroy(s.sc);
  if (s.se) (void)hipStreamDestroy(s.se);
  if (s.sx) (void)hipStreamDestroy(s.sx);
}

int check_params(const lbm_param* p) {
  if (!p) return fail(LBM_EINVAL, "params is NULL");
  if (p->nx < 1 || p->ny < 2) return fail(LBM_EINVAL, "lattice must be at least 1 x 2 (got %d x %d)", p->nx, p->ny);
  if ((long)p->nx * p->ny > (1L << 31) - 1) return fail(LBM_EINVAL, "lattice too large for int cell indices");
  return LBM_OK;
}

long count_fluid(const int* obstacles, long n) {
  long f = 0;
  for (long i = 0; i < n; ++i) f += obstacles[i] ? 0 : 1;
  return f;
}

// Launches one sweep over rows y_begin + i*y_stride of slab s.
template <int V, int MODE>
void launch_sweep_vm(const lbm::SweepArgs& a, hipStream_t st) {
  const long threads = (long)a.y_count * (a.nx / V);
  const int grid = cdiv(threads, lbm::kBlock);
  hipLaunchKernelGGL((lbm::lbm_sweep<V, MODE>), dim3(grid), dim3(lbm::kBlock), 0, st, a);
}

// variant = lbm::kFastMath | kNtStore | kNtLoad bits (option "kernel_variant"); bit 3 (value 8) = the step's average
// speed re-summed from the stored populations, the reference's form (lbm::kSpeedFromStored; one-step kernel only)
template <int V>
void launch_sweep_v(const lbm::SweepArgs& a, hipStream_t st, long variant) {
  constexpr int S = lbm::kSpeedFromStored;
  switch (variant & 15) {
    case 0: launch_sweep_vm<V, 0>(a, st); break;
    case 1: launch_sweep_vm<V, 1>(a, st); break;
    case 2: launch_sweep_vm<V, 2>(a, st); break;
    case 3: launch_sweep_vm<V, 3>(a, st); break;
    case 4: launch_sweep_vm<V, 4>(a, st); break;
    case 5: launch_sweep_vm<V, 5>(a, st); break;
    case 6: launch_sweep_vm<V, 6>(a, st); break;
    case 7: launch_sweep_vm<V, 7>(a, st); break;
    case 8: launch_sweep_vm<V, S | 0>(a, st); break;
    case 9: launch_sweep_vm<V, S | 1>(a, st); break;
    case 10: launch_sweep_vm<V, S | 2>(a, st); break;
    case 11: launch_sweep_vm<V, S | 3>(a, st); break;
    case 12: launch_sweep_vm<V, S | 4>(a, st); break;
    case 13: launch_sweep_vm<V, S | 5>(a, st); break;
    case 14: launch_sweep_vm<V, S | 6>(a, st); break;
    default: launch_sweep_vm<V, S | 7>(a, st); break;
  }
}

int sweep_blocks(const lbm_ctx* c, int y_count) {
  return cdiv((long)y_count * (c->p.nx / c->V), lbm::kBlock);
}

void launch_sweep(const lbm_ctx* c, const lbm::SweepArgs& a, hipStream_t st) {
  switch (c->V) {
    case 4: launch_sweep_v<4>(a, st, c->variant); break;
    case 2: launch_sweep_v<2>(a, st, c->variant); break;
    default: launch_sweep_v<1>(a, st, c->variant); break;
  }
}

// Halo exchange for parity q on the exchange streams (all local slabs): slots
// [slot0, slot0 + nslots) of the nine-slot halo buffers (lbm_kernels.hip.h, kHaloSlots).
int exchange_halos(lbm_ctx* c, int q, int slot0, int nslots) {
  const size_t off = (size_t)slot0 * c->p.nx;
  const size_t n = (size_t)nslots * c->p.nx;
  const int ns = (int)c->slabs.size();
  if (c->exchange == LBM_EXCHANGE_RCCL) {
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      HIPC(hipStreamWaitEvent(s.sx, s.ev_bnd[q], 0));
    }
    NCCLC(rccl::GroupStart());
    for (int i = 0; i < ns; ++i) {
      Slab& s = c->slabs[i];
      const int me = c->rank_mode ? c->rank : i;
      const int south = (me + c->nranks - 1) % c->nranks, north = (me + 1) % c->nranks;
      // order matters when south == north (2 ranks): sends S then N, receives N then S
      NCCLC(rccl::Send(s.send_s[q] + off, n, rccl::kFloat32, south, s.comm, s.sx));
      NCCLC(rccl::Send(s.send_n[q] + off, n, rccl::kFloat32, north, s.comm, s.sx));
      NCCLC(rccl::Recv(s.ghost_n[q] + off, n, rccl::kFloat32, north, s.comm, s.sx));
      NCCLC(rccl::Recv(s.ghost_s[q] + off, n, rccl::kFloat32, south, s.comm, s.sx));
    }
    NCCLC(rccl::GroupEnd());
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      HIPC(hipEventRecord(s.ev_recv[q], s.sx));
    }
  } else {  // LBM_EXCHANGE_COPY: the receiver pulls from its neighbours' send buffers
    for (int i = 0; i < ns; ++i) {
      Slab& s = c->slabs[i];
      Slab& so = c->slabs[(i + ns - 1) % ns];
      Slab& no = c->slabs[(i + 1) % ns];
      HIPC(hipSetDevice(s.dev));
      HIPC(hipStreamWaitEvent(s.sx, s.ev_bnd[q], 0));
      HIPC(hipStreamWaitEvent(s.sx, so.ev_bnd[q], 0));
      HIPC(hipStreamWaitEvent(s.sx, no.ev_bnd[q], 0));
      HIPC(hipMemcpyAsync(s.ghost_s[q] + off, so.send_n[q] + off, sizeof(float) * n, hipMemcpyDeviceToDevice, s.sx));
      HIPC(hipMemcpyAsync(s.ghost_n[q] + off, no.send_s[q] + off, sizeof(float) * n, hipMemcpyDeviceToDevice, s.sx));
      HIPC(hipEventRecord(s.ev_recv[q], s.sx));
    }
  }
  return LBM_OK;
}

// Per-step sums of a run.  Sized once at lbm_create for the deck's own maxIters and grown
// geometrically, so that a run never allocates unless it is longer than anything before it (an
// allocation inside lbm_run costs more than a 20-step run of 1024^2).  Where no collective reduces
// them on the device (every context but a rank of a process-per-GPU job), the folding blocks store
// each step's double straight into pinned host memory -- one posted 8-byte write per step -- and a
// run ends without a device-to-host copy (two queued copies cost ~10 us of a 150-us 20-step run);
// a rank keeps a device array (RCCL reduces it) and a pinned staging copy.
int ensure_sums(Slab& s, int nsteps) {
  if (s.sums_cap >= nsteps) return LBM_OK;
  HIPC(hipSetDevice(s.dev));
  int cap = std::max(1024, s.sums_cap);
  while (cap < nsteps) cap = (cap > (1 << 29)) ? nsteps : cap * 2;
  if (s.sums && !s.sums_direct) HIPC(hipFree(s.sums));
  if (s.sums_host) HIPC(hipHostFree(s.sums_host));
  s.sums = nullptr; s.sums_host = nullptr; s.sums_cap = 0;
  HIPC(hipHostMalloc((void**)&s.sums_host, sizeof(double) * cap, hipHostMallocPortable | hipHostMallocMapped));
  if (s.sums_direct) HIPC(hipHostGetDevicePointer((void**)&s.sums, s.sums_host, 0));
  else HIPC(hipMalloc((void**)&s.sums, sizeof(double) * cap));
  if (!s.err_host) {
    HIPC(hipHostMalloc((void**)&s.err_host, 64, hipHostMallocPortable | hipHostMallocMapped));
    memset(s.err_host, 0, 64);
  }
  s.sums_cap = cap;
  return LBM_OK;
}

bool plan_regtile(lbm_ctx* c);    // resident engine, below
bool plan_regtile_slabs(lbm_ctx* c);
bool regtile_tiling_rule(int nx, int rows, int per_dev, int ncu, int* ty_out, int* r_out);
struct Slab;
int regtile_slab_mail_alloc(lbm_ctx* c, Slab& s);
size_t regtile_slab_mail_bytes(const lbm_ctx* c);
typedef void (*wave_fn)(const lbm::WaveArgs);
wave_fn wave_kernel(int K, bool slab, int C, int flavour);   // the lbm_wave instantiations, below
bool march_slabs_setup(lbm_ctx* c);   // marching kernel across slabs, below
int wave_blocks_per_cu(int K, int C);        // occupancy of lbm_wave<K, ., ., C>, below
bool p2p_march_pays(const lbm_ctx* c);
bool slab_wave_pays(const lbm_ctx* c, int rows, int K = 8);
int slab_wave_rows(const lbm_ctx* c, int ny_rows, int K = 8, int extra_waves = 0);
double slab_wave_efficiency(const lbm_ctx* c, int ny_rows, int h, int K = 8, int extra_waves = 0);
int march_rows_for(const lbm_ctx* c, int ny_rows);
int wave_slots(const lbm_ctx* c, int K);

// lbm_wave<8>: one or two columns per lane?  Each form is priced by what it does with a full chip -- 394 GLUPS with one
// column (three waves per SIMD, 48 of 64 lanes delivered), 456 with two (two waves per SIMD, 112 of 128 delivered, the two
// cells of a lane issued statement by statement; both figures rose by 6 / 11 % with the 70-instruction collision, the
// measurements quoted below are from before it) -- times the useful share of its wave-slot time with its best chunk
// height on `rows` rows.  Measured in one call (profiles/r03_wave_two_columns.log): 8192^2 348 against 320 GLUPS (149-row
// chunks: 4070 waves for 2 x 2048 slots), 6144^2 324 against 314, 4096^2 219 against 274 (too few waves for two rounds);
// 8192-wide slabs of N = 2 / 4 / 8: 113 / 62.8 / 35.5 us per step against 119 / 66.7 / 37.4.  What the two-column form wants
// is a chunk height that fills WHOLE rounds of its 2048 wave slots -- one round is as good as two: 4096^2 with 75-row chunks
// (2035 waves) 321 GLUPS, with 74-row chunks (2072 waves: a second round for 24 of them) 208; lbm_march there: 278.
// Sets c->wave_cols (unless the caller fixed it) and returns the predicted rate of the form chosen.
double wave_pick_cols(lbm_ctx* c, int rows, int* rows_per_chunk) {
  const bool fixed = getenv("LBM_WAVE_COLS") != nullptr;
  double best = -1.0;
  int best_c = c->wave_cols, best_h = 0;
  for (int cols : {1, 2}) {
    if (fixed && cols != c->wave_cols) continue;
    if (cols == 2 && !(c->p.nx % 2 == 0 && c->p.nx >= 128)) continue;
    const int was = c->wave_cols;
    c->wave_cols = cols;
    const int h = slab_wave_rows(c, rows, 8);
    const double rate = (cols == 2 ? 456.0 : 394.0) * slab_wave_efficiency(c, rows, h, 8);
    c->wave_cols = was;
    if (rate > best) { best = rate; best_c = cols; best_h = h; }
  }
  c->wave_cols = best_c;
  if (rows_per_chunk) *rows_per_chunk = best_h;
  return best;
}

int finish_create(lbm_ctx* c, const int* obstacles, const float* cells) {
  const bool exchanging = c->exchange != 0;
  const int ay = c->p.ny - 2;  // the accelerate row of the global lattice (d2q9-bgk.c:240)
  for (auto& s : c->slabs) {
    s.accel_row = (ay >= s.row0 && ay < s.row0 + s.nyl) ? ay - s.row0 : -1;
    int rc = slab_alloc(c, s, exchanging);
    if (rc) return rc;
    rc = slab_upload(c, s, obstacles, cells);
    if (rc) return rc;
    s.sums_direct = !c->rank_mode;
    rc = ensure_sums(s, std::max(c->p.maxIters, 1));
    if (rc) return rc;
  }
  pick_defaults(c);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->slabs[0].dev) == hipSuccess) c->ncu = prop.multiProcessorCount;
    else (void)hipGetLastError();
    const char* e = getenv("LBM_ENGINE");
    if (e) c->engine = (atoi(e) == 0 || atoi(e) == 1 || atoi(e) == 3) ? atoi(e) : 0;
    if ((e = getenv("LBM_REGTILE_ASYNC"))) c->regtile_async = atoi(e) ? 1 : 0;
    if (!exchanging && c->slabs.size() == 1) plan_regtile(c);
    if (exchanging && plan_regtile_slabs(c) && c->rank_mode && c->nranks > 1) {
      // one process per GPU: the neighbours find this slab's mail area through a hipIpc handle in its halo block
      Slab& s0 = c->slabs[0];
      int rc = regtile_slab_mail_alloc(c, s0);
      if (rc) { (void)hipGetLastError(); c->splan.ty = 0; }
      else {
        hipIpcMemHandle_t h;
        if (hipIpcGetMemHandle(&h, s0.tmail) != hipSuccess) { (void)hipGetLastError(); c->splan.ty = 0; }
        else HIPC(hipMemcpy(s0.comm_block + 4 * s0.halo_bytes + 512 + 192, &h, sizeof(h), hipMemcpyHostToDevice));
      }
    }
    // Four steps per pass (lbm_march) where its strips and chunks fill the chip: measured 1.5-1.6x
    // lbm_sweep2 from 2048^2 up (194 / 226 / 238 GLUPS at 2048^2 / 4096^2 / 8192^2 against 129 / 146 /
    // 150), 0.8x at 1024^2, where 5 strips x 32-row chunks leave the CUs 49 % busy by the estimate
    // below (2048^2: 88 %).
    if (!getenv("LBM_TIME_BLOCK") && c->time_block == 2 && exchanging && c->exchange == LBM_EXCHANGE_P2P) {
      // peer-to-peer halos (one process or one process per GPU): lbm_march where the smallest slab fills the chip
      // (a function of the lattice and the number of slabs only: every rank decides alike)
      if (c->p.nx % 4 == 0 && c->p.nx >= lbm::MarchCfg<kMarchK>::W && c->p.ny / c->nranks >= 4 * kMarchK && p2p_march_pays(c)) c->time_block = 4;
      if (c->time_block == 4 && c->march_kernel != 0 && !getenv("LBM_MARCH_KERNEL")) {
        // lbm_wave<8> (one or two columns per lane, whichever the model prices higher) where its waves fill a round of the chip
        const int rows = c->p.ny / c->nranks;
        wave_pick_cols(c, rows, nullptr);
        if (slab_wave_pays(c, rows, 8)) c->time_block = 8; else c->wave_cols = 1;
      }
    } else
    if (!getenv("LBM_TIME_BLOCK") && c->time_block == 2 && exchanging && c->exchange == LBM_EXCHANGE_RCCL) {
      // RCCL halos (one process per GPU, or one process with a slab per GPU): lbm_wave<8> with ghost bands -- K rows of all
      // nine planes per direction per K steps -- where the smallest slab fills the chip's wave slots (every rank decides alike)
      // (one column per lane: with the slab cut into an edge launch and an interior launch the two-column form's fewer, fatter
      // waves measured behind -- ring of one, 8192-wide slabs of N = 1 / 2 / 4 / 8: 256 / 134 / 71.0 / 40.4 us per step against
      // 219 / 122 / 68.3 / 39.1)
      if (c->march_kernel != 0 && slab_wave_pays(c, c->p.ny / c->nranks)) c->time_block = 8;
    } else
    if (!getenv("LBM_TIME_BLOCK") && c->time_block == 2 && exchanging) {
      // slabs of one process: lbm_march with the neighbours' rows read in place, where every slab fills the chip
      c->time_block = 4;
      bool ok = c->march_kernel != 1 && march_slabs_setup(c);
      for (auto& s : c->slabs) {
        if (!ok) break;
        const int h = march_rows_for(c, s.nyl), ns = cdiv(c->p.nx, lbm::MarchCfg<kMarchK>::WOUT);
        const long blocks = (long)ns * cdiv(s.nyl, h), rounds = (blocks + c->ncu - 1) / std::max(c->ncu, 1);
        if ((double)s.nyl * ns / ((double)rounds * std::max(c->ncu, 1) * (h + 3 * (kMarchK - 1))) < 0.65) ok = false;
      }
      if (!ok) { c->time_block = 2; c->march_slabs = -1; }
      else if (c->march_kernel != 0 && !getenv("LBM_MARCH_KERNEL")) {
        int smallest = c->p.ny;
        for (auto& s : c->slabs) smallest = std::min(smallest, s.nyl);
        wave_pick_cols(c, smallest, nullptr);
        bool w8 = true;
        for (auto& s : c->slabs) w8 = w8 && slab_wave_pays(c, s.nyl);
        if (w8) c->time_block = 8; else c->wave_cols = 1;
      }
    } else
    if (!getenv("LBM_TIME_BLOCK") && c->time_block == 2) {
      c->time_block = 4;
      if (!march_eligible(c)) c->time_block = 2;
      else if (!use_wave_kernel(c)) { if (march_efficiency(c, march_pick_rows(c)) < 0.65) c->time_block = 2; }
      else if ((long)c->p.nx * c->p.ny < (3L << 20)) c->time_block = 2;   // lbm_wave needs a few thousand waves: from about 2048^2
      else c->time_block = 6;
      // Eight steps per pass in registers (lbm_wave<8>, 10.5 B per update, bound by its arithmetic) against lbm_march
      // (19.4 B, bound by HBM): each priced by what it does with a full chip -- 371 and 300 GLUPS -- times the share of
      // its slot-iterations that are useful with the best chunk height.  Measured in one call, lbm_wave<8> with the
      // chunk height of the model / lbm_march: 8192^2 328 / 284, 7168^2 329 / 299, 6144^2 325 / 302, 5120^2 310 / 294
      // (91-row chunks = 1.99 rounds; 87 rows = 2.06 rounds: 282), 4608^2 308 / 295, 4096^2 287 / 280, 3072^2 265 / 259,
      // 2048^2 201 / 230.
      if (c->time_block == 4 && c->march_kernel < 0 && !getenv("LBM_MARCH_KERNEL") && c->p.nx >= 64 && c->p.ny >= 32 &&
          (double)c->p.ny * c->slabs[0].pitch * 4.0 < 4.0e9) {
        int h = 0;
        const int cols_was = c->wave_cols;
        // (x 0.88: for a single short round the wave-slot model is about a tenth too pessimistic -- 2048^2 predicted 205, measured
        // 247, lbm_march 231; 2560^2 254 / 277-281 / 254 -- profiles/r03_default_kernel_by_size.log)
        if (wave_pick_cols(c, c->p.ny, &h) >= 0.88 * 300.0 * march_efficiency(c, march_pick_rows(c))) {
          c->time_block = 8; c->march_kernel = 1;
          if (c->wave_rows <= 0) c->wave_rows = h;
        } else c->wave_cols = cols_was;
      }
    }
  }
  return LBM_OK;
}

}  // namespace

// ----------------------------------------------------------------- C ABI
// The tiling lbm_regtile would use: host arithmetic only (no device is touched), so that the rule can be tested where there
// is no GPU (tests/test_abi.py).
extern "C" int lbm_plan_tiles(int nx, int rows, int slabs_per_device, int compute_units, int* tile_rows, int* rows_per_wave) {
  if (!tile_rows || !rows_per_wave) return fail(LBM_EINVAL, "NULL argument");
  int ty = 0, r = 0;
  if (!regtile_tiling_rule(nx, rows, slabs_per_device, compute_units, &ty, &r))
    return fail(LBM_EINVAL, "no register tiling: %d columns x %d rows, %d slab(s) per device, %d compute units", nx, rows, slabs_per_device, compute_units);
  *tile_rows = ty; *rows_per_wave = r;
  return LBM_OK;
}

extern "C" int lbm_device_count(int* count) {
  if (!count) return fail(LBM_EINVAL, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
  *count = n;
  return LBM_OK;
}

extern "C" int lbm_rccl_unique_id(void* id128) {
  if (!id128) return fail(LBM_EINVAL, "id buffer is NULL");
  int rc = rccl::load();
  if (rc) return rc;
  rccl::unique_id id;
  NCCLC(rccl::GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return LBM_OK;
}

namespace {

// Peer-to-peer: neighbour comm blocks of slabs that live in THIS process.
int p2p_connect_local(lbm_ctx* c) {
  const int ns = (int)c->slabs.size();
  for (int i = 0; i < ns; ++i) {
    Slab& s = c->slabs[i];
    Slab& so = c->slabs[(i + ns - 1) % ns];
    Slab& no = c->slabs[(i + 1) % ns];
    for (Slab* o : {&so, &no}) {
      if (o->dev == s.dev) continue;
      int can = 0;
      HIPC(hipDeviceCanAccessPeer(&can, s.dev, o->dev));
      if (!can) return fail(LBM_EHIP, "device %d cannot access device %d peer-to-peer", s.dev, o->dev);
      HIPC(hipSetDevice(s.dev));
      hipError_t e = hipDeviceEnablePeerAccess(o->dev, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        return fail(LBM_EHIP, "hipDeviceEnablePeerAccess(%d -> %d): %s", s.dev, o->dev, hipGetErrorString(e));
      (void)hipGetLastError();
    }
    s.peer_s = so.comm_block;
    s.peer_n = no.comm_block;
    Slab* nb[2] = {&so, &no};
    for (int side = 0; side < 2; ++side) {
      s.nb_lat[side][0] = nb[side]->lat[0]; s.nb_lat[side][1] = nb[side]->lat[1];
      s.nb_blocked[side] = nb[side]->blocked; s.nb_plane[side] = nb[side]->plane; s.nb_nyl[side] = nb[side]->nyl;
    }
  }
  c->p2p_connected = true;
  return LBM_OK;
}

// Peer-to-peer, one process per GPU: map the two neighbours' comm blocks from their hipIpc handles.
int p2p_connect_ipc(lbm_ctx* c, const char* handles, int nranks) {
  if (nranks != c->nranks) return fail(LBM_EINVAL, "expected %d handles, got %d", c->nranks, nranks);
  Slab& s = c->slabs[0];
  HIPC(hipSetDevice(s.dev));
  const int south = (c->rank + nranks - 1) % nranks, north = (c->rank + 1) % nranks;
  auto open = [&](int r, char** out, bool* ipc) -> int {
    if (r == c->rank) { *out = s.comm_block; *ipc = false; return LBM_OK; }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)r * LBM_P2P_HANDLE_BYTES, sizeof(h));
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(LBM_EHIP, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e)); }
    *out = (char*)ptr; *ipc = true;
    return LBM_OK;
  };
  int rc = open(south, &s.peer_s, &s.peer_s_ipc);
  if (rc) return rc;
  if (north == south) { s.peer_n = s.peer_s; s.peer_n_ipc = false; }
  else if ((rc = open(north, &s.peer_n, &s.peer_n_ipc))) return rc;
  // the neighbours' lattices and obstacle maps, for the marching launches (their handles sit in their halo blocks)
  const int nbr[2] = {south, north};
  char* blocks[2] = {s.peer_s, s.peer_n};
  for (int side = 0; side < 2; ++side) {
    const int r = nbr[side];
    const int r0 = (int)((long)r * c->p.ny / nranks), nyl = (int)((long)(r + 1) * c->p.ny / nranks) - r0;
    s.nb_nyl[side] = nyl;
    s.nb_plane[side] = (long)nyl * s.pitch + 5184;       // (slab_alloc's rule)
    if (r == c->rank) {
      s.nb_lat[side][0] = s.lat[0]; s.nb_lat[side][1] = s.lat[1]; s.nb_blocked[side] = s.blocked;
    } else if (side == 1 && north == south) {
      for (int i = 0; i < 3; ++i) s.nb_ipc[1][i] = s.nb_ipc[0][i];
      s.nb_lat[1][0] = s.nb_lat[0][0]; s.nb_lat[1][1] = s.nb_lat[0][1]; s.nb_blocked[1] = s.nb_blocked[0];
    } else {
      hipIpcMemHandle_t h[3];
      HIPC(hipMemcpy(h, blocks[side] + 4 * s.halo_bytes + 512, sizeof(h), hipMemcpyDeviceToHost));
      for (int i = 0; i < 3; ++i) {
        void* ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, h[i], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(LBM_EHIP, "hipIpcOpenMemHandle(lattice of rank %d): %s", r, hipGetErrorString(e)); }
        s.nb_ipc[side][i] = ptr;
      }
      s.nb_lat[side][0] = (const float*)s.nb_ipc[side][0]; s.nb_lat[side][1] = (const float*)s.nb_ipc[side][1];
      s.nb_blocked[side] = (const uint8_t*)s.nb_ipc[side][2];
    }
    // the neighbour's mail area (register tiles across slabs): a fourth handle behind the three, zero if it has none
    if (c->splan.ty > 0 && r != c->rank) {
      if (side == 1 && north == south) { s.tmail_nb[1] = s.tmail_nb[0]; s.tmail_nb_bytes[1] = s.tmail_nb_bytes[0]; s.tmail_nb_ipc[1] = false; }
      else {
        hipIpcMemHandle_t h, zero;
        memset(&zero, 0, sizeof(zero));
        HIPC(hipMemcpy(&h, blocks[side] + 4 * s.halo_bytes + 512 + 192, sizeof(h), hipMemcpyDeviceToHost));
        void* ptr = nullptr;
        if (memcmp(&h, &zero, sizeof(h)) != 0 && hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess) {
          s.tmail_nb[side] = (char*)ptr; s.tmail_nb_bytes[side] = regtile_slab_mail_bytes(c); s.tmail_nb_ipc[side] = true;
        } else { (void)hipGetLastError(); c->splan.ty = 0; }      // (no mail area over there, or not mappable: the streaming kernels)
      }
    }
  }
  c->p2p_connected = true;
  return LBM_OK;
}

int p2p_export(lbm_ctx* c, char* handle64) {
  Slab& s = c->slabs[0];
  if (!s.comm_block) return fail(LBM_EINVAL, "context has no peer-to-peer halo block");
  HIPC(hipSetDevice(s.dev));
  hipIpcMemHandle_t h;
  HIPC(hipIpcGetMemHandle(&h, s.comm_block));
  memcpy(handle64, &h, LBM_P2P_HANDLE_BYTES);
  return LBM_OK;
}

int create_fail(lbm_ctx* c, int rc) {
  std::string keep = g_err;
  lbm_destroy(c);
  snprintf(g_err, sizeof(g_err), "%s", keep.c_str());
  return rc;
}

}  // namespace

extern "C" int lbm_create(const lbm_param* params, const int* obstacles, const float* cells,
                          int nslabs, const int* devices, int exchange, lbm_ctx** out) {
  if (!out) return fail(LBM_EINVAL, "out is NULL");
  *out = nullptr;
  int rc = check_params(params);
  if (rc) return rc;
  if (!obstacles) return fail(LBM_EINVAL, "obstacles is NULL");
  if (nslabs < 1 || nslabs > params->ny) return fail(LBM_EINVAL, "nslabs must be in [1, ny] (got %d)", nslabs);
  if (exchange < 0 || exchange > LBM_EXCHANGE_P2P) return fail(LBM_EINVAL, "unknown exchange mode %d", exchange);
  int ndev = 0;
  lbm_device_count(&ndev);
  if (ndev < 1) return fail(LBM_ENODEV, "no HIP device visible; this library has no CPU path");

  lbm_ctx* c = new lbm_ctx();
  c->p = *params;
  c->nranks = nslabs;
  const char* force = getenv("LBM_FORCE_EXCHANGE");
  if (nslabs == 1 && !(force && atoi(force)))
    c->exchange = 0;
  else if (exchange == LBM_EXCHANGE_AUTO) {
    // RCCL wants one rank per device (ncclCommInitAll rejects a repeated device): several slabs
    // on one GPU trade their halos with peer copies instead
    bool repeated = false;
    for (int i = 0; i < nslabs && !repeated; ++i)
      for (int j = 0; j < i; ++j)
        if ((devices ? devices[i] : i) == (devices ? devices[j] : j)) { repeated = true; break; }
    c->exchange = repeated ? LBM_EXCHANGE_COPY : LBM_EXCHANGE_RCCL;
  } else {
    c->exchange = exchange;
  }
  if (c->exchange == LBM_EXCHANGE_P2P && params->ny / nslabs < 2) {
    delete c;
    return fail(LBM_EINVAL, "peer-to-peer halos need at least 2 rows per slab");
  }
  c->tot_fluid = count_fluid(obstacles, (long)params->nx * params->ny);
  c->slabs.resize(nslabs);
  std::vector<int> devs(nslabs);
  for (int i = 0; i < nslabs; ++i) {
    Slab& s = c->slabs[i];
    s.dev = devices ? devices[i] : i;
    devs[i] = s.dev;
    if (s.dev < 0 || s.dev >= ndev) {
      delete c;
      return fail(LBM_ENODEV, "slab %d wants HIP device %d but only %d visible", i, s.dev, ndev);
    }
    s.row0 = (int)((long)i * params->ny / nslabs);
    s.nyl = (int)((long)(i + 1) * params->ny / nslabs) - s.row0;
  }
  rc = finish_create(c, obstacles, cells);
  if (!rc && c->exchange == LBM_EXCHANGE_RCCL) {
    rc = rccl::load();
    if (!rc) {
      std::vector<rccl::comm_t> comms(nslabs);
      int r = rccl::CommInitAll(comms.data(), nslabs, devs.data());
      if (r != 0) rc = fail(LBM_ERCCL, "ncclCommInitAll failed: %s", rccl::GetErrorString(r));
      else for (int i = 0; i < nslabs; ++i) c->slabs[i].comm = comms[i];
    }
  }
  if (!rc && c->exchange == LBM_EXCHANGE_P2P) rc = p2p_connect_local(c);
  if (rc) return create_fail(c, rc);
  *out = c;
  return LBM_OK;
}

extern "C" int lbm_create_rank_ex(const lbm_param* params, const int* obstacles, const float* cells,
                                  int rank, int nranks, int device, const void* unique_id, int exchange,
                                  lbm_ctx** out) {
  if (!out) return fail(LBM_EINVAL, "out is NULL");
  *out = nullptr;
  int rc = check_params(params);
  if (rc) return rc;
  if (!obstacles) return fail(LBM_EINVAL, "obstacles is NULL");
  if (nranks < 1 || nranks > params->ny || rank < 0 || rank >= nranks)
    return fail(LBM_EINVAL, "bad rank %d of %d (ny = %d)", rank, nranks, params->ny);
  if (exchange != LBM_EXCHANGE_RCCL && exchange != LBM_EXCHANGE_P2P)
    return fail(LBM_EINVAL, "rank contexts trade halos by RCCL or peer-to-peer (got mode %d)", exchange);
  int ndev = 0;
  lbm_device_count(&ndev);
  if (ndev < 1) return fail(LBM_ENODEV, "no HIP device visible; this library has no CPU path");
  if (device < 0 || device >= ndev) return fail(LBM_ENODEV, "HIP device %d not visible (%d devices)", device, ndev);
  const char* force = getenv("LBM_FORCE_EXCHANGE");
  const bool exchanging = nranks > 1 || (force && atoi(force));
  const bool want_p2p = exchanging && exchange == LBM_EXCHANGE_P2P;
  if (exchanging && !unique_id && !want_p2p) return fail(LBM_EINVAL, "unique_id is NULL");
  if (want_p2p && params->ny / nranks < 2) return fail(LBM_EINVAL, "peer-to-peer halos need at least 2 rows per slab");

  lbm_ctx* c = new lbm_ctx();
  c->p = *params;
  c->rank_mode = true;
  c->rank = rank;
  c->nranks = nranks;
  c->exchange = exchanging ? (want_p2p ? LBM_EXCHANGE_P2P : LBM_EXCHANGE_RCCL) : 0;
  c->no_comm = (unique_id == nullptr) && nranks > 1;
  c->tot_fluid = count_fluid(obstacles, (long)params->nx * params->ny);
  c->slabs.resize(1);
  Slab& s = c->slabs[0];
  s.dev = device;
  s.row0 = (int)((long)rank * params->ny / nranks);
  s.nyl = (int)((long)(rank + 1) * params->ny / nranks) - s.row0;
  int p2p_rc = LBM_OK;   // a failed peer-to-peer set-up is survivable when RCCL is there to fall back on
  if (want_p2p && unique_id) {
    // probe the uncached allocation first, so that a refusal does not abort the slab set-up half way
    void* probe = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipExtMallocWithFlags(&probe, 4096, hipDeviceMallocUncached) != hipSuccess) {
      (void)hipGetLastError();
      p2p_rc = fail(LBM_EHIP, "uncached device memory not available");
      c->exchange = LBM_EXCHANGE_RCCL;
    }
    if (probe) (void)hipFree(probe);
  }
  rc = finish_create(c, obstacles, cells);
  if (!rc && unique_id && (exchanging || nranks > 1)) {
    rc = rccl::load();
    if (!rc) {
      rccl::unique_id id;
      memcpy(&id, unique_id, sizeof(id));
      if (hipSetDevice(device) != hipSuccess) rc = fail(LBM_EHIP, "hipSetDevice(%d) failed", device);
      if (!rc) {
        int r = rccl::CommInitRank(&s.comm, nranks, id, rank);
        if (r != 0) rc = fail(LBM_ERCCL, "ncclCommInitRank failed: %s", rccl::GetErrorString(r));
      }
    }
  }
  if (!rc && want_p2p && unique_id) {
    // trade the hipIpc handles through the communicator; every rank learns whether ALL succeeded
    const size_t hb = LBM_P2P_HANDLE_BYTES;
    std::vector<char> all((size_t)nranks * hb, 0), mine(hb, 0);
    if (!p2p_rc && nranks > 1) p2p_rc = p2p_export(c, mine.data());
    char* d_buf = nullptr;
    if (hipMalloc((void**)&d_buf, (size_t)(nranks + 1) * hb) != hipSuccess) rc = fail(LBM_EHIP, "hipMalloc failed");
    if (!rc) {
      (void)hipMemcpy(d_buf + (size_t)nranks * hb, mine.data(), hb, hipMemcpyHostToDevice);
      int r = rccl::AllGather(d_buf + (size_t)nranks * hb, d_buf, hb, rccl::kInt8, s.comm, s.sc);
      if (r != 0) rc = fail(LBM_ERCCL, "ncclAllGather failed: %s", rccl::GetErrorString(r));
      if (!rc && hipStreamSynchronize(s.sc) != hipSuccess) rc = fail(LBM_EHIP, "handle all-gather failed");
      if (!rc) (void)hipMemcpy(all.data(), d_buf, (size_t)nranks * hb, hipMemcpyDeviceToHost);
    }
    if (!rc && !p2p_rc) p2p_rc = (nranks > 1) ? p2p_connect_ipc(c, all.data(), nranks) : p2p_connect_local(c);
    if (!rc) {  // agreement: sum of failures over all ranks -- [0] the halo blocks, [1] the mail areas of the register tiles
      // (a rank whose mail area could not be allocated or mapped must not be the only one to know: the others would launch
      // tiles that wait for its mail)
      bool tiles_ok = c->splan.ty > 0;
      if (tiles_ok && nranks > 1) for (int side = 0; side < 2; ++side) tiles_ok = tiles_ok && s.tmail_nb[side] != nullptr;
      double both[2] = {p2p_rc ? 1.0 : 0.0, tiles_ok ? 0.0 : 1.0}, *d_f = (double*)d_buf;
      (void)hipMemcpy(d_f, both, sizeof(both), hipMemcpyHostToDevice);
      int r = rccl::AllReduce(d_f, d_f, 2, rccl::kFloat64, rccl::kSum, s.comm, s.sc);
      if (r != 0 || hipStreamSynchronize(s.sc) != hipSuccess) rc = fail(LBM_ERCCL, "peer-to-peer agreement failed");
      else (void)hipMemcpy(both, d_f, sizeof(both), hipMemcpyDeviceToHost);
      const double fails = both[0];
      if (!rc && both[1] > 0.0) c->splan.ty = 0;       // somebody has no register tiling: nobody uses it
      if (!rc && fails > 0.0) {
        // somebody could not map a neighbour: everyone trades halos by RCCL instead
        if (getenv("LBM_VERBOSE")) fprintf(stderr, "lbm: peer-to-peer halos unavailable (%s); using RCCL\n", p2p_rc ? g_err : "another rank failed");
        slab_free_halos(s);
        c->exchange = LBM_EXCHANGE_RCCL;
        c->p2p_connected = false;
        rc = slab_alloc_halos(c, s);
        if (!rc) rc = upload_ghost_masks(c, s, obstacles);
      }
    }
    if (d_buf) (void)hipFree(d_buf);
  } else if (!rc && want_p2p && nranks == 1) {
    rc = p2p_connect_local(c);   // ring of one (LBM_FORCE_EXCHANGE)
  }
  if (rc) return create_fail(c, rc);
  *out = c;
  return LBM_OK;
}

extern "C" int lbm_create_rank(const lbm_param* params, const int* obstacles, const float* cells,
                               int rank, int nranks, int device, const void* unique_id, lbm_ctx** out) {
  const char* e = getenv("LBM_RANK_EXCHANGE");
  const int mode = (e && !strcmp(e, "p2p")) ? LBM_EXCHANGE_P2P : LBM_EXCHANGE_RCCL;
  return lbm_create_rank_ex(params, obstacles, cells, rank, nranks, device, unique_id, mode, out);
}

extern "C" int lbm_p2p_handle(lbm_ctx* c, void* handle64) {
  if (!c || !handle64) return fail(LBM_EINVAL, "NULL argument");
  if (!c->rank_mode || c->exchange != LBM_EXCHANGE_P2P) return fail(LBM_EINVAL, "not a peer-to-peer rank context");
  return p2p_export(c, (char*)handle64);
}

extern "C" int lbm_p2p_connect(lbm_ctx* c, const void* handles, int nranks) {
  if (!c || !handles) return fail(LBM_EINVAL, "NULL argument");
  if (!c->rank_mode || c->exchange != LBM_EXCHANGE_P2P) return fail(LBM_EINVAL, "not a peer-to-peer rank context");
  if (c->p2p_connected) return fail(LBM_EINVAL, "already connected");
  return p2p_connect_ipc(c, (const char*)handles, nranks);
}

extern "C" int lbm_num_slabs(const lbm_ctx* ctx) { return ctx ? (int)ctx->slabs.size() : 0; }

extern "C" int lbm_slab_rows(const lbm_ctx* ctx, int slab, int* row_begin, int* row_end) {
  if (!ctx || slab < 0 || slab >= (int)ctx->slabs.size()) return fail(LBM_EINVAL, "bad slab index");
  if (row_begin) *row_begin = ctx->slabs[slab].row0;
  if (row_end) *row_end = ctx->slabs[slab].row0 + ctx->slabs[slab].nyl;
  return LBM_OK;
}

namespace {

// Edge launches run on their own high-priority stream, concurrent with the interior launch, only
// when the interior is long enough to pay for the extra cross-stream events (measured on one GPU,
// RCCL self-ring: 8192^2 577 -> 537 us/step, but 1024^2 24 -> 33 us/step): big slabs only.
inline bool split_edge_stream(const lbm_ctx* c, const Slab& s) { return (long)s.nyl * c->p.nx >= (1L << 22); }
inline hipStream_t edge_stream(const lbm_ctx* c, const Slab& s) { return split_edge_stream(c, s) ? s.se : s.sc; }

// One single-step launch group (all local slabs) for step tt, launch index li.
int launch_single(lbm_ctx* c, int li, int tt, bool last, bool fold_prev, float a1, float a2) {
  const int nx = c->p.nx;
  const int q = li & 1, qp = q ^ 1;
  const bool ex = c->exchange != 0;
  const long h3 = 3L * nx;  // one-step halos live in slots 3..5 of the nine-slot buffers
  int rc;
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    lbm::SweepArgs a;
    a.src = s.lat[c->cur];
    a.dst = s.lat[c->cur ^ 1];
    a.plane = s.plane; a.pitch = s.pitch; a.nx = nx; a.nyl = s.nyl;
    a.blocked = s.blocked;
    a.omega = c->p.omega;
    a.accel_row = last ? -1 : s.accel_row;
    a.a1 = a1; a.a2 = a2;
    a.partials = s.partials[q];
    a.prev_partials = nullptr; a.prev_count = 0; a.prev_sum = nullptr;
    if (!ex) {
      // one slab, periodic self-wrap: the halo rows are the slab's own edge rows
      const long top = (long)(s.nyl - 1) * s.pitch;
      a.south2 = a.src + 2 * s.plane + top; a.south5 = a.src + 5 * s.plane + top; a.south6 = a.src + 6 * s.plane + top;
      a.north4 = a.src + 4 * s.plane; a.north7 = a.src + 7 * s.plane; a.north8 = a.src + 8 * s.plane;
      a.send_south = a.send_north = nullptr;
      a.y_begin = 0; a.y_count = s.nyl; a.y_stride = 1;
      const int nb = sweep_blocks(c, a.y_count);
      if (fold_prev) { a.prev_partials = s.partials[qp]; a.prev_count = nb; a.prev_sum = s.sums + (tt - 1); }
      launch_sweep(c, a, s.sc);
      HIPC(hipGetLastError());
    } else {
      a.south2 = s.ghost_s[qp] + h3; a.south5 = s.ghost_s[qp] + h3 + nx; a.south6 = s.ghost_s[qp] + h3 + 2 * nx;
      a.north4 = s.ghost_n[qp] + h3; a.north7 = s.ghost_n[qp] + h3 + nx; a.north8 = s.ghost_n[qp] + h3 + 2 * nx;
      a.send_south = s.send_s[q] + h3; a.send_north = s.send_n[q] + h3;
      // boundary rows first: they feed the exchange
      const int nb_rows = s.nyl >= 2 ? 2 : 1;
      a.y_begin = 0; a.y_count = nb_rows; a.y_stride = s.nyl >= 2 ? s.nyl - 1 : 1;
      const int nbb = sweep_blocks(c, nb_rows);
      const int nbi = s.nyl > 2 ? sweep_blocks(c, s.nyl - 2) : 0;
      if (fold_prev) { a.prev_partials = s.partials[qp]; a.prev_count = nbb + nbi; a.prev_sum = s.sums + (tt - 1); }
      // edge stream: after the halos of the previous launch arrived and its interior finished
      hipStream_t es = edge_stream(c, s);
      HIPC(hipStreamWaitEvent(es, s.ev_recv[qp], 0));
      if (es != s.sc) HIPC(hipStreamWaitEvent(es, s.ev_int[qp], 0));
      launch_sweep(c, a, es);
      HIPC(hipGetLastError());
      HIPC(hipEventRecord(s.ev_bnd[q], es));
    }
  }
  if (ex) {
    if ((rc = exchange_halos(c, q, 3, 3))) return rc;
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      // interior stream: after the previous launch's edge rows are in place (ev_bnd of launch li-1)
      const bool split = split_edge_stream(c, s);
      if (split) HIPC(hipStreamWaitEvent(s.sc, s.ev_bnd[qp], 0));
      if (s.nyl <= 2) { if (split) HIPC(hipEventRecord(s.ev_int[q], s.sc)); continue; }
      lbm::SweepArgs a;
      a.src = s.lat[c->cur];
      a.dst = s.lat[c->cur ^ 1];
      a.plane = s.plane; a.pitch = s.pitch; a.nx = nx; a.nyl = s.nyl;
      a.blocked = s.blocked;
      a.omega = c->p.omega;
      a.accel_row = last ? -1 : s.accel_row;
      a.a1 = a1; a.a2 = a2;
      a.south2 = a.south5 = a.south6 = a.north4 = a.north7 = a.north8 = nullptr;  // interior rows never touch halos
      a.send_south = a.send_north = nullptr;
      a.y_begin = 1; a.y_count = s.nyl - 2; a.y_stride = 1;
      a.partials = s.partials[q] + sweep_blocks(c, 2);
      a.prev_partials = nullptr; a.prev_count = 0; a.prev_sum = nullptr;
      launch_sweep(c, a, s.sc);
      HIPC(hipGetLastError());
      if (split) HIPC(hipEventRecord(s.ev_int[q], s.sc));
    }
  }
  c->cur ^= 1;
  return LBM_OK;
}

int single_partial_count(const lbm_ctx* c, const Slab& s) {
  if (c->exchange == 0) return sweep_blocks(c, s.nyl);
  return sweep_blocks(c, s.nyl >= 2 ? 2 : 1) + (s.nyl > 2 ? sweep_blocks(c, s.nyl - 2) : 0);
}

// One two-step launch group (all local slabs) for steps tt, tt+1, launch index li.
int launch_pair(lbm_ctx* c, int li, int tt, bool accel_out, bool fold_prev, float a1, float a2) {
  const int nx = c->p.nx;
  const int q = li & 1, qp = q ^ 1;
  const bool ex = c->exchange != 0;
  const int ntx = cdiv(nx, kT2X);   // (partial tiles only when the slab is alone)
  int rc;
  auto fill = [&](Slab& s, lbm::Sweep2Args& a) {
    const int nbtot = ntx * cdiv(s.nyl, kT2Y);
    a.src = s.lat[c->cur]; a.dst = s.lat[c->cur ^ 1];
    a.plane = s.plane; a.pitch = s.pitch; a.nx = nx; a.ny = s.nyl;
    a.blocked = s.blocked; a.omega = c->p.omega;
    a.accel_row = s.accel_row >= 0 ? s.accel_row : lbm::kNoRow;
    a.accel_out = accel_out ? 1 : 0;
    a.a1 = a1; a.a2 = a2;
    a.partials1 = s.partials[q]; a.partials2 = s.partials[q] + nbtot;
    a.prev1 = a.prev2 = nullptr; a.prev_count = 0; a.prev_sum = nullptr;
    a.ghost_s = a.ghost_n = nullptr; a.blocked_gs = a.blocked_gn = nullptr; a.send_s = a.send_n = nullptr;
    return nbtot;
  };
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    lbm::Sweep2Args a;
    const int nbtot = fill(s, a);
    const int nty = cdiv(s.nyl, kT2Y);
    if (fold_prev) { a.prev1 = s.partials[qp]; a.prev2 = s.partials[qp] + nbtot; a.prev_count = nbtot; a.prev_sum = s.sums + (tt - 2); }
    if (!ex) {
      a.by_begin = 0; a.by_count = nty; a.by_stride = 1;
      launch_sweep2(c, a, nbtot, s.sc, false);
      HIPC(hipGetLastError());
    } else {
      // edge tile rows first: they consume the halos of the previous pair and pack the next ones
      a.by_begin = 0; a.by_count = nty >= 2 ? 2 : 1; a.by_stride = nty >= 2 ? nty - 1 : 1;
      a.ghost_s = s.ghost_s[qp]; a.ghost_n = s.ghost_n[qp];
      a.blocked_gs = s.blocked_gs; a.blocked_gn = s.blocked_gn;
      a.send_s = s.send_s[q]; a.send_n = s.send_n[q];
      hipStream_t es = edge_stream(c, s);
      HIPC(hipStreamWaitEvent(es, s.ev_recv[qp], 0));
      if (es != s.sc) HIPC(hipStreamWaitEvent(es, s.ev_int[qp], 0));
      launch_sweep2(c, a, ntx * a.by_count, es, true);
      HIPC(hipGetLastError());
      HIPC(hipEventRecord(s.ev_bnd[q], es));
    }
  }
  if (ex) {
    if ((rc = exchange_halos(c, q, 0, lbm::kHaloSlots))) return rc;
    for (auto& s : c->slabs) {
      const int nty = cdiv(s.nyl, kT2Y);
      HIPC(hipSetDevice(s.dev));
      const bool split = split_edge_stream(c, s);
      if (split) HIPC(hipStreamWaitEvent(s.sc, s.ev_bnd[qp], 0));
      if (nty <= 2) { if (split) HIPC(hipEventRecord(s.ev_int[q], s.sc)); continue; }
      lbm::Sweep2Args a;
      fill(s, a);
      a.by_begin = 1; a.by_count = nty - 2; a.by_stride = 1;
      a.partials1 += 2 * ntx; a.partials2 += 2 * ntx;   // after the two edge tile rows
      launch_sweep2(c, a, ntx * (nty - 2), s.sc, false);
      HIPC(hipGetLastError());
      if (split) HIPC(hipEventRecord(s.ev_int[q], s.sc));
    }
  }
  c->cur ^= 1;
  return LBM_OK;
}

// One marching launch: steps tt .. tt+K-1 of the lone slab, launch index li.
int launch_march(lbm_ctx* c, int li, int tt, bool accel_out, bool fold_prev) {
  using Cfg = lbm::MarchCfg<kMarchK>;
  Slab& s = c->slabs[0];
  HIPC(hipSetDevice(s.dev));
  if (c->march_rows <= 0) c->march_rows = march_pick_rows(c);
  const int q = li & 1, qp = q ^ 1;
  lbm::MarchArgs a;
  a.src = s.lat[c->cur]; a.dst = s.lat[c->cur ^ 1];
  a.plane = s.plane; a.pitch = s.pitch; a.nx = c->p.nx; a.ny = c->p.ny;
  a.blocked = s.blocked; a.omega = c->p.omega;
  a.accel_row = c->p.ny - 2; a.accel_out = accel_out ? 1 : 0;
  a.a1 = c->p.density * c->p.accel / 9.f; a.a2 = c->p.density * c->p.accel / 36.f;
  a.H = c->march_rows;
  a.nstrips = cdiv(c->p.nx, Cfg::WOUT); a.nchunks = cdiv(c->p.ny, a.H);
  const int nb = a.nstrips * a.nchunks;
  if ((long)kMarchK * nb > s.partial_cap) return fail(LBM_EINVAL, "marching kernel: %d blocks exceed the partial-sum buffer", nb);
  a.partials = s.partials[q];
  a.prev = nullptr; a.prev_count = 0; a.prev_sum = nullptr;
  if (fold_prev) { a.prev = s.partials[qp]; a.prev_count = nb; a.prev_sum = s.sums + (tt - kMarchK); }
  switch ((int)(c->variant & (lbm::kFastMath | lbm::kNtStore))) {
    case 0: hipLaunchKernelGGL((lbm::lbm_march<kMarchK, 0>), dim3(nb), dim3(kMarchK * 256), 0, s.sc, a); break;
    case 1: hipLaunchKernelGGL((lbm::lbm_march<kMarchK, 1>), dim3(nb), dim3(kMarchK * 256), 0, s.sc, a); break;
    case 2: hipLaunchKernelGGL((lbm::lbm_march<kMarchK, 2>), dim3(nb), dim3(kMarchK * 256), 0, s.sc, a); break;
    default: hipLaunchKernelGGL((lbm::lbm_march<kMarchK, 3>), dim3(nb), dim3(kMarchK * 256), 0, s.sc, a); break;
  }
  HIPC(hipGetLastError());
  c->cur ^= 1;
  return LBM_OK;
}

// ---- lbm_march across the slabs of ONE process.  A slab's K ghost rows on either side are not copied anywhere: the
// kernel reads them out of the neighbouring slab's lattice (same device, or a peer device over xGMI once peer
// access is on).  Launch n+1 of a slab waits for launch n of both neighbours: that orders the rows it reads and
// the rows of its own source lattice (next launch's destination) the neighbours were reading.
bool march_slabs_setup(lbm_ctx* c) {
  if (c->march_slabs >= 0) return c->march_slabs == 1;
  c->march_slabs = 0;
  if (c->rank_mode || c->exchange == 0 || c->exchange == LBM_EXCHANGE_RCCL) return false;
  const int K = (c->time_block == 8) ? 8 : kMarchK;
  if (K != kMarchK ? c->p.nx < 64 : (c->p.nx % 4 != 0 || c->p.nx < lbm::MarchCfg<kMarchK>::W)) return false;
  const int ns = (int)c->slabs.size();
  for (auto& s : c->slabs)
    if (s.nyl < 4 * K || (double)s.nyl * s.pitch * 4.0 >= 4.0e9) return false;
  for (int i = 0; i < ns; ++i)
    for (int d : {(i + ns - 1) % ns, (i + 1) % ns}) {
      const int a = c->slabs[i].dev, b = c->slabs[d].dev;
      if (a == b) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) { (void)hipGetLastError(); return false; }
      if (hipSetDevice(a) != hipSuccess) return false;
      const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
      (void)hipGetLastError();
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return false;
    }
  c->march_slabs = 1;
  return true;
}
// Steps per marching pass of a context whose slabs trade rows: 8 = lbm_wave<8>, 4 = lbm_march, 0 = no marching.
inline int slab_K(const lbm_ctx* c) {
  if (c->exchange == 0 || (c->variant & 8)) return 0;
  if (c->time_block == 8) return (c->p.nx >= 64) ? 8 : 0;
  if (c->time_block == kMarchK) return (c->p.nx % 4 == 0 && c->p.nx >= lbm::MarchCfg<kMarchK>::W) ? kMarchK : 0;
  return 0;
}
inline bool march_slabs_on(lbm_ctx* c) { return slab_K(c) != 0 && march_slabs_setup(c); }

int march_rows_for(const lbm_ctx* c, int ny_rows) {
  const int ns = cdiv(c->p.nx, lbm::MarchCfg<kMarchK>::WOUT), ncu = std::max(c->ncu, 1);
  int best_h = std::min(ny_rows, 256);
  double best = -1.0;
  for (int h = std::min(ny_rows, 32); h <= std::min(ny_rows, 1024); ++h) {
    const long blocks = (long)ns * cdiv(ny_rows, h);
    const long rounds = (blocks + ncu - 1) / ncu;
    const double eff = (double)ny_rows * ns / ((double)rounds * ncu * (h + 3 * (kMarchK - 1)));
    if (eff > best + 1e-9) { best = eff; best_h = h; }
  }
  return best_h;
}
// Rows per chunk of lbm_wave<8> on a slab of ny_rows rows, and the share of the chip's wave-slot time that is useful
// work with it: a chunk costs 2K fill iterations, and waves that do not fill the last round leave slots idle.
inline bool slab_is_wave(int K) { return K == 8; }
// wave slots of the chip for lbm_wave<K> with the context's columns per lane
int wave_slots(const lbm_ctx* c, int K) { return std::max(c->ncu, 1) * std::max(wave_blocks_per_cu(K, wave_C(c, K)), 1) * (lbm::kWaveBlock / 64); }
// The share of the chip's wave-slot time that is useful work with chunks of h rows: a chunk costs its 2K fill
// iterations on top of its h, and the waves come in rounds of `slots`: up to three rounds a partial round costs a whole
// one (4096^2: 118-row chunks = 0.98 rounds 285 GLUPS, 114-row chunks = 1.008 rounds 231; 5120^2: 91 rows = 1.99 rounds
// 310, 87 rows = 2.06 rounds 282), beyond that the rounds blur into each other.  A single round, in which every wave
// fills at the same time, runs ~0.87 of what this predicts, several rounds ~0.94 (371 GLUPS x this figure against the
// measured rates of 2048^2 ... 8192^2 and of the 8192-wide slabs).
double slab_wave_efficiency(const lbm_ctx* c, int ny_rows, int h, int K, int extra_waves) {
  const int nwc = cdiv(c->p.nx, wave_out_cols(c, K));
  const long waves = (long)nwc * cdiv(ny_rows, h) + extra_waves, slots = wave_slots(c, K);
  const double r = (double)waves / slots;
  const double rounds = r <= 3.0 ? std::ceil(r) : r;
  // (two columns per lane: 0.91 - 0.96 measured for one round and for two alike)
  const double shape = wave_C(c, K) == 2 ? 0.93 : rounds <= 1.0 ? 0.87 : 0.94;
  // (a level's fill rows are half empty on average: K of the 2K fill iterations' worth of work)
  return shape * (double)nwc * ny_rows / (rounds * slots * (h + 2.0 * K));
}
int slab_wave_rows(const lbm_ctx* c, int ny_rows, int K, int extra_waves) {
  if (c->wave_rows > 0) return std::min(c->wave_rows, ny_rows);
  const int hmax = wave_C(c, K) == 2 ? 320 : 128;
  int best_h = std::min(ny_rows, 128);
  double best = -1.0;
  for (int h = std::min(ny_rows, 32); h <= std::min(ny_rows, hmax); ++h) {        // (beyond 128 rows the one-column chunks get slower: measured)
    const double e = slab_wave_efficiency(c, ny_rows, h, K, extra_waves);
    if (e > best + 1e-9) { best = e; best_h = h; }
  }
  return best_h;
}
// partial-sum slots (blocks) of one marching launch on a slab
inline int march_slab_blocks(const lbm_ctx* c, const Slab& s) {
  const int K = slab_K(c);
  if (slab_is_wave(K)) return cdiv((long)cdiv(c->p.nx, wave_out_cols(c, K)) * cdiv(s.nyl, slab_wave_rows(c, s.nyl, K)), lbm::kWaveBlock / 64);
  return cdiv(c->p.nx, lbm::MarchCfg<kMarchK>::WOUT) * cdiv(s.nyl, march_rows_for(c, s.nyl));
}

// The neighbours of a slab as a marching launch sees them.
struct SlabNb { const float* src_s; const float* src_n; long plane_s, plane_n; int ny_s, ny_n; const uint8_t* blk_s; const uint8_t* blk_n; };

// One marching launch on one slab: steps tt .. tt+K-1, partial sums into buffer q (folding the previous launch's).
int launch_slab_pass(lbm_ctx* c, Slab& s, const SlabNb& nbr, int K, int q, int tt, bool accel_out, bool fold_prev) {
  const float a1 = c->p.density * c->p.accel / 9.f, a2 = c->p.density * c->p.accel / 36.f;
  const int nb = march_slab_blocks(c, s), qp = q ^ 1;
  if ((long)K * nb > s.partial_cap) return fail(LBM_EINVAL, "marching kernel: %d blocks exceed the partial-sum buffer", nb);
  // the lattice's accelerate row (ny-2) in this slab's row numbers, and its periodic images: one of them may fall
  // into the K rows this slab recomputes on a neighbour's behalf
  const int ar = (c->p.ny - 2) - s.row0;
  const int flavour = (int)(c->variant & (lbm::kFastMath | lbm::kNtStore));
  if (slab_is_wave(K)) {
    lbm::WaveArgs a;
    a.src = s.lat[c->cur]; a.dst = s.lat[c->cur ^ 1];
    a.plane = s.plane; a.pitch = s.pitch; a.nx = c->p.nx; a.ny = s.nyl;
    a.blocked = s.blocked; a.omega = c->p.omega;
    a.accel_row = lbm::kNoRow; a.accel_out = accel_out ? 1 : 0; a.a1 = a1; a.a2 = a2;
    a.H = slab_wave_rows(c, s.nyl, K);
    a.y_begin = 0; a.y_end = s.nyl;
    a.nwc = cdiv(c->p.nx, wave_out_cols(c, K)); a.nchunks = cdiv(s.nyl, a.H);
    a.nchunks_a = a.nchunks; a.yb_begin = a.yb_end = 0;
    a.partials = s.partials[q]; a.pstride = nb; a.pbase = 0;
    a.prev = nullptr; a.prev_count = 0; a.prev_sum = nullptr;
    if (fold_prev) { a.prev = s.partials[qp]; a.prev_count = nb; a.prev_sum = s.sums + (tt - K); }
    a.src_s = nbr.src_s; a.src_n = nbr.src_n; a.plane_s = nbr.plane_s; a.plane_n = nbr.plane_n;
    a.ny_s = nbr.ny_s; a.ny_n = nbr.ny_n; a.blocked_s = nbr.blk_s; a.blocked_n = nbr.blk_n;
    a.acc_rows[0] = ar; a.acc_rows[1] = ar - c->p.ny; a.acc_rows[2] = ar + c->p.ny;
    hipLaunchKernelGGL(wave_kernel(K, true, wave_C(c, K), flavour), dim3(nb), dim3(lbm::kWaveBlock), 0, s.sc, a);
  } else {
    using Cfg = lbm::MarchCfg<kMarchK>;
    lbm::MarchArgs a;
    a.src = s.lat[c->cur]; a.dst = s.lat[c->cur ^ 1];
    a.plane = s.plane; a.pitch = s.pitch; a.nx = c->p.nx; a.ny = s.nyl;
    a.blocked = s.blocked; a.omega = c->p.omega;
    a.accel_row = lbm::kNoRow; a.accel_out = accel_out ? 1 : 0; a.a1 = a1; a.a2 = a2;
    a.H = march_rows_for(c, s.nyl);
    a.nstrips = cdiv(c->p.nx, Cfg::WOUT); a.nchunks = cdiv(s.nyl, a.H);
    a.partials = s.partials[q];
    a.prev = nullptr; a.prev_count = 0; a.prev_sum = nullptr;
    if (fold_prev) { a.prev = s.partials[qp]; a.prev_count = nb; a.prev_sum = s.sums + (tt - K); }
    a.src_s = nbr.src_s; a.src_n = nbr.src_n; a.plane_s = nbr.plane_s; a.plane_n = nbr.plane_n;
    a.ny_s = nbr.ny_s; a.ny_n = nbr.ny_n; a.blocked_s = nbr.blk_s; a.blocked_n = nbr.blk_n;
    a.acc_rows[0] = ar; a.acc_rows[1] = ar - c->p.ny; a.acc_rows[2] = ar + c->p.ny;
    switch (flavour) {
      case 0: hipLaunchKernelGGL((lbm::lbm_march<kMarchK, 0, true>), dim3(nb), dim3(kMarchK * 256), 0, s.sc, a); break;
      case 1: hipLaunchKernelGGL((lbm::lbm_march<kMarchK, 1, true>), dim3(nb), dim3(kMarchK * 256), 0, s.sc, a); break;
      case 2: hipLaunchKernelGGL((lbm::lbm_march<kMarchK, 2, true>), dim3(nb), dim3(kMarchK * 256), 0, s.sc, a); break;
      default: hipLaunchKernelGGL((lbm::lbm_march<kMarchK, 3, true>), dim3(nb), dim3(kMarchK * 256), 0, s.sc, a); break;
    }
  }
  HIPC(hipGetLastError());
  return LBM_OK;
}

// One marching launch group over all slabs: steps tt .. tt+K-1, launch index li.
int launch_march_slabs(lbm_ctx* c, int li, int tt, bool accel_out, bool fold_prev) {
  const int ns = (int)c->slabs.size(), q = li & 1, qp = q ^ 1, K = slab_K(c);
  for (int i = 0; i < ns; ++i) {
    Slab& s = c->slabs[i];
    Slab& so = c->slabs[(i + ns - 1) % ns];
    Slab& no = c->slabs[(i + 1) % ns];
    HIPC(hipSetDevice(s.dev));
    HIPC(hipStreamWaitEvent(s.sc, so.ev_march[qp], 0));
    HIPC(hipStreamWaitEvent(s.sc, no.ev_march[qp], 0));
    const SlabNb nbr{so.lat[c->cur], no.lat[c->cur], so.plane, no.plane, so.nyl, no.nyl, so.blocked, no.blocked};
    const int rc = launch_slab_pass(c, s, nbr, K, q, tt, accel_out, fold_prev);
    if (rc) return rc;
    HIPC(hipEventRecord(s.ev_march[q], s.sc));
  }
  c->cur ^= 1;
  return LBM_OK;
}

// ---- lbm_wave across slabs under the RCCL transport (LBM_EXCHANGE_RCCL, one process per GPU or one process with one
// slab per GPU): GHOST BANDS.  The K rows below and above a slab are kept as the neighbours hold them in two bands per
// lattice ([9 planes][K rows][pitch], contiguous) and travel once per K steps: after the slab's two EDGE launches (a
// short chunk of rows at the bottom and one at the top, which produce the K rows each neighbour needs) the rows are
// packed and sent with ncclSend / ncclRecv on the exchange stream, overlapped with the INTERIOR launch, which touches
// no ghost row (reference rows: d2q9-bgk.c:971-998 names the planes that cross a row boundary; K steps need all nine
// planes of K rows).  The kernel is the SLAB flavour that also reads neighbours' rows in place: a band looks to it
// like a neighbour's lattice of K rows.  Message protocol: tests/test_slab_gloo.py::test_k_row_ghost_zone_of_the_marching_kernels.
struct BandPlan { int K, he, H, nwc, nb_e, nb_i; };   // edge chunk rows, interior chunk rows, blocks of the edge launch (both edge chunks) and of the interior launch

bool march_bands_on(const lbm_ctx* c) {
  const int K = slab_K(c);
  if (c->exchange != LBM_EXCHANGE_RCCL || !slab_is_wave(K)) return false;
  const int rows = c->p.ny / c->nranks;                     // the smallest slab: every rank decides alike
  if (rows < 4 * K || (double)(rows + 1) * c->slabs[0].pitch * 4.0 >= 4.0e9) return false;
  for (auto& s : c->slabs) if (!s.band_blk_s) return false;
  return true;
}

// Edge chunks about half the height of the interior's, so that an edge launch plus the exchange it feeds is over before
// the interior launch is; interior chunks such that all three launches fit the chip's wave slots in whole rounds.
BandPlan band_plan(const lbm_ctx* c, const Slab& s) {
  BandPlan b;
  b.K = slab_K(c);
  b.nwc = cdiv(c->p.nx, wave_out_cols(c, b.K));
  // (an edge chunk is one wave per strip marching alone on its SIMD: ~2.2 us per row at K = 8, against ~3.7 for the rows of the
  // interior launch's three waves per SIMD; LBM_BAND_EDGE_ROWS overrides)
  static const int edge_rows_env = getenv("LBM_BAND_EDGE_ROWS") ? atoi(getenv("LBM_BAND_EDGE_ROWS")) : 0;
  const int hu = slab_wave_rows(c, s.nyl, b.K);
  b.he = std::max(b.K, std::min(edge_rows_env > 0 ? edge_rows_env : hu / 3, s.nyl / 4));
  b.H = slab_wave_rows(c, s.nyl - 2 * b.he, b.K, 2 * b.nwc);
  const int per = lbm::kWaveBlock / 64;
  b.nb_e = cdiv(2 * b.nwc, per);
  b.nb_i = cdiv((long)b.nwc * cdiv(s.nyl - 2 * b.he, b.H), per);
  return b;
}

int bands_setup(lbm_ctx* c, int K) {
  for (auto& s : c->slabs) {
    if (s.band_K == K) continue;
    HIPC(hipSetDevice(s.dev));
    const size_t bytes = sizeof(float) * 9 * (size_t)K * s.pitch;
    for (int i = 0; i < 2; ++i) {
      if (s.band_s[i]) HIPC(hipFree(s.band_s[i]));
      if (s.band_n[i]) HIPC(hipFree(s.band_n[i]));
      s.band_s[i] = s.band_n[i] = nullptr;
      HIPC(hipMalloc((void**)&s.band_s[i], bytes));
      HIPC(hipMalloc((void**)&s.band_n[i], bytes));
    }
    if (s.band_send_s) HIPC(hipFree(s.band_send_s));
    if (s.band_send_n) HIPC(hipFree(s.band_send_n));
    s.band_send_s = s.band_send_n = nullptr;
    HIPC(hipMalloc((void**)&s.band_send_s, bytes));
    HIPC(hipMalloc((void**)&s.band_send_n, bytes));
    s.band_K = K;
  }
  return LBM_OK;
}

// The K edge rows of lattice `par` of every local slab to the neighbours' bands of the same parity: on the exchange
// streams, behind event ev_bnd[q] ("the launches that wrote those rows are over"); ev_recv[q] says the bands have arrived.
int exchange_bands(lbm_ctx* c, int q, int par, int K) {
  const int ns = (int)c->slabs.size();
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    HIPC(hipStreamWaitEvent(s.sx, s.ev_bnd[q], 0));
    hipLaunchKernelGGL(lbm::lbm_pack_band_rows, dim3(cdiv((long)K * s.pitch, 256)), dim3(256), 0, s.sx,
                       s.lat[par], s.plane, s.pitch, s.nyl, K, s.band_send_s, s.band_send_n);
    HIPC(hipGetLastError());
  }
  NCCLC(rccl::GroupStart());
  for (int i = 0; i < ns; ++i) {
    Slab& s = c->slabs[i];
    const size_t n = 9 * (size_t)K * s.pitch;
    const int me = c->rank_mode ? c->rank : i;
    const int south = (me + c->nranks - 1) % c->nranks, north = (me + 1) % c->nranks;
    // order matters when south == north (1 or 2 ranks): sends S then N, receives N then S
    NCCLC(rccl::Send(s.band_send_s, n, rccl::kFloat32, south, s.comm, s.sx));
    NCCLC(rccl::Send(s.band_send_n, n, rccl::kFloat32, north, s.comm, s.sx));
    NCCLC(rccl::Recv(s.band_n[par], n, rccl::kFloat32, north, s.comm, s.sx));
    NCCLC(rccl::Recv(s.band_s[par], n, rccl::kFloat32, south, s.comm, s.sx));
  }
  NCCLC(rccl::GroupEnd());
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    HIPC(hipEventRecord(s.ev_recv[q], s.sx));
  }
  return LBM_OK;
}

// One launch of lbm_wave<K, ., SLAB> on a slab whose ghost rows live in bands: the interior rows [he, nyl - he) in chunks of
// H, or (edge) the two edge chunks [0, he) and [nyl - he, nyl) together.
int launch_band_rows(lbm_ctx* c, Slab& s, const BandPlan& b, bool edge, int q, int tt, bool accel_out, bool fold_prev, hipStream_t st) {
  const int K = b.K, qp = q ^ 1, ntot = b.nb_e + b.nb_i;
  const int nblocks = edge ? b.nb_e : b.nb_i, pbase = edge ? 0 : b.nb_e;
  lbm::WaveArgs a;
  a.src = s.lat[c->cur]; a.dst = s.lat[c->cur ^ 1];
  a.plane = s.plane; a.pitch = s.pitch; a.nx = c->p.nx; a.ny = s.nyl;
  a.blocked = s.blocked; a.omega = c->p.omega;
  a.accel_row = lbm::kNoRow; a.accel_out = accel_out ? 1 : 0;
  a.a1 = c->p.density * c->p.accel / 9.f; a.a2 = c->p.density * c->p.accel / 36.f;
  if (edge) {
    a.H = b.he; a.y_begin = 0; a.y_end = b.he; a.nchunks_a = 1; a.yb_begin = s.nyl - b.he; a.yb_end = s.nyl; a.nchunks = 2;
  } else {
    a.H = b.H; a.y_begin = b.he; a.y_end = s.nyl - b.he; a.nchunks = cdiv(a.y_end - a.y_begin, b.H); a.nchunks_a = a.nchunks;
    a.yb_begin = a.yb_end = 0;
  }
  a.nwc = b.nwc;
  a.partials = s.partials[q]; a.pstride = ntot; a.pbase = pbase;
  a.prev = nullptr; a.prev_count = 0; a.prev_sum = nullptr;
  if (fold_prev) { a.prev = s.partials[qp]; a.prev_count = ntot; a.prev_sum = s.sums + (tt - K); }
  const long bplane = (long)K * s.pitch;
  a.src_s = s.band_s[c->cur]; a.src_n = s.band_n[c->cur]; a.plane_s = bplane; a.plane_n = bplane;
  a.ny_s = K; a.ny_n = K;
  a.blocked_s = s.band_blk_s + (size_t)(kBandRows - K) * s.pitch; a.blocked_n = s.band_blk_n;
  const int ar = (c->p.ny - 2) - s.row0;
  a.acc_rows[0] = ar; a.acc_rows[1] = ar - c->p.ny; a.acc_rows[2] = ar + c->p.ny;
  hipLaunchKernelGGL(wave_kernel(K, true, wave_C(c, K), (int)(c->variant & (lbm::kFastMath | lbm::kNtStore))),
                     dim3(nblocks), dim3(lbm::kWaveBlock), 0, st, a);
  HIPC(hipGetLastError());
  return LBM_OK;
}

// One marching group (K steps) of every local slab: edge launches, exchange of the new edge rows, interior launch.
int launch_band_group(lbm_ctx* c, int li, int tt, bool accel_out, bool fold_prev) {
  const int q = li & 1, qp = q ^ 1, K = slab_K(c);
  int rc;
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    const BandPlan b = band_plan(c, s);
    hipStream_t es = edge_stream(c, s);
    HIPC(hipStreamWaitEvent(es, s.ev_recv[qp], 0));                       // the bands of the source lattice have arrived
    if (es != s.sc) HIPC(hipStreamWaitEvent(es, s.ev_int[qp], 0));        // the previous interior launch is over
    if ((rc = launch_band_rows(c, s, b, true, q, tt, accel_out, fold_prev, es))) return rc;
    HIPC(hipEventRecord(s.ev_bnd[q], es));
  }
  if ((rc = exchange_bands(c, q, c->cur ^ 1, K))) return rc;
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    const BandPlan b = band_plan(c, s);
    const bool split = split_edge_stream(c, s);
    if (split) HIPC(hipStreamWaitEvent(s.sc, s.ev_bnd[qp], 0));           // the previous edge launches are over
    if ((rc = launch_band_rows(c, s, b, false, q, tt, accel_out, false, s.sc))) return rc;
    if (split) HIPC(hipEventRecord(s.ev_int[q], s.sc));
  }
  c->cur ^= 1;
  return LBM_OK;
}

// ---- lbm_wave: K steps per pass, one wave per strip of 64 (or 128) columns
// The instantiations: a lattice alone (K = 4, 6, 8 with one column per lane, K = 8 with two) and a slab with neighbours
// (K = 8, one or two columns); flavour = IEEE or fast rcp / sqrt (bit 0) x nontemporal stores (bit 1).
template <int K, bool SLAB, int C>
wave_fn wave_kernel_f(int flavour) {
  switch (flavour & 3) {
    case 0: return lbm::lbm_wave<K, 0, SLAB, C>;
    case 1: return lbm::lbm_wave<K, 1, SLAB, C>;
    case 2: return lbm::lbm_wave<K, 2, SLAB, C>;
    default: return lbm::lbm_wave<K, 3, SLAB, C>;
  }
}
wave_fn wave_kernel(int K, bool slab, int C, int flavour) {
  if (slab) return C == 2 ? wave_kernel_f<8, true, 2>(flavour) : wave_kernel_f<8, true, 1>(flavour);     // (K = 8 only)
  if (K == 8) return C == 2 ? wave_kernel_f<8, false, 2>(flavour) : wave_kernel_f<8, false, 1>(flavour);
  if (K == 6) return wave_kernel_f<6, false, 1>(flavour);
  return wave_kernel_f<4, false, 1>(flavour);
}

int wave_blocks_per_cu(int K, int C) {
  static int cache[16][3] = {};                 // (the answer does not change; lbm_set_option asks often)
  if (K < 16 && C < 3 && cache[K][C] > 0) return cache[K][C];
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(wave_kernel(K, false, C, 1)), lbm::kWaveBlock, 0) != hipSuccess) { (void)hipGetLastError(); n = 0; }
  if (K < 16 && C < 3) cache[K][C] = n;
  return n;
}

// Waves the device holds at once (occupancy query), and the rows per chunk.  Short chunks win although each
// pays 2K fill iterations: many more waves than the device holds keep every SIMD's wave slots full from the
// first row to the last (measured at 8192^2, GLUPS for 24 / 32 / 48 / 64 / 96 rows: K = 4: 193 219 191 204 179;
// K = 6: 229 237 244 246 237; K = 8: 198 212 224 227 228; one resident round of 511-row chunks: 153 at K = 4).
void wave_plan(lbm_ctx* c) {
  const int K = c->time_block;
  int bpc = wave_blocks_per_cu(K, wave_C(c, K));
  if (bpc < 1) bpc = 4;
  c->wave_capacity = std::max(c->ncu, 1) * bpc * (lbm::kWaveBlock / 64);
  if (c->wave_rows > 0) return;
  c->wave_rows = (K == 8) ? slab_wave_rows(c, c->p.ny, 8) : std::min(c->p.ny, K >= 6 ? 64 : 32);   // (K = 8: whole rounds of the chip's wave slots)
}

// One lbm_wave launch: steps tt .. tt+K-1 of the lone slab, launch index li.
int launch_wave(lbm_ctx* c, int li, int tt, bool accel_out, bool fold_prev) {
  Slab& s = c->slabs[0];
  HIPC(hipSetDevice(s.dev));
  const int K = c->time_block;
  if (c->wave_rows <= 0 || c->wave_capacity <= 0) wave_plan(c);
  const int q = li & 1, qp = q ^ 1;
  lbm::WaveArgs a;
  a.src = s.lat[c->cur]; a.dst = s.lat[c->cur ^ 1];
  a.plane = s.plane; a.pitch = s.pitch; a.nx = c->p.nx; a.ny = c->p.ny;
  a.blocked = s.blocked; a.omega = c->p.omega;
  a.accel_row = c->p.ny - 2; a.accel_out = accel_out ? 1 : 0;
  a.a1 = c->p.density * c->p.accel / 9.f; a.a2 = c->p.density * c->p.accel / 36.f;
  a.H = c->wave_rows;
  a.y_begin = 0; a.y_end = c->p.ny;
  a.nwc = cdiv(c->p.nx, wave_out_cols(c, K)); a.nchunks = cdiv(c->p.ny, a.H);
  a.nchunks_a = a.nchunks; a.yb_begin = a.yb_end = 0;
  const int nb = cdiv((long)a.nwc * a.nchunks, lbm::kWaveBlock / 64);
  if ((long)K * nb > s.partial_cap) return fail(LBM_EINVAL, "lbm_wave: %d blocks exceed the partial-sum buffer (raise wave_rows)", nb);
  a.partials = s.partials[q]; a.pstride = nb; a.pbase = 0;
  a.prev = nullptr; a.prev_count = 0; a.prev_sum = nullptr;
  if (fold_prev) { a.prev = s.partials[qp]; a.prev_count = nb; a.prev_sum = s.sums + (tt - K); }
  // (development: LBM_WAVE_PAD_LDS = bytes of unused dynamic LDS per block, to hold fewer blocks on a CU than the registers
  // allow -- how the rate depends on the waves per SIMD: profiles/r03_wave_occupancy.log)
  static const int pad_lds = getenv("LBM_WAVE_PAD_LDS") ? atoi(getenv("LBM_WAVE_PAD_LDS")) : 0;
  hipLaunchKernelGGL(wave_kernel(K, false, wave_C(c, K), (int)(c->variant & (lbm::kFastMath | lbm::kNtStore))), dim3(nb), dim3(lbm::kWaveBlock), pad_lds, s.sc, a);
  HIPC(hipGetLastError());
  c->cur ^= 1;
  return LBM_OK;
}
inline int wave_blocks(const lbm_ctx* c) {
  return cdiv((long)cdiv(c->p.nx, wave_out_cols(c, c->time_block)) * cdiv(c->p.ny, c->wave_rows), lbm::kWaveBlock / 64);
}

// The blocks of one marching launch must fit the per-block partial sums (K floats per block).  Asked BEFORE anything
// of a run is queued -- behind the prologue the lattice would already carry the accelerate phase of a step that is
// then never taken -- and when the chunk heights are set.
int check_march_partials(lbm_ctx* c, bool slabs_march) {
  if (slabs_march) {
    const int K = slab_K(c);
    for (auto& s : c->slabs) {
      const long nb = march_slab_blocks(c, s);
      if ((long)K * nb > s.partial_cap) return fail(LBM_EINVAL, "marching kernel: %ld blocks exceed the partial-sum buffer (raise wave_rows / march_rows)", nb);
    }
  } else if (march_eligible(c)) {
    const int K = c->time_block;
    long nb;
    if (use_wave_kernel(c)) {
      if (c->wave_rows <= 0 || c->wave_capacity <= 0) wave_plan(c);
      nb = wave_blocks(c);
    } else {
      if (c->march_rows <= 0) c->march_rows = march_pick_rows(c);
      nb = (long)cdiv(c->p.nx, lbm::MarchCfg<kMarchK>::WOUT) * cdiv(c->p.ny, c->march_rows);
    }
    if ((long)K * nb > c->slabs[0].partial_cap)
      return fail(LBM_EINVAL, "marching kernel: %ld blocks exceed the partial-sum buffer (raise wave_rows / march_rows)", nb);
  }
  return LBM_OK;
}

}  // namespace

namespace {

// End of a run: reduce across ranks (if there is a communicator), fetch the per-step sums and the
// peer-to-peer error word through pinned staging with async copies queued behind the step loop,
// then ONE wait per slab (s.sc has joined the edge and exchange streams by then).
int collect_sums(lbm_ctx* c, int nsteps, float* av_vels, std::chrono::steady_clock::time_point wall0, int extra = 0) {
  // (extra: doubles behind the per-step sums that are reduced and fetched with them -- run_regtile_slabs' "somebody gave up")
  if (c->rank_mode && c->slabs[0].comm != nullptr) {   // (a ring of one rank has a communicator too: identity)
    Slab& s = c->slabs[0];
    NCCLC(rccl::AllReduce(s.sums, s.sums, (size_t)(nsteps + extra), rccl::kFloat64, rccl::kSum, s.comm, s.sc));
  }
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    if ((av_vels || extra) && !s.sums_direct) HIPC(hipMemcpyAsync(s.sums_host, s.sums, sizeof(double) * (nsteps + extra), hipMemcpyDeviceToHost, s.sc));
    if (s.counters) HIPC(hipMemcpyAsync(s.err_host, s.counters + 32, sizeof(uint32_t), hipMemcpyDeviceToHost, s.sc));
  }
  double gpu_ms = 0.0;
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    if (c->exchange != 0) {
      HIPC(hipStreamSynchronize(s.sx));
      HIPC(hipStreamSynchronize(s.se));
    }
    HIPC(hipStreamSynchronize(s.sc));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, s.ev_t0, s.ev_t1));
    if (ms > gpu_ms) gpu_ms = ms;
  }
  c->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  c->gpu_ms = gpu_ms;
  for (auto& s : c->slabs)
    if (s.counters && *s.err_host) {
      c->p2p_failed = true;
      return fail(LBM_EHIP, "peer-to-peer halo wait timed out (a neighbouring slab stopped)");
    }
  if (av_vels) {
    const double nf = (double)c->tot_fluid;
    const size_t ns = c->slabs.size();
    for (int i = 0; i < nsteps; ++i) {
      double acc = 0.0;
      for (size_t k = 0; k < ns; ++k) acc += c->slabs[k].sums_host[i];
      av_vels[i] = (float)(acc / nf);  // d2q9-bgk.c:1811
    }
  }
  return LBM_OK;
}

// ----------------------------------------------------------------- resident engine
void resident_free(lbm_ctx* c) {
  if (c->slabs.empty()) return;
  (void)hipSetDevice(c->slabs[0].dev);
  if (c->tmail) (void)hipFree(c->tmail);
  c->tmail = nullptr; c->rpartials_tiles = 0;
  if (c->rpartials) (void)hipFree(c->rpartials);
  if (c->rabort) (void)hipFree(c->rabort);
  c->rpartials = nullptr; c->rabort = nullptr; c->rpartials_cap = 0;
}

// The resident engine cannot be used on this context (any more): remember why, say so ONCE on stderr (a run that quietly
// takes twice as long is worse than a line of text), carry on with the streaming kernels.
void resident_give_up(lbm_ctx* c, const char* why) {
  c->resident_broken = true;
  snprintf(c->resident_why, sizeof(c->resident_why), "%s", why);
  static bool said = false;
  if (!said || getenv("LBM_VERBOSE")) fprintf(stderr, "lbm: register-tile engine not used (%s); running the streaming kernels instead\n", why);
  said = true;
}

// ---- register-tile engine (lbm_regtile.hip.h): 64-column tiles of nw x r rows, one per CU
bool regtile_ok(const lbm_ctx* c, int ty, int r) {
  if (c->p.nx % 64 != 0 || ty < 1 || ty > c->p.ny || c->p.ny % ty != 0) return false;
  if (!(r == 1 || r == 2 || r == 4) || ty % r != 0 || ty / r > 16) return false;
  // every tile must be resident at once: a CU takes 16 waves of this kernel (128 VGPRs) and 160 KB of its blocks' LDS
  const int nw = ty / r;
  const int per_cu = std::min({3, 16 / nw, (160 * 1024) / lbm::regtile_lds_bytes(nw, r)});
  return (long)(c->p.nx / 64) * (c->p.ny / ty) <= (long)c->ncu * per_cu;
}
void regtile_set(lbm_ctx* c, int ty, int r) {
  c->tplan.ty = ty; c->tplan.r = r; c->tplan.nw = ty / r; c->tplan.ntx = c->p.nx / 64; c->tplan.nty = c->p.ny / ty;
  c->tplan.bpc = 0;          // (the residency of this tiling has not been asked yet)
}
// Default tiling (of a lattice alone and of equal slabs alike; `per_dev` = slabs sharing a device, `rows` = rows per slab).
// Measured with the mailboxes in uncached memory (profiles/r03_regtile_tilings.log), us per step: the SHORTEST tiles that
// still fit one per CU win -- 1024x512: 32 rows 2.26, 64 rows 2.94; 1024x256: 16 rows 1.94, 32 rows 2.20; 1024x128: 8 rows
// 1.58, 16 rows 1.87; 256x256: 4 rows 1.34, 8 rows 1.38 -- but not one-row tiles (128x128: 2 rows 1.26, 1 row 1.31); and
// within a tile height, as few rows per wave as leave at most EIGHT waves (they meet at a barrier every step; 1024x256,
// 16-row tiles: 16 x 1 rows 2.00, 8 x 2 1.94, 4 x 4 1.98; 1024x128, 8-row tiles: 8 x 1 1.58, 4 x 2 1.76, 2 x 4 1.93),
// sixteen where the lattice leaves no choice (1024x1024: 16 waves x 4 rows, 2.94).
bool regtile_tiling_rule(int nx, int rows, int per_dev, int ncu, int* ty_out, int* r_out) {
  if (nx < 64 || nx % 64 != 0 || rows < 1 || per_dev < 1 || ncu < 1) return false;
  const long ntx = nx / 64;
  for (int ty = (rows >= 2 ? 2 : 1); ty <= std::min(rows, 64); ++ty) {
    if (rows % ty != 0 || (long)per_dev * ntx * (rows / ty) > (long)ncu) continue;
    for (int waves : {8, 16})
      for (int r : {1, 2, 4}) {
        if (ty % r != 0 || ty / r > waves) continue;
        if ((160 * 1024) / lbm::regtile_lds_bytes(ty / r, r) < 1) continue;
        *ty_out = ty; *r_out = r;
        return true;
      }
  }
  return false;
}
bool regtile_default_tiling(const lbm_ctx* c, int rows, int per_dev, int* ty_out, int* r_out) {
  return regtile_tiling_rule(c->p.nx, rows, per_dev, c->ncu, ty_out, r_out);
}
bool plan_regtile(lbm_ctx* c) {
  c->tplan.ty = 0;
  int ty = 0, r = 0;
  if (!regtile_default_tiling(c, c->p.ny, 1, &ty, &r) || !regtile_ok(c, ty, r)) return false;
  regtile_set(c, ty, r);
  return true;
}

// The instantiation of lbm_regtile for a tiling and flavour (dbg: the LBM_RESIDENT_DEBUG timing experiments, R = 4 only).
typedef void (*regtile_fn)(const lbm::RegTileArgs);
regtile_fn regtile_kernel(int r, bool fast, int dbg, bool trace, bool async) {
  constexpr int NW_ = lbm::kResDebugNoWait, NS_ = lbm::kResDebugNoSend, AS_ = lbm::kRegAsync;
  if (async && dbg == 0 && !trace && r == 4) return fast ? lbm::lbm_regtile<4, AS_ | 1> : lbm::lbm_regtile<4, AS_>;
  if (async && dbg == 0 && !trace && r == 2) return fast ? lbm::lbm_regtile<2, AS_ | 1> : lbm::lbm_regtile<2, AS_>;
  if (async && trace && r == 4) return lbm::lbm_regtile<4, AS_ | 2048 | 1>;
  if (r == 4 && dbg == 1) return lbm::lbm_regtile<4, NW_>;
  if (r == 4 && dbg == 2) return lbm::lbm_regtile<4, NW_ | NS_>;
  if (r == 4 && dbg == 3) return lbm::lbm_regtile<4, NW_ | NS_ | 256>;
  if (r == 4 && dbg == 4) return lbm::lbm_regtile<4, NW_ | 512>;
  if (r == 4 && dbg == 5) return lbm::lbm_regtile<4, NW_ | 1024>;
  if (r == 4 && trace) return lbm::lbm_regtile<4, 2048>;
  switch (r) {
    case 4: return fast ? lbm::lbm_regtile<4, 1> : lbm::lbm_regtile<4, 0>;
    case 2: return fast ? lbm::lbm_regtile<2, 1> : lbm::lbm_regtile<2, 0>;
    default: return fast ? lbm::lbm_regtile<1, 1> : lbm::lbm_regtile<1, 0>;
  }
}

// Before the first launch of a tiling on a device: let the kernel have its dynamic LDS (beyond the 64 KB a kernel gets
// without asking) and ASK the runtime how many of its blocks a CU takes.  The tiles wait on each other, so all of them must
// be resident at once: blocks per CU x CUs >= tiles, or the launch would stall until its waits time out.  Returns the
// blocks per CU, or -1 with the reason in lbm_last_error.
int regtile_prepare(const lbm_ctx* c, const void* fn, int dev, int threads, unsigned shm) {
  struct Seen { const void* fn; int dev; };
  static std::mutex mu;
  static std::vector<Seen> raised;
  {
    std::lock_guard<std::mutex> g(mu);
    bool have = false;
    for (auto& e : raised) have = have || (e.fn == fn && e.dev == dev);
    if (!have) {
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        if (shm > 64u * 1024u) { fail(LBM_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e)); return -1; }
      } else raised.push_back({fn, dev});
    }
  }
  int n = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, shm);
  if (e != hipSuccess) { (void)hipGetLastError(); fail(LBM_EHIP, "occupancy query failed: %s", hipGetErrorString(e)); return -1; }
  (void)c;
  return n;
}

int run_regtile(lbm_ctx* c, int nsteps, float* av_vels, bool* done) {
  *done = false;
  Slab& s = c->slabs[0];
  HIPC(hipSetDevice(s.dev));
  const auto& t = c->tplan;
  const int ntiles = t.ntx * t.nty;
  const size_t mail_bytes = (size_t)ntiles * 2 * (size_t)lbm::regtile_box(t.ty);
  const bool fast = (c->variant & lbm::kFastMath) != 0;
  const dim3 grid(ntiles), block(64 * t.nw);
  const unsigned shm = (unsigned)lbm::regtile_lds_bytes(t.nw, t.r);
  const char* dbg = getenv("LBM_RESIDENT_DEBUG");   // timing experiments (wrong results): see lbm_regtile.hip.h
  static const bool want_stats = getenv("LBM_REGTILE_STATS") != nullptr;   // development: missed polls per run, and a trace
  const regtile_fn fn = regtile_kernel(t.r, fast, dbg ? atoi(dbg) : 0, want_stats && getenv("LBM_REGTILE_TRACE"), c->regtile_async != 0);
  if (c->tplan.bpc == 0) {                             // first run of this tiling: is every tile resident at once?
    const int n = regtile_prepare(c, reinterpret_cast<const void*>(fn), s.dev, (int)block.x, shm);
    c->tplan.bpc = (n < 0) ? -1 : n;
    if (n < 0) snprintf(c->resident_why, sizeof(c->resident_why), "%s", lbm_last_error());
    else if ((long)n * std::max(c->ncu, 1) < (long)ntiles) {
      c->tplan.bpc = -1;
      snprintf(c->resident_why, sizeof(c->resident_why), "%d tiles of %d waves, but the device takes %d block(s) per CU on %d CUs at once", ntiles, t.nw, n, c->ncu);
    }
  }
  if (c->tplan.bpc < 0) return fail(LBM_EINVAL, "register tiling not usable: %s", c->resident_why);
  if (!c->tmail) {
    // Uncached device memory where the device offers it: the granules are written once and read once, by another CU, and
    // every access is sc1 anyway -- without the L2 allocation a hand-off is shorter (1024x1024: 4.14 -> 3.48 us per step, found
    // when the slabs' mail areas, uncached for the sake of stores from other GPUs, ran faster than this one;
    // LBM_REGTILE_MAIL_CACHED=1: ordinary device memory)
    static const bool cached = getenv("LBM_REGTILE_MAIL_CACHED") && atoi(getenv("LBM_REGTILE_MAIL_CACHED"));
    if (cached || hipExtMallocWithFlags((void**)&c->tmail, mail_bytes, hipDeviceMallocUncached) != hipSuccess) {
      (void)hipGetLastError();
      c->tmail = nullptr;
      HIPC(hipMalloc((void**)&c->tmail, mail_bytes));
    }
    HIPC(hipMemsetAsync(c->tmail, 0, mail_bytes, s.sc));
    if (!c->rabort) {
      HIPC(hipMalloc((void**)&c->rabort, 64));
      HIPC(hipMemsetAsync(c->rabort, 0, 64, s.sc));
    }
  }
  // Tags only ever grow (a freshly zeroed mailbox is valid for any tag >= 1), except here: before they would wrap, the
  // mailboxes are cleared and the count starts over.
  if ((unsigned long long)c->rtag + (unsigned long long)nsteps >= 0x7fffff00ull) {
    HIPC(hipMemsetAsync(c->tmail, 0, mail_bytes, s.sc));
    c->rtag = 1;
  }
  // per-step tile sums
  const int rtiles = ntiles;
  if (c->rpartials_cap < nsteps || c->rpartials_tiles < rtiles) {
    long cap = std::max(1024L, c->rpartials_cap);
    while (cap < nsteps) cap *= 2;
    if (c->rpartials) HIPC(hipFree(c->rpartials));
    c->rpartials = nullptr; c->rpartials_cap = 0;
    HIPC(hipMalloc((void**)&c->rpartials, sizeof(float) * (size_t)cap * rtiles));
    c->rpartials_cap = cap; c->rpartials_tiles = rtiles;
  }
  int rc = ensure_sums(s, nsteps);
  if (rc) return rc;
  lbm::RegTileArgs a;
  a.src = s.lat[c->cur]; a.dst = s.lat[c->cur ^ 1];
  a.plane = s.plane; a.pitch = s.pitch; a.nx = c->p.nx; a.ny = c->p.ny;
  a.blocked = s.blocked; a.omega = c->p.omega;
  a.accel_row = c->p.ny - 2;
  a.a1 = c->p.density * c->p.accel / 9.f; a.a2 = c->p.density * c->p.accel / 36.f;
  a.ty = t.ty; a.ntx = t.ntx; a.nty = t.nty;
  a.nsteps = nsteps; a.tag0 = c->rtag;
  a.mail = c->tmail; a.mail_bytes = (unsigned)mail_bytes; a.partials = c->rpartials; a.abort_word = c->rabort;
  a.fault = getenv("LBM_REGTILE_FAULT") ? 1 : 0;   // (tests: a tile that never starts)
  a.stats = nullptr;
  static unsigned long long* stats_buf = nullptr;
  constexpr size_t kStatsWords = 4 + 16 * 4 * 16 + 72;    // (+ the first wave that gave up: lbm_regtile.hip.h, await)
  if (want_stats) {
    if (!stats_buf) HIPC(hipMalloc((void**)&stats_buf, kStatsWords * 8));
    unsigned long long head[4] = {0, 0, (unsigned long long)(getenv("LBM_REGTILE_TRACE_TILE") ? atoi(getenv("LBM_REGTILE_TRACE_TILE")) : ntiles / 2 + t.ntx / 2),
                                  (unsigned long long)(getenv("LBM_REGTILE_TRACE_STEP") ? atoi(getenv("LBM_REGTILE_TRACE_STEP")) : nsteps / 2)};
    HIPC(hipMemsetAsync(stats_buf, 0, kStatsWords * 8, s.sc));
    HIPC(hipMemcpyAsync(stats_buf, head, sizeof head, hipMemcpyHostToDevice, s.sc));
    HIPC(hipStreamSynchronize(s.sc));
    a.stats = stats_buf;
  }
  c->rtag += (uint32_t)nsteps + 1u;   // (the last step's mail is sent too, and must never be taken for the next run's state 0)
  s.err_host[1] = 0;   // lbm_fold_steps stores the abort word here
  const auto wall0 = std::chrono::steady_clock::now();
  HIPC(hipEventRecord(s.ev_t0, s.sc));
  hipLaunchKernelGGL(fn, grid, block, shm, s.sc, a);
  HIPC(hipGetLastError());
  hipLaunchKernelGGL(lbm::lbm_fold_steps, dim3(cdiv(nsteps, lbm::kBlock / 64)), dim3(lbm::kBlock), 0, s.sc,
                     c->rpartials, ntiles, nsteps, s.sums, c->rabort, s.err_host + 1);
  HIPC(hipGetLastError());
  HIPC(hipEventRecord(s.ev_t1, s.sc));
  rc = collect_sums(c, nsteps, av_vels, wall0);
  if (rc) return rc;
  if (want_stats) {
    std::vector<unsigned long long> st(kStatsWords);
    HIPC(hipMemcpy(st.data(), stats_buf, kStatsWords * 8, hipMemcpyDeviceToHost));
    fprintf(stderr, "lbm_regtile: %d steps, %d waves: %llu waits found their mail missing (%.3f per wave and step), %llu extra fetches\n",
            nsteps, ntiles * t.nw, st[0], (double)st[0] / ((double)nsteps * ntiles * t.nw), st[1]);
    if (st[1028] != 0) {
      fprintf(stderr, "lbm_regtile: the first wave to give up: tile %llu (of %d x %d) wave %llu row %llu, waiting for tag %llu (run's tag0 %u); tags it holds, lane: couriers / edge row\n",
              st[1028] - 1, t.ntx, t.nty, st[1029], st[1030], st[1031], a.tag0);
      for (int l : {0, 1, 2, 3, 31, 60, 61, 62, 63}) fprintf(stderr, "   lane %2d: %llu / %llu\n", l, st[1032 + l] >> 32, st[1032 + l] & 0xffffffffull);
    }
    if (getenv("LBM_REGTILE_TRACE")) {
      unsigned long long t0 = ~0ull;
      for (size_t i = 4; i < 4 + 16 * 4 * 16; ++i) if (st[i] && st[i] < t0) t0 = st[i];
      fprintf(stderr, "trace of tile %llu from step %llu (shader clocks / 100 since the first stamp; slots: barrier | per row: start, mail, done | end)\n", st[2], st[3]);
      for (int ww = 0; ww < t.nw; ++ww)
        for (int q = 0; q < 4; ++q) {
          fprintf(stderr, "  wave %2d step +%d:", ww, q);
          for (int k = 0; k < 14; ++k) {
            const unsigned long long v = st[4 + ((ww * 4 + q) * 16 + k)];
            if (k == 1 || k == 13 || (k > 1 && (k - 1) % 3 == 0)) fprintf(stderr, " |");
            fprintf(stderr, " %6.1f", v ? (double)(v - t0) / 100.0 : -1.0);
          }
          fprintf(stderr, "\n");
        }
    }
  }
  if (s.err_host[1] != 0) {
    HIPC(hipMemsetAsync(c->rabort, 0, 64, s.sc));
    HIPC(hipStreamSynchronize(s.sc));
    resident_give_up(c, "a tile waited 1 s for a neighbour: not every tile was running at once");
    return LBM_OK;
  }
  c->cur ^= 1;
  *done = true;
  return LBM_OK;
}

// ---- register tiles ACROSS SLABS (the multi-GPU half): every slab keeps its rows in the registers of its
// own GPU for the whole run, and the granules that leave a slab through its bottom / top edge go straight into the
// neighbouring slab's mailboxes (lbm_regtile.hip.h, kRegSlab) -- over xGMI when that slab lives on another GPU.  Same
// tiling on every slab (equal slabs, 64-column tiles of ty rows); the slabs of one device go in ONE launch (their tiles
// wait for each other, so they must be resident together).  Contexts whose neighbours can store into each other's
// memory: slabs of one process (copy and peer-to-peer contexts: pointers, peer access across devices), and one process
// per GPU with peer-to-peer halos (hipIpc mappings, handles in the halo block; needs the communicator, through which the
// ranks agree after every run whether anybody gave up).
int regtile_slab_count(const lbm_ctx* c) { return c->rank_mode ? c->nranks : (int)c->slabs.size(); }

bool regtile_slabs_possible(const lbm_ctx* c) {
  if (c->exchange != LBM_EXCHANGE_P2P && c->exchange != LBM_EXCHANGE_COPY) return false;
  // (ranks without a communicator cannot agree on whether anybody gave up: the streaming kernels, unless a test that adds up
  // the ranks' results itself says otherwise)
  static const bool trust = getenv("LBM_REGTILE_SLABS_NO_AGREEMENT") && atoi(getenv("LBM_REGTILE_SLABS_NO_AGREEMENT"));
  if (c->rank_mode && c->nranks > 1 && ((c->no_comm && !trust) || c->exchange != LBM_EXCHANGE_P2P)) return false;
  const int n = regtile_slab_count(c);
  return c->p.nx % 64 == 0 && n >= 1 && c->p.ny % n == 0;
}

// Tiling: as for a lattice alone (as few rows per wave as fit, on at most half the CUs where possible), counted per device.
bool plan_regtile_slabs(lbm_ctx* c) {
  c->splan.ty = 0;
  if (!regtile_slabs_possible(c)) return false;
  const char* off = getenv("LBM_REGTILE_SLABS");
  if (off && atoi(off) == 0) return false;
  const int nyl = c->p.ny / regtile_slab_count(c);
  int per_dev = 1;
  for (auto& a : c->slabs) {
    int n = 0;
    for (auto& b : c->slabs) n += (b.dev == a.dev) ? 1 : 0;
    per_dev = std::max(per_dev, n);
  }
  // (development: LBM_REGTILE_SLAB_TILING = rows per tile x 10 + rows per wave, as the `regtile` option of a lone lattice)
  const int forced = getenv("LBM_REGTILE_SLAB_TILING") ? atoi(getenv("LBM_REGTILE_SLAB_TILING")) : 0;
  int ty = 0, r = 0;
  if (forced > 0) {
    ty = forced / 10; r = forced % 10;
    if (!(r == 1 || r == 2 || r == 4) || ty < r || ty % r != 0 || ty / r > 16 || nyl % ty != 0 ||
        (long)per_dev * (c->p.nx / 64) * (nyl / ty) > (long)c->ncu) return false;
  } else if (!regtile_default_tiling(c, nyl, per_dev, &ty, &r)) return false;
  c->splan.ty = ty; c->splan.r = r; c->splan.nw = ty / r; c->splan.ntx = c->p.nx / 64; c->splan.nty = nyl / ty; c->splan.bpc = 0;
  return true;
}

typedef void (*regtile_slabs_fn)(const lbm::RegTileArgs*);
regtile_slabs_fn regtile_slabs_kernel(int r, bool fast, bool async) {
  constexpr int AS_ = lbm::kRegAsync, SL_ = lbm::kRegSlab;
  if (async && r == 4) return fast ? lbm::lbm_regtile_slabs<4, SL_ | AS_ | 1> : lbm::lbm_regtile_slabs<4, SL_ | AS_>;
  if (async && r == 2) return fast ? lbm::lbm_regtile_slabs<2, SL_ | AS_ | 1> : lbm::lbm_regtile_slabs<2, SL_ | AS_>;
  switch (r) {
    case 4: return fast ? lbm::lbm_regtile_slabs<4, SL_ | 1> : lbm::lbm_regtile_slabs<4, SL_>;
    case 2: return fast ? lbm::lbm_regtile_slabs<2, SL_ | 1> : lbm::lbm_regtile_slabs<2, SL_>;
    default: return fast ? lbm::lbm_regtile_slabs<1, SL_ | 1> : lbm::lbm_regtile_slabs<1, SL_>;
  }
}

size_t regtile_slab_mail_bytes(const lbm_ctx* c) {
  return (size_t)c->splan.ntx * c->splan.nty * 2 * (size_t)lbm::regtile_box(c->splan.ty);
}

// A slab's mail area.  Uncached device memory, like the peer-to-peer halo blocks: a neighbour on another GPU stores into it
// behind this GPU's L2 (LBM_REGTILE_MAIL_CACHED=1: ordinary device memory, to measure what that costs on one GPU).
int regtile_slab_mail_alloc(lbm_ctx* c, Slab& s) {
  if (s.tmail) return LBM_OK;
  HIPC(hipSetDevice(s.dev));
  const size_t bytes = regtile_slab_mail_bytes(c);
  static const bool cached = getenv("LBM_REGTILE_MAIL_CACHED") && atoi(getenv("LBM_REGTILE_MAIL_CACHED"));
  hipError_t e = cached ? hipMalloc((void**)&s.tmail, bytes) : hipExtMallocWithFlags((void**)&s.tmail, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); s.tmail = nullptr; return fail(LBM_EHIP, "cannot allocate the mail area of a slab: %s", hipGetErrorString(e)); }
  HIPC(hipMemset(s.tmail, 0, bytes));
  HIPC(hipDeviceSynchronize());
  s.tmail_bytes = bytes;
  return LBM_OK;
}

void regtile_slabs_free(lbm_ctx* c) {
  for (auto& s : c->slabs) {
    (void)hipSetDevice(s.dev);
    for (int side = 0; side < 2; ++side) {
      if (s.tmail_nb_ipc[side] && s.tmail_nb[side] && (side == 0 || s.tmail_nb[1] != s.tmail_nb[0])) (void)hipIpcCloseMemHandle(s.tmail_nb[side]);
      s.tmail_nb[side] = nullptr; s.tmail_nb_ipc[side] = false;
    }
    if (s.tmail) (void)hipFree(s.tmail);
    if (s.rpartials) (void)hipFree(s.rpartials);
    if (s.rabort) (void)hipFree(s.rabort);
    if (s.ev_rt) (void)hipEventDestroy(s.ev_rt);
    s.tmail = nullptr; s.rpartials = nullptr; s.rabort = nullptr; s.ev_rt = nullptr; s.rpartials_cap = 0;
  }
  if (c->rtable) (void)hipHostFree(c->rtable);
  c->rtable = c->rtable_dev = nullptr;
}

bool regtile_slabs_usable(const lbm_ctx* c) {
  if (c->splan.ty <= 0 || c->resident_broken || (c->variant & 8) != 0 || !(c->engine == 0 || c->engine == 3)) return false;
  if (!regtile_slabs_possible(c)) return false;
  if (c->rank_mode && c->nranks > 1) {
    if (!c->p2p_connected) return false;
    for (int side = 0; side < 2; ++side) if (!c->slabs[0].tmail_nb[side]) return false;
  }
  return true;
}

int run_regtile_slabs(lbm_ctx* c, int nsteps, float* av_vels, bool* done) {
  *done = false;
  const auto& t = c->splan;
  const int ns = (int)c->slabs.size(), ntiles = t.ntx * t.nty;
  const bool fast = (c->variant & lbm::kFastMath) != 0;
  const dim3 block(64 * t.nw);
  const unsigned shm = (unsigned)lbm::regtile_lds_bytes(t.nw, t.r);
  const regtile_slabs_fn fn = regtile_slabs_kernel(t.r, fast, c->regtile_async != 0);
  int rc;
  // device groups: the local slabs in the order of their devices' first appearance
  std::vector<int> order, gstart;          // order[k] = slab index; gstart[g] = first k of group g (+ end)
  {
    std::vector<bool> taken(ns, false);
    for (int i = 0; i < ns; ++i) {
      if (taken[i]) continue;
      gstart.push_back((int)order.size());
      for (int j = i; j < ns; ++j) if (!taken[j] && c->slabs[j].dev == c->slabs[i].dev) { taken[j] = true; order.push_back(j); }
    }
    gstart.push_back((int)order.size());
  }
  const int ngroups = (int)gstart.size() - 1;
  if (c->splan.bpc == 0) {           // first run: is every tile of every device resident at once?
    int worst = 1 << 30, most = 1;
    for (int g = 0; g < ngroups; ++g) {
      Slab& l = c->slabs[order[gstart[g]]];
      HIPC(hipSetDevice(l.dev));
      const int n = regtile_prepare(c, reinterpret_cast<const void*>(fn), l.dev, (int)block.x, shm);
      if (n < 0) { c->splan.bpc = -1; snprintf(c->resident_why, sizeof(c->resident_why), "%s", lbm_last_error()); break; }
      worst = std::min(worst, n); most = std::max(most, gstart[g + 1] - gstart[g]);
    }
    if (c->splan.bpc == 0) {
      c->splan.bpc = worst;
      if ((long)worst * std::max(c->ncu, 1) < (long)ntiles * most) {
        c->splan.bpc = -1;
        snprintf(c->resident_why, sizeof(c->resident_why), "%d slabs x %d tiles of %d waves on one device, but it takes %d block(s) per CU on %d CUs at once", most, ntiles, t.nw, worst, c->ncu);
      }
    }
  }
  if (c->splan.bpc < 0) return fail(LBM_EINVAL, "register tiling across slabs not usable: %s", c->resident_why);
  // peer access between the devices of neighbouring slabs (one process)
  if (!c->rank_mode)
    for (int i = 0; i < ns; ++i)
      for (int d : {(i + ns - 1) % ns, (i + 1) % ns}) {
        const int a = c->slabs[i].dev, b = c->slabs[d].dev;
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) { (void)hipGetLastError(); return fail(LBM_EHIP, "device %d cannot store into device %d", a, b); }
        HIPC(hipSetDevice(a));
        const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
        (void)hipGetLastError();
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(LBM_EHIP, "hipDeviceEnablePeerAccess(%d -> %d): %s", a, b, hipGetErrorString(e));
      }
  for (auto& s : c->slabs) {
    if ((rc = regtile_slab_mail_alloc(c, s))) return rc;
    HIPC(hipSetDevice(s.dev));
    if (!s.ev_rt) HIPC(hipEventCreateWithFlags(&s.ev_rt, hipEventDisableTiming));
    if (s.rpartials_cap < nsteps) {
      long cap = std::max(1024L, s.rpartials_cap);
      while (cap < nsteps) cap *= 2;
      if (s.rpartials) HIPC(hipFree(s.rpartials));
      s.rpartials = nullptr; s.rpartials_cap = 0;
      HIPC(hipMalloc((void**)&s.rpartials, sizeof(float) * (size_t)cap * ntiles));
      s.rpartials_cap = cap;
    }
    if ((rc = ensure_sums(s, nsteps + 1))) return rc;
  }
  for (int g = 0; g < ngroups; ++g) {
    Slab& l = c->slabs[order[gstart[g]]];
    if (!l.rabort) {
      HIPC(hipSetDevice(l.dev));
      HIPC(hipMalloc((void**)&l.rabort, 64));
      HIPC(hipMemset(l.rabort, 0, 64));
    }
  }
  if (!c->rtable) {
    HIPC(hipHostMalloc((void**)&c->rtable, sizeof(lbm::RegTileArgs) * ns, hipHostMallocPortable | hipHostMallocMapped));
    HIPC(hipHostGetDevicePointer((void**)&c->rtable_dev, c->rtable, 0));
  }
  // tags: as for a lattice alone; every slab (every rank) counts the same runs, so all hold the same tag0
  if ((unsigned long long)c->rtag + (unsigned long long)nsteps >= 0x7fffff00ull)
    return fail(LBM_EINVAL, "mailbox tags of the slabs exhausted (2^31 steps on one context)");
  for (int g = 0; g < ngroups; ++g)
    for (int k = gstart[g]; k < gstart[g + 1]; ++k) {
      const int i = order[k];
      Slab& s = c->slabs[i];
      lbm::RegTileArgs& a = c->rtable[k];
      a.src = s.lat[c->cur]; a.dst = s.lat[c->cur ^ 1];
      a.plane = s.plane; a.pitch = s.pitch; a.nx = c->p.nx; a.ny = s.nyl;
      a.blocked = s.blocked; a.omega = c->p.omega;
      a.accel_row = s.accel_row;
      a.a1 = c->p.density * c->p.accel / 9.f; a.a2 = c->p.density * c->p.accel / 36.f;
      a.ty = t.ty; a.ntx = t.ntx; a.nty = t.nty;
      a.nsteps = nsteps; a.tag0 = c->rtag;
      a.mail = s.tmail; a.mail_bytes = (unsigned)s.tmail_bytes;
      a.partials = s.rpartials; a.abort_word = c->slabs[order[gstart[g]]].rabort;
      a.fault = (getenv("LBM_REGTILE_FAULT") && i == 0) ? 1 : 0;
      a.stats = nullptr;
      if (c->rank_mode && c->nranks > 1) {
        a.mail_s = s.tmail_nb[0]; a.mail_n = s.tmail_nb[1];
        a.mail_bytes_s = (unsigned)s.tmail_nb_bytes[0]; a.mail_bytes_n = (unsigned)s.tmail_nb_bytes[1];
      } else {                        // (one process, or a ring of one rank: the neighbours are local slabs)
        Slab& so = c->slabs[(i + ns - 1) % ns];
        Slab& no = c->slabs[(i + 1) % ns];
        a.mail_s = so.tmail; a.mail_n = no.tmail;
        a.mail_bytes_s = (unsigned)so.tmail_bytes; a.mail_bytes_n = (unsigned)no.tmail_bytes;
      }
      a.nty_s = t.nty; a.nty_n = t.nty;
    }
  c->rtag += (uint32_t)nsteps + 1u;
  const auto wall0 = std::chrono::steady_clock::now();
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    s.err_host[1] = 0;
    HIPC(hipEventRecord(s.ev_t0, s.sc));
  }
  for (int g = 0; g < ngroups; ++g) {
    Slab& l = c->slabs[order[gstart[g]]];
    HIPC(hipSetDevice(l.dev));
    for (int k = gstart[g] + 1; k < gstart[g + 1]; ++k) HIPC(hipStreamWaitEvent(l.sc, c->slabs[order[k]].ev_t0, 0));
    hipLaunchKernelGGL(fn, dim3(ntiles, gstart[g + 1] - gstart[g]), block, shm, l.sc, c->rtable_dev + gstart[g]);
    HIPC(hipGetLastError());
    HIPC(hipEventRecord(l.ev_rt, l.sc));
    for (int k = gstart[g]; k < gstart[g + 1]; ++k) {
      Slab& s = c->slabs[order[k]];
      if (k > gstart[g]) HIPC(hipStreamWaitEvent(s.sc, l.ev_rt, 0));
      hipLaunchKernelGGL(lbm::lbm_fold_steps, dim3(cdiv(nsteps, lbm::kBlock / 64)), dim3(lbm::kBlock), 0, s.sc,
                         s.rpartials, ntiles, nsteps, s.sums, l.rabort, s.err_host + 1);
      HIPC(hipGetLastError());
      HIPC(hipEventRecord(s.ev_t1, s.sc));
    }
  }
  // did anybody give up?  One process: the abort words are all here.  One process per GPU: the ranks must agree (a rank whose
  // neighbour stopped notices a second later; one far away in a short run might not at all), so the word rides as one more
  // double behind the per-step sums through the all-reduce that ends the run.
  const bool agree = c->rank_mode && c->slabs[0].comm != nullptr;
  if (agree) {
    Slab& s = c->slabs[0];
    hipLaunchKernelGGL(lbm::lbm_abort_to_sum, dim3(1), dim3(64), 0, s.sc, s.rabort, s.sums + nsteps);
    HIPC(hipGetLastError());
  }
  rc = collect_sums(c, nsteps, av_vels, wall0, agree ? 1 : 0);
  if (rc) return rc;
  bool gave_up = false;
  for (auto& s : c->slabs) gave_up = gave_up || s.err_host[1] != 0;
  if (agree) gave_up = gave_up || c->slabs[0].sums_host[nsteps] != 0.0;
  if (gave_up) {
    for (int g = 0; g < ngroups; ++g) {
      Slab& l = c->slabs[order[gstart[g]]];
      HIPC(hipSetDevice(l.dev));
      HIPC(hipMemsetAsync(l.rabort, 0, 64, l.sc));
      HIPC(hipStreamSynchronize(l.sc));
    }
    resident_give_up(c, "a tile waited 1 s for a neighbour (register tiles across slabs): not every tile was running at once");
    return LBM_OK;
  }
  c->cur ^= 1;
  *done = true;
  return LBM_OK;
}

// Peer-to-peer pointers of a slab for launch group `seq` (see lbm::P2PSync); bumps the completion
// targets by the number of blocks that will count themselves done on each side.
lbm::P2PSync p2p_sync(Slab& s, uint32_t seq, int blocks_s, int blocks_n) {
  lbm::P2PSync y;
  const size_t f = 4 * s.halo_bytes;
  y.flag_s = (const uint32_t*)(s.comm_block + f);
  y.flag_n = (const uint32_t*)(s.comm_block + f + 256);
  y.rem_flag_s = (uint32_t*)(s.peer_s + f + 256);   // I am the south neighbour's NORTH side
  y.rem_flag_n = (uint32_t*)(s.peer_n + f);
  y.cnt_s = s.counters; y.cnt_n = s.counters + 16; y.err = s.counters + 32;
  s.cnt_s_total += (uint32_t)blocks_s; s.cnt_n_total += (uint32_t)blocks_n;
  y.cnt_target_s = s.cnt_s_total; y.cnt_target_n = s.cnt_n_total;
  y.seq = seq;
  return y;
}
inline float* p2p_remote_s(const Slab& s, uint32_t seq) { return (float*)(s.peer_s + (size_t)(2 + (seq & 1)) * s.halo_bytes); }  // its ghost_n
inline float* p2p_remote_n(const Slab& s, uint32_t seq) { return (float*)(s.peer_n + (size_t)(seq & 1) * s.halo_bytes); }        // its ghost_s

// Peer-to-peer contexts march too (lbm_march, neighbours' rows read in place over xGMI) where every slab fills the
// chip.  The decision uses the lattice, the number of slabs and the options only -- every rank must come to the same
// answer, the two protocols do not mix.
bool p2p_march_on(const lbm_ctx* c) {
  const int K = slab_K(c);
  if (K == 0 || c->exchange != LBM_EXCHANGE_P2P) return false;
  const int rows = c->p.ny / c->nranks;                     // the smallest slab
  if (rows < 4 * K || (double)(rows + 1) * c->slabs[0].pitch * 4.0 >= 4.0e9) return false;
  for (auto& s : c->slabs) if (!s.nb_lat[0][0] || !s.nb_lat[1][0]) return false;   // (connect failed: an error everywhere)
  return true;
}
bool p2p_march_pays(const lbm_ctx* c) {                      // same estimate as for a lone lattice, on the smallest slab
  const int rows = c->p.ny / c->nranks, h = march_rows_for(c, rows), ns = cdiv(c->p.nx, lbm::MarchCfg<kMarchK>::WOUT), ncu = std::max(c->ncu, 1);
  const long blocks = (long)ns * cdiv(rows, h), rounds = (blocks + ncu - 1) / ncu;
  return (double)rows * ns / ((double)rounds * ncu * (h + 3 * (kMarchK - 1))) >= 0.65;
}
// lbm_wave<8> instead of lbm_march on slabs of `rows` rows?  When its waves fill at least most of one round of the
// chip's wave slots.  Measured on one GPU (tools/strong_scaling_proxy.py, us per step, lbm_wave<8> against lbm_march):
// 8192 x 4096 126 / 135, 8192 x 2048 69.3 / 70.3, 8192 x 1024 38.7 / 39.4 -- a little ahead everywhere, with half as
// many launches (and flag hand-offs over xGMI) per step; 1024-wide slabs (22 wave columns) are far too narrow: 12.4 / 5.4.
bool slab_wave_pays(const lbm_ctx* c, int rows, int K) {
  if (c->p.nx < 64 || rows < 32 || c->march_kernel == 0) return false;
  const int h = slab_wave_rows(c, rows, K);
  return (double)cdiv(c->p.nx, wave_out_cols(c, K)) * cdiv(rows, h) >= 0.85 * wave_slots(c, K);
}

// The step loop with peer-to-peer halos: one stream per slab, no events, no host-side exchange.
// Two-step launches carry the hand-off themselves (edge tiles first); single steps are bracketed
// by a wait launch and a push launch.
int run_p2p(lbm_ctx* c, int nsteps, float* av_vels) {
  if (!c->p2p_connected) return fail(LBM_EINVAL, "peer-to-peer halos are not connected (lbm_p2p_connect)");
  const int nx = c->p.nx;
  const float a1 = c->p.density * c->p.accel / 9.f, a2 = c->p.density * c->p.accel / 36.f;
  const bool pairs = t2_eligible(c) && nsteps >= 2;
  const int ntx = nx / kT2X;
  const int push_grid = cdiv(nx, lbm::kBlock);
  int rc;
  for (auto& s : c->slabs)
    if ((rc = ensure_sums(s, nsteps))) return rc;
  if (p2p_march_on(c) && nsteps >= slab_K(c) && (rc = check_march_partials(c, true))) return rc;   // (before anything is queued)

  auto push = [&](Slab& s, const float* lat, uint32_t seq, bool do_push) -> int {
    const int grid = do_push ? push_grid : 1;
    lbm::P2PSync y = p2p_sync(s, seq, do_push ? grid : 0, do_push ? grid : 0);
    hipLaunchKernelGGL(lbm::lbm_p2p_push, dim3(grid), dim3(lbm::kBlock), 0, s.sc, lat, s.plane, s.pitch, nx, s.nyl,
                       p2p_remote_s(s, seq), p2p_remote_n(s, seq), y, do_push ? 1 : 0);
    HIPC(hipGetLastError());
    return LBM_OK;
  };

  // ---- prologue: accelerate phase of the first step
  uint32_t seq = 0;
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    if (s.accel_row >= 0) {
      hipLaunchKernelGGL(lbm::lbm_accelerate_row, dim3(cdiv(nx, 256)), dim3(256), 0, s.sc,
                         s.lat[c->cur], s.plane, s.pitch, nx, s.accel_row, s.blocked, a1, a2);
      HIPC(hipGetLastError());
    }
  }
  const auto wall0 = std::chrono::steady_clock::now();
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    HIPC(hipEventRecord(s.ev_t0, s.sc));
  }

  int li = 0, tt = 0;
  // ---- groups of K steps with lbm_march: the K ghost rows either side are read straight out of the neighbours'
  // lattices.  Launch group seq of a slab starts once both neighbours have raised seq-1 ("my launch seq-1 is over":
  // their rows are final, and they no longer read the lattice this launch overwrites) and raises seq when it is over.
  if (p2p_march_on(c) && nsteps >= slab_K(c)) {
    const int K = slab_K(c);
    auto raise = [&](Slab& s, uint32_t q) -> int {
      const size_t f = 4 * s.halo_bytes;
      hipLaunchKernelGGL(lbm::lbm_p2p_raise, dim3(1), dim3(64), 0, s.sc, (uint32_t*)(s.peer_s + f + 256), (uint32_t*)(s.peer_n + f), q);
      HIPC(hipGetLastError());
      return LBM_OK;
    };
    seq = ++c->seq;                                     // "the starting lattice is in place"
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      if ((rc = push(s, nullptr, seq, false))) return rc;   // (the neighbours are through with the previous run)
      if ((rc = raise(s, seq))) return rc;
    }
    const int ngroups = nsteps / K;
    for (int g = 0; g < ngroups; ++g, ++li, tt += K) {
      seq = ++c->seq;
      const int q = li & 1;
      for (auto& s : c->slabs) {
        HIPC(hipSetDevice(s.dev));
        if ((rc = push(s, nullptr, seq, false))) return rc;   // wait for both neighbours' seq-1
        const SlabNb nbr{s.nb_lat[0][c->cur], s.nb_lat[1][c->cur], s.nb_plane[0], s.nb_plane[1], s.nb_nyl[0], s.nb_nyl[1],
                         s.nb_blocked[0], s.nb_blocked[1]};
        if ((rc = launch_slab_pass(c, s, nbr, K, q, tt, tt + K < nsteps, g > 0))) return rc;
        if ((rc = raise(s, seq))) return rc;
      }
      c->cur ^= 1;
    }
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      const int nb = march_slab_blocks(c, s);
      hipLaunchKernelGGL(lbm::lbm_fold_partials, dim3(K), dim3(lbm::kBlock), 0, s.sc, s.partials[(li - 1) & 1], nb,
                         s.sums + (tt - K), nb);
      HIPC(hipGetLastError());
    }
  }
  // ---- the remaining steps trade halos: push those of the lattice as it stands
  if (tt < nsteps) {
    seq = ++c->seq;
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      if ((rc = push(s, s.lat[c->cur], seq, true))) return rc;
    }
  }
  if (pairs && nsteps - tt >= 2) {
    const int npairs = (nsteps - tt) / 2;
    for (int j = 0; j < npairs; ++j, ++li, tt += 2) {
      seq = ++c->seq;
      const int q = li & 1, qp = q ^ 1;
      for (auto& s : c->slabs) {
        HIPC(hipSetDevice(s.dev));
        const int nty = s.nyl / kT2Y, nbtot = ntx * nty;
        lbm::Sweep2Args a;
        a.src = s.lat[c->cur]; a.dst = s.lat[c->cur ^ 1];
        a.plane = s.plane; a.pitch = s.pitch; a.nx = nx; a.ny = s.nyl;
        a.blocked = s.blocked; a.omega = c->p.omega;
        a.accel_row = s.accel_row >= 0 ? s.accel_row : lbm::kNoRow;
        a.accel_out = (tt + 2 < nsteps) ? 1 : 0;
        a.a1 = a1; a.a2 = a2;
        a.partials1 = s.partials[q]; a.partials2 = s.partials[q] + nbtot;
        a.prev1 = a.prev2 = nullptr; a.prev_count = 0; a.prev_sum = nullptr;
        if (j > 0) { a.prev1 = s.partials[qp]; a.prev2 = s.partials[qp] + nbtot; a.prev_count = nbtot; a.prev_sum = s.sums + (tt - 2); }
        a.by_begin = 0; a.by_count = nty; a.by_stride = 1;
        a.ghost_s = s.ghost_s[(seq - 1) & 1]; a.ghost_n = s.ghost_n[(seq - 1) & 1];
        a.blocked_gs = s.blocked_gs; a.blocked_gn = s.blocked_gn;
        a.send_s = p2p_remote_s(s, seq); a.send_n = p2p_remote_n(s, seq);
        a.sync = p2p_sync(s, seq, ntx, ntx);
        launch_sweep2_k<lbm::kSweep2P2P>(c, a, nbtot, s.sc);
        HIPC(hipGetLastError());
      }
      c->cur ^= 1;
    }
    const int ql = (li - 1) & 1;
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      const int nbtot = ntx * (s.nyl / kT2Y);
      hipLaunchKernelGGL(lbm::lbm_fold_partials, dim3(2), dim3(lbm::kBlock), 0, s.sc, s.partials[ql], nbtot, s.sums + (tt - 2), nbtot);
      HIPC(hipGetLastError());
    }
  }
  const int first_single = tt;
  for (; tt < nsteps; ++tt, ++li) {
    seq = ++c->seq;
    const int q = li & 1, qp = q ^ 1;
    const bool last = (tt == nsteps - 1);
    const long h3 = 3L * nx;
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      if ((rc = push(s, nullptr, seq, false))) return rc;   // wait for the halos of launch seq-1
      lbm::SweepArgs a;
      a.src = s.lat[c->cur]; a.dst = s.lat[c->cur ^ 1];
      a.plane = s.plane; a.pitch = s.pitch; a.nx = nx; a.nyl = s.nyl;
      a.blocked = s.blocked; a.omega = c->p.omega;
      a.accel_row = last ? -1 : s.accel_row;
      a.a1 = a1; a.a2 = a2;
      a.partials = s.partials[q];
      const float* gs = s.ghost_s[(seq - 1) & 1] + h3;
      const float* gn = s.ghost_n[(seq - 1) & 1] + h3;
      a.south2 = gs; a.south5 = gs + nx; a.south6 = gs + 2 * nx;
      a.north4 = gn; a.north7 = gn + nx; a.north8 = gn + 2 * nx;
      a.send_south = a.send_north = nullptr;
      a.y_begin = 0; a.y_count = s.nyl; a.y_stride = 1;
      const int nb = sweep_blocks(c, s.nyl);
      a.prev_partials = nullptr; a.prev_count = 0; a.prev_sum = nullptr;
      if (tt > first_single) { a.prev_partials = s.partials[qp]; a.prev_count = nb; a.prev_sum = s.sums + (tt - 1); }
      launch_sweep(c, a, s.sc);
      HIPC(hipGetLastError());
      if ((rc = push(s, s.lat[c->cur ^ 1], seq, true))) return rc;   // the new edge rows, packed and pushed
    }
    c->cur ^= 1;
  }
  const int ql = (li - 1) & 1;
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    if (first_single < nsteps) {
      hipLaunchKernelGGL(lbm::lbm_fold_partials, dim3(1), dim3(lbm::kBlock), 0, s.sc, s.partials[ql],
                         sweep_blocks(c, s.nyl), s.sums + (nsteps - 1), 0);
      HIPC(hipGetLastError());
    }
    HIPC(hipEventRecord(s.ev_t1, s.sc));
  }
  return collect_sums(c, nsteps, av_vels, wall0);
}

}  // namespace

extern "C" int lbm_run(lbm_ctx* c, int nsteps, float* av_vels) {
  if (!c) return fail(LBM_EINVAL, "ctx is NULL");
  if (nsteps < 0) return fail(LBM_EINVAL, "nsteps < 0");
  if (nsteps == 0) { c->gpu_ms = c->wall_ms = 0.0; return LBM_OK; }
  if (c->p2p_failed) return fail(LBM_EHIP, "a peer-to-peer halo wait timed out earlier: this lattice is no longer defined");
  const int nx = c->p.nx;
  const float a1 = c->p.density * c->p.accel / 9.f;   // d2q9-bgk.c:230-231
  const float a2 = c->p.density * c->p.accel / 36.f;
  if (c->exchange != 0 && regtile_slabs_usable(c)) {
    bool done = false;
    int rr = run_regtile_slabs(c, nsteps, av_vels, &done);
    if (rr && c->engine == 0) {          // (as below: set-up failures of the automatic engine are not the caller's problem)
      (void)hipGetLastError();
      resident_give_up(c, lbm_last_error());
      rr = LBM_OK;
    }
    if (rr) return rr;
    if (done) { c->engine_last = 3; return LBM_OK; }
  }
  if (c->exchange != 0 && c->engine >= 2) return fail(LBM_EINVAL, "register tiles across slabs cannot run here (%s) (engine = %d)",
                                                      c->resident_why[0] ? c->resident_why : "no tiling", c->engine);
  if (c->exchange == LBM_EXCHANGE_P2P) { c->engine_last = 1; return run_p2p(c, nsteps, av_vels); }
  if (c->exchange == 0 && c->slabs.size() == 1 && (c->engine == 3 || c->engine == 0) && c->tplan.ty > 0 && !c->resident_broken &&
      (c->variant & 8) == 0) {
    bool done = false;
    int rr = run_regtile(c, nsteps, av_vels, &done);
    if (rr && c->engine == 0) {
      // automatic engine: a set-up or launch failure of the resident kernel (LDS attribute refused, tiles not all
      // resident, allocation failed) is not the caller's problem -- the source lattice is untouched, the streaming kernels run
      (void)hipGetLastError();
      resident_give_up(c, lbm_last_error());
      rr = LBM_OK;
    }
    if (rr) return rr;
    if (done) { c->engine_last = 3; return LBM_OK; }
  }
  if (c->engine >= 2) return fail(LBM_EINVAL, "the resident kernel cannot run here (%s), or this lattice has no resident tiling (engine = %d)",
                                  c->resident_why[0] ? c->resident_why : "no tiling", c->engine);
  c->engine_last = 1;
  const bool ex = c->exchange != 0;
  const bool pairs = t2_eligible(c) && nsteps >= 2;
  int rc;

  for (auto& s : c->slabs)
    if ((rc = ensure_sums(s, nsteps))) return rc;

  const bool slabs_march = ex && march_slabs_on(c) && nsteps >= slab_K(c);
  const bool bands = ex && !slabs_march && march_bands_on(c) && nsteps >= slab_K(c);   // RCCL transport: ghost bands
  if ((slabs_march || nsteps >= c->time_block) && (rc = check_march_partials(c, slabs_march))) return rc;   // (before anything is queued)
  if (bands) {
    if ((rc = bands_setup(c, slab_K(c)))) return rc;
    for (auto& s : c->slabs) {
      const BandPlan b = band_plan(c, s);
      if ((long)b.K * (b.nb_e + b.nb_i) > s.partial_cap)
        return fail(LBM_EINVAL, "marching kernel: %d blocks exceed the partial-sum buffer (raise wave_rows)", b.nb_e + b.nb_i);
    }
  }

  // ---- prologue: accelerate phase of the first step
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    if (s.accel_row >= 0) {
      hipLaunchKernelGGL(lbm::lbm_accelerate_row, dim3(cdiv(nx, 256)), dim3(256), 0, s.sc,
                         s.lat[c->cur], s.plane, s.pitch, nx, s.accel_row, s.blocked, a1, a2);
      HIPC(hipGetLastError());
    }
    if (slabs_march) HIPC(hipEventRecord(s.ev_march[1], s.sc));   // "launch -1": the starting lattice is in place
  }
  // halo buffers of the launches that trade halos (lbm_sweep2 / lbm_sweep on slabs), filled from the current lattice
  // as "launch par" (the launch before the first one that reads them)
  auto prime_halos = [&](int par) -> int {
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      if (s.nyl >= 2)
        hipLaunchKernelGGL(lbm::lbm_pack_halos9, dim3(cdiv(nx, 256)), dim3(256), 0, s.sc,
                           s.lat[c->cur], s.plane, s.pitch, nx, s.nyl, s.send_s[par], s.send_n[par]);
      else
        hipLaunchKernelGGL(lbm::lbm_pack_halos, dim3(cdiv(nx, 256)), dim3(256), 0, s.sc,
                           s.lat[c->cur], s.plane, s.pitch, nx, s.nyl, s.send_s[par] + 3L * nx, s.send_n[par] + 3L * nx);
      HIPC(hipGetLastError());
      HIPC(hipEventRecord(s.ev_bnd[par], s.sc));   // "edge rows of launch -1 are in place"
      if (split_edge_stream(c, s)) HIPC(hipEventRecord(s.ev_int[par], s.sc));   // "interior of launch -1 is done"
    }
    return pairs ? exchange_halos(c, par, 0, lbm::kHaloSlots) : exchange_halos(c, par, 3, 3);
  };
  if (ex && !slabs_march && !bands && (rc = prime_halos(1))) return rc;
  if (bands) {
    // the ghost bands of the starting lattice, as "group -1" (parity 1 of the events)
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      HIPC(hipEventRecord(s.ev_bnd[1], s.sc));
      if (split_edge_stream(c, s)) HIPC(hipEventRecord(s.ev_int[1], s.sc));
    }
    if ((rc = exchange_bands(c, 1, c->cur, slab_K(c)))) return rc;
  }

  const auto wall0 = std::chrono::steady_clock::now();
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    HIPC(hipEventRecord(s.ev_t0, s.sc));
  }

  // ---- the step loop (reference d2q9-bgk.c:180-201); no host sync inside.
  // Launch index li numbers the launch groups (a pair of steps or a single step); its parity
  // selects the halo / partial-sum buffers.
  int li = 0, tt = 0;
  if (bands) {                                         // groups of K steps, ghost bands by RCCL once per group
    const int K = slab_K(c), ngroups = nsteps / K;
    for (int g = 0; g < ngroups; ++g, ++li, tt += K)
      if ((rc = launch_band_group(c, li, tt, tt + K < nsteps, g > 0))) return rc;
    const int ql = (li - 1) & 1;
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      if (split_edge_stream(c, s)) HIPC(hipStreamWaitEvent(s.sc, s.ev_bnd[ql], 0));   // join the edge stream
      const BandPlan b = band_plan(c, s);
      const int nb = b.nb_e + b.nb_i;
      hipLaunchKernelGGL(lbm::lbm_fold_partials, dim3(K), dim3(lbm::kBlock), 0, s.sc, s.partials[ql], nb, s.sums + (tt - K), nb);
      HIPC(hipGetLastError());
    }
    if (tt < nsteps && (rc = prime_halos((li & 1) ^ 1))) return rc;   // the remaining steps trade halos
  } else
  if (slabs_march) {                                   // groups of K steps, row-marching, every slab of this process
    const int K = slab_K(c), ngroups = nsteps / K;
    for (int g = 0; g < ngroups; ++g, ++li, tt += K)
      if ((rc = launch_march_slabs(c, li, tt, tt + K < nsteps, g > 0))) return rc;
    for (auto& s : c->slabs) {
      HIPC(hipSetDevice(s.dev));
      const int nb = march_slab_blocks(c, s);
      hipLaunchKernelGGL(lbm::lbm_fold_partials, dim3(K), dim3(lbm::kBlock), 0, s.sc, s.partials[(li - 1) & 1], nb,
                         s.sums + (tt - K), nb);
      HIPC(hipGetLastError());
    }
    if (tt < nsteps) {
      // the remaining steps trade halos: every slab's marching launches must be over before a neighbour packs / copies
      for (auto& s : c->slabs) {
        HIPC(hipSetDevice(s.dev));
        for (auto& o : c->slabs) HIPC(hipStreamWaitEvent(s.sc, o.ev_march[(li - 1) & 1], 0));
      }
      if ((rc = prime_halos((li & 1) ^ 1))) return rc;
    }
  } else
  if (march_eligible(c) && nsteps >= c->time_block) {   // groups of K steps, row-marching (lone slab)
    const int K = c->time_block, ngroups = nsteps / K;
    const bool wave = use_wave_kernel(c);
    for (int g = 0; g < ngroups; ++g, ++li, tt += K)
      if ((rc = wave ? launch_wave(c, li, tt, tt + K < nsteps, g > 0) : launch_march(c, li, tt, tt + K < nsteps, g > 0))) return rc;
    Slab& s = c->slabs[0];
    const int nb = wave ? wave_blocks(c) : cdiv(nx, lbm::MarchCfg<kMarchK>::WOUT) * cdiv(c->p.ny, c->march_rows);
    hipLaunchKernelGGL(lbm::lbm_fold_partials, dim3(K), dim3(lbm::kBlock), 0, s.sc, s.partials[(li - 1) & 1], nb,
                       s.sums + (tt - K), nb);
    HIPC(hipGetLastError());
  }
  if (pairs && nsteps - tt >= 2) {
    const int npairs = (nsteps - tt) / 2;
    for (int j = 0; j < npairs; ++j, ++li, tt += 2)
      if ((rc = launch_pair(c, li, tt, tt + 2 < nsteps, j > 0, a1, a2))) return rc;
    const int ql = (li - 1) & 1;
    for (auto& s : c->slabs) {  // fold the last pair's partials
      HIPC(hipSetDevice(s.dev));
      if (ex && split_edge_stream(c, s)) HIPC(hipStreamWaitEvent(s.sc, s.ev_bnd[ql], 0));   // join the edge stream
      const int nbtot = cdiv(nx, kT2X) * cdiv(s.nyl, kT2Y);
      hipLaunchKernelGGL(lbm::lbm_fold_partials, dim3(2), dim3(lbm::kBlock), 0, s.sc, s.partials[ql], nbtot, s.sums + (tt - 2), nbtot);
      HIPC(hipGetLastError());
    }
  }
  const int first_single = tt;
  for (; tt < nsteps; ++tt, ++li)
    if ((rc = launch_single(c, li, tt, tt == nsteps - 1, tt > first_single, a1, a2))) return rc;

  // ---- epilogue: fold the last single step's partials, collect the per-step sums
  const int ql = (li - 1) & 1;
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    if (ex && split_edge_stream(c, s)) HIPC(hipStreamWaitEvent(s.sc, s.ev_bnd[ql], 0));     // join the edge stream
    if (first_single < nsteps) {
      hipLaunchKernelGGL(lbm::lbm_fold_partials, dim3(1), dim3(lbm::kBlock), 0, s.sc, s.partials[ql],
                         single_partial_count(c, s), s.sums + (nsteps - 1), 0);
      HIPC(hipGetLastError());
    }
    HIPC(hipEventRecord(s.ev_t1, s.sc));
    if (ex) HIPC(hipStreamWaitEvent(s.sc, s.ev_recv[ql], 0));  // drain the last exchange
  }
  return collect_sums(c, nsteps, av_vels, wall0);
}

extern "C" int lbm_last_run_ms(const lbm_ctx* c, double* gpu_ms, double* wall_ms) {
  if (!c) return fail(LBM_EINVAL, "ctx is NULL");
  if (gpu_ms) *gpu_ms = c->gpu_ms;
  if (wall_ms) *wall_ms = c->wall_ms;
  return LBM_OK;
}

extern "C" int lbm_read_state(lbm_ctx* c, float* out) {
  if (!c || !out) return fail(LBM_EINVAL, "NULL argument");
  const int nx = c->p.nx;
  const int base_row = c->rank_mode ? c->slabs[0].row0 : 0;
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    const long ncell = (long)s.nyl * nx;
    DeviceTemp t;
    HIPC(hipMalloc(&t.p, sizeof(float) * 9 * ncell));
    float* d_aos = (float*)t.p;
    hipLaunchKernelGGL(lbm::lbm_soa_to_aos, dim3(cdiv(ncell, 256)), dim3(256), 0, s.sc, s.lat[c->cur], d_aos, s.plane, s.pitch, nx, ncell);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(s.sc));
    HIPC(hipMemcpy(out + 9L * (s.row0 - base_row) * nx, d_aos, sizeof(float) * 9 * ncell, hipMemcpyDeviceToHost));
  }
  return LBM_OK;
}

// Runs lbm_derive on every local slab; returns global speed sum and mass.
static int derive_all(lbm_ctx* c, float* out4, double* speed_sum, double* mass) {
  const int nx = c->p.nx;
  const int base_row = c->rank_mode ? c->slabs[0].row0 : 0;
  double tot[2] = {0.0, 0.0};
  for (auto& s : c->slabs) {
    HIPC(hipSetDevice(s.dev));
    const long ncell = (long)s.nyl * nx;
    const int grid = cdiv(ncell, lbm::kBlock);
    DeviceTemp t;
    if (out4) HIPC(hipMalloc(&t.p, sizeof(float) * 4 * ncell));
    float* d_out = (float*)t.p;
    float* part = s.partials[0];  // idle between runs; capacity >= grid
    double* mpart = s.scratch_d;
    double* res = s.scratch_d + s.scratch_cap;  // 2 doubles: speed, mass
    hipLaunchKernelGGL(lbm::lbm_derive, dim3(grid), dim3(lbm::kBlock), 0, s.sc, s.lat[c->cur], s.plane, s.pitch, nx, ncell,
                       s.blocked, c->p.density, d_out, part, mpart);
    HIPC(hipGetLastError());
    hipLaunchKernelGGL(lbm::lbm_fold_partials, dim3(1), dim3(lbm::kBlock), 0, s.sc, part, grid, res, 0);
    hipLaunchKernelGGL(lbm::lbm_fold_double, dim3(1), dim3(lbm::kBlock), 0, s.sc, mpart, grid, res + 1);
    HIPC(hipGetLastError());
    if (c->rank_mode && s.comm != nullptr)
      NCCLC(rccl::AllReduce(res, res, 2, rccl::kFloat64, rccl::kSum, s.comm, s.sc));
    HIPC(hipStreamSynchronize(s.sc));
    double h[2];
    HIPC(hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost));
    tot[0] += h[0]; tot[1] += h[1];
    if (out4) {
      HIPC(hipMemcpy(out4 + 4L * (s.row0 - base_row) * nx, d_out, sizeof(float) * 4 * ncell, hipMemcpyDeviceToHost));
    }
  }
  if (speed_sum) *speed_sum = tot[0];
  if (mass) *mass = tot[1];
  return LBM_OK;
}

extern "C" int lbm_av_velocity(lbm_ctx* c, float* out) {
  if (!c || !out) return fail(LBM_EINVAL, "NULL argument");
  double sp = 0.0;
  int rc = derive_all(c, nullptr, &sp, nullptr);
  if (rc) return rc;
  *out = (float)(sp / (double)c->tot_fluid);  // d2q9-bgk.c:2713
  return LBM_OK;
}

extern "C" int lbm_reynolds(lbm_ctx* c, float* out) {
  if (!c || !out) return fail(LBM_EINVAL, "NULL argument");
  float av = 0.f;
  int rc = lbm_av_velocity(c, &av);
  if (rc) return rc;
  const float viscosity = 1.f / 6.f * (2.f / c->p.omega - 1.f);  // d2q9-bgk.c:2895
  *out = av * c->p.reynolds_dim / viscosity;
  return LBM_OK;
}

extern "C" int lbm_total_density(lbm_ctx* c, double* out) {
  if (!c || !out) return fail(LBM_EINVAL, "NULL argument");
  return derive_all(c, nullptr, nullptr, out);
}

extern "C" int lbm_final_state(lbm_ctx* c, float* out) {
  if (!c || !out) return fail(LBM_EINVAL, "NULL argument");
  return derive_all(c, out, nullptr, nullptr);
}

extern "C" int lbm_destroy(lbm_ctx* c) {
  if (!c) return LBM_OK;
  resident_free(c);
  regtile_slabs_free(c);
  for (auto& s : c->slabs) slab_free(s);
  delete c;
  return LBM_OK;
}

extern "C" int lbm_timestep(const lbm_param* params, float* cells, float* tmp_cells,
                            const int* obstacles, float* av_vel) {
  if (!cells || !tmp_cells) return fail(LBM_EINVAL, "NULL lattice");
  lbm_ctx* c = nullptr;
  int rc = lbm_create(params, obstacles, cells, 1, nullptr, LBM_EXCHANGE_AUTO, &c);
  if (rc) return rc;
  float av = 0.f;
  rc = lbm_run(c, 1, &av);
  if (!rc) rc = lbm_read_state(c, tmp_cells);
  lbm_destroy(c);
  if (rc) return rc;
  // the reference mutates `cells` (accelerate, row ny-2, d2q9-bgk.c:230-260): do the same
  // host-side so that callers relying on that side effect see it
  {
    const float a1 = params->density * params->accel / 9.f, a2 = params->density * params->accel / 36.f;
    const int jj = params->ny - 2;
    for (int ii = 0; ii < params->nx; ++ii) {
      float* s = cells + 9L * (ii + (long)jj * params->nx);
      if (!obstacles[ii + jj * params->nx] && (s[3] - a1) > 0.f && (s[6] - a2) > 0.f && (s[7] - a2) > 0.f) {
        s[1] += a1; s[5] += a2; s[8] += a2; s[3] -= a1; s[6] -= a2; s[7] -= a2;
      }
    }
  }
  if (av_vel) *av_vel = av;
  return LBM_OK;
}

extern "C" int lbm_set_option(lbm_ctx* c, const char* key, long value) {
  if (!c || !key) return fail(LBM_EINVAL, "NULL argument");
  if (!strcmp(key, "vector_width")) {
    if (!(value == 1 || (value == 2 && c->p.nx % 2 == 0 && c->p.nx >= 4) || (value == 4 && c->p.nx % 4 == 0 && c->p.nx >= 8)))
      return fail(LBM_EINVAL, "vector_width %ld not usable with nx = %d", value, c->p.nx);
    c->V = (int)value;
    c->engine = 1;   // (choosing among the streaming kernels selects the streaming engine)
    return LBM_OK;
  }
  if (!strcmp(key, "t2_threads")) {
    if (value != 256 && value != 512 && value != 1024) return fail(LBM_EINVAL, "t2_threads must be 256, 512 or 1024");
    c->t2_threads = (int)value;
    c->engine = 1;
    return LBM_OK;
  }
  if (!strcmp(key, "march_rows")) {
    if (value < 1 || value > c->p.ny) return fail(LBM_EINVAL, "march_rows must be in [1, ny]");
    { const int was = c->march_rows; c->march_rows = (int)value;
      if (check_march_partials(c, c->exchange != 0 && c->exchange != LBM_EXCHANGE_P2P && march_slabs_on(c))) { c->march_rows = was; return LBM_EINVAL; } }
    return LBM_OK;
  }
  if (!strcmp(key, "march_kernel")) {
    if (value < -1 || value > 1) return fail(LBM_EINVAL, "march_kernel must be 0 (lbm_march), 1 (lbm_wave) or -1 (automatic)");
    c->march_kernel = (int)value;
    return LBM_OK;
  }
  if (!strcmp(key, "wave_rows")) {
    if (value < 1 || value > c->p.ny) return fail(LBM_EINVAL, "wave_rows must be in [1, ny]");
    { const int was = c->wave_rows; c->wave_rows = (int)value;
      if (check_march_partials(c, c->exchange != 0 && (p2p_march_on(c) || (c->exchange != LBM_EXCHANGE_P2P && march_slabs_on(c))))) { c->wave_rows = was; return LBM_EINVAL; } }
    return LBM_OK;
  }
  if (!strcmp(key, "wave_cols")) {
    if (value != 1 && value != 2) return fail(LBM_EINVAL, "wave_cols must be 1 or 2 (columns per lane of lbm_wave)");
    if (value != c->wave_cols) { c->wave_rows = 0; c->wave_capacity = 0; }
    c->wave_cols = (int)value;
    return LBM_OK;
  }
  if (!strcmp(key, "time_block")) {
    if (value != 1 && value != 2 && value != 4 && value != 6 && value != 8) return fail(LBM_EINVAL, "time_block must be 1, 2, 4, 6 or 8");
    if (value != c->time_block) { c->wave_rows = 0; c->wave_capacity = 0; if (c->march_slabs == 0) c->march_slabs = -1; }   // (what the slabs can march depends on K)
    c->time_block = (int)value;
    c->engine = 1;
    return LBM_OK;
  }
  if (!strcmp(key, "engine")) {
    if (value != 0 && value != 1 && value != 3)
      return fail(LBM_EINVAL, "engine must be 0 (auto), 1 (streaming kernels) or 3 (resident in registers); 2, the LDS-resident engine, was removed");
    if (value == 3 && c->exchange != 0 && c->splan.ty == 0)
      return fail(LBM_EINVAL, "register tiles across slabs need equal slabs that tile onto the CUs and neighbours that can store into each other's memory");
    if (value == 3 && c->exchange == 0 && (c->slabs.size() != 1 || c->tplan.ty == 0))
      return fail(LBM_EINVAL, "the resident kernel needs a lattice alone on its GPU that tiles onto the CUs");
    c->engine = (int)value;
    if (value == 3) {
      c->resident_broken = false; c->resident_why[0] = 0;
      if (c->tplan.bpc < 0) c->tplan.bpc = 0;
      if (c->splan.bpc < 0) c->splan.bpc = 0;
    }
    return LBM_OK;
  }
  if (!strcmp(key, "regtile_async")) {
    if (value != 0 && value != 1) return fail(LBM_EINVAL, "regtile_async must be 0 or 1");
    c->regtile_async = (int)value;
    c->tplan.bpc = c->tplan.bpc < 0 ? c->tplan.bpc : 0;     // (another instantiation: ask about its residency again)
    c->splan.bpc = c->splan.bpc < 0 ? c->splan.bpc : 0;
    return LBM_OK;
  }
  if (!strcmp(key, "regtile")) {   // rows per tile * 10 + rows per wave
    const int ty = (int)(value / 10), r = (int)(value % 10);
    if (c->exchange != 0 || c->slabs.size() != 1 || !regtile_ok(c, ty, r))
      return fail(LBM_EINVAL, "register tile of %d rows, %d per wave, does not fit this lattice / device", ty, r);
    if (c->tmail) { (void)hipFree(c->tmail); c->tmail = nullptr; }
    regtile_set(c, ty, r);
    return LBM_OK;
  }
  if (!strcmp(key, "kernel_variant")) {
    if (value < 0 || value > 15) return fail(LBM_EINVAL, "kernel_variant must be in [0, 15]");
    c->variant = value;      // (bit 3: the one-step kernel with the reference's speed sum; see t2_eligible / march_eligible / lbm_run)
    return LBM_OK;
  }
  return fail(LBM_EINVAL, "unknown option %s", key);
}

extern "C" int lbm_get_info(const lbm_ctx* c, const char* key, double* value) {
  if (!c || !key || !value) return fail(LBM_EINVAL, "NULL argument");
  if (!strcmp(key, "vector_width")) { *value = c->V; return LBM_OK; }
  if (!strcmp(key, "kernel_variant")) { *value = (double)c->variant; return LBM_OK; }
  if (!strcmp(key, "time_block")) { *value = c->time_block; return LBM_OK; }
  if (!strcmp(key, "t2_threads")) { *value = c->t2_threads; return LBM_OK; }
  if (!strcmp(key, "time_block_active")) {
    *value = (march_eligible(c) || p2p_march_on(c) || march_bands_on(c) || (c->exchange != 0 && c->exchange != LBM_EXCHANGE_P2P && march_slabs_on(const_cast<lbm_ctx*>(c))))
                 ? c->time_block : t2_eligible(c) ? 2 : 1;
    return LBM_OK;
  }
  if (!strcmp(key, "march_kernel")) { *value = ((march_eligible(c) && use_wave_kernel(c)) || (c->exchange != 0 && slab_is_wave(slab_K(c)))) ? 1 : 0; return LBM_OK; }
  if (!strcmp(key, "wave_rows")) { *value = c->wave_rows; return LBM_OK; }
  if (!strcmp(key, "wave_cols")) { *value = c->wave_cols; return LBM_OK; }
  if (!strcmp(key, "wave_cols_active")) { *value = wave_C(c, c->time_block); return LBM_OK; }   // what lbm_wave<time_block> would run with
  if (!strcmp(key, "wave_out_cols")) { *value = wave_out_cols(c, c->time_block); return LBM_OK; }
  if (!strcmp(key, "wave_capacity")) { *value = c->wave_capacity; return LBM_OK; }
  if (!strcmp(key, "march_rows")) { *value = c->march_rows > 0 ? c->march_rows : march_pick_rows(c); return LBM_OK; }
  if (!strcmp(key, "fluid_cells")) { *value = (double)c->tot_fluid; return LBM_OK; }
  if (!strcmp(key, "engine")) { *value = c->engine; return LBM_OK; }
  if (!strcmp(key, "engine_last")) { *value = c->engine_last; return LBM_OK; }
  if (!strcmp(key, "engine_next")) {   // what the next lbm_run will try first
    if (c->exchange != 0) *value = regtile_slabs_usable(c) ? 3 : 1;
    else *value = (c->slabs.size() == 1 && !c->resident_broken)
                 ? (((c->engine == 3 || c->engine == 0) && c->tplan.ty > 0 && (c->variant & 8) == 0) ? 3 : 1) : 1;
    return LBM_OK;
  }
  if (!strcmp(key, "resident_fallback")) { *value = c->resident_broken ? 1 : 0; return LBM_OK; }   // 1: the resident kernel could not run here
  if (!strcmp(key, "regtile_blocks_per_cu")) { *value = c->exchange != 0 ? c->splan.bpc : c->tplan.bpc; return LBM_OK; }   // occupancy answer (0: not asked yet)
  if (!strcmp(key, "compute_units")) { *value = c->ncu; return LBM_OK; }
  if (!strcmp(key, "regtile")) { *value = c->exchange != 0 ? c->splan.ty * 10.0 + c->splan.r : c->tplan.ty * 10.0 + c->tplan.r; return LBM_OK; }
  if (!strcmp(key, "regtile_async")) { *value = c->regtile_async; return LBM_OK; }
  if (!strcmp(key, "exchange")) { *value = c->exchange; return LBM_OK; }
  if (!strcmp(key, "pitch")) { *value = c->slabs[0].pitch; return LBM_OK; }
  if (!strcmp(key, "hbm_bytes")) {
    double b = 0;
    for (auto& s : c->slabs) b += 2.0 * 9 * 4 * (double)s.plane + (double)s.plane;
    *value = b; return LBM_OK;
  }
  return fail(LBM_EINVAL, "unknown info key %s", key);
}
