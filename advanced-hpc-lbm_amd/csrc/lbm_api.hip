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
  bool splan_peers = false;             // peer access between the neighbouring slabs' devices has been switched on
  lbm::RegTileArgs* rtable = nullptr;   // pinned, device-mapped: one entry per local slab, grouped by device
  lbm::RegTileArgs* rtable_dev = nullptr;
  int ncu = 0;                 // CUs of slab 0's device
  double gpu_ms = 0.0, wall_ms = 0.0;
};

// tile of the two-step kernel
constexpr int kT2X = 64, kT2Y = 16;
// rows of obstacle bytes kept either side of a slab for the ghost bands of the marching kernels (the largest K)
constexpr int kBandRows = 8;

#include "lbm_host_slabs.inc"

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

#include "lbm_host_march.inc"

#include "lbm_host_run.inc"

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
  if (!strcmp(key, "regtile_tag")) {      // test hook: the next mailbox tag (they only grow: tests reach the restart paths with it)
    if (value < (long)c->rtag || value >= 0x7fffff00L) return fail(LBM_EINVAL, "regtile_tag must not go back (now %u) and must stay below 2^31", c->rtag);
    c->rtag = (uint32_t)value;
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
  if (!strcmp(key, "regtile_tag")) { *value = c->rtag; return LBM_OK; }
  if (!strcmp(key, "exchange")) { *value = c->exchange; return LBM_OK; }
  if (!strcmp(key, "pitch")) { *value = c->slabs[0].pitch; return LBM_OK; }
  if (!strcmp(key, "hbm_bytes")) {
    double b = 0;
    for (auto& s : c->slabs) b += 2.0 * 9 * 4 * (double)s.plane + (double)s.plane;
    *value = b; return LBM_OK;
  }
  return fail(LBM_EINVAL, "unknown info key %s", key);
}
