/*
 * lbm_mi355x.h -- C ABI of the MI355X-native D2Q9-BGK time-step path.
 *
 * This is the drop-in boundary for the hot path of ChuyueL/advanced-hpc-lbm:
 * the per-time-step sweep `timestep_new2` and the loop that drives it
 * (reference d2q9-bgk.c:180-201, 228-1813).  The reference has no FFI of its
 * own; its boundary is one in-process call site (d2q9-bgk.c:182,190)
 *
 *     av_vels[tt] = timestep_new2(params, cells, tmp_cells, obstacles);
 *     swap(&cells, &tmp_cells);
 *
 * plus the functions main calls once around it.  A per-step host-returning
 * call would force a device sync every step, so the unit here is "run N
 * steps": the library keeps both lattices resident in HBM (SoA planes), and
 * the host hands over / takes back the reference's own host layouts
 * (t_speed AoS cells, int obstacles, t_param).
 *
 * Conventions
 *   - Plain C types only.  Every function returns 0 on success and a non-zero
 *     LBM_E* code on failure; lbm_last_error() returns the message for the
 *     calling thread (the CLI host prints it through the reference's die()
 *     convention, d2q9-bgk.c:3001-3007).
 *   - Host arrays use the reference's layouts exactly:
 *       cells      float[ny*nx*9]  = t_speed[ny*nx]   (d2q9-bgk.c:76-79), index ii + jj*nx
 *       obstacles  int[ny*nx], 0 = fluid, non-zero = blocked (d2q9-bgk.c:2797,2856)
 *   - There is no CPU fallback: without a usable HIP device every compute
 *     entry point fails with LBM_ENODEV.
 */
#ifndef LBM_MI355X_H
#define LBM_MI355X_H

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_NSPEEDS 9

/* error codes */
#define LBM_OK        0
#define LBM_EINVAL    1  /* bad argument / unsupported size */
#define LBM_ENODEV    2  /* no HIP device, or fewer than requested */
#define LBM_EHIP      3  /* a HIP runtime call failed */
#define LBM_ERCCL     4  /* RCCL missing or a RCCL call failed */
#define LBM_ENOMEM    5

/* Same fields, order and types as the reference's t_param (d2q9-bgk.c:64-73). */
typedef struct {
  int   nx;            /* no. of cells in x-direction */
  int   ny;            /* no. of cells in y-direction */
  int   maxIters;      /* no. of iterations */
  int   reynolds_dim;  /* dimension for Reynolds number */
  float density;       /* density per link */
  float accel;         /* density redistribution */
  float omega;         /* relaxation parameter */
} lbm_param;

typedef struct lbm_ctx lbm_ctx;

/* How neighbouring row slabs trade their one-row halos each step. */
#define LBM_EXCHANGE_AUTO   0  /* 1 slab: none (periodic self-wrap); >1 slabs: RCCL, or peer
                                  copies when the device list repeats a device (RCCL wants
                                  one rank per GPU) */
#define LBM_EXCHANGE_COPY   1  /* hipMemcpyAsync between slabs of ONE process (peer copies) */
#define LBM_EXCHANGE_RCCL   2  /* ncclSend/ncclRecv pairs over xGMI, own stream, overlapped */
#define LBM_EXCHANGE_P2P    3  /* kernels store their halo rows straight into the neighbour's
                                  buffers over xGMI and hand off through flags (no host, no
                                  collective call in the step loop); needs peer access (one
                                  process) or hipIpc (one process per GPU) and uncached device
                                  memory.  A halo wait gives up after 4 s: ranks must enter
                                  lbm_run within 4 s of each other, and when a neighbour stops,
                                  lbm_run returns LBM_EHIP within seconds (every queued launch
                                  sees the sticky error word and drains); the lattice is then
                                  undefined and later lbm_run calls on the context fail */

/* Message of the last failure on this thread ("" if none). */
const char* lbm_last_error(void);

/* Number of visible HIP devices (0 is a valid answer). */
int lbm_device_count(int* count);

/*
 * Replaces: initialise()'s allocation + the first-touch of cells/tmp_cells/
 * obstacles (d2q9-bgk.c:2787-2857), for a whole lattice owned by ONE process.
 *   params     global lattice parameters
 *   obstacles  int[ny*nx] blocked map (copied; caller keeps ownership)
 *   cells      float[ny*nx*9] initial lattice, or NULL for the reference's
 *              rest-equilibrium start (d2q9-bgk.c:2802-2823)
 *   nslabs     number of row slabs (>=1): slab r owns rows [r*ny/nslabs, (r+1)*ny/nslabs)
 *   devices    int[nslabs] HIP device per slab, or NULL for 0..nslabs-1
 *              (repeating a device is allowed: several slabs on one GPU)
 *   exchange   LBM_EXCHANGE_*
 */
int lbm_create(const lbm_param* params, const int* obstacles, const float* cells,
               int nslabs, const int* devices, int exchange, lbm_ctx** out);

/*
 * One-process-per-GPU form (bench.py under torch.distributed.run): this
 * process owns slab `rank` of `nranks` on HIP device `device`; halos travel
 * by RCCL.  `unique_id` is the 128-byte ncclUniqueId made by
 * lbm_rccl_unique_id() on rank 0 and broadcast by the caller.
 * `obstacles`/`cells` are the GLOBAL arrays (each rank keeps its rows only);
 * cells may be NULL as above.
 */
int lbm_rccl_unique_id(void* id128);
int lbm_create_rank(const lbm_param* params, const int* obstacles, const float* cells,
                    int rank, int nranks, int device, const void* unique_id, lbm_ctx** out);

/*
 * Same, with the halo transport chosen by the caller (LBM_EXCHANGE_RCCL or LBM_EXCHANGE_P2P;
 * lbm_create_rank = RCCL unless the environment says LBM_RANK_EXCHANGE=p2p).
 *   unique_id != NULL: the library forms a RCCL communicator (used for the end-of-run
 *     reductions, for RCCL halos, and to trade the hipIpc handles of peer-to-peer halos);
 *     if peer-to-peer set-up fails on ANY rank, all ranks fall back to RCCL halos together.
 *   unique_id == NULL (peer-to-peer only): no RCCL at all.  The caller trades the 64-byte
 *     handles itself -- lbm_p2p_handle() on every rank, all-gather by any means,
 *     lbm_p2p_connect() with all nranks handles in rank order -- before the first lbm_run;
 *     lbm_run / lbm_av_velocity / lbm_total_density then return this rank's CONTRIBUTION
 *     (slab sum over the global fluid-cell count), to be added across ranks by the caller.
 */
int lbm_create_rank_ex(const lbm_param* params, const int* obstacles, const float* cells,
                       int rank, int nranks, int device, const void* unique_id, int exchange,
                       lbm_ctx** out);
#define LBM_P2P_HANDLE_BYTES 64
int lbm_p2p_handle(lbm_ctx* ctx, void* handle64);
int lbm_p2p_connect(lbm_ctx* ctx, const void* handles, int nranks);

/* Rows [row_begin, row_end) of the global lattice held by slab `slab` of this context. */
int lbm_slab_rows(const lbm_ctx* ctx, int slab, int* row_begin, int* row_end);
int lbm_num_slabs(const lbm_ctx* ctx);

/*
 * Replaces: the time-step loop, d2q9-bgk.c:180-201 --
 *     for (tt...) { av_vels[tt] = timestep_new2(params, cells, tmp_cells, obstacles); swap(); }
 * Advances the resident lattice by `nsteps` steps and writes the per-step
 * average fluid speed (timestep_new2's return value, d2q9-bgk.c:1811) to
 * av_vels[0..nsteps) (may be NULL).  Synchronous: returns when the GPU work
 * is complete.  In rank mode every rank receives the global av_vels.
 */
int lbm_run(lbm_ctx* ctx, int nsteps, float* av_vels);

/* GPU time of the step loop of the last lbm_run, from HIP events on the
 * compute stream of slab 0 (ms), and host wall time of the same region. */
int lbm_last_run_ms(const lbm_ctx* ctx, double* gpu_ms, double* wall_ms);

/* Copies the current lattice back in the reference's AoS layout
 * (what `cells` holds after the swap at d2q9-bgk.c:190).
 * Single-process contexts: the whole lattice, float[ny*nx*9].
 * Rank contexts: this rank's rows only, float[(row_end-row_begin)*nx*9]. */
int lbm_read_state(lbm_ctx* ctx, float* cells_out);

/* Replaces: av_velocity() and calc_reynolds(), d2q9-bgk.c:2665-2714, 2893-2898,
 * evaluated on the resident lattice (global value in every mode). */
int lbm_av_velocity(lbm_ctx* ctx, float* out);
int lbm_reynolds(lbm_ctx* ctx, float* out);

/* Sum of all 9*nx*ny distribution values (total_density(), d2q9-bgk.c:2900-2916);
 * constant from step to step.  Accumulated in double. */
int lbm_total_density(lbm_ctx* ctx, double* out);

/* Replaces: the per-cell arithmetic of write_values(), d2q9-bgk.c:2935-2976.
 * out[4*(ii + jj*nx) + {0,1,2,3}] = u_x, u_y, |u|, pressure, computed on the GPU
 * (blocked cells: 0, 0, 0, density/3).  Same slab-local convention as
 * lbm_read_state in rank mode. */
int lbm_final_state(lbm_ctx* ctx, float* out);

/* Replaces: finalise(), d2q9-bgk.c:2871-2890.
 * Rank contexts with peer-to-peer halos: the neighbours' last launches still store into this
 * context's halo block, so destroy it only after EVERY rank's last lbm_run has returned
 * (a barrier of the caller's choice); the RCCL and single-process forms need no such care. */
int lbm_destroy(lbm_ctx* ctx);

/*
 * Parity shim with the reference's own call shape (d2q9-bgk.c:98,228):
 * one step on host arrays.  Like the reference it applies the accelerate
 * phase to `cells` IN PLACE (row ny-2), fully overwrites `tmp_cells`, reads
 * `obstacles`, and returns the average speed through *av_vel.  Uploads and
 * downloads every call: for tests, not for speed.
 */
int lbm_timestep(const lbm_param* params, float* cells, float* tmp_cells,
                 const int* obstacles, float* av_vel);

/*
 * The tiling the register-resident engine (lbm_regtile: the whole lbm_run in one launch, replacing the loop of
 * d2q9-bgk.c:180-201) would give a lattice -- or each of several equal slabs -- of nx columns x rows rows on a device
 * with `compute_units` CUs that holds `slabs_per_device` such slabs: 64-column tiles of *tile_rows rows, one per CU,
 * *rows_per_wave rows per wavefront.  Host arithmetic only (no device needed).  LBM_EINVAL: it does not tile.
 */
int lbm_plan_tiles(int nx, int rows, int slabs_per_device, int compute_units, int* tile_rows, int* rows_per_wave);

/* Tuning / introspection (never needed for correctness; the table of keys is INTEGRATION.md section 3).
 * Options: "engine" (0 auto, 1 streaming kernels, 3 register tiles -- alone or across slabs -- or fail), "time_block"
 * (1, 2, 4, 6, 8 steps per pass), "march_kernel", "march_rows", "wave_rows", "wave_cols" (1, 2), "regtile" (tiling),
 * "regtile_async" (0, 1), "regtile_tag" (test hook: the next mailbox tag), "kernel_variant" (bits: 1 fast rcp / sqrt, 2 / 4 nontemporal stores / loads, 8 the reference's
 * form of the speed sum, d2q9-bgk.c:1783-1811, 256 one-step kernel only), "vector_width", "t2_threads".
 * Info: "engine_last", "engine_next", "resident_fallback", "time_block_active", "march_kernel", "wave_rows",
 * "wave_cols_active", "wave_out_cols", "regtile", "regtile_blocks_per_cu", "exchange", "compute_units", "fluid_cells",
 * "pitch", "hbm_bytes". */
int lbm_set_option(lbm_ctx* ctx, const char* key, long value);  /* e.g. "kernel_variant" */
int lbm_get_info(const lbm_ctx* ctx, const char* key, double* value);

#ifdef __cplusplus
}
#endif
#endif /* LBM_MI355X_H */
