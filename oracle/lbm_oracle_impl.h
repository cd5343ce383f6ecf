/*
 * oracle/lbm_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (the CPU oracle).
 *
 * Serial CPU restatement of the reference's D2Q9-BGK time step, written from
 * the language-neutral spec, one generic cell body with
 * modulo-style periodic wrap instead of the reference's nine peeled copies.
 * Included twice by lbm_oracle.c: once with ORC_REAL=float, once =double.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/.  The product path (advanced-hpc-lbm_amd/) never
 * links, loads or calls this.
 *
 * Every function cites the reference lines (/root/reference/d2q9-bgk.c) whose
 * arithmetic and statement order it follows.  Statement order matters: the
 * double flavour, built without FP contraction or reassociation, reproduces
 * the reference's shipped golden files digit for digit, and the float flavour
 * matches a strict-IEEE build of the reference bit for bit.
 */

#ifndef ORC_REAL
#error "include from lbm_oracle.c with ORC_REAL and ORC_SUFFIX defined"
#endif

#define ORC_CAT2(a, b) a##b
#define ORC_CAT(a, b) ORC_CAT2(a, b)
#define ORC_FN(name) ORC_CAT(name, ORC_SUFFIX)

/* Rest-equilibrium initial lattice: every cell, blocked or not.
 * Reference: initialise(), d2q9-bgk.c:2802-2823. */
void ORC_FN(orc_init_cells_)(const orc_param* p, ORC_REAL* cells)
{
  const ORC_REAL w0 = (ORC_REAL)p->density * (ORC_REAL)4 / (ORC_REAL)9;
  const ORC_REAL w1 = (ORC_REAL)p->density / (ORC_REAL)9;
  const ORC_REAL w2 = (ORC_REAL)p->density / (ORC_REAL)36;
  const long n = (long)p->nx * p->ny;
  for (long c = 0; c < n; c++) {
    ORC_REAL* s = cells + 9 * c;
    s[0] = w0;
    s[1] = s[2] = s[3] = s[4] = w1;
    s[5] = s[6] = s[7] = s[8] = w2;
  }
}

/* Accelerate phase, in place on the source lattice, row ny-2 only.
 * Reference: timestep_new2 d2q9-bgk.c:230-260 (same text as accelerate_flow
 * 1888-1918). */
void ORC_FN(orc_accelerate_)(const orc_param* p, ORC_REAL* cells, const int* obstacles)
{
  const ORC_REAL a1 = (ORC_REAL)p->density * (ORC_REAL)p->accel / (ORC_REAL)9;
  const ORC_REAL a2 = (ORC_REAL)p->density * (ORC_REAL)p->accel / (ORC_REAL)36;
  const int jj = p->ny - 2;
  for (int ii = 0; ii < p->nx; ii++) {
    ORC_REAL* s = cells + 9 * ((long)ii + (long)jj * p->nx);
    if (!obstacles[ii + jj * p->nx]
        && (s[3] - a1) > (ORC_REAL)0
        && (s[6] - a2) > (ORC_REAL)0
        && (s[7] - a2) > (ORC_REAL)0) {
      s[1] += a1; s[5] += a2; s[8] += a2;
      s[3] -= a1; s[6] -= a2; s[7] -= a2;
    }
  }
}

/* One cell of the fused sweep: pull-stream, then bounce-back or BGK collide,
 * then the post-collision speed |u'| for the average.  Returns 1 and sets *speed for a
 * fluid cell, 0 for a blocked one.
 * Reference: representative copy d2q9-bgk.c:971-1131 (the other eight peeled
 * copies are textually identical apart from neighbour index formation);
 * gather map = propagate() 2139-2147. */
static int ORC_FN(orc_cell_)(const orc_param* p, const ORC_REAL* cells, ORC_REAL* out,
                             int blocked, long c0, long cE, long cN, long cW, long cS,
                             long cNE, long cNW, long cSW, long cSE, ORC_REAL* speed)
{
  /* pulled values: direction k arrives from the neighbour opposite to k */
  const ORC_REAL p0 = cells[9 * c0 + 0];
  const ORC_REAL p1 = cells[9 * cW + 1];
  const ORC_REAL p2 = cells[9 * cS + 2];
  const ORC_REAL p3 = cells[9 * cE + 3];
  const ORC_REAL p4 = cells[9 * cN + 4];
  const ORC_REAL p5 = cells[9 * cSW + 5];
  const ORC_REAL p6 = cells[9 * cSE + 6];
  const ORC_REAL p7 = cells[9 * cNE + 7];
  const ORC_REAL p8 = cells[9 * cNW + 8];

  if (blocked) {
    /* d2q9-bgk.c:971-981: mirrored store of the pulled values */
    out[0] = p0; out[1] = p3; out[2] = p4; out[3] = p1; out[4] = p2;
    out[5] = p7; out[6] = p8; out[7] = p5; out[8] = p6;
    return 0;
  }

  const ORC_REAL one = (ORC_REAL)1, two = (ORC_REAL)2;
  const ORC_REAL c_sq = one / (ORC_REAL)3;   /* d2q9-bgk.c:983-986 */
  const ORC_REAL w0 = (ORC_REAL)4 / (ORC_REAL)9;
  const ORC_REAL w1 = one / (ORC_REAL)9;
  const ORC_REAL w2 = one / (ORC_REAL)36;
  const ORC_REAL omega = (ORC_REAL)p->omega;

  /* d2q9-bgk.c:988-998: density summed in direction order 0..8 */
  ORC_REAL rho = (ORC_REAL)0;
  rho += p0; rho += p1; rho += p2; rho += p3; rho += p4;
  rho += p5; rho += p6; rho += p7; rho += p8;

  /* d2q9-bgk.c:1002-1016 */
  const ORC_REAL u_x = (p1 + p5 + p8 - (p3 + p6 + p7)) / rho;
  const ORC_REAL u_y = (p2 + p5 + p6 - (p4 + p7 + p8)) / rho;
  const ORC_REAL u_sq = u_x * u_x + u_y * u_y;      /* :1019 */

  ORC_REAL u[9];                                     /* :1023-1030 */
  u[1] = u_x;         u[2] = u_y;
  u[3] = -u_x;        u[4] = -u_y;
  u[5] = u_x + u_y;   u[6] = -u_x + u_y;
  u[7] = -u_x - u_y;  u[8] = u_x - u_y;

  ORC_REAL d[9];                                     /* :1035-1062 */
  d[0] = w0 * rho * (one - u_sq / (two * c_sq));
  for (int k = 1; k < 9; k++) {
    const ORC_REAL w = (k < 5) ? w1 : w2;
    d[k] = w * rho * (one + u[k] / c_sq
                      + (u[k] * u[k]) / (two * c_sq * c_sq)
                      - u_sq / (two * c_sq));
  }

  /* relaxation, d2q9-bgk.c:1066-1100 */
  out[0] = p0 + omega * (d[0] - p0);
  out[1] = p1 + omega * (d[1] - p1);
  out[2] = p2 + omega * (d[2] - p2);
  out[3] = p3 + omega * (d[3] - p3);
  out[4] = p4 + omega * (d[4] - p4);
  out[5] = p5 + omega * (d[5] - p5);
  out[6] = p6 + omega * (d[6] - p6);
  out[7] = p7 + omega * (d[7] - p7);
  out[8] = p8 + omega * (d[8] - p8);

  /* av-velocity contribution from the STORED values, d2q9-bgk.c:1103-1130 */
  ORC_REAL rho2 = (ORC_REAL)0;
  for (int k = 0; k < 9; k++) rho2 += out[k];
  const ORC_REAL vx = (out[1] + out[5] + out[8] - (out[3] + out[6] + out[7])) / rho2;
  const ORC_REAL vy = (out[2] + out[5] + out[6] - (out[4] + out[7] + out[8])) / rho2;
  *speed = ORC_FN(orc_sqrt_)((vx * vx) + (vy * vy));
  return 1;
}

/* The fused sweep WITHOUT the accelerate phase: reads `cells`, fully
 * overwrites `tmp_cells`; returns the per-step average speed.
 * Reference: timestep_new2 d2q9-bgk.c:262-1811 (row-major jj outer / ii inner,
 * serial accumulation of tot_u in that order, return tot_u/(real)tot_cells). */
ORC_REAL ORC_FN(orc_sweep_)(const orc_param* p, const ORC_REAL* cells, ORC_REAL* tmp_cells,
                            const int* obstacles)
{
  const int nx = p->nx, ny = p->ny;
  int tot_cells = 0;
  ORC_REAL tot_u = (ORC_REAL)0;
  for (int jj = 0; jj < ny; jj++) {
    const int y_n = (jj + 1) % ny;                 /* propagate() :2132-2135 */
    const int y_s = (jj == 0) ? (ny - 1) : (jj - 1);
    for (int ii = 0; ii < nx; ii++) {
      const int x_e = (ii + 1) % nx;
      const int x_w = (ii == 0) ? (nx - 1) : (ii - 1);
      ORC_REAL speed;
      const long c0 = (long)ii + (long)jj * nx;
      if (ORC_FN(orc_cell_)(p, cells, tmp_cells + 9 * c0, obstacles[c0], c0,
                            (long)x_e + (long)jj * nx, (long)ii + (long)y_n * nx,
                            (long)x_w + (long)jj * nx, (long)ii + (long)y_s * nx,
                            (long)x_e + (long)y_n * nx, (long)x_w + (long)y_n * nx,
                            (long)x_w + (long)y_s * nx, (long)x_e + (long)y_s * nx,
                            &speed)) {
        tot_u += speed;
        ++tot_cells;
      }
    }
  }
  return tot_u / (ORC_REAL)tot_cells;
}

/* Row-range form of the sweep, for slab-decomposition tests: processes rows
 * [row_begin, row_end) of a lattice of p->ny rows (periodic indexing over those
 * p->ny rows, exactly as above), leaves other rows of tmp_cells untouched and
 * returns the UN-normalised speed sum and the fluid-cell count of the range.
 * A slab with one halo row on each side is a lattice of nyl+2 rows swept over
 * [1, nyl+1).  Same cell arithmetic as orc_sweep_ (d2q9-bgk.c:971-1131). */
void ORC_FN(orc_sweep_rows_)(const orc_param* p, const ORC_REAL* cells, ORC_REAL* tmp_cells,
                             const int* obstacles, int row_begin, int row_end,
                             ORC_REAL* tot_u_out, int* tot_cells_out)
{
  const int nx = p->nx, ny = p->ny;
  int tot_cells = 0;
  ORC_REAL tot_u = (ORC_REAL)0;
  for (int jj = row_begin; jj < row_end; jj++) {
    const int y_n = (jj + 1) % ny;
    const int y_s = (jj == 0) ? (ny - 1) : (jj - 1);
    for (int ii = 0; ii < nx; ii++) {
      const int x_e = (ii + 1) % nx;
      const int x_w = (ii == 0) ? (nx - 1) : (ii - 1);
      ORC_REAL speed;
      const long c0 = (long)ii + (long)jj * nx;
      if (ORC_FN(orc_cell_)(p, cells, tmp_cells + 9 * c0, obstacles[c0], c0,
                            (long)x_e + (long)jj * nx, (long)ii + (long)y_n * nx,
                            (long)x_w + (long)jj * nx, (long)ii + (long)y_s * nx,
                            (long)x_e + (long)y_n * nx, (long)x_w + (long)y_n * nx,
                            (long)x_w + (long)y_s * nx, (long)x_e + (long)y_s * nx,
                            &speed)) {
        tot_u += speed;
        ++tot_cells;
      }
    }
  }
  *tot_u_out = tot_u;
  *tot_cells_out = tot_cells;
}

/* Accelerate phase on an arbitrary row (slab tests: the global row ny-2 has a
 * different local index).  Same arithmetic as orc_accelerate_. */
void ORC_FN(orc_accelerate_row_)(const orc_param* p, ORC_REAL* cells, const int* obstacles, int jj)
{
  const ORC_REAL a1 = (ORC_REAL)p->density * (ORC_REAL)p->accel / (ORC_REAL)9;
  const ORC_REAL a2 = (ORC_REAL)p->density * (ORC_REAL)p->accel / (ORC_REAL)36;
  for (int ii = 0; ii < p->nx; ii++) {
    ORC_REAL* s = cells + 9 * ((long)ii + (long)jj * p->nx);
    if (!obstacles[ii + jj * p->nx]
        && (s[3] - a1) > (ORC_REAL)0
        && (s[6] - a2) > (ORC_REAL)0
        && (s[7] - a2) > (ORC_REAL)0) {
      s[1] += a1; s[5] += a2; s[8] += a2;
      s[3] -= a1; s[6] -= a2; s[7] -= a2;
    }
  }
}

/* One full reference time step (accelerate + sweep).  The caller swaps the
 * lattices afterwards, as main does at d2q9-bgk.c:182,190.
 * Reference signature: timestep_new2, d2q9-bgk.c:98,228. */
ORC_REAL ORC_FN(orc_timestep_)(const orc_param* p, ORC_REAL* cells, ORC_REAL* tmp_cells,
                               const int* obstacles)
{
  ORC_FN(orc_accelerate_)(p, cells, obstacles);
  return ORC_FN(orc_sweep_)(p, cells, tmp_cells, obstacles);
}

/* nsteps time steps with ping-pong; on return `cells` holds the final lattice
 * (an odd nsteps is handled by a copy so the caller never has to track the
 * swap).  Reference: main loop d2q9-bgk.c:180-201. */
void ORC_FN(orc_run_)(const orc_param* p, ORC_REAL* cells, ORC_REAL* tmp_cells,
                      const int* obstacles, int nsteps, ORC_REAL* av_vels)
{
  ORC_REAL* a = cells;
  ORC_REAL* b = tmp_cells;
  for (int tt = 0; tt < nsteps; tt++) {
    av_vels[tt] = ORC_FN(orc_timestep_)(p, a, b, obstacles);
    ORC_REAL* t = a; a = b; b = t;
  }
  if (a != cells) memcpy(cells, a, sizeof(ORC_REAL) * 9 * (size_t)p->nx * p->ny);
}

/* Average speed over fluid cells of a lattice.
 * Reference: av_velocity(), d2q9-bgk.c:2665-2714. */
ORC_REAL ORC_FN(orc_av_velocity_)(const orc_param* p, const ORC_REAL* cells, const int* obstacles)
{
  int tot_cells = 0;
  ORC_REAL tot_u = (ORC_REAL)0;
  const long n = (long)p->nx * p->ny;
  for (long c = 0; c < n; c++) {
    if (obstacles[c]) continue;
    const ORC_REAL* s = cells + 9 * c;
    ORC_REAL rho = (ORC_REAL)0;
    for (int k = 0; k < 9; k++) rho += s[k];
    const ORC_REAL u_x = (s[1] + s[5] + s[8] - (s[3] + s[6] + s[7])) / rho;
    const ORC_REAL u_y = (s[2] + s[5] + s[6] - (s[4] + s[7] + s[8])) / rho;
    tot_u += ORC_FN(orc_sqrt_)((u_x * u_x) + (u_y * u_y));
    ++tot_cells;
  }
  return tot_u / (ORC_REAL)tot_cells;
}

/* Reference: calc_reynolds(), d2q9-bgk.c:2893-2898. */
ORC_REAL ORC_FN(orc_reynolds_)(const orc_param* p, const ORC_REAL* cells, const int* obstacles)
{
  const ORC_REAL viscosity = (ORC_REAL)1 / (ORC_REAL)6 * ((ORC_REAL)2 / (ORC_REAL)p->omega - (ORC_REAL)1);
  return ORC_FN(orc_av_velocity_)(p, cells, obstacles) * p->reynolds_dim / viscosity;
}

/* Total mass; constant from step to step (debug invariant of the reference).
 * Reference: total_density(), d2q9-bgk.c:2900-2916. */
ORC_REAL ORC_FN(orc_total_density_)(const orc_param* p, const ORC_REAL* cells)
{
  ORC_REAL total = (ORC_REAL)0;
  const long n = 9L * p->nx * p->ny;
  for (long i = 0; i < n; i++) total += cells[i];
  return total;
}

/* Derived fields written to final_state.dat: out[4*c + {0,1,2,3}] =
 * u_x, u_y, |u|, pressure for cell c (row-major).
 * Reference: write_values(), d2q9-bgk.c:2935-2976. */
void ORC_FN(orc_final_state_)(const orc_param* p, const ORC_REAL* cells, const int* obstacles,
                              ORC_REAL* out)
{
  const ORC_REAL c_sq = (ORC_REAL)1 / (ORC_REAL)3;
  const long n = (long)p->nx * p->ny;
  for (long c = 0; c < n; c++) {
    ORC_REAL u_x, u_y, u, pressure;
    if (obstacles[c]) {
      u_x = u_y = u = (ORC_REAL)0;
      pressure = (ORC_REAL)p->density * c_sq;
    } else {
      const ORC_REAL* s = cells + 9 * c;
      ORC_REAL rho = (ORC_REAL)0;
      for (int k = 0; k < 9; k++) rho += s[k];
      u_x = (s[1] + s[5] + s[8] - (s[3] + s[6] + s[7])) / rho;
      u_y = (s[2] + s[5] + s[6] - (s[4] + s[7] + s[8])) / rho;
      u = ORC_FN(orc_sqrt_)((u_x * u_x) + (u_y * u_y));
      pressure = rho * c_sq;
    }
    out[4 * c + 0] = u_x; out[4 * c + 1] = u_y; out[4 * c + 2] = u; out[4 * c + 3] = pressure;
  }
}

#undef ORC_FN
#undef ORC_CAT
#undef ORC_CAT2
