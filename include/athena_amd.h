/* include/athena_amd.h -- C-ABI of the MI355X-native hot path of Atmospheric Athena.
 *
 * Plain C, plain pointers and sizes; no torch / HIP types in any signature (a HIP stream is
 * passed as `void*`).  One `aa_grid` = one Grid of the reference (one slab of the root
 * Domain, resident on one GPU).  Every entry point names the reference interface it
 * replaces (paths under /root/reference/src).  All functions return 0 on success and a
 * negative code on error; `aa_last_error()` returns the message the reference would have
 * passed to ath_error() (utils.c:118).  Nothing here falls back to a CPU path: if the GPU
 * or the library is missing, creation fails.
 *
 * Host-side state is exchanged in the reference's own layout: `GridS.U` is one contiguous
 * block of ConsS {d,M1,M2,M3,E[,s0]} indexed [k][j][i] over Nx+2*nghost zones per direction
 * (athena.h:81-100,:290; ath_array.c:78-120) and `GridS.EdgeFlux` is [Nx3+1][Nx2+1][Nx1+1]
 * (init_grid.c:242-250).  On the device the state is struct-of-arrays.
 */
#ifndef ATHENA_AMD_H
#define ATHENA_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

enum { AA_NGHOST = 4 };                     /* defs.h.in:129-140 */
enum { AA_BC_NONE = 0, AA_BC_REFLECT = 1, AA_BC_OUTFLOW = 2, AA_BC_PERIODIC = 4 }; /* bvals_mhd.c:560-586 */

typedef struct aa_params {
  int    Nx[3];            /* active zones of this Grid (all > 1: 3-D only)                     */
  int    rootNx[3];        /* active zones of the root Domain                                   */
  double xmin[3], xmax[3]; /* root Domain extent; dx = (xmax-xmin)/rootNx (init_mesh.c:225)     */
  double MinX[3];          /* lower edge of this Grid (init_grid.c:104-111)                     */
  int    bc[6];            /* ix1,ox1,ix2,ox2,ix3,ox3; AA_BC_NONE = neighbour Grid fills it     */
  int    nscal;            /* NSCALARS (0 or 1; 1 whenever ion != 0)                            */
  int    ion;              /* ION_RADIATION + ION_RADPLANE                                      */
  double gamma, cour_no, tlim;   /* <problem>gamma, <time>cour_no, <time>tlim (main.c:368-378)  */
  /* <ionradiation> block, ionrad_3d.c:742-757 */
  double sigma_ph, m_H, mu, e_gamma, alpha_C, k_B, time_unit;
  double max_de_iter, max_de_therm_iter, max_dx_iter;
  double max_de_step, max_de_therm_step, max_dx_step;
  double tfloor, tceil;
  int    maxiter;
  int    device;           /* HIP device ordinal                                                */
  int    integrator;       /* 0: CTU + H-correction (configure default + --enable-h-correction);
                              1: van Leer, no H-correction (--with-integrator=vl);
                              2: CTU without H-correction (configure default, NO_H_CORRECTION)  */
  int    level;            /* DomainS.Level (static mesh refinement): dx = root dx / 2^level
                              (init_mesh.c:245); 0 for a single-level run                      */
  int    order;            /* configure --with-order: 2 (or 0) piecewise linear, lr_states_plm.c;
                              3 piecewise parabolic, lr_states_ppm.c (CTU integrator only)    */
  int    ion_path;         /* radiation sub-cycle: 0 by size (one kernel from rays of 48 zones, or as
                              AA_ION_FUSED says), 1 the one-kernel form (aa_ion_pass ...), 2 the
                              two-kernel form (aa_ion_rates / aa_ion_update)                   */
  int    nslab;            /* > 1: this ONE Grid of the caller is cut into that many x3 slabs, one per
                              GPU (devices `device`, `device`+1, ... modulo the visible ones, or
                              AA_SLAB_DEVICES=0,1,..), behind this same interface: the library does
                              what init_mesh.c:583-620, bvals_mhd.c:423-493 and the MPI_Allreduce
                              calls of new_dt.c / ionrad_3d.c do for the reference's ranks.  0: one
                              GPU unless AA_NGPU is set in the environment; 1: one GPU          */
} aa_params;

typedef struct aa_grid aa_grid;

/* ---- lifecycle: init_grid.c (U, EdgeFlux), integrate_init_3d (integrate_3d_ctu.c:3374),
 *      ion_radtransfer_init_3d (ionrad_3d.c:739), ion_radtransfer_init (ionrad.c:64)      */
/* aa_params.nslab > 1 (or AA_NGPU=N in the environment): the handle stands for the caller's ONE Grid cut into N x3 slabs, one
 * per HIP device (AA_SLAB_DEVICES=0,1,.. or p->device, +1, .. modulo the visible ones; csrc/slabs.hip).  Root level only
 * (aa_mesh_create refuses it), >= 4 planes per slab, <= 64 slabs; aa_pack_x3 / aa_unpack_x3 / aa_set_stream return an error
 * on it and aa_halo_doubles 0 (the slabs exchange their halos themselves); the caller's current device is left unchanged. */
int         aa_create(const aa_params *p, aa_grid **out);
void        aa_destroy(aa_grid *g);   /* integrate_destruct_3d :3498 */
const char *aa_last_error(void);
int         aa_set_stream(aa_grid *g, void *hip_stream);   /* run on a caller-owned stream */
int         aa_sync(aa_grid *g);
long long   aa_device_bytes(const aa_grid *g);

/* ---- state transfer (host view coherence points: problem(), Userwork_*, outputs) */
int aa_upload_cons(aa_grid *g, const double *U_aos);      /* host ConsS block -> device SoA */
int aa_download_cons(aa_grid *g, double *U_aos);
/* Only the ghost zones of the block (its active zones are left alone): for a caller whose host copy of the active zones is
 * current, i.e. nothing but aa_bvals_* / aa_new_dt ran on the device since the block last travelled either way.  The library keeps
 * track itself: if any call since the last aa_upload_cons / aa_download_cons wrote active zones (integrators, ion step, pinned
 * zones, restriction / flux correction) the whole block is downloaded instead. */
int aa_download_ghost_zones(aa_grid *g, double *U_aos);
int aa_upload_edgeflux(aa_grid *g, const double *ef);
int aa_download_edgeflux(aa_grid *g, double *ef);
int aa_get_mesh_state(const aa_grid *g, double *time, double *dt, int *nstep);  /* MeshS.time/dt/nstep */
int aa_set_mesh_state(aa_grid *g, double time, double dt, int nstep);

/* ---- hooks the problem file installs */
/* globals.h:24 StaticGravPot: the callback is evaluated ONCE on the host at cell centres and
 * face centres (cc_pos.c:36-43) and kept as device tables; pass NULL to remove it.           */
typedef double (*aa_gravpot_fn)(double x1, double x2, double x3);
int aa_set_static_grav_pot(aa_grid *g, aa_gravpot_fn fn);
int aa_set_static_grav_tables(aa_grid *g, const double *phi_cc, const double *phi_f1,
                              const double *phi_f2, const double *phi_f3); /* [N3][N2][N1] each */
/* globals.h:25 CoolingFunc (optically thin cooling in integrate_3d_ctu.c: Steps 1c-3c :359-368 / :662-671 / :846-855 on the L/R
 * states, 8b :2133-2266 P^{n+1/2}, 11c :2943-2953 the energy).  The reference calls a host function pointer per state; a device
 * integrator needs the function itself, so the library carries the one the reference ships: AA_COOL_KOYINUT = KoyInut
 * (microphysics/cool.c:48, cgs units).  CTU integrator only -- the reference's integrate_3d_vl.c has no cooling terms.         */
#define AA_COOL_NONE    0
#define AA_COOL_KOYINUT 1
int aa_set_cooling(aa_grid *g, int kind);
/* Userwork_in_loop of prob/ioniz_sphere.c:255-306 re-imposes fixed values on a fixed set of
 * cells every step; the device-side equivalent is a list of pinned cells (linear index into
 * the [k][j][i] block incl. ghosts, nvar values each) applied after the integrator.          */
int aa_set_pinned_cells(aa_grid *g, long long n, const long long *index, const double *values);
int aa_apply_pinned_cells(aa_grid *g);
/* ionradplane_3d.c:56 add_radplane_3d (called by problem()); dir = -1 (rays along +x1) or -2 (along +x2: two-kernel
 * sub-cycle).  dir = -3 and dir > 0 are refused: the reference has no defined behaviour for them. */
int aa_add_radplane_3d(aa_grid *g, int dir, double flux);
int aa_has_radplane(const aa_grid *g);   /* main.c:546 `radplanelist[0].nradplane > 0`: whether a step includes the ion step */

/* ---- the reference's per-step call sites */
int aa_bvals_mhd(aa_grid *g);                       /* bvals_mhd.c:174 (physical BCs only)        */
int aa_bvals_mhd_side(aa_grid *g, int dir, int side); /* one (*BCFun)(pGrid) call of bvals_mhd.c:196-420:
                                                        dir 0..2, side 0 inner / 1 outer; for drivers that
                                                        interleave user boundary functions (bvals_mhd_fun :917) */
int aa_bvals_ionrad(aa_grid *g);                    /* bvals_ionrad.c:63                          */
int aa_new_dt(aa_grid *g);                          /* new_dt.c:32                                */
int aa_integrate_3d_ctu(aa_grid *g);                /* integrate_3d_ctu.c:110, dt = Grid dt       */
int aa_integrate_3d_vl(aa_grid *g);                 /* integrate_3d_vl.c:96 (NO_H_CORRECTION)     */
int aa_ion_radtransfer_3d(aa_grid *g, int *niter);  /* ionrad_3d.c:862; may shrink the Grid dt    */
int aa_start(aa_grid *g);                           /* main.c:412-451: bvals, bvals_ionrad, new_dt */
int aa_step(aa_grid *g, int *niter);                /* one pass of main.c:519-669                 */

/* ---- phases, for drivers that interleave neighbour exchange / global reductions
 *      (the places where the reference calls MPI)                          */
/* The part of integrate_3d_ctu that reads none of the x3 neighbours' planes (the first-pass x1 / x2 sweeps of the
 * planes ks..ke, integrate_3d_ctu.c:196-620): call it between posting the x3 halo (bvals_mhd.c:423-493) and waiting
 * for it; aa_integrate_3d_ctu then does the rest.  Same bits with or without; a no-op where the split does not apply. */
int aa_integrate_begin(aa_grid *g);      /* (aa_upload_cons, aa_download_cons and aa_history between the two calls are allowed:
                                           * they stage through the face-state area and make the integrator redo these sweeps) */
/* new_dt.c:72-140 inside the integrator: with on != 0 the caller promises that between aa_integrate_3d_ctu / _vl and the
 * next aa_new_dt_local / aa_cfl_max_v nothing but aa_apply_pinned_cells changes the active zones (the order of main.c:572-629
 * when Userwork_in_loop only pins zones); the update kernel then leaves max(|v_d| + a) behind while the new state is in
 * registers and new_dt reads it instead of sweeping the Grid again.  aa_step does this by itself.                     */
int aa_cfl_in_update(aa_grid *g, int on);
int aa_new_dt_local(aa_grid *g, double *dt_cfl);                /* new_dt.c:72-170 before Allreduce */
int aa_ion_begin(aa_grid *g);                                   /* ionrad_3d.c:896-905            */
int aa_ion_rates(aa_grid *g, double *dt_chem, double *dt_therm);/* :922-938 before Allreduce      */
int aa_ion_update(aa_grid *g, double dt, long long *cellcount, double *dt_hydro); /* :965-1002    */
/* Before the first aa_ion_pass of an ion step (after aa_ion_begin): `limit` = the value aa_ion_pick will be given.  The
 * first pass then also applies the first sub-cycle's update with that whole step (the zone is in registers); where the
 * reduction confirms the step -- a quiet radiation field, ONE sub-cycle per hydro step -- the closing update pass has nothing
 * left to do; otherwise the next pass restarts from the state the entry saved.  Same results bit for bit; optional
 * (AA_ION_SPECULATE=0 in the environment turns it off). */
int aa_ion_speculate(aa_grid *g, double limit);
/* The same loop for Grids that run the ONE-KERNEL sub-cycle (aa_ion_is_fused: rays of 48 zones or more; the two
 * calls above refuse such a Grid).  The loop is cut at its only true barrier, the reduction that yields the step:
 * pass n applies update(n-1) and runs sweep(n) + rates(n) on the updated zones; its reduction words (MIN dt_chem,
 * MIN dt_therm, MAX (|v|+a)/dx, SUM cells out of range, OR negative-dt_chem -- the operands of the reference's four
 * MPI_Allreduce calls ionrad_3d.c:275,399,554,672) land in DEVICE memory, AA_ION_WORDS doubles per rank.  aa_ion_pick
 * (a one-thread kernel) folds the words of all ranks, books the update just applied -- the time covered so far stays on
 * the device -- and picks the next step; a pass whose step was cut back to the limit skips its sweep by itself.  So a
 * multi-rank driver issues ONE all-gather per sub-cycle on the Grid's stream and the host reads back ONCE per sub-cycle:
 *   aa_ion_begin; aa_ion_pass(0,1); [all-gather]; aa_ion_pick(first=1);
 *   repeat: aa_ion_pass(1,1); [all-gather]; aa_ion_pick(first=0); aa_ion_fetch -> step, limit flag and stop criteria of
 *           the update that pass applied; the reference's stop tests (:974-1002);
 *   aa_ion_finish (GridS.EdgeFlux of the last sweep that counted; the one after a stop was speculative).       */
#define AA_ION_WORDS 8
int aa_ion_is_fused(const aa_grid *g);
int aa_ion_pass(aa_grid *g, int update, int sweep, double *dev_words);
int aa_ion_pick(aa_grid *g, const double *dev_words_all, int nranks, int first, double limit);
int aa_ion_fetch(aa_grid *g, double *dt, int *limit_hit, double *dt_chem, double *dt_therm, long long *cellcount,
                 double *dt_hydro, int *neg_dt_chem);
int aa_ion_finish(aa_grid *g);
/* ion_radtransfer_3d (ionrad_3d.c:862-1047, root level) of a rank of a multi-rank run in ONE call: the loop above with the
 * driver's collective as a callback.  `gather(ctx)` must all-gather this rank's AA_ION_WORDS doubles at dev_words into
 * dev_words_all (nranks x AA_ION_WORDS, DEVICE memory) on the Grid's stream and return 0; it is called once per pass, between
 * aa_ion_pass and aa_ion_pick.  pGrid->dt (aa_get_mesh_state) is cut back as the reference does (:985-1000, :1019-1023).
 * gather == NULL: one rank (dev_words / dev_words_all may be NULL).  Needs the one-kernel sub-cycle (aa_ion_is_fused). */
typedef int (*aa_gather_fn)(void *ctx);
int aa_ion_radtransfer_3d_gather(aa_grid *g, double *dev_words, const double *dev_words_all, int nranks, aa_gather_fn gather,
                                 void *ctx, int *niter);
int aa_host_syncs(aa_grid *g, int reset);   /* stream synchronisations that returned scalars to the host so far */
/* x3 halo: pack_ix3/pack_ox3/unpack_* (bvals_mhd.c:2608-3175).  side 0 = inner (ks..ks+3),
 * side 1 = outer; buffers are DEVICE pointers of aa_halo_doubles() doubles.                */
long long aa_halo_doubles(const aa_grid *g);
int aa_pack_x3(aa_grid *g, int side, double *dev_buf);
int aa_unpack_x3(aa_grid *g, int side, const double *dev_buf);
/* x2 x x3 pencils (init_mesh.c:526-620 NGrid_x2 x NGrid_x3; x1 is never cut: the rays): the x2 halo, four rows x all i incl.
 * the x1 ghost zones x the ACTIVE k-planes (bvals_mhd.c:2462 pack_ix2 / pack_ox2, :2896 unpack_ix2 / unpack_ox2).  A bvals_mhd
 * call of such a driver: aa_bvals_mhd_side for x1; this exchange and aa_bvals_mhd_side for the physical x2 sides; then x3. */
long long aa_halo_doubles_x2(const aa_grid *g);
int aa_pack_x2(aa_grid *g, int side, double *dev_buf);
int aa_unpack_x2(aa_grid *g, int side, const double *dev_buf);
/* The same halos through HOST buffers, for a driver that moves them itself (the reference's own MPI ranks linked on the shim:
 * MPI_Isend / MPI_Irecv of bvals_mhd.c:296-493): dir 1 = x2, 2 = x3; aa_halo_doubles_dir doubles per buffer. */
long long aa_halo_doubles_dir(const aa_grid *g, int dir);
int aa_halo_get(aa_grid *g, int dir, int side, double *host_buf);        /* pack_i* (side 0) / pack_o* (side 1) -> host_buf */
int aa_halo_put(aa_grid *g, int dir, int side, const double *host_buf);  /* host_buf -> the ghost planes of that side      */
int aa_device_count(void);                                               /* visible HIP devices (0: none)                  */

/* ---- static mesh refinement (reference built with --enable-smr): nested levels, all resident on one GPU.
 *      levels[] holds one Grid per Domain in the order of the reference's loops (MeshS.Domain[nl][nd], athena.h:355-361):
 *      level by level from the root, Domains of a level in deck order (they neither overlap nor touch, init_mesh.c:398-418;
 *      up to three children per Grid).  Each was created with aa_params.level = its level, Nx = the
 *      Domain's zones, MinX = its lower edge (init_mesh.c:281-286), bc = 0 on fine/coarse sides
 *      (bvals_mhd.c:193-361 ProlongateLater); disp[3*g+d] = <domainN> iDisp/jDisp/kDisp in zones
 *      of its level.  Fill every level (problem(), hooks) before aa_mesh_start().  The Mesh takes
 *      over the levels' streams; destroy it before the levels.                               */
typedef struct aa_mesh aa_mesh;
int  aa_mesh_create(int nlevels, aa_grid **levels, const int *disp, aa_mesh **out); /* init_grid.c overlap
                                                          tables (:150-560) + SMR_init (smr.c:2931)  */
void aa_mesh_destroy(aa_mesh *m);
int  aa_mesh_get_state(const aa_mesh *m, double *time, double *dt, int *nstep);     /* MeshS          */
int  aa_mesh_set_state(aa_mesh *m, double time, double dt, int nstep);
int  aa_mesh_set_stream(aa_mesh *m, void *hip_stream);
int  aa_mesh_restrict_correct(aa_mesh *m);        /* smr.c:1207 RestrictCorrect                      */
int  aa_mesh_restrict_correct_pair(aa_mesh *m, int l); /* its step for one pair: level l+1 -> level l  */
int  aa_mesh_ionrad_restrict_correct(aa_mesh *m); /* smr.c:85 ionradRestrictCorrect                  */
int  aa_mesh_prolongate(aa_mesh *m);              /* smr.c:2359 Prolongate                           */
int  aa_mesh_new_dt(aa_mesh *m);                  /* new_dt.c:32 over all levels                     */
int  aa_mesh_ion_radtransfer(aa_mesh *m, int level, int *niter); /* ionrad_3d.c:862 on Domain[level],
                                                     incl. ionrad_prolong_rcv/_snd (ionrad_smr.c)   */
int  aa_mesh_ionflux_prolong(aa_mesh *m, int level); /* ionrad_prolong_snd(level-1) + _rcv(level) alone    */
int  aa_mesh_start(aa_mesh *m);                   /* main.c:395-447                                  */
int  aa_mesh_step(aa_mesh *m, int *niter);        /* one pass of main.c:519-669; niter[nlevels]      */

/* Multi-GPU SMR: every level is cut into x3 slabs at the SAME root planes, so that restriction, the
 * radiation hand-off and prolongation stay inside a rank.  aa_mesh_create_local builds one rank's stack
 * of slabs from explicit links (21 ints per link l: cs[3] = first parent zone under the child slab as a
 * local index incl. ghosts, n[3] parent zones, prol[6] = that side of the child slab is a fine/coarse
 * boundary of the LEVEL, corr[6] = the parent zone outside that side is on this rank, cdisp[3] = child
 * origin - 2 x parent origin).  Where a level ends exactly at a cut (prol=1, corr=0 on an x3 side) the
 * parent plane outside belongs to the neighbouring rank: the child's restricted boundary flux travels
 * as a message of (Nx1/2)(Nx2/2) x 6 doubles (smr.c:1592-1640 is the same message under MPI).        */
int  aa_mesh_create_local(int nlevels, aa_grid **levels, const int *links, aa_mesh **out);
int  aa_flux_x3_export(aa_grid *child, int side, double *dev_buf);   /* side 0 lower / 1 upper boundary */
int  aa_flux_x3_apply(aa_grid *parent, int side, int i0, int j0, int n1, int n2, const double *dev_buf);
int  aa_cfl_max_v(aa_grid *g, double v[3]);       /* new_dt.c:72-140 of one Grid (the MAX over the slabs of
                                                     a level and the carry from level to level are the driver's) */

/* ---- function-level kernels on device arrays' host mirrors (parity tests): n states of
 *      nvar = 5+nscal doubles each, same conventions as fluxes()/lr_states()              */
int aa_test_fluxes(int nscal, double gamma, int n, const double *Ul, const double *Ur,
                   const double *etah, double *F);                      /* roe.c:59        */
int aa_test_lr_states(int nscal, double gamma, int n, const double *W, double dt, double dx,
                      int il, int iu, double *Wl, double *Wr);          /* lr_states_plm.c:62 */
int aa_test_lr_states_ppm(int nscal, double gamma, int n, const double *W, double dt, double dx,
                          int il, int iu, double *Wl, double *Wr);      /* lr_states_ppm.c:91  */

int aa_test_xdiv(int n, const double *a, const double *b, double *out);  /* csrc/hydro_dev.h x_div / x_sqrt beside the compiler's
                                                                            a/b and sqrt(a): out[5][n] = x_div, a/b, x_sqrt,
                                                                            sqrt, x_div_r(a, b, 1/b) */
int aa_test_explog(int n, const double *x, double *y_exp, double *y_log);  /* the exp / ln of csrc/ion_pass.hip
                                                                             (n a multiple of 4): exp(x), ln|x|  */

/* ---- dump_history.c:157-200: volume integrals over the active zones of this Grid, in the column
 *      order of the .hst file: mass, total E, x1/x2/x3 Mom., x1/x2/x3-KE, scalar 0 (0 if NSCALARS=0).
 *      Sums over Grids (MPI_Reduce :257) and the division by the Domain volume (:271-279) are the
 *      caller's.                                                                               */
int aa_history(aa_grid *g, double sums[9]);

/* ---- measurement: per-kernel accumulated device time (hipEvent pairs on the stream) */
int         aa_profile_enable(aa_grid *g, int on);
int         aa_profile_reset(aa_grid *g);
int         aa_profile_count(const aa_grid *g);
const char *aa_profile_name(const aa_grid *g, int i);
int         aa_profile_get(aa_grid *g, int i, double *total_ms, long long *launches);

#ifdef __cplusplus
}
#endif
#endif
