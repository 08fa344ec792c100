/* host/athena_shim.c -- the reference's hot-path entry points in C, on top of the
 * MI355X library.  Linked in place of the reference's integrators/, reconstruction/, rsolvers/,
 * ionradiation/, bvals_mhd.o and new_dt.o, it lets the reference's own driver (main.c), mesh
 * construction, parameter reader, outputs and problem files run unchanged with the per-step
 * physics on the GPU.  Single level, single Grid per process (NO_MPI / NO_SMR configuration).
 *
 * Host/device coherence (the reference's problem files and outputs index pG->U on the host):
 *   AA_COHERENCE=step  (default) the host block is refreshed after Integrate() (so that
 *                      Userwork_in_loop sees and may edit it; re-uploaded before new_dt) and
 *                      after the end-of-step bvals_mhd (so data_output sees ghost zones too).
 *                      Always correct; costs three PCIe transfers of U per step.
 *   AA_COHERENCE=learn first step as above, and the cells Userwork_in_loop changed are recorded;
 *                      from then on they are re-imposed on the device (aa_apply_pinned_cells)
 *                      and the host block is refreshed only at the end-of-step bvals_mhd every
 *                      AA_SYNC_EVERY steps (default 1).  Valid when Userwork writes the same
 *                      values every step, as prob/ioniz_sphere.c:255-306 does.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/athena_compat.h"
#include "../../include/athena_amd.h"

/* provided by the driver this shim is linked into (globals.h:14-25, prototypes.h:118,178-184) */
extern Real CourNo, Gamma, Gamma_1;
extern GravPotFun_t StaticGravPot;
extern CoolingFun_t CoolingFunc;
extern double par_getd(char *block, char *name);
extern void ath_error(char *fmt, ...);

static aa_grid *G = NULL;
static MeshS *M = NULL, *M0 = NULL;
static GridS *PG = NULL;
static int host_newer = 1;          /* the host block holds data the device has not seen */
static int learn = 0, learned = 0, sync_every = 1;
static double *snap = NULL;         /* copy of U after Integrate (learn mode) */
static size_t ncell = 0;

#define CHK(call) do { if ((call) != 0) ath_error("[athena_amd]: %s\n", aa_last_error()); } while (0)

static double *host_block(void) { return (double*)&(PG->U[0][0][0]); }   /* ath_array.c:100-117 */
#if AA_ION_RADPLANE
static double *host_edgeflux(void) { return (double*)&(PG->EdgeFlux[0][0][0]); }
#endif

/* configure --with-integrator=vl  <->  -DAA_VL_INTEGRATOR at compile time, or AA_INTEGRATOR=vl */
static int use_vl(void)
{
#ifdef AA_VL_INTEGRATOR
  return 1;
#else
  const char *e = getenv("AA_INTEGRATOR");
  return (e && strcmp(e, "vl") == 0);
#endif
}

static void ensure_grid(MeshS *pM)
{
  aa_params p; DomainS *pD; int d; const char *env;
  if (G) return;
  if (pM->NLevels != 1 || pM->DomainsPerLevel[0] != 1)
    ath_error("[athena_amd]: single level / single Domain only (SMR is a later round)\n");
  if (CoolingFunc != NULL) ath_error("[athena_amd]: CoolingFunc is not supported on this path\n");
  M = pM; pD = &pM->Domain[0][0]; PG = pD->Grid;
  if (sizeof(ConsS) != (5 + AA_NSCALARS)*sizeof(double)) ath_error("[athena_amd]: ConsS layout\n");
  memset(&p, 0, sizeof p);
  for (d = 0; d < 3; d++) {
    p.Nx[d] = PG->Nx[d]; p.rootNx[d] = pM->Nx[d];
    p.xmin[d] = pM->RootMinX[d]; p.xmax[d] = pM->RootMaxX[d]; p.MinX[d] = PG->MinX[d];
  }
  p.bc[0] = pM->BCFlag_ix1; p.bc[1] = pM->BCFlag_ox1; p.bc[2] = pM->BCFlag_ix2;
  p.bc[3] = pM->BCFlag_ox2; p.bc[4] = pM->BCFlag_ix3; p.bc[5] = pM->BCFlag_ox3;
  for (d = 0; d < 6; d++)
    if (p.bc[d] != 1 && p.bc[d] != 2 && p.bc[d] != 4) ath_error("[bvals_init]: bc flag = %d unknown\n", p.bc[d]);
  p.nscal = AA_NSCALARS;
#if AA_ION_RADPLANE
  p.ion = (pM->radplanelist != NULL && pM->radplanelist->nradplane > 0);
#else
  p.ion = 0;
#endif
  p.gamma = Gamma; p.cour_no = CourNo; p.tlim = par_getd("time", "tlim");
  if (p.ion) {                                   /* ionrad_3d.c:742-757 */
    p.sigma_ph = par_getd("ionradiation", "sigma_ph"); p.m_H = par_getd("ionradiation", "m_H");
    p.mu = par_getd("ionradiation", "mu"); p.e_gamma = par_getd("ionradiation", "e_gamma");
    p.alpha_C = par_getd("ionradiation", "alpha_C"); p.k_B = par_getd("ionradiation", "k_B");
    p.time_unit = par_getd("ionradiation", "time_unit");
    p.max_de_iter = par_getd("ionradiation", "max_de_iter");
    p.max_de_therm_iter = par_getd("ionradiation", "max_de_therm_iter");
    p.max_dx_iter = par_getd("ionradiation", "max_dx_iter");
    p.max_de_step = par_getd("ionradiation", "max_de_step");
    p.max_de_therm_step = par_getd("ionradiation", "max_de_therm_step");
    p.max_dx_step = par_getd("ionradiation", "max_dx_step");
    p.tfloor = par_getd("ionradiation", "tfloor"); p.tceil = par_getd("ionradiation", "tceil");
    p.maxiter = (int)par_getd("ionradiation", "maxiter");
  }
  env = getenv("AA_DEVICE"); p.device = env ? atoi(env) : 0;
  p.integrator = use_vl();
  CHK(aa_create(&p, &G));
  ncell = (size_t)(PG->Nx[0] + 2*AA_NGHOST)*(PG->Nx[1] + 2*AA_NGHOST)*(PG->Nx[2] + 2*AA_NGHOST);
#if AA_ION_RADPLANE
  if (p.ion) CHK(aa_add_radplane_3d(G, pM->radplanelist->dir[0], pM->radplanelist->flux_i));
#endif
  if (StaticGravPot != NULL) CHK(aa_set_static_grav_pot(G, StaticGravPot));
  env = getenv("AA_COHERENCE"); learn = (env && strcmp(env, "learn") == 0);
  env = getenv("AA_SYNC_EVERY"); sync_every = env ? atoi(env) : 1; if (sync_every < 1) sync_every = 1;
  fprintf(stderr, "[athena_amd] Grid %dx%dx%d on HIP device %d, %.2f GB resident, coherence=%s\n",
          p.Nx[0], p.Nx[1], p.Nx[2], p.device, aa_device_bytes(G)/1e9, learn ? "learn" : "step");
}

static void to_device(void)
{
  if (host_newer) { CHK(aa_upload_cons(G, host_block())); host_newer = 0; }
  CHK(aa_set_mesh_state(G, M->time, M->dt, M->nstep));
}

static void to_host(void)
{
  CHK(aa_download_cons(G, host_block()));
#if AA_ION_RADPLANE
  if (M->radplanelist != NULL && M->radplanelist->nradplane > 0) CHK(aa_download_edgeflux(G, host_edgeflux()));
#endif
}

/* ---- reconstruction / integrator ---------------------------------------------------------- */
void lr_states_init(MeshS *pM) { (void)pM; }
void lr_states_destruct(void) {}

static void integrate_3d_ctu_amd(DomainS *pD)
{
  GridS *pG = pD->Grid;
  to_device();
  CHK(aa_set_mesh_state(G, M->time, pG->dt, M->nstep));
  if (use_vl()) CHK(aa_integrate_3d_vl(G)); else CHK(aa_integrate_3d_ctu(G));
  if (learn && learned) { CHK(aa_apply_pinned_cells(G)); return; }
  to_host();                                    /* Userwork_in_loop reads and may write pG->U */
  host_newer = 1;
  if (learn) {
    if (!snap) snap = (double*)malloc(ncell*sizeof(ConsS));
    memcpy(snap, host_block(), ncell*sizeof(ConsS));
  }
}

VDFun_t integrate_init(MeshS *pM)
{
  if (CourNo > 0.5)     /* integrate.c:66-68 */
    ath_error("<time>cour_no was set to %g: must be <= 0.5 with 3D integrator\n", CourNo);
  ensure_grid(pM);
  return integrate_3d_ctu_amd;
}

void integrate_destruct(void)
{
  if (G) { aa_destroy(G); G = NULL; }
  free(snap); snap = NULL;
}

/* ---- boundaries / time step ------------------------------------------------------------- */
void bvals_mhd_init(MeshS *pM) { M0 = pM; }   /* main.c:412, before the first bvals_mhd; the
                                                  non-ion DomainS has no Mesh back-pointer */

/* bvals_mhd.c:917: problem() may enrol its own boundary function for a side (it runs on the host
 * block, like every problem-file hook) */
void bvals_mhd_fun(DomainS *pD, enum BCDirection dir, VGFun_t prob_bc)
{
  switch (dir) {
  case left_x1:  pD->ix1_BCFun = prob_bc; break;
  case right_x1: pD->ox1_BCFun = prob_bc; break;
  case left_x2:  pD->ix2_BCFun = prob_bc; break;
  case right_x2: pD->ox2_BCFun = prob_bc; break;
  case left_x3:  pD->ix3_BCFun = prob_bc; break;
  case right_x3: pD->ox3_BCFun = prob_bc; break;
  default: ath_error("[bvals_fun]: Unknown direction = %d\n", (int)dir);
  }
}

static int after_new_dt = 0, steps_since_sync = 0;
void bvals_mhd(DomainS *pD)
{
  VGFun_t usr[6]; int d, side, any = 0;
  ensure_grid(M0);
  to_device();
  usr[0] = pD->ix1_BCFun; usr[1] = pD->ox1_BCFun; usr[2] = pD->ix2_BCFun;
  usr[3] = pD->ox2_BCFun; usr[4] = pD->ix3_BCFun; usr[5] = pD->ox3_BCFun;
  for (d = 0; d < 6; d++) any |= (usr[d] != NULL);
  if (!any) CHK(aa_bvals_mhd(G));
  else {
    /* bvals_mhd.c:196-420: ix1, ox1, ix2, ox2, ix3, ox3 in this order; a user function sees the host
     * block with everything filled so far and its ghost zones travel back before the next side */
    for (d = 0; d < 3; d++) for (side = 0; side < 2; side++) {
      if (usr[2*d + side] == NULL) { CHK(aa_bvals_mhd_side(G, d, side)); continue; }
      CHK(aa_download_cons(G, host_block()));
      (*usr[2*d + side])(PG);
      CHK(aa_upload_cons(G, host_block()));
    }
  }
  /* main.c calls bvals_mhd after the ion step (:552; nothing on the host looks at U before
   * Integrate) and after new_dt (:638; data_output() at the top of the next cycle reads the host
   * block): only the latter refreshes the host view */
  if (after_new_dt) {
    after_new_dt = 0;
    if (!(learn && learned) || (++steps_since_sync >= sync_every)) { to_host(); steps_since_sync = 0; }
  }
}

void new_dt(MeshS *pM)
{
  int nl, nd; double t, dt; int n;
  ensure_grid(pM);
  if (learn && !learned && snap) {              /* what did Userwork_in_loop change? */
    const double *h = host_block(); long long cnt = 0, c; size_t i; int v;
    const int nv = 5 + AA_NSCALARS; long long *idx; double *val;
    for (i = 0; i < ncell; i++) if (memcmp(h + i*nv, snap + i*nv, nv*sizeof(double)) != 0) cnt++;
    idx = (long long*)malloc((size_t)(cnt + 1)*sizeof(long long)); val = (double*)malloc((size_t)(cnt + 1)*nv*sizeof(double));
    for (i = 0, c = 0; i < ncell; i++) if (memcmp(h + i*nv, snap + i*nv, nv*sizeof(double)) != 0) {
      idx[c] = (long long)i; for (v = 0; v < nv; v++) val[c*nv + v] = h[i*nv + v]; c++;
    }
    CHK(aa_set_pinned_cells(G, cnt, idx, val));
    free(idx); free(val); free(snap); snap = NULL; learned = 1;
    CHK(aa_apply_pinned_cells(G));              /* this step's Userwork, on the device */
    fprintf(stderr, "[athena_amd] Userwork_in_loop pins %lld cells; re-imposed on the device from now on\n", cnt);
  }
  if (learn && learned) host_newer = 0;         /* the stale host copy must not travel back */
  to_device();
  CHK(aa_new_dt(G));
  CHK(aa_get_mesh_state(G, &t, &dt, &n));
  pM->dt = dt;
  after_new_dt = 1;
  for (nl = 0; nl < pM->NLevels; nl++) for (nd = 0; nd < pM->DomainsPerLevel[nl]; nd++)
    if (pM->Domain[nl][nd].Grid != NULL) pM->Domain[nl][nd].Grid->dt = dt;      /* new_dt.c:189-195 */
}

/* ---- ion radiation ------------------------------------------------------------------------ */
void ion_radtransfer_init_domain(MeshS *pM) { (void)pM; }

#if AA_ION_RADPLANE
static void ion_radtransfer_3d_amd(DomainS *pD)
{
  GridS *pG = pD->Grid; MeshS *pM = pD->Mesh; int niter = 0; double t, dt; int n;
  to_device();
  CHK(aa_set_mesh_state(G, pM->time, pG->dt, pM->nstep));
  CHK(aa_ion_radtransfer_3d(G, &niter));
  CHK(aa_get_mesh_state(G, &t, &dt, &n));
  pG->dt = dt; pM->dt = dt;                     /* ionrad_3d.c:1033 */
  fprintf(stderr, "Radiation done in %d iterations; new dt = %e\n", niter, dt);
}

VDFun_t ion_radtransfer_init(MeshS *pM, int ires)
{
  (void)ires;
  ensure_grid(pM);
  return ion_radtransfer_3d_amd;
}

void bvals_ionrad_init(MeshS *pM) { (void)pM; }
void bvals_ionrad(DomainS *pD) { ensure_grid(pD->Mesh); CHK(aa_bvals_ionrad(G)); }
void set_coarse_time(void) {}
void clear_coarse_time(void) {}

void add_radplane_3d(GridS *pGrid, int dir, Real flux)   /* ionradplane_3d.c:56-66 */
{
  MeshS *pMesh = pGrid->Mesh;
  pMesh->radplanelist->dir[0] = dir;
  pMesh->radplanelist->flux_i = flux;
  if (G) CHK(aa_add_radplane_3d(G, dir, flux));
}
#endif /* AA_ION_RADPLANE */
