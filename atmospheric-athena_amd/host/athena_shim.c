/* host/athena_shim.c -- the reference's hot-path entry points in C, on top of the
 * MI355X library.  Linked in place of the reference's integrators/, reconstruction/, rsolvers/,
 * ionradiation/, bvals_mhd.o and new_dt.o, it lets the reference's own driver (main.c), mesh
 * construction, parameter reader, outputs and problem files run unchanged with the per-step
 * physics on the GPU.  One Grid per Domain and one Domain per level, all on one GPU: the single
 * level configuration, or (compiled with -DAA_SMR against the reference's --enable-smr build) the
 * nested levels of static mesh refinement, where it also provides SMR_init, RestrictCorrect,
 * Prolongate and ionradRestrictCorrect (smr.c) on top of aa_mesh_*.
 *
 * Compiled with -DAA_MPI against the reference's --enable-mpi build (single level), every MPI rank of the reference drives
 * ONE GPU through this shim: the Domain is cut by the reference's own init_mesh (NGrid_x2 x NGrid_x3; NGrid_x1 must be 1:
 * the rays), the ghost zones between neighbouring Grids travel as in bvals_mhd.c:296-493 -- pack on the device, MPI_Isend /
 * MPI_Irecv of host buffers in pD->Comm_Domain, unpack on the device, x1 then x2 then x3 --, new_dt reduces with
 * MPI_Allreduce(MIN) (new_dt.c:177) and the radiation sub-cycle with the reference's own two rounds of reductions
 * (ionrad_3d.c:275,399,554,672).  HIP device of a rank: AA_DEVICE, else rank modulo the visible devices.
 *
 * -DAA_MPI and -DAA_SMR together (the reference's README.rst:25 configuration, --enable-mpi ... --enable-smr): every rank holds at
 * most one Grid per Domain (init_mesh.c:583-700) and drives them as ONE stack of nested slabs on its GPU (aa_mesh_create_local).
 * Supported decompositions: one Domain per level, every Domain cut along x3 only (NGrid_x1 = NGrid_x2 = 1), and the Grids a rank
 * holds nested in each other -- the child slab of a rank lies over that rank's parent slab, which the reference's equal division
 * gives when a refined Domain is centred on the cuts of its parent (two ranks) or spans the whole x3 extent (any number).  Then
 * RestrictCorrect, Prolongate, ionradRestrictCorrect and the coarse -> fine radiation hand-off never leave the rank
 * (smr.c:136,176,1169,1191 and ionrad_smr.c:105,122,452 exchange with a rank's own Grids only); what crosses ranks is the x3 halo
 * of every level in its Domain's communicator (bvals_mhd.c:423-493), new_dt's MIN (new_dt.c:177; max_v carried from Grid to
 * Grid on each rank first, :33, as the reference does) and the sub-cycle reductions of each level in its Comm_Domain
 * (ionrad_3d.c:275,399,554,672).  Anything else is refused with a message.  Host/device coherence: `step` only.
 *
 * Host/device coherence (the reference's problem files and outputs index pG->U on the host):
 *   AA_COHERENCE=step  (default) the host block is refreshed after Integrate() (so that
 *                      Userwork_in_loop sees and may edit it; re-uploaded before new_dt) and
 *                      after the end-of-step bvals_mhd (so data_output sees ghost zones too).
 *                      Always correct, whatever the problem file does; costs three PCIe transfers of U per step.
 *   AA_COHERENCE=auto  (opt-in: a CONTRACT with the problem file) the first two steps run as `step` while the zones
 *                      Userwork_in_loop writes are recorded; if both steps wrote the same values into the same zones
 *                      (prob/ioniz_sphere.c:255-306 does; a problem without Userwork trivially does) they are re-imposed
 *                      on the device from then on (aa_apply_pinned_cells) and the host block is refreshed only when
 *                      main() is about to read it: when an <outputN> block is due at the next data_output (its schedule,
 *                      output.c:205-208 and :507-522, is mirrored from the same par table), when the loop is about to end
 *                      (tlim / nlim: the forced final output, main.c:743) and after SIGTERM (ath_signal.c).  The imprint
 *                      is RE-VALIDATED every AA_REVALIDATE_EVERY steps (default 8) and on the step after every such
 *                      refresh: that step runs as `step`, and the host block after Userwork_in_loop must equal the
 *                      downloaded state with the recorded values imposed, zone for zone; if not (a Userwork that becomes
 *                      active later, e.g. `if (time > t0)`, or whose values drift) the shim says so on stderr and stays
 *                      `step` for the rest of the run -- the writes of the steps since the last validation were NOT
 *                      seen by the device, which is why this mode is not the default.
 *                      Not seen at all: outputs enrolled by the problem file itself at run time, and reads of pG->U by
 *                      problem-file code other than Userwork_in_loop's writes -- use `step` for those.
 *   AA_COHERENCE=learn first step as `step`, and the cells Userwork_in_loop changed are recorded;
 *                      from then on they are re-imposed on the device (aa_apply_pinned_cells)
 *                      and the host block is refreshed only at the end-of-step bvals_mhd every
 *                      AA_SYNC_EVERY steps (default 1).  Unverified (round 1's form): valid only when Userwork writes
 *                      the same values every step.
 */
#include <math.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/athena_compat.h"
#include "../../include/athena_amd.h"

/* provided by the driver this shim is linked into (globals.h:14-25, prototypes.h:118,178-184) */
extern Real CourNo, Gamma, Gamma_1;
extern GravPotFun_t StaticGravPot;
extern CoolingFun_t CoolingFunc;
extern Real KoyInut(const Real dens, const Real Press, const Real dt) __attribute__((weak));   /* microphysics/cool.c:48 */
extern double par_getd(char *block, char *name);
extern double par_getd_def(char *block, char *name, double def);
extern int par_geti_def(char *block, char *name, int def);
extern int par_exist(char *block, char *name);
extern void ath_error(char *fmt, ...);
#define MAXLEV 16
#ifdef AA_MPI
#ifdef AA_SMR
#define AA_MPISMR 1
#endif
extern int myID_Comm_world;                 /* globals.h:27 */
static MPI_Comm comm_lev[MAXLEV];           /* pD->Comm_Domain of every Domain this rank holds a Grid of */
static int nb_lev[MAXLEV][4];               /* lx2, rx2, lx3, rx3 neighbour Grids in that communicator (ID_Comm_Domain; -1: none) */
static double *hbuf[MAXLEV][2][2][2];       /* [level][dir - 1][side][send, recv] host buffers of the halo */
static int nranks_lev[MAXLEV];
#define comm_dom comm_lev[0]
#define nb_id nb_lev[0]
#define nranks_dom nranks_lev[0]
#endif

static int NL = 0;                          /* Grids: one per Domain, level by level (1 without SMR) */
static int LBASE[MAXLEV + 1];               /* index of the first Domain of a level in that order */
#define GI(pD) (LBASE[(pD)->Level] + (pD)->DomNumber)
static aa_grid *G[MAXLEV];
static GridS *PG[MAXLEV];
static MeshS *M = NULL, *M0 = NULL;
#ifdef AA_SMR
static aa_mesh *MM = NULL;
#endif
#ifdef AA_MPISMR
static int LI[MAXLEV];              /* Domain index -> place in this rank's stack of nested slabs (aa_mesh_create_local) */
static int NLOC = 0;
static double tcoarse_ = 0.0;       /* ionrad_3d.c:44 */
#endif
#define HAVE(l) (G[l] != NULL)      /* (MPI + SMR: a rank need not hold a Grid of every Domain) */
static int host_newer[MAXLEV];      /* the host block of this level holds data the device has not seen */
static int active_same[MAXLEV];     /* host and device agree on the ACTIVE zones of this level (only ghost zones may differ) */
static int learn = 0, learned = 0, sync_every = 1;
/* AA_COHERENCE=auto: the write set of Userwork_in_loop must repeat before it is trusted; the output schedule */
static int automode = 0, gave_up = 0;
/* auto, after the imprint is trusted: the copy of it that re-validation compares against, and when to look again */
static long long pin_n[MAXLEV]; static long long *pin_idx[MAXLEV]; static double *pin_val[MAXLEV];
static int reval_every = 8, steps_since_reval = 0, verify_now = 0, verify_next = 0;
static long long nw_prev[MAXLEV]; static long long *iw_prev[MAXLEV]; static double *vw_prev[MAXLEV];
#define MAXOUT_MIRROR 64
static int nout = 0; static double out_t[MAXOUT_MIRROR], out_dt[MAXOUT_MIRROR];
static double tlim_ = 0; static int nlim_ = -1;
static volatile sig_atomic_t term_seen = 0;
static void (*old_term)(int) = SIG_DFL;
static int term_hooked = 0;
static double *snap[MAXLEV];        /* copy of U after Integrate (learn mode) */
static size_t ncell[MAXLEV];
static int integrated = 0;          /* Integrate ran since the last host refresh (SMR) */

#define CHK(call) do { if ((call) != 0) ath_error("[athena_amd]: %s\n", aa_last_error()); } while (0)

static double *host_block(int l) { return (double*)&(PG[l]->U[0][0][0]); }   /* ath_array.c:100-117 */
#if AA_ION_RADPLANE
static double *host_edgeflux(int l) { return (double*)&(PG[l]->EdgeFlux[0][0][0]); }
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

/* configure without --enable-h-correction (its default)  <->  -DAA_NO_H_CORRECTION at compile time, or AA_H_CORRECTION=0 */
static int no_h_correction(void)
{
#ifdef AA_NO_H_CORRECTION
  return 1;
#else
  const char *e = getenv("AA_H_CORRECTION");
  return (e && atoi(e) == 0);
#endif
}

/* configure --with-order=3  <->  -DAA_THIRD_ORDER at compile time, or AA_ORDER=3 */
static int recon_order(void)
{
#ifdef AA_THIRD_ORDER
  return 3;
#else
  const char *e = getenv("AA_ORDER");
  return (e && atoi(e) == 3) ? 3 : 2;
#endif
}

static void ensure_grid(MeshS *pM)
{
  aa_params p; DomainS *pD; int d, l, irefine; const char *env;
  int disp[3*MAXLEV];
  if (NL) return;
#ifdef AA_SMR
  { int nl_, ng = 0; for (nl_ = 0; nl_ < pM->NLevels; nl_++) ng += pM->DomainsPerLevel[nl_];
    if (pM->NLevels > MAXLEV || ng > MAXLEV) ath_error("[athena_amd]: more than %d Domains\n", MAXLEV); }
#else
  if (pM->NLevels != 1) ath_error("[athena_amd]: this shim was compiled without -DAA_SMR: single level only\n");
#endif
  /* several Domains on a level (athena.h:355-361, init_mesh.c:131-235): one device Grid each, in the order of the
   * reference's loops over (nl, nd) */
  { int nl_; LBASE[0] = 0; for (nl_ = 0; nl_ < pM->NLevels; nl_++) LBASE[nl_ + 1] = LBASE[nl_] + pM->DomainsPerLevel[nl_]; }
  /* globals.h:25: the device integrator carries the cooling function the reference ships (microphysics/cool.c:48); any other
   * host function cannot run inside a kernel */
  if (CoolingFunc != NULL && (KoyInut == NULL || CoolingFunc != KoyInut))
    ath_error("[athena_amd]: CoolingFunc is a function the device integrator does not carry (only KoyInut, microphysics/cool.c)\n");
  if (CoolingFunc != NULL && use_vl()) fprintf(stderr, "[athena_amd] CoolingFunc is enrolled, but the van Leer integrator has no cooling terms (integrate_3d_vl.c): ignored, as in the reference\n");
  if (sizeof(ConsS) != (5 + AA_NSCALARS)*sizeof(double)) ath_error("[athena_amd]: ConsS layout\n");
  M = pM;
  env = getenv("AA_COHERENCE");
  automode = (env && strcmp(env, "auto") == 0);          /* default: step (always correct) */
  learn = automode || (env && strcmp(env, "learn") == 0);
  { const char *r = getenv("AA_REVALIDATE_EVERY"); reval_every = r ? atoi(r) : 8; if (reval_every < 1) reval_every = 1; }
  if (env && strcmp(env, "step") != 0 && strcmp(env, "learn") != 0 && strcmp(env, "auto") != 0)
    ath_error("[athena_amd]: AA_COHERENCE=%s (auto, step or learn)\n", env);
  if (automode) {            /* the <outputN> schedule main()'s data_output will follow (output.c:183-208) */
    int outn, maxout = par_geti_def("job", "maxout", 10); char block[32];
    for (outn = 1; outn <= maxout && nout < MAXOUT_MIRROR; outn++) {
      sprintf(block, "output%d", outn);
      if (!par_exist(block, "out_fmt") && !par_exist(block, "name")) continue;
      out_dt[nout] = par_getd(block, "dt");
      /* a fresh run's forced first dump (main.c:500) moves every next-output time on by one dt (output.c:509-511); a
       * block that carries its own "time" (restart files do) is taken as is: at worst one refresh too many */
      out_t[nout] = par_exist(block, "time") ? par_getd(block, "time") : pM->time + out_dt[nout];
      nout++;
    }
    tlim_ = par_getd("time", "tlim"); nlim_ = par_geti_def("time", "nlim", -1);
  }
  env = getenv("AA_SYNC_EVERY"); sync_every = env ? atoi(env) : 1; if (sync_every < 1) sync_every = 1;
  for (l = 0; l < LBASE[pM->NLevels]; l++) {
    { int nl_ = 0; while (LBASE[nl_ + 1] <= l) nl_++; pD = &pM->Domain[nl_][l - LBASE[nl_]]; irefine = 1 << nl_; }
    PG[l] = pD->Grid; G[l] = NULL;
#ifdef AA_MPI
    if (pD->NGrid[0] != 1) ath_error("[athena_amd]: NGrid_x1 = %d: x1 is never cut (the rays travel along it)\n", pD->NGrid[0]);
#ifdef AA_MPISMR
    if (pD->NGrid[1] != 1) ath_error("[athena_amd]: MPI + SMR: NGrid_x2 = %d in <domain%d>: nested levels are cut along x3 only\n", pD->NGrid[1], pD->InputBlock);
    if (pM->DomainsPerLevel[pD->Level] != 1) ath_error("[athena_amd]: MPI + SMR: several Domains on level %d are not cut across ranks\n", pD->Level);
    if (PG[l] == NULL) {                              /* this rank holds no Grid of this Domain (and then of no finer one: checked below) */
      if (l == 0) ath_error("[athena_amd]: this rank holds no Grid of the root Domain\n");
      continue;
    }
#endif
    comm_lev[l] = pD->Comm_Domain;
    MPI_Comm_size(comm_lev[l], &nranks_lev[l]);
    { /* bvals_init (bvals_mhd.c:537-821, which this shim replaces): on a periodic Domain the Grids at either end of a
       * direction are each other's neighbours */
      int L = -1, Mi = -1, Ni = -1, a, b, c;
      for (c = 0; c < pD->NGrid[2]; c++) for (b = 0; b < pD->NGrid[1]; b++) for (a = 0; a < pD->NGrid[0]; a++)
        if (pD->GData[c][b][a].ID_Comm_world == myID_Comm_world) { L = a; Mi = b; Ni = c; }
      if (L < 0) ath_error("[athena_amd]: this rank holds no Grid of the root Domain\n");
      if (pD->Level > 0) { /* (a refined Domain has no periodic wrap of its own) */ }
      else
      if (pM->BCFlag_ix2 == 4 && pM->BCFlag_ox2 == 4 && pD->NGrid[1] > 1) {
        if (Mi == 0 && PG[l]->lx2_id < 0) PG[l]->lx2_id = pD->GData[Ni][pD->NGrid[1] - 1][L].ID_Comm_Domain;
        if (Mi == pD->NGrid[1] - 1 && PG[l]->rx2_id < 0) PG[l]->rx2_id = pD->GData[Ni][0][L].ID_Comm_Domain;
      }
      if (pD->Level == 0 && pM->BCFlag_ix3 == 4 && pM->BCFlag_ox3 == 4 && pD->NGrid[2] > 1) {
        if (Ni == 0 && PG[l]->lx3_id < 0) PG[l]->lx3_id = pD->GData[pD->NGrid[2] - 1][Mi][L].ID_Comm_Domain;
        if (Ni == pD->NGrid[2] - 1 && PG[l]->rx3_id < 0) PG[l]->rx3_id = pD->GData[0][Mi][L].ID_Comm_Domain;
      }
    }
    nb_lev[l][0] = PG[l]->lx2_id; nb_lev[l][1] = PG[l]->rx2_id; nb_lev[l][2] = PG[l]->lx3_id; nb_lev[l][3] = PG[l]->rx3_id;
#else
    if (pD->NGrid[0]*pD->NGrid[1]*pD->NGrid[2] != 1) ath_error("[athena_amd]: one Grid per Domain only\n");
#endif
    memset(&p, 0, sizeof p);
    for (d = 0; d < 3; d++) {
      p.Nx[d] = PG[l]->Nx[d]; p.rootNx[d] = pM->Nx[d];
      p.xmin[d] = pM->RootMinX[d]; p.xmax[d] = pM->RootMaxX[d]; p.MinX[d] = PG[l]->MinX[d];
      disp[3*l + d] = pD->Disp[d];
    }
    p.bc[0] = pM->BCFlag_ix1; p.bc[1] = pM->BCFlag_ox1; p.bc[2] = pM->BCFlag_ix2;
    p.bc[3] = pM->BCFlag_ox2; p.bc[4] = pM->BCFlag_ix3; p.bc[5] = pM->BCFlag_ox3;
    for (d = 0; d < 6; d++)
      if (p.bc[d] != 1 && p.bc[d] != 2 && p.bc[d] != 4) ath_error("[bvals_init]: bc flag = %d unknown\n", p.bc[d]);
    for (d = 0; d < 3; d++) {                      /* bvals_mhd.c:193-361: ProlongateLater on fine/coarse sides */
      if (pD->Disp[d] != 0) p.bc[2*d] = 0;
      if ((pD->Disp[d] + pD->Nx[d])/irefine != pM->Nx[d]) p.bc[2*d + 1] = 0;
    }
#ifdef AA_MPI
    for (d = 0; d < 4; d++) if (nb_lev[l][d] >= 0) p.bc[2 + d] = 0;     /* a neighbour Grid fills these ghost zones */
    p.nslab = 1;                                   /* the reference's ranks ARE the decomposition */
    if (nranks_lev[l] > 1) p.ion_path = 2;         /* the two-kernel sub-cycle: its reductions sit where the reference's are */
#ifdef AA_MPISMR
    p.ion_path = 2;                                /* ... on every level (the levels of a rank's stack use one protocol) */
    /* a Grid inside its Domain: the fine/coarse sides are those of the DOMAIN (above, pD->Disp); an x3 side that is a cut between
     * two Grids of the Domain is filled by the halo exchange */
#endif
#endif
    p.level = pD->Level;
#ifdef AA_SMR
    p.nslab = 1;                                   /* nested levels stay on one GPU */
#endif
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
#ifdef AA_MPI
    if (!env) { const int nd = aa_device_count(); p.device = nd > 0 ? myID_Comm_world % nd : 0; }
#endif
    p.integrator = use_vl() ? 1 : (no_h_correction() ? 2 : 0);
    p.order = recon_order();
    CHK(aa_create(&p, &G[l]));
    ncell[l] = (size_t)(PG[l]->Nx[0] + 2*AA_NGHOST)*(PG[l]->Nx[1] + 2*AA_NGHOST)*(PG[l]->Nx[2] + 2*AA_NGHOST);
    host_newer[l] = 1; snap[l] = NULL;
#if AA_ION_RADPLANE
    if (p.ion) CHK(aa_add_radplane_3d(G[l], pM->radplanelist->dir[0], pM->radplanelist->flux_i));
#endif
    if (StaticGravPot != NULL) CHK(aa_set_static_grav_pot(G[l], StaticGravPot));
    if (CoolingFunc != NULL && !use_vl()) CHK(aa_set_cooling(G[l], AA_COOL_KOYINUT));
    env = getenv("AA_NGPU");
    if (env && atoi(env) > 1 && pM->NLevels == 1)   /* the library cuts the Grid into x3 slabs, one per GPU (csrc/slabs.hip) */
      fprintf(stderr, "[athena_amd] Grid %dx%dx%d in %d slabs along x3 on HIP devices %d.., %.2f GB resident, coherence=%s\n",
              p.Nx[0], p.Nx[1], p.Nx[2], atoi(env), p.device, aa_device_bytes(G[l])/1e9, automode ? "auto" : (learn ? "learn" : "step"));
    else
    fprintf(stderr, "[athena_amd] Grid %dx%dx%d (level %d) on HIP device %d, %.2f GB resident, coherence=%s\n",
            p.Nx[0], p.Nx[1], p.Nx[2], l, p.device, aa_device_bytes(G[l])/1e9, automode ? "auto" : (learn ? "learn" : "step"));
  }
  NL = LBASE[pM->NLevels];
#ifdef AA_MPISMR
  {
    /* this rank's stack: the Grids it holds, root first, each nested in the one before (links as config.mesh_slabs /
     * aa_mesh_create_local want them: the child's overlap on the parent in local parent indices incl. ghost zones, zones per
     * direction, which of its six sides are fine/coarse boundaries of the LEVEL (prolonged) and corrected on this rank) */
    aa_grid *loc[MAXLEV]; int links[21*MAXLEV]; int nloc = 0;
    (void)disp;
    if (learn) { fprintf(stderr, "[athena_amd] MPI + SMR: AA_COHERENCE=step (auto / learn are one-process modes)\n"); learn = 0; automode = 0; }
    for (l = 0; l < NL; l++) {
      if (!HAVE(l)) { LI[l] = -1; continue; }
      if (l > 0 && !HAVE(l - 1)) ath_error("[athena_amd]: MPI + SMR: this rank holds a Grid of level %d but none of level %d\n", l, l - 1);
      LI[l] = nloc; loc[nloc++] = G[l];
    }
    for (l = 0; l + 1 < nloc; l++) {
      const GridS *P = PG[l], *C = PG[l + 1]; const DomainS *cD = &pM->Domain[l + 1][0];
      int *q = links + 21*l; const int ir = 1 << (l + 1);
      for (d = 0; d < 3; d++) {
        const int a = C->Disp[d]/2 - P->Disp[d], b = (C->Disp[d] + C->Nx[d])/2 - P->Disp[d];
        const int glo = cD->Disp[d], ghi = cD->Disp[d] + cD->Nx[d];
        const int at_lo = (C->Disp[d] == glo) && glo != 0;
        const int at_hi = (C->Disp[d] + C->Nx[d] == ghi) && (ghi/ir != pM->Nx[d]);
        if ((C->Disp[d] & 1) || (C->Nx[d] & 1) || a < 0 || b > P->Nx[d])
          ath_error("[athena_amd]: MPI + SMR: the level-%d Grid of this rank (x%d zones %d..%d) does not lie over its level-%d Grid (%d..%d): "
                    "choose NGrid_x3 so that every rank's Grids are nested (DESIGN.md 5)\n", l + 1, d + 1, C->Disp[d]/2, (C->Disp[d] + C->Nx[d])/2,
                    l, P->Disp[d], P->Disp[d] + P->Nx[d]);
        /* a fine/coarse boundary of the level that coincides with a cut of the parent would be corrected on the neighbouring rank
         * (aa_flux_x3_export / _apply): the reference's equal division of every Domain cannot produce it with nested Grids */
        if (d == 2 && ((at_lo && a == 0) || (at_hi && b == P->Nx[d])))
          ath_error("[athena_amd]: MPI + SMR: level %d ends exactly on a cut of level %d along x3: not supported\n", l + 1, l);
        q[d] = a + AA_NGHOST; q[3 + d] = b - a;
        q[6 + 2*d] = at_lo; q[6 + 2*d + 1] = at_hi; q[12 + 2*d] = at_lo; q[12 + 2*d + 1] = at_hi;
        q[18 + d] = C->Disp[d] - 2*P->Disp[d];
      }
    }
    NLOC = nloc;
    CHK(aa_mesh_create_local(nloc, loc, links, &MM));
  }
#elif defined(AA_SMR)
  CHK(aa_mesh_create(NL, G, disp, &MM));
#else
  (void)disp;
#endif
}

static void to_device(int l)
{
  if (!HAVE(l)) return;
  if (host_newer[l]) { CHK(aa_upload_cons(G[l], host_block(l))); host_newer[l] = 0; active_same[l] = 1; }
  CHK(aa_set_mesh_state(G[l], M->time, PG[l]->dt, M->nstep));
}

/* ghosts_only: the caller is refresh_for_output, i.e. only boundary calls have run on the device since the block last
 * travelled (active_same): then the ghost shell is all that differs */
static void to_host_x(int l, int ghosts_only)
{
  static int shell_ok = -1;
  if (!HAVE(l)) return;      /* AA_GHOST_REFRESH=0: always the whole block (A/B measurements) */
  if (shell_ok < 0) { const char *e = getenv("AA_GHOST_REFRESH"); shell_ok = !(e && atoi(e) == 0); }
  if (shell_ok && ghosts_only && active_same[l] && !host_newer[l]) CHK(aa_download_ghost_zones(G[l], host_block(l)));
  else CHK(aa_download_cons(G[l], host_block(l)));
  active_same[l] = 1;
#if AA_ION_RADPLANE
  if (M->radplanelist != NULL && M->radplanelist->nradplane > 0) CHK(aa_download_edgeflux(G[l], host_edgeflux(l)));
#endif
}

static void to_host(int l) { to_host_x(l, 0); }

/* after the integrator (and, with SMR, RestrictCorrect): Userwork_in_loop reads and may write pG->U */
static void refresh_for_userwork(int l)
{
  if (!HAVE(l)) return;
  if (learn && learned && automode && l == 0) {        /* is this a step on which the imprint is looked at again? */
    verify_now = verify_next || (++steps_since_reval >= reval_every);
    verify_next = 0;
  }
  if (learn && learned && !(automode && verify_now)) { CHK(aa_apply_pinned_cells(G[l])); active_same[l] = 0; return; }
  to_host(l);
  host_newer[l] = 1;
  if (learn) {
    if (!snap[l]) snap[l] = (double*)malloc(ncell[l]*sizeof(ConsS));
    if (!snap[l]) ath_error("[athena_amd]: out of host memory for the snapshot of the host block\n");
    memcpy(snap[l], host_block(l), ncell[l]*sizeof(ConsS));
  }
}

/* ---- reconstruction / integrator ---------------------------------------------------------- */
void lr_states_init(MeshS *pM) { (void)pM; }
void lr_states_destruct(void) {}

static void integrate_3d_amd(DomainS *pD)
{
  const int l = GI(pD);
  to_device(l);
  active_same[l] = 0;
  if (use_vl()) CHK(aa_integrate_3d_vl(G[l])); else CHK(aa_integrate_3d_ctu(G[l]));
  integrated = 1;
  if (NL == 1) refresh_for_userwork(0);         /* with SMR, RestrictCorrect (main.c:591) still follows */
}

VDFun_t integrate_init(MeshS *pM)
{
  if (CourNo > 0.5)     /* integrate.c:66-68 */
    ath_error("<time>cour_no was set to %g: must be <= 0.5 with 3D integrator\n", CourNo);
  ensure_grid(pM);
  return integrate_3d_amd;
}

void integrate_destruct(void)
{
  int l;
#ifdef AA_SMR
  if (MM) { aa_mesh_destroy(MM); MM = NULL; }
#endif
  for (l = 0; l < NL; l++) { if (G[l]) { aa_destroy(G[l]); G[l] = NULL; } free(snap[l]); snap[l] = NULL; }
#ifdef AA_MPI
  { int a, b, c; for (l = 0; l < MAXLEV; l++) for (a = 0; a < 2; a++) for (b = 0; b < 2; b++) for (c = 0; c < 2; c++) { free(hbuf[l][a][b][c]); hbuf[l][a][b][c] = NULL; } }
#endif
  NL = 0;
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

static void on_term(int sgn)
{
  term_seen = 1;
  if (old_term != SIG_DFL && old_term != SIG_IGN && old_term != on_term) (*old_term)(sgn);     /* ath_signal.c's handler */
}

/* AA_COHERENCE=auto: will main() read the host blocks before the next step?  data_output at the top of the next cycle
 * (an <outputN> block is due: output.c:507-522), the forced final output when the loop ends (main.c:519, :743), SIGTERM */
static int host_read_due(void)
{
  int n, due = 0;
  for (n = 0; n < nout; n++) if (M->time >= out_t[n]) { out_t[n] += out_dt[n]; due = 1; }
  if (M->time >= tlim_ || (nlim_ >= 0 && M->nstep >= nlim_)) due = 1;
  if (term_seen) due = 1;
  return due;
}

/* data_output() at the top of the next cycle reads the host blocks */
static void refresh_for_output(void)
{
  int l, due;
  after_new_dt = 0;
  if (!(learn && learned)) due = 1;
  else if (automode) due = host_read_due();
  else due = (++steps_since_sync >= sync_every);
  if (due) { for (l = 0; l < NL; l++) to_host_x(l, 1); steps_since_sync = 0; if (automode && learned) verify_next = 1; }
}

#ifdef AA_MPI
/* bvals_mhd.c:296-493 for one direction (dir 1 = x2, 2 = x3): the four planes either side travel between neighbouring Grids.
 * Tag = the way the message travels (down / up), so that two Grids that are each other's neighbour on both sides (periodic
 * Domain cut in two) keep their messages apart. */
static void halo_exchange(int l, int dir)
{
  const int lo = nb_lev[l][2*(dir - 1)], hi = nb_lev[l][2*(dir - 1) + 1];
  const long long n = aa_halo_doubles_dir(G[l], dir);
  MPI_Request rq[4]; MPI_Status st[4]; int nrq = 0, side, w;
  double *(*hb)[2] = hbuf[l][dir - 1];
  if (lo < 0 && hi < 0) return;
  for (side = 0; side < 2; side++) for (w = 0; w < 2; w++)
    if (!hb[side][w] && !(hb[side][w] = (double*)malloc((size_t)n*sizeof(double))))
      ath_error("[athena_amd]: out of host memory for the MPI halo buffers\n");
  if (lo >= 0) MPI_Irecv(hb[0][1], (int)n, MPI_DOUBLE, lo, 100 + 2*dir + 1, comm_lev[l], &rq[nrq++]);   /* travelled up   */
  if (hi >= 0) MPI_Irecv(hb[1][1], (int)n, MPI_DOUBLE, hi, 100 + 2*dir,     comm_lev[l], &rq[nrq++]);   /* travelled down */
  if (lo >= 0) { CHK(aa_halo_get(G[l], dir, 0, hb[0][0])); MPI_Isend(hb[0][0], (int)n, MPI_DOUBLE, lo, 100 + 2*dir, comm_lev[l], &rq[nrq++]); }
  if (hi >= 0) { CHK(aa_halo_get(G[l], dir, 1, hb[1][0])); MPI_Isend(hb[1][0], (int)n, MPI_DOUBLE, hi, 100 + 2*dir + 1, comm_lev[l], &rq[nrq++]); }
  MPI_Waitall(nrq, rq, st);
  if (lo >= 0) CHK(aa_halo_put(G[l], dir, 0, hb[0][1]));
  if (hi >= 0) CHK(aa_halo_put(G[l], dir, 1, hb[1][1]));
}
#endif

void bvals_mhd(DomainS *pD)
{
  VGFun_t usr[6]; int d, side, any = 0; const int l = GI(pD);
  ensure_grid(M0);
  if (!HAVE(l)) return;
  to_device(l);
  usr[0] = pD->ix1_BCFun; usr[1] = pD->ox1_BCFun; usr[2] = pD->ix2_BCFun;
  usr[3] = pD->ox2_BCFun; usr[4] = pD->ix3_BCFun; usr[5] = pD->ox3_BCFun;
  for (d = 0; d < 6; d++) any |= (usr[d] != NULL);
#ifdef AA_MPI
  any = 1;                      /* side by side: the exchanges sit between the directions */
#endif
  if (!any) CHK(aa_bvals_mhd(G[l]));
  else {
    /* bvals_mhd.c:196-420: ix1, ox1, ix2, ox2, ix3, ox3 in this order; a user function sees the host
     * block with everything filled so far and its ghost zones travel back before the next side */
    for (d = 0; d < 3; d++) {
      for (side = 0; side < 2; side++) {
#ifdef AA_MPI
        if (d > 0 && nb_lev[l][2*(d - 1) + side] >= 0) continue;     /* not a physical side of this Grid */
#endif
        if (usr[2*d + side] == NULL) { CHK(aa_bvals_mhd_side(G[l], d, side)); continue; }
        CHK(aa_download_cons(G[l], host_block(l)));
        (*usr[2*d + side])(PG[l]);
        CHK(aa_upload_cons(G[l], host_block(l)));
      }
#ifdef AA_MPI
      if (d > 0) halo_exchange(l, d);
#endif
    }
  }
  /* main.c calls bvals_mhd after the ion step (:552; nothing on the host looks at U before
   * Integrate) and after new_dt (:638): only the latter refreshes the host view -- with SMR the
   * refresh waits for Prolongate (:647), which still writes the fine ghost zones */
  if (after_new_dt && NL == 1) refresh_for_output();
}

void new_dt(MeshS *pM)
{
  int nl, nd, l; double t, dt; int n;
  ensure_grid(pM);
  if (learn && learned && automode && verify_now && snap[0]) {
    /* re-validation: the host block after Userwork_in_loop must be the downloaded state with the recorded values
     * imposed -- nothing more (a Userwork that woke up), nothing less, nothing else (values that drift) */
    const int nv = 5 + AA_NSCALARS;
    int ok = 1;
    for (l = 0; l < NL && ok; l++) {
      long long c; int v;
      for (c = 0; c < pin_n[l]; c++) for (v = 0; v < nv; v++) snap[l][pin_idx[l][c]*nv + v] = pin_val[l][c*nv + v];
      if (memcmp(host_block(l), snap[l], ncell[l]*nv*sizeof(double)) != 0) ok = 0;
    }
    for (l = 0; l < NL; l++) { free(snap[l]); snap[l] = NULL; }
    verify_now = 0; steps_since_reval = 0;
    if (ok) {
      for (l = 0; l < NL; l++) { CHK(aa_apply_pinned_cells(G[l])); active_same[l] = 0; }   /* this step's Userwork, on the device */
    } else {
      /* the host blocks hold what Userwork_in_loop really did on this step's state: they travel back (host_newer is set),
       * and from here on every step does */
      learn = 0; learned = 0; gave_up = 1;
#ifndef AA_SMR
      CHK(aa_cfl_in_update(G[0], 0));
#endif
      fprintf(stderr, "[athena_amd] WARNING: Userwork_in_loop no longer writes the imprint recorded at the start of the run "
                      "(step %d): coherence falls back to `step` for the rest of the run; its writes since the last "
                      "validation (at most %d steps) did not reach the device\n", pM->nstep, reval_every);
    }
  }
  if (learn && !learned && !gave_up && snap[0]) {           /* what did Userwork_in_loop change? */
    const int nv = 5 + AA_NSCALARS;
    int same = 1;
    long long cnt_l[MAXLEV]; long long *idx_l[MAXLEV]; double *val_l[MAXLEV];
    for (l = 0; l < NL; l++) {
      const double *h = host_block(l); long long cnt = 0, c; size_t i; int v;
      for (i = 0; i < ncell[l]; i++) if (memcmp(h + i*nv, snap[l] + i*nv, nv*sizeof(double)) != 0) cnt++;
      idx_l[l] = (long long*)malloc((size_t)(cnt + 1)*sizeof(long long)); val_l[l] = (double*)malloc((size_t)(cnt + 1)*nv*sizeof(double));
      if (idx_l[l] == NULL || val_l[l] == NULL) ath_error("[athena_amd]: out of host memory for the %lld zones Userwork_in_loop wrote\n", cnt);
      for (i = 0, c = 0; i < ncell[l]; i++) if (memcmp(h + i*nv, snap[l] + i*nv, nv*sizeof(double)) != 0) {
        idx_l[l][c] = (long long)i; for (v = 0; v < nv; v++) val_l[l][c*nv + v] = h[i*nv + v]; c++;
      }
      cnt_l[l] = cnt;
      free(snap[l]); snap[l] = NULL;
      /* auto: trusted only when two consecutive steps wrote the same values into the same zones */
      if (automode) {
        if (iw_prev[l] == NULL) same = 0;
        else if (nw_prev[l] != cnt || memcmp(iw_prev[l], idx_l[l], (size_t)cnt*sizeof(long long)) != 0 ||
                 memcmp(vw_prev[l], val_l[l], (size_t)cnt*nv*sizeof(double)) != 0) same = -1;
      }
    }
    if (automode && same == 0) {               /* first step: remember, look again after the next one */
      for (l = 0; l < NL; l++) { nw_prev[l] = cnt_l[l]; iw_prev[l] = idx_l[l]; vw_prev[l] = val_l[l]; }
    } else if (automode && same < 0) {          /* Userwork_in_loop is not a fixed imprint: keep the host in the loop every step */
      for (l = 0; l < NL; l++) { free(idx_l[l]); free(val_l[l]); free(iw_prev[l]); free(vw_prev[l]); iw_prev[l] = NULL; vw_prev[l] = NULL; }
      gave_up = 1; learn = 0;
      fprintf(stderr, "[athena_amd] Userwork_in_loop writes differ from step to step: coherence stays `step`\n");
    } else {
      for (l = 0; l < NL; l++) {
        CHK(aa_set_pinned_cells(G[l], cnt_l[l], idx_l[l], val_l[l]));
        CHK(aa_apply_pinned_cells(G[l])); active_same[l] = 0;   /* this step's Userwork, on the device */
        fprintf(stderr, "[athena_amd] Userwork_in_loop pins %lld cells on level %d; re-imposed on the device from now on\n", cnt_l[l], l);
        if (automode) { pin_n[l] = cnt_l[l]; pin_idx[l] = idx_l[l]; pin_val[l] = val_l[l]; }     /* kept for re-validation */
        else { free(idx_l[l]); free(val_l[l]); }
        if (iw_prev[l]) { free(iw_prev[l]); free(vw_prev[l]); iw_prev[l] = NULL; vw_prev[l] = NULL; }
      }
      learned = 1;
#ifndef AA_SMR
      /* from now on nothing but the pinned zones changes between Integrate() and new_dt(): new_dt's maxima can come
       * out of the update kernel (AA_CFL_FUSED=0 keeps the separate sweep) */
      { const char *e = getenv("AA_CFL_FUSED"); CHK(aa_cfl_in_update(G[0], !(e && atoi(e) == 0))); }
#endif
      if (automode && !term_hooked) {              /* from here on the host block is only as fresh as host_read_due() makes it */
        void (*h)(int) = signal(SIGTERM, on_term);
        if (h != on_term) old_term = h;
        term_hooked = 1;
      }
    }
  }
  for (l = 0; l < NL; l++) {
    if (learn && learned) host_newer[l] = 0;    /* the stale host copy must not travel back */
    to_device(l);
  }
#ifdef AA_MPISMR
  { /* new_dt.c:32-185: max(|v| + a) per direction carried from Grid to Grid of THIS rank (:33: max_v1.. are set to zero once,
     * before the loop over levels), (cum / dx_level) of every level, then the MIN over all ranks (:177) */
    double cum[3] = {0.0, 0.0, 0.0}, max_dti = 0.0, mine, all, v[3];
    for (l = 0; l < NL; l++) {
      if (!HAVE(l)) continue;
      CHK(aa_cfl_max_v(G[l], v));
      { const double dxl[3] = {PG[l]->dx1, PG[l]->dx2, PG[l]->dx3}; int d_;
        for (d_ = 0; d_ < 3; d_++) { cum[d_] = (cum[d_] > v[d_]) ? cum[d_] : v[d_]; }
        for (d_ = 0; d_ < 3; d_++) { const double q_ = cum[d_]/dxl[d_]; max_dti = (max_dti > q_) ? max_dti : q_; } }
    }
    mine = CourNo/max_dti;
    mine = (pM->nstep == 0) ? mine : ((2.0*pM->dt < mine) ? 2.0*pM->dt : mine);       /* :169-173 before the reduction (:177) */
    MPI_Allreduce(&mine, &all, 1, MPI_DOUBLE, MPI_MIN, MPI_COMM_WORLD);
    dt = all; t = pM->time; n = pM->nstep;
    { const double tlim = par_getd("time", "tlim"); if (t < tlim && (tlim - t) < dt) dt = tlim - t; }
    for (l = 0; l < NL; l++) if (HAVE(l)) CHK(aa_set_mesh_state(G[l], t, dt, n));
  }
#elif defined(AA_SMR)
  CHK(aa_mesh_set_state(MM, pM->time, pM->dt, pM->nstep));
  CHK(aa_mesh_new_dt(MM));
  CHK(aa_mesh_get_state(MM, &t, &dt, &n));
#elif defined(AA_MPI)
  { double mine, all;                                /* new_dt.c:159-185 with the MIN over the Grids of the Domain (:177) */
    CHK(aa_new_dt_local(G[0], &mine));
    MPI_Allreduce(&mine, &all, 1, MPI_DOUBLE, MPI_MIN, MPI_COMM_WORLD);
    dt = (pM->nstep == 0) ? all : ((2.0*pM->dt < all) ? 2.0*pM->dt : all);
    t = pM->time; n = pM->nstep;
    { const double tlim = par_getd("time", "tlim"); if (t < tlim && (tlim - t) < dt) dt = tlim - t; }
    CHK(aa_set_mesh_state(G[0], t, dt, n)); }
#else
  CHK(aa_new_dt(G[0]));
  CHK(aa_get_mesh_state(G[0], &t, &dt, &n));
#endif
  pM->dt = dt;
  after_new_dt = 1;
  for (nl = 0; nl < pM->NLevels; nl++) for (nd = 0; nd < pM->DomainsPerLevel[nl]; nd++)
    if (pM->Domain[nl][nd].Grid != NULL) pM->Domain[nl][nd].Grid->dt = dt;      /* new_dt.c:189-195 */
}

#ifdef AA_SMR
/* ---- static mesh refinement (smr.c) ------------------------------------------------------- */
void SMR_init(MeshS *pM) { ensure_grid(pM); }          /* main.c:400, after problem() on every level */

void RestrictCorrect(MeshS *pM)                        /* main.c:401, :591 */
{
  int l;
  ensure_grid(pM);
  for (l = 0; l < NL; l++) { to_device(l); active_same[l] = 0; }
  CHK(aa_mesh_restrict_correct(MM));      /* (MPI + SMR: the pairs of this rank's own stack; nothing crosses a cut, see ensure_grid) */
  if (integrated) { integrated = 0; for (l = 0; l < NL; l++) refresh_for_userwork(l); }
}

void Prolongate(MeshS *pM)                             /* main.c:446, :647 */
{
  int l;
  ensure_grid(pM);
  for (l = 0; l < NL; l++) to_device(l);
  CHK(aa_mesh_prolongate(MM));
  /* :647 ends the cycle (data_output follows at the top of the next one); :446 precedes the very
   * first output, which must already see the restricted parent zones */
  if (after_new_dt || pM->nstep == 0) refresh_for_output();
}

void ionradRestrictCorrect(MeshS *pM)                  /* main.c:561 */
{
  int l;
  (void)pM;
  for (l = 0; l < NL; l++) active_same[l] = 0;
  CHK(aa_mesh_ionrad_restrict_correct(MM));
}
#endif /* AA_SMR */

/* ---- ion radiation ------------------------------------------------------------------------ */
void ion_radtransfer_init_domain(MeshS *pM) { (void)pM; }

#if AA_ION_RADPLANE
static void ion_radtransfer_3d_amd(DomainS *pD)
{
  GridS *pG = pD->Grid; MeshS *pM = pD->Mesh; int niter = 0; double t, dt; int n; const int l = GI(pD);
  to_device(l);
  active_same[l] = 0;
#ifdef AA_MPISMR
  {
    /* ionrad_3d.c:862-1047 on the Grid this rank holds of level pD->Level, with the reference's own reductions in the Domain's
     * communicator: MIN of dt_chem / dt_therm behind the rates (:399, :554) on every level; on the root also SUM of the cells out
     * of range and MIN of dt_hydro behind the update (:275, :672).  A refined level takes its incoming flux from this rank's own
     * parent Grid (ionrad_smr.c:34, :345: the rays run along x1, which is never cut) and sub-cycles to the time the root covered. */
    const int fine = (pD->Level != 0);
    double dt_done = 0.0, hdt = pG->dt; int stop = 0;
    if (!HAVE(l)) return;
    if (fine) CHK(aa_mesh_ionflux_prolong(MM, LI[l])); else tcoarse_ = 0.0;
    CHK(aa_set_mesh_state(G[l], pM->time, pG->dt, pM->nstep));
    CHK(aa_ion_begin(G[l]));
    while (fine || !stop) {
      double a[2], b[2], dts, dth, dth_all; long long cnt; long cl, cl_all;
      CHK(aa_ion_rates(G[l], &a[0], &a[1]));
      MPI_Allreduce(a, b, 2, MPI_DOUBLE, MPI_MIN, comm_lev[l]);
      dts = (b[1] < b[0]) ? b[1] : b[0];
      if (!fine) { if (dt_done + dts > hdt) { dts = hdt - dt_done; stop = 1; } }
      else if (dt_done + dts > tcoarse_) { dts = tcoarse_ - dt_done; stop = 1; }
      CHK(aa_ion_update(G[l], dts, &cnt, &dth));
      dt_done += dts; niter++;
      if (!fine) {
        cl = (long)cnt;
        MPI_Allreduce(&cl, &cl_all, 1, MPI_LONG, MPI_SUM, comm_lev[l]);
        if (cl_all > 20) { hdt = dt_done; break; }                 /* MAXCELLCOUNT, ionrad.h:38 */
        if (stop) break;
        MPI_Allreduce(&dth, &dth_all, 1, MPI_DOUBLE, MPI_MIN, comm_lev[l]);
        if (dth_all < dt_done) { hdt = dt_done; break; }
      } else if (stop) { hdt = dt_done; break; }
    }
    if (!fine) {
      if (niter == (int)par_getd("ionradiation", "maxiter")) hdt = dt_done;
      tcoarse_ = dt_done;
      if (hdt < 0) ath_error("[ion_radtransfer_3d]: dt = %e, dt_done = %e\n", hdt, dt_done);
    }
    CHK(aa_set_mesh_state(G[l], pM->time, hdt, pM->nstep));
  }
#elif defined(AA_SMR)
  CHK(aa_mesh_set_state(MM, pM->time, pM->dt, pM->nstep));
  CHK(aa_mesh_ion_radtransfer(MM, l, &niter));
#elif defined(AA_MPI)
  if (nranks_dom > 1) {
    /* ionrad_3d.c:862-1047 with the reference's own reductions: MIN of dt_chem / dt_therm behind the rates (:275, :399),
     * SUM of the cells out of range and MIN of dt_hydro behind the update (:554, :672); MAXCELLCOUNT ionrad.h:38 */
    double dt_done = 0.0, hdt = pG->dt; int hydro_done = 0;
    CHK(aa_ion_begin(G[l]));
    while (!hydro_done) {
      double a[2], b[2], dts, dth, dth_all; long long cnt; long cl, cl_all;
      CHK(aa_ion_rates(G[l], &a[0], &a[1]));
      MPI_Allreduce(a, b, 2, MPI_DOUBLE, MPI_MIN, comm_dom);
      dts = (b[1] < b[0]) ? b[1] : b[0];
      if (dt_done + dts > hdt) { dts = hdt - dt_done; hydro_done = 1; }
      CHK(aa_ion_update(G[l], dts, &cnt, &dth));
      cl = (long)cnt;
      MPI_Allreduce(&cl, &cl_all, 1, MPI_LONG, MPI_SUM, comm_dom);
      MPI_Allreduce(&dth, &dth_all, 1, MPI_DOUBLE, MPI_MIN, comm_dom);
      dt_done += dts; niter++;
      if (cl_all > 20) { hdt = dt_done; break; }
      if (hydro_done) break;
      if (dth_all < dt_done) { hdt = dt_done; break; }
    }
    if (niter == (int)par_getd("ionradiation", "maxiter")) hdt = dt_done;
    if (hdt < 0) ath_error("[ion_radtransfer_3d]: dt = %e, dt_done = %e\n", hdt, dt_done);
    CHK(aa_set_mesh_state(G[l], pM->time, hdt, pM->nstep));
  } else CHK(aa_ion_radtransfer_3d(G[l], &niter));
#else
  CHK(aa_ion_radtransfer_3d(G[l], &niter));
#endif
  CHK(aa_get_mesh_state(G[l], &t, &dt, &n));
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
void bvals_ionrad(DomainS *pD) { ensure_grid(pD->Mesh); if (HAVE(GI(pD))) CHK(aa_bvals_ionrad(G[GI(pD)])); }
void set_coarse_time(void) {}
void clear_coarse_time(void) {}

void add_radplane_3d(GridS *pGrid, int dir, Real flux)   /* ionradplane_3d.c:56-66 */
{
  MeshS *pMesh = pGrid->Mesh; int l;
  pMesh->radplanelist->dir[0] = dir;
  pMesh->radplanelist->flux_i = flux;
  for (l = 0; l < NL; l++) if (HAVE(l)) CHK(aa_add_radplane_3d(G[l], dir, flux));
}
#endif /* AA_ION_RADPLANE */
