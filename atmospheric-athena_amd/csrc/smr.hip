// smr.hip -- static mesh refinement: the nested levels of a Mesh resident on one GPU.
//
// Every level is an ordinary aa_grid (its own SoA pool, dx = root dx / 2^level).  This unit adds
// what the reference adds between the levels, as kernels that read one level's arrays and write
// the other's directly in HBM -- there are no send/receive buffers (smr.c packs them only to
// feed MPI):
//   restriction + flux correction   smr.c:1207  RestrictCorrect
//   restriction of E and s[0]       smr.c:85    ionradRestrictCorrect
//   prolongation into ghost zones   smr.c:2359  Prolongate, :3068 ProCon, :3478 mcd_slope
//   radiation hand-off to the child ionrad_smr.c:345/:34  ionrad_prolong_snd / _rcv
// and the host control flow of the SMR branches of main.c:519-669, new_dt.c:32 and
// ionrad_3d.c:862.  The fluxes the reference copies into CGrid/PGrid.myFlx in Step 12e of the
// integrator (integrate_3d_ctu.c:3072) are read straight from each level's flux arrays, which
// keep the second-pass fluxes until the next integrator call.
//
// Summation orders follow the reference term by term, so the strict (-ffp-contract=off) build
// reproduces the CPU results bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "api_internal.h"

using namespace aa;

#define NG AA_NGHOST
#define AA_MAXLEV 16     /* Grids of a Mesh (one per Domain) */

namespace {

struct Link {                 // level l+1 seen from level l (init_grid.c: CGrid.ijks/ijke, myFlx != NULL)
  int cs[3], ce[3];           // overlap on the parent, parent indices incl. ghost offset
  int n[3];                   // overlap size in parent zones
  int prol[6];                // the child has a fine/coarse boundary on this side: ghost zones prolonged
  int corr[6];                // ... and the parent zone outside is on this Grid: flux-corrected here
  int cdisp[3];               // child origin minus 2 x parent origin, zones of the child's level
};

__device__ __forceinline__ Real *fld(const DevGrid &g, Real *base, int v) { return base + (long)v*g.nc; }

// ---- restriction: 2x2x2 fine zones -> one parent zone, summed as smr.c:1391-1458 -------------
// varmask bit v = restrict variable v (RestrictCorrect: all; ionradRestrictCorrect: E and s0)
__global__ void __launch_bounds__(256)
k_restrict(DevGrid f, DevGrid c, Link L, unsigned varmask)
{
  const long n = (long)L.n[0]*L.n[1]*L.n[2];
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= n) return;
  const int a = (int)(lin % L.n[0]), b = (int)((lin / L.n[0]) % L.n[1]), cc = (int)(lin / ((long)L.n[0]*L.n[1]));
  const long mc = (long)(L.cs[2] + cc)*c.sK + (long)(L.cs[1] + b)*c.sJ + (L.cs[0] + a);
  const long mf = (long)(f.ks + 2*cc)*f.sK + (long)(f.js + 2*b)*f.sJ + (f.is + 2*a);
  for (int v = 0; v < 6; v++) {
    if (!(varmask >> v & 1u)) continue;
    const Real *q = fld(f, f.U, v) + mf;
    Real s = q[0] + q[1];
    s += q[f.sJ] + q[f.sJ + 1];
    s += q[f.sK] + q[f.sK + 1] + q[f.sK + f.sJ] + q[f.sK + f.sJ + 1];
    s *= 0.125;
    fld(c, c.U, v)[mc] = s;
  }
}

// ---- flux correction (smr.c:1277-1340) on the parent zones just outside the child, one side per
// launch; the child's flux through a parent face is the average of its 2x2 faces (:1464-1640) ----
__global__ void __launch_bounds__(256)
k_flux_correct(DevGrid f, DevGrid c, Link L, int dim_arg, int nvar, Real dt)
{
  // dim_arg < 0: all sides in one launch, side = blockIdx.y (the sides correct disjoint parent zones: the zone outside each face)
  const int dim = (dim_arg >= 0) ? dim_arg : (int)blockIdx.y;
  if (dim_arg < 0 && !L.corr[dim]) return;
  const int d = dim >> 1, d1 = (d == 0) ? 1 : 0, d2 = (d == 2) ? 1 : 2;      // fast, slow transverse
  const long n = (long)L.n[d1]*L.n[d2];
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= n) return;
  const int a = (int)(lin % L.n[d1]), b = (int)(lin / L.n[d1]);
  const long sc[3] = {1, c.sJ, c.sK}, sf[3] = {1, f.sJ, f.sK};
  const int flo[3] = {f.is, f.js, f.ks}, fhi[3] = {f.ie, f.je, f.ke};
  int ic[3]; ic[d1] = L.cs[d1] + a; ic[d2] = L.cs[d2] + b;
  Real q;
  long mface_c, mcell, mface_f;
  if (dim & 1) { ic[d] = L.ce[d] + 1; q =  (dt/c.dx[d]); }
  else         { ic[d] = L.cs[d] - 1; q = -(dt/c.dx[d]); }
  mcell = ic[2]*sc[2] + ic[1]*sc[1] + ic[0];
  ic[d] = (dim & 1) ? L.ce[d] + 1 : L.cs[d];
  mface_c = ic[2]*sc[2] + ic[1]*sc[1] + ic[0];
  {
    int jf[3]; jf[d1] = flo[d1] + 2*a; jf[d2] = flo[d2] + 2*b; jf[d] = (dim & 1) ? fhi[d] + 1 : flo[d];
    mface_f = jf[2]*sf[2] + jf[1]*sf[1] + jf[0];
  }
  for (int v = 0; v < nvar; v++) {
    const Real mine = fld(c, c.F, d*6 + v)[mface_c];
    const Real *qf = fld(f, f.F, d*6 + v) + mface_f;
    Real fine = qf[0] + qf[sf[d1]];
    fine += qf[sf[d2]] + qf[sf[d2] + sf[d1]];
    fine *= 0.25;
    fld(c, c.U, v)[mcell] -= q*(mine - fine);
  }
}

// ---- prolongation ----------------------------------------------------------------------------
// snapshot of the parent zones cs-3 .. ce+3 around the child, taken when the parent "sends"
// (smr.c:2397-2470: before the parent's own ghost zones are refreshed in the same call)
__global__ void __launch_bounds__(256)
k_box_copy(DevGrid c, Link L, Real *box)
{
  const int b0 = L.n[0] + 6, b1 = L.n[1] + 6, b2 = L.n[2] + 6;
  const long nb = (long)b0*b1*b2;
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= nb) return;
  const int i = (int)(lin % b0), j = (int)((lin / b0) % b1), k = (int)(lin / ((long)b0*b1));
  const long m = (long)(L.cs[2] - 3 + k)*c.sK + (long)(L.cs[1] - 3 + j)*c.sJ + (L.cs[0] - 3 + i);
  for (int v = 0; v < 6; v++) box[(long)v*nb + lin] = fld(c, c.U, v)[m];
}

__device__ __forceinline__ Real mcd_slope(const Real vl, const Real vc, const Real vr)   // smr.c:3478
{
  const Real dvl = (vc - vl), dvr = (vr - vc);
  if (dvl > 0.0 && dvr > 0.0) {
    const Real dv = 2.0*(dvl < dvr ? dvl : dvr), dvm = 0.5*(dvl + dvr);
    return (dvm < dv ? dvm : dv);
  } else if (dvl < 0.0 && dvr < 0.0) {
    const Real dv = 2.0*(dvl > dvr ? dvl : dvr), dvm = 0.5*(dvl + dvr);
    return (dvm > dv ? dvm : dv);
  }
  return 0.0;
}

// one thread = one parent zone under 2x2x2 ghost zones of the child (ProCon, smr.c:3068); one
// launch per boundary side, in the reference's order of sides (regions overlap at edges and
// corners with identical values)
__global__ void __launch_bounds__(128)
k_prolong(DevGrid f, Link L, const Real *box, int dim_arg, int nvar)
{
  // dim_arg < 0: all sides in one launch, side = blockIdx.y (where two sides' regions overlap, at edges and corners, both write the
  // same values: every ghost zone is a function of the snapshot `box` alone)
  const int dim = (dim_arg >= 0) ? dim_arg : (int)blockIdx.y;
  if (dim_arg < 0 && !L.prol[dim]) return;
  const int lo[3] = {f.is, f.js, f.ks}, hi[3] = {f.ie, f.je, f.ke};
  int ps[3], cnt[3];
  for (int d = 0; d < 3; d++) { ps[d] = lo[d] - NG; cnt[d] = (hi[d] - lo[d] + 1 + 2*NG)/2; }
  if (dim & 1) ps[dim >> 1] = hi[dim >> 1] + 1;
  cnt[dim >> 1] = NG/2;
  const long n = (long)cnt[0]*cnt[1]*cnt[2];
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= n) return;
  const int a = (int)(lin % cnt[0]), b = (int)((lin / cnt[0]) % cnt[1]), cc = (int)(lin / ((long)cnt[0]*cnt[1]));
  const int i = ps[0] + 2*a, j = ps[1] + 2*b, k = ps[2] + 2*cc;              // lower fine zone of the 2x2x2 block
  const int b0 = L.n[0] + 6, b1 = L.n[1] + 6;
  const long nb = (long)b0*b1*(L.n[2] + 6);
  const long sb1 = b0, sb2 = (long)b0*b1;
  // parent zone in box coordinates: fine lo-4 lies in parent cs-2 = box index 1
  const long mb = (long)((k - (lo[2] - NG))/2 + 1)*sb2 + (long)((j - (lo[1] - NG))/2 + 1)*sb1 + ((i - (lo[0] - NG))/2 + 1);
  Real P[6][8];
  // d, M1, M2, M3, s0: conserved variable + limited slopes
  for (int v = 0; v < 6; v++) {
    if (v == 4 || v >= nvar) continue;
    const Real *q = box + (long)v*nb + mb;
    const Real uc = q[0];
    const Real dq1 = mcd_slope(q[-1], uc, q[1]);
    const Real dq2 = mcd_slope(q[-sb1], uc, q[sb1]);
    const Real dq3 = mcd_slope(q[-sb2], uc, q[sb2]);
    for (int kk = 0; kk < 2; kk++) for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 2; ii++)
      P[v][kk*4 + jj*2 + ii] = uc + (0.5*ii - 0.25)*dq1 + (0.5*jj - 0.25)*dq2 + (0.5*kk - 0.25)*dq3;
  }
  {  // the internal energy, not E, is interpolated (:3146-3168)
    const Real *qd = box + mb, *q1 = box + nb + mb, *q2 = box + 2*nb + mb, *q3 = box + 3*nb + mb, *qe = box + 4*nb + mb;
#define EINT(o) (qe[o] - 0.5*(q1[o]*q1[o] + q2[o]*q2[o] + q3[o]*q3[o])/qd[o])
    const Real Pi = EINT(0);
    const Real dq1 = mcd_slope(EINT(-1), Pi, EINT(1));
    const Real dq2 = mcd_slope(EINT(-sb1), Pi, EINT(sb1));
    const Real dq3 = mcd_slope(EINT(-sb2), Pi, EINT(sb2));
#undef EINT
    for (int kk = 0; kk < 2; kk++) for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 2; ii++) {
      const int o = kk*4 + jj*2 + ii;
      Real e = Pi + (0.5*ii - 0.25)*dq1 + (0.5*jj - 0.25)*dq2 + (0.5*kk - 0.25)*dq3;
      e += 0.5*(P[1][o]*P[1][o] + P[2][o]*P[2][o] + P[3][o]*P[3][o])/P[0][o];
      P[4][o] = e;
    }
  }
  for (int v = 0; v < nvar; v++) {
    Real *u = fld(f, f.U, v);
    for (int kk = 0; kk < 2; kk++) for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 2; ii++)
      u[(long)(k + kk)*f.sK + (long)(j + jj)*f.sJ + (i + ii)] = P[v][kk*4 + jj*2 + ii];
  }
}

// ---- radiation hand-off (ionrad_smr.c:345 + :34, rays along +x1): the flux the parent left at
// the upstream face of the child, copied piecewise-constant onto the child's 2x2 rays.  The
// reference moves it through CGrid.ionFlx; nothing touches the parent's EdgeFlux in between. ----
__global__ void __launch_bounds__(256)
k_ionflux_prolong(DevGrid f, DevGrid c, Link L)
{
  const int w = L.n[1] + 1, h = L.n[2] + 1;
  const int lin = blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= w*h) return;
  const int jj = lin % w, kk = lin / w;
  const int j = L.cs[1] - NG + jj, k = L.cs[2] - NG + kk;                 // parent active indices
  const long cfp = (long)(c.Nx1 + 1), cfrow = (long)(c.Nx2 + 1)*cfp;
  const long ffp = (long)(f.Nx1 + 1), ffrow = (long)(f.Nx2 + 1)*ffp;
  const Real v = c.edgeflux[(long)k*cfrow + (long)j*cfp + (L.cs[0] - NG)];
  const int ks = k*2 - L.cdisp[2], js = j*2 - L.cdisp[1];                 // :97-98
#define EF(a,b) f.edgeflux[(long)(a)*ffrow + (long)(b)*ffp]
  EF(ks, js) = v;
  if (jj < w - 1) {
    if (kk < h - 1) { EF(ks+1, js+1) = v; EF(ks, js+1) = v; EF(ks+1, js) = v; }
    else EF(ks, js+1) = v;
  } else if (kk < h - 1) EF(ks+1, js) = v;
#undef EF
}

// ---- flux correction across slabs: the child's restricted x3 boundary flux as a message
// ([b][a][6]), and its application on the neighbouring slab of the parent (smr.c:1322-1340) ----
__global__ void __launch_bounds__(256)
k_flux_x3_export(DevGrid f, int side, Real *buf)
{
  const int n1 = f.Nx1/2, n2 = f.Nx2/2;
  const int lin = blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= n1*n2) return;
  const int a = lin % n1, b = lin / n1;
  const long m = (long)(side ? f.ke + 1 : f.ks)*f.sK + (long)(f.js + 2*b)*f.sJ + (f.is + 2*a);
  for (int v = 0; v < 6; v++) {
    const Real *q = fld(f, f.F, 2*6 + v) + m;
    Real s = q[0] + q[1];
    s += q[f.sJ] + q[f.sJ + 1];
    s *= 0.25;
    buf[(long)lin*6 + v] = s;
  }
}

__global__ void __launch_bounds__(256)
k_flux_x3_apply(DevGrid c, int side, int i0, int j0, int n1, int n2, int nvar, Real dt, const Real *buf)
{
  const int lin = blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= n1*n2) return;
  const int a = lin % n1, b = lin / n1;
  const int kc = side ? c.ks : c.ke, kf = side ? c.ks : c.ke + 1;
  const Real q = side ? (dt/c.dx[2]) : -(dt/c.dx[2]);
  const long mc = (long)kc*c.sK + (long)(j0 + b)*c.sJ + (i0 + a), mf = (long)kf*c.sK + (long)(j0 + b)*c.sJ + (i0 + a);
  for (int v = 0; v < nvar; v++)
    fld(c, c.U, v)[mc] -= q*(fld(c, c.F, 2*6 + v)[mf] - buf[(long)lin*6 + v]);
}

inline unsigned nblk(long n, int b) { return (unsigned)((n + b - 1)/b); }

}  // namespace

// Grids in the order of the reference's loops: level by level from the root, Domains of a level in deck order
// (MeshS.Domain[nl][nd]).  Link L joins grid L+1 (the child) to grid par[L] (its parent); with one Domain per level par[L] = L.
// Domains of a level neither overlap nor touch (init_mesh.c:398-418): a child has ONE parent.
struct aa_mesh {
  int nl = 0;                      // number of Grids
  aa_grid *lev[AA_MAXLEV];
  int disp[AA_MAXLEV][3];
  int par[AA_MAXLEV];              // par[L]: parent grid of grid L+1
  int f2c[AA_MAXLEV];              // the links finest level first, deck order inside a level (smr.c:1224)
  Link link[AA_MAXLEV];            // link[L]: grid L+1 on grid par[L]
  Real *box[AA_MAXLEV];            // prolongation snapshot of the parent's zones around grid L+1
  hipStream_t st = nullptr; bool own_stream = true;
  // aa_mesh_step: the integrators of the levels read and write their own level only (main.c:572-585 calls them one after the other
  // and couples the levels afterwards), and the Grids of the reference's decks are too small to fill the GPU alone: level l > 0
  // integrates on side[l], forked from and joined to `st` by events (AA_MESH_OVERLAP=0: one after the other on `st`)
  hipStream_t side[AA_MAXLEV] = {nullptr}; hipEvent_t ev_fork = nullptr, ev_join[AA_MAXLEV] = {nullptr}; bool overlap = true;
  bool one_launch = true;          // the six sides of a flux correction / prolongation in one launch each
  double tcoarse = 0;              // ionrad_3d.c:44
  double time = 0, dt = 0; int nstep = 0;   // MeshS
};

extern "C" {

void aa_mesh_destroy(aa_mesh *m);
// error paths of the create functions: the caller keeps its Grids as they were (mesh_alloc marked them as levels of a Mesh)
static void mesh_drop(aa_mesh *m)
{
  for (int l = 0; l < m->nl; l++) if (m->lev[l]) { m->lev[l]->keep_flux = false; m->lev[l]->keep = {0, {{0}}}; }
  delete m;
}
static int mesh_finish(aa_mesh *m, aa_grid **levels, aa_mesh **out)
{
  hipError_t e = hipSetDevice(levels[0]->p.device);
  if (e == hipSuccess) e = hipStreamCreate(&m->st);
  if (e != hipSuccess) { mesh_drop(m); return aa_fail(-2, "[aa_mesh_create]: %s", hipGetErrorString(e)); }
  for (int l = 0; l < m->nl; l++) aa_set_stream(m->lev[l], (void*)m->st);
  { const char *ev = getenv("AA_MESH_OVERLAP"); m->overlap = ev ? atoi(ev) != 0 : true; }
  { const char *ev = getenv("AA_SMR_ONE_LAUNCH"); m->one_launch = ev ? atoi(ev) != 0 : true; }
  if (m->overlap && m->nl > 1) {
    e = hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming);
    for (int l = 1; l < m->nl && e == hipSuccess; l++) {
      e = hipStreamCreateWithFlags(&m->side[l], hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_join[l], hipEventDisableTiming);
    }
    if (e != hipSuccess) { aa_mesh_destroy(m); return aa_fail(-2, "[aa_mesh_create]: %s", hipGetErrorString(e)); }
  }
  for (int l = 0; l + 1 < m->nl; l++) {
    const Link &L = m->link[l];
    const size_t nb = (size_t)(L.n[0] + 6)*(L.n[1] + 6)*(L.n[2] + 6)*6;
    if (hipMalloc(&m->box[l], nb*sizeof(Real)) != hipSuccess) { aa_mesh_destroy(m); return aa_fail(-2, "[aa_mesh_create]: hipMalloc box"); }
    (void)hipMemset(m->box[l], 0, nb*sizeof(Real));
  }
  // the face planes whose second-pass fluxes k_flux_correct / k_flux_x3_export / k_flux_x3_apply read:
  // a level's own boundary faces and the outline of its child
  for (int l = 0; l < m->nl; l++) {
    aa_grid *g = m->lev[l];
    const int lo[3] = {g->d.is, g->d.js, g->d.ks}, hi[3] = {g->d.ie, g->d.je, g->d.ke};
    g->keep.n = 8;
    int nch = 0;
    for (int d = 0; d < 3; d++) { for (int q = 0; q < 8; q++) g->keep.p[d][q] = lo[d]; g->keep.p[d][1] = hi[d] + 1; }
    for (int L = 0; L + 1 < m->nl; L++) {
      if (m->par[L] != l) continue;
      if (nch == 3) { aa_mesh_destroy(m); return aa_fail(-1, "[aa_mesh_create]: more than three child Domains on one Grid"); }
      for (int d = 0; d < 3; d++) { g->keep.p[d][2 + 2*nch] = m->link[L].cs[d]; g->keep.p[d][3 + 2*nch] = m->link[L].ce[d] + 1; }
      nch++;
    }
  }
  { // the links finest level first, deck order inside a level
    int n = 0, maxlev = 0;
    for (int L = 0; L + 1 < m->nl; L++) if (m->lev[L + 1]->level > maxlev) maxlev = m->lev[L + 1]->level;
    for (int lev = maxlev; lev >= 1; lev--) for (int L = 0; L + 1 < m->nl; L++) if (m->lev[L + 1]->level == lev) m->f2c[n++] = L;
  }
  *out = m;
  return 0;
}

static aa_mesh *mesh_alloc(int nlevels, aa_grid **levels)
{
  if (!levels || nlevels < 1 || nlevels > AA_MAXLEV) { aa_fail(-1, "[aa_mesh_create]: bad arguments"); return nullptr; }
  aa_mesh *m = new aa_mesh();
  m->nl = nlevels;
  for (int l = 0; l < AA_MAXLEV; l++) { m->lev[l] = nullptr; m->box[l] = nullptr; }
  for (int l = 0; l < nlevels; l++) {
    aa_grid *g = levels[l];
    if (!g || (l == 0 ? g->level != 0 : (g->level < 1 || g->level < levels[l - 1]->level || g->level > levels[l - 1]->level + 1))) {
      mesh_drop(m); aa_fail(-1, "[aa_mesh_create]: grids must come level by level from the root (grid %d)", l); return nullptr; }
    if (!g->slab.empty()) { mesh_drop(m); aa_fail(-1, "[aa_mesh_create]: levels[%d] is cut into slabs (aa_params.nslab / AA_NGPU): nested levels stay on one device", l); return nullptr; }
    if (g->p.device != levels[0]->p.device) { mesh_drop(m); aa_fail(-1, "[aa_mesh_create]: all levels must live on one device"); return nullptr; }
    m->lev[l] = g; m->box[l] = nullptr; m->par[l] = l;
    g->keep_flux = true;
  }
  return m;
}

// init_grid.c (overlap tables) + SMR_init (smr.c:2931).  Takes over the levels' streams: all
// levels run on one stream so that inter-level kernels are ordered without events.
int aa_mesh_create(int nlevels, aa_grid **levels, const int *disp, aa_mesh **out)
{
  if (!disp || !out) return aa_fail(-1, "[aa_mesh_create]: bad arguments");
  aa_mesh *m = mesh_alloc(nlevels, levels);
  if (!m) return -1;
  for (int c = 1; c < nlevels; c++) {
    const aa_grid *C = m->lev[c];
    const int *dc = disp + 3*c;
    // the Domain of the level below that contains this one
    int pi = -1;
    for (int q = 0; q < c && pi < 0; q++) {
      const aa_grid *Q = m->lev[q];
      bool inside = (Q->level == C->level - 1);
      for (int d = 0; d < 3 && inside; d++) {
        const int dq = Q->level ? disp[3*q + d] : 0;
        if (dc[d]/2 < dq || (dc[d] + C->p.Nx[d])/2 > dq + Q->p.Nx[d]) inside = false;
      }
      if (inside) pi = q;
    }
    if (pi < 0) { mesh_drop(m); return aa_fail(-1, "[aa_mesh_create]: grid %d (level %d) is not nested in a Domain of level %d", c, C->level, C->level - 1); }
    const int l = c - 1;
    m->par[l] = pi;
    const aa_grid *P = m->lev[pi];
    Link &L = m->link[l];
    const int lo[3] = {P->d.is, P->d.js, P->d.ks};
    const int irefine = 1 << C->level;
    const int dp[3] = {P->level ? disp[3*pi] : 0, P->level ? disp[3*pi + 1] : 0, P->level ? disp[3*pi + 2] : 0};
    for (int d = 0; d < 3; d++) {
      const int a = dc[d]/2 - dp[d], b = (dc[d] + C->p.Nx[d])/2 - dp[d];
      if ((dc[d] & 1) || (C->p.Nx[d] & 1) || a < 0 || b > P->p.Nx[d]) {
        mesh_drop(m); return aa_fail(-1, "[aa_mesh_create]: grid %d is not nested in grid %d along x%d", c, pi, d + 1);
      }
      L.cs[d] = a + lo[d]; L.ce[d] = b + lo[d] - 1; L.n[d] = b - a; L.cdisp[d] = dc[d];
      L.prol[2*d]     = L.corr[2*d]     = (dc[d] != 0);
      L.prol[2*d + 1] = L.corr[2*d + 1] = ((dc[d] + C->p.Nx[d])/irefine != C->p.rootNx[d]);
      // init_mesh.c:320-360: a child may touch its parent's edge only on the root boundary
      if ((a == 0 && L.prol[2*d]) || (b == P->p.Nx[d] && L.prol[2*d + 1])) {
        mesh_drop(m); return aa_fail(-1, "[init_mesh] child Domain (grid %d) touches its parent in x%d", c, d + 1);
      }
    }
    // init_mesh.c:398-418: Domains on the same level neither overlap nor touch
    for (int q = 1; q < c; q++) {
      const aa_grid *Q = m->lev[q];
      if (Q->level != C->level) continue;
      bool sep = false;
      for (int d = 0; d < 3; d++) if (dc[d] > disp[3*q + d] + Q->p.Nx[d] || disp[3*q + d] > dc[d] + C->p.Nx[d]) sep = true;
      if (!sep) { mesh_drop(m); return aa_fail(-1, "[init_mesh]: Domains at the same level overlap or touch (grids %d and %d)", q, c); }
    }
    // ionrad_smr.c:97-98 mixes a parent-local index with the child's root-relative Disp: with a displaced
    // parent (3+ levels) the reference writes out of bounds.  Refused by default; AA_SMR_DEEP_RADIATION=fixed
    // uses the index the formula evidently means (child origin - 2 x parent origin), a documented
    // departure from the reference for decks like its own 5-level one.
    if (P->p.ion && (dp[1] || dp[2])) {
      const char *e = getenv("AA_SMR_DEEP_RADIATION");
      if (!(e && strcmp(e, "fixed") == 0)) {
        mesh_drop(m); return aa_fail(-1, "[aa_mesh_create]: radiation across a displaced parent (grid %d) is undefined in the reference "
                                     "(set AA_SMR_DEEP_RADIATION=fixed for the corrected hand-off)", pi);
      }
      for (int d = 0; d < 3; d++) L.cdisp[d] = dc[d] - 2*dp[d];
    }
  }
  return mesh_finish(m, levels, out);
}

// One rank's stack of x3 slabs of a Mesh whose levels are all cut at the same planes (multi-GPU
// SMR).  links[21*l..]: cs[3] (local parent index incl. ghosts), n[3], prol[6], corr[6], cdisp[3]
// of level l+1 on level l.  A child slab may reach the edge of its parent slab; where the level
// itself ends at that edge (prol=1, corr=0) the parent plane outside is corrected on the
// neighbouring slab with aa_flux_x3_export / aa_flux_x3_apply.
int aa_mesh_create_local(int nlevels, aa_grid **levels, const int *links, aa_mesh **out)
{
  if ((nlevels > 1 && !links) || !out) return aa_fail(-1, "[aa_mesh_create_local]: bad arguments");
  aa_mesh *m = mesh_alloc(nlevels, levels);
  if (!m) return -1;
  // links join grid l+1 to grid l here (par[l] = l): ONE Domain per level, a strict chain (mesh_alloc alone would also take siblings)
  for (int l = 0; l < nlevels; l++)
    if (m->lev[l]->level != l) { mesh_drop(m); return aa_fail(-1, "[aa_mesh_create_local]: grid %d is on level %d: a rank's stack holds one Domain per level", l, levels[l]->level); }
  for (int l = 0; l + 1 < nlevels; l++) {
    const int *q = links + 21*l;
    const aa_grid *P = m->lev[l], *C = m->lev[l + 1];
    Link &L = m->link[l];
    for (int d = 0; d < 3; d++) {
      L.cs[d] = q[d]; L.n[d] = q[3 + d]; L.ce[d] = q[d] + q[3 + d] - 1; L.cdisp[d] = q[18 + d];
      if (L.n[d]*2 != C->p.Nx[d] || L.cs[d] < AA_NGHOST || L.ce[d] >= AA_NGHOST + P->p.Nx[d]) {
        mesh_drop(m); return aa_fail(-1, "[aa_mesh_create_local]: link %d does not match the slabs along x%d", l, d + 1);
      }
    }
    for (int d = 0; d < 6; d++) { L.prol[d] = q[6 + d]; L.corr[d] = q[12 + d]; }
  }
  return mesh_finish(m, levels, out);
}

void aa_mesh_destroy(aa_mesh *m)      // the levels stay alive and go back to the default stream
{
  if (!m) return;
  hipStreamSynchronize(m->st);
  for (int l = 0; l < m->nl; l++) {
    m->lev[l]->st = nullptr; m->lev[l]->own_stream = false; if (m->box[l]) hipFree(m->box[l]);
    m->lev[l]->keep_flux = false; m->lev[l]->keep = {0, {{0}}};      // no longer a level of a Mesh
  }
  for (int l = 1; l < m->nl; l++) { if (m->side[l]) hipStreamDestroy(m->side[l]); if (m->ev_join[l]) hipEventDestroy(m->ev_join[l]); }
  if (m->ev_fork) hipEventDestroy(m->ev_fork);
  if (m->own_stream) hipStreamDestroy(m->st);
  delete m;
}

int aa_mesh_set_stream(aa_mesh *m, void *hip_stream)   // run every level on a caller-owned stream
{
  hipStreamSynchronize(m->st);
  if (m->own_stream) { hipStreamDestroy(m->st); m->own_stream = false; }
  m->st = (hipStream_t)hip_stream;
  for (int l = 0; l < m->nl; l++) { m->lev[l]->st = m->st; m->lev[l]->own_stream = false; }
  return 0;
}

int aa_mesh_get_state(const aa_mesh *m, double *time, double *dt, int *nstep)
{ if (time) *time = m->time; if (dt) *dt = m->dt; if (nstep) *nstep = m->nstep; return 0; }
int aa_mesh_set_state(aa_mesh *m, double time, double dt, int nstep)
{ m->time = time; m->dt = dt; m->nstep = nstep; return 0; }

// smr.c:1207.  Before the first step (main.c:401) the reference's myFlx arrays are all zero and the
// flux correction adds q*(0-0); here the flux arrays are zero-initialised, to the same effect.
int aa_mesh_restrict_correct_pair(aa_mesh *m, int l)      // level l+1 -> level l
{
  if (l >= 0 && l + 1 < m->nl) m->lev[m->par[l]]->active_dirty = true;
  if (l < 0 || l + 1 >= m->nl) return aa_fail(-1, "[aa_mesh_restrict_correct_pair]: pair %d", l);
  aa_grid *P = m->lev[m->par[l]], *C = m->lev[l + 1];       // link l: grid l+1 on its parent
  const Link &L = m->link[l];
  const int nvar = 5 + P->p.nscal;
  Scope s(P, "smr_restrict_correct");
  hipLaunchKernelGGL(k_restrict, dim3(nblk((long)L.n[0]*L.n[1]*L.n[2], 256)), dim3(256), 0, m->st,
                     C->d, P->d, L, (1u << nvar) - 1u);
  if (m->one_launch) {      // the six sides in ONE launch (AA_SMR_ONE_LAUNCH=0: one per side)
    long nmax = 0; bool any = false;
    for (int dim = 0; dim < 6; dim++) {
      const int d = dim >> 1, d1 = (d == 0) ? 1 : 0, d2 = (d == 2) ? 1 : 2;
      if (L.corr[dim]) { any = true; const long n = (long)L.n[d1]*L.n[d2]; if (n > nmax) nmax = n; }
    }
    if (any) hipLaunchKernelGGL(k_flux_correct, dim3(nblk(nmax, 256), 6), dim3(256), 0, m->st, C->d, P->d, L, -1, nvar, (Real)P->dt);
  } else
  for (int dim = 0; dim < 6; dim++) {
    if (!L.corr[dim]) continue;
    const int d = dim >> 1, d1 = (d == 0) ? 1 : 0, d2 = (d == 2) ? 1 : 2;
    hipLaunchKernelGGL(k_flux_correct, dim3(nblk((long)L.n[d1]*L.n[d2], 256)), dim3(256), 0, m->st,
                       C->d, P->d, L, dim, nvar, (Real)P->dt);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int aa_mesh_restrict_correct(aa_mesh *m)
{
  // finest pair first: a level is restricted for its parent after it received its own child's
  // solution and flux correction (the order of the loop over levels at smr.c:1224)
  for (int l = 0; l + 1 < m->nl; l++) { int rc = aa_mesh_restrict_correct_pair(m, m->f2c[l]); if (rc) return rc; }
  return 0;
}

// smr.c:85: E and s[0] only, after the radiation step
int aa_mesh_ionrad_restrict_correct(aa_mesh *m)
{
  for (int l = 0; l < m->nl; l++) m->lev[l]->active_dirty = true;
  for (int q = 0; q + 1 < m->nl; q++) {
    const int l = m->f2c[q];
    aa_grid *P = m->lev[m->par[l]], *C = m->lev[l + 1];
    const Link &L = m->link[l];
    Scope s(P, "smr_ion_restrict");
    hipLaunchKernelGGL(k_restrict, dim3(nblk((long)L.n[0]*L.n[1]*L.n[2], 256)), dim3(256), 0, m->st,
                       C->d, P->d, L, (1u << 4) | (1u << 5));
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// smr.c:2359
int aa_mesh_prolongate(aa_mesh *m)
{
  for (int l = 0; l < m->nl; l++) {
    for (int c = 0; c + 1 < m->nl; c++) {      // Step 1: hand the zones around every child over
      if (m->par[c] != l) continue;
      const Link &L = m->link[c];
      const long nb = (long)(L.n[0] + 6)*(L.n[1] + 6)*(L.n[2] + 6);
      hipLaunchKernelGGL(k_box_copy, dim3(nblk(nb, 256)), dim3(256), 0, m->st, m->lev[l]->d, L, m->box[c]);
    }
    if (l > 0) {                               // Steps 2-3: own ghost zones from the parent's zones
      aa_grid *C = m->lev[l];
      const Link &L = m->link[l - 1];
      const int nvar = 5 + C->p.nscal;
      Scope s(C, "smr_prolongate");
      if (m->one_launch) {
        long nmax = 0;
        for (int dim = 0; dim < 6; dim++) {
          if (!L.prol[dim]) continue;
          long cnt = 1;
          for (int d = 0; d < 3; d++) cnt *= (d == (dim >> 1)) ? NG/2 : (C->p.Nx[d] + 2*NG)/2;
          if (cnt > nmax) nmax = cnt;
        }
        if (nmax > 0) hipLaunchKernelGGL(k_prolong, dim3(nblk(nmax, 128), 6), dim3(128), 0, m->st, C->d, L, m->box[l - 1], -1, nvar);
      } else
      for (int dim = 0; dim < 6; dim++) {
        if (!L.prol[dim]) continue;
        long cnt = 1;
        for (int d = 0; d < 3; d++) cnt *= (d == (dim >> 1)) ? NG/2 : (C->p.Nx[d] + 2*NG)/2;
        hipLaunchKernelGGL(k_prolong, dim3(nblk(cnt, 128)), dim3(128), 0, m->st, C->d, L, m->box[l - 1], dim, nvar);
      }
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// new_dt.c:32 over all levels: max_v is carried from Grid to Grid (:33), one dt for the Mesh
int aa_mesh_new_dt(aa_mesh *m)
{
  double cum[3] = {0.0, 0.0, 0.0}, max_dti = 0.0;
  for (int l = 0; l < m->nl; l++) {
    aa_grid *g = m->lev[l];
    { Scope s(g, "new_dt");
      HIPCHK(hipMemsetAsync(g->sc->max_v, 0, 3*sizeof(unsigned long long), g->st));
      launch_cfl(g->d, g->sc, g->st); }
    int rc = aa_fetch_scalars(g); if (rc) return rc;
    for (int d = 0; d < 3; d++) {
      const double v = bits_to_double(g->sc_host->max_v[d]);
      cum[d] = (cum[d] > v) ? cum[d] : v;
    }
    for (int d = 0; d < 3; d++) { const double q = cum[d]/g->d.dx[d]; max_dti = (max_dti > q) ? max_dti : q; }
  }
  const aa_params &p = m->lev[0]->p;
  const double dtc = p.cour_no/max_dti;
  if (m->nstep == 0) m->dt = dtc; else m->dt = (2.0*m->dt < dtc) ? 2.0*m->dt : dtc;
  if ((m->time < p.tlim) && ((p.tlim - m->time) < m->dt)) m->dt = p.tlim - m->time;
  for (int l = 0; l < m->nl; l++) m->lev[l]->dt = m->dt;
  return 0;
}

// ionrad_smr.c:345 + :34: the flux level l-1 left at the upstream face of level l, onto level l's rays
int aa_mesh_ionflux_prolong(aa_mesh *m, int l)
{
  if (l < 1 || l >= m->nl) return aa_fail(-1, "[aa_mesh_ionflux_prolong]: level %d", l);
  const Link &L = m->link[l - 1];
  aa_grid *P = m->lev[m->par[l - 1]];
  { int rc = aa_edgeflux_ready(P); if (rc) return rc; }       // the parent's EdgeFlux of its last sweep
  if (L.prol[0])
    hipLaunchKernelGGL(k_ionflux_prolong, dim3(nblk((long)(L.n[1] + 1)*(L.n[2] + 1), 256)), dim3(256), 0, m->st,
                       m->lev[l]->d, P->d, L);
  HIPCHK(hipGetLastError());
  return 0;
}

// multi-GPU SMR: the child's restricted boundary flux (DEVICE buffer of (Nx1/2)(Nx2/2)*6 doubles) ...
int aa_flux_x3_export(aa_grid *child, int side, double *dev_buf)
{
  if (!child->slab.empty()) return aa_fail(-1, "[aa_flux_x3_export]: not available on a Grid cut into slabs");
  const long n = (long)(child->p.Nx[0]/2)*(child->p.Nx[1]/2);
  hipLaunchKernelGGL(k_flux_x3_export, dim3(nblk(n, 256)), dim3(256), 0, child->st, child->d, side, dev_buf);
  HIPCHK(hipGetLastError());
  return 0;
}
// ... and its application to the plane of the parent slab across the cut (Grid dt as in RestrictCorrect)
int aa_flux_x3_apply(aa_grid *parent, int side, int i0, int j0, int n1, int n2, const double *dev_buf)
{
  parent->active_dirty = true;
  if (!parent->slab.empty()) return aa_fail(-1, "[aa_flux_x3_apply]: not available on a Grid cut into slabs");
  if (i0 < AA_NGHOST || j0 < AA_NGHOST || i0 + n1 > AA_NGHOST + parent->p.Nx[0] || j0 + n2 > AA_NGHOST + parent->p.Nx[1])
    return aa_fail(-1, "[aa_flux_x3_apply]: region outside the Grid");
  hipLaunchKernelGGL(k_flux_x3_apply, dim3(nblk((long)n1*n2, 256)), dim3(256), 0, parent->st, parent->d, side, i0, j0, n1, n2,
                     5 + parent->p.nscal, (Real)parent->dt, dev_buf);
  HIPCHK(hipGetLastError());
  return 0;
}

// ionrad_3d.c:862 with STATIC_MESH_REFINEMENT: the root sub-cycles to its own stopping criteria and
// publishes the time it covered; a refined level sub-cycles until it has covered exactly that
int aa_mesh_ion_radtransfer(aa_mesh *m, int l, int *niter_out)
{
  aa_grid *g = m->lev[l];
  const bool finegrid = (g->level != 0);
  double dt_done = 0.0;
  int niter = 0, rc;
  if (finegrid) { if ((rc = aa_mesh_ionflux_prolong(m, l))) return rc; }
  else m->tcoarse = 0;
  if ((rc = aa_ion_run(g, finegrid ? 1 : 0, finegrid ? m->tcoarse : g->dt, &niter, &dt_done))) return rc;
  if (!finegrid) {
    if (niter == g->p.maxiter) g->dt = dt_done;
    m->tcoarse = dt_done;
  }
  m->dt = g->dt;                             // :1030 pMesh->dt = pGrid->dt
  if (niter_out) *niter_out = niter;
  return 0;
}

// main.c:395-447 after problem() has filled every level
int aa_mesh_start(aa_mesh *m)
{
  int rc;
  if ((rc = aa_mesh_restrict_correct(m))) return rc;
  for (int l = 0; l < m->nl; l++) {
    if ((rc = aa_bvals_mhd(m->lev[l]))) return rc;
    if ((rc = aa_bvals_ionrad(m->lev[l]))) return rc;
  }
  if ((rc = aa_mesh_prolongate(m))) return rc;
  return aa_mesh_new_dt(m);
}

// one pass of main.c:519-669 with STATIC_MESH_REFINEMENT; niter[l] = radiation sub-cycles of level l
int aa_mesh_step(aa_mesh *m, int *niter)
{
  int rc;
  aa_grid *root = m->lev[0];
  if (root->p.ion && root->nradplane > 0) {                          // :546-562
    for (int l = 0; l < m->nl; l++) {
      int n = 0;
      m->lev[l]->time = m->time;
      if ((rc = aa_mesh_ion_radtransfer(m, l, &n))) return rc;
      if (niter) niter[l] = n;
      if ((rc = aa_bvals_mhd(m->lev[l]))) return rc;
    }
    if ((rc = aa_mesh_ionrad_restrict_correct(m))) return rc;
  } else if (niter) for (int l = 0; l < m->nl; l++) niter[l] = 0;
  const bool fork = m->overlap && m->nl > 1 && m->side[1];
  if (fork) {
    HIPCHK(hipEventRecord(m->ev_fork, m->st));
    for (int l = 1; l < m->nl; l++) HIPCHK(hipStreamWaitEvent(m->side[l], m->ev_fork, 0));
  }
  rc = 0;
  for (int l = 0; l < m->nl && !rc; l++) {                            // :572-585
    aa_grid *g = m->lev[l];
    if (fork && l > 0) g->st = m->side[l];
    rc = (g->p.integrator == 1 ? aa_integrate_3d_vl(g) : aa_integrate_3d_ctu(g));
    g->st = m->st;
  }
  if (fork)
    for (int l = 1; l < m->nl; l++) { HIPCHK(hipEventRecord(m->ev_join[l], m->side[l])); HIPCHK(hipStreamWaitEvent(m->st, m->ev_join[l], 0)); }
  if (rc) return rc;
  if ((rc = aa_mesh_restrict_correct(m))) return rc;                  // :591
  for (int l = 0; l < m->nl; l++)                                     // :597 Userwork_in_loop
    if (m->lev[l]->npin > 0 && (rc = aa_apply_pinned_cells(m->lev[l]))) return rc;
  m->nstep++; m->time += m->dt;                                       // :618-626
  for (int l = 0; l < m->nl; l++) { m->lev[l]->time = m->time; m->lev[l]->nstep = m->nstep; }
  if ((rc = aa_mesh_new_dt(m))) return rc;                            // :629
  for (int l = 0; l < m->nl; l++) if ((rc = aa_bvals_mhd(m->lev[l]))) return rc;   // :635-644
  return aa_mesh_prolongate(m);                                       // :647
}

}  // extern "C"
