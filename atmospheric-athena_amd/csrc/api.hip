// api.hip -- the C-ABI of include/athena_amd.h: owns the device mirror of one Grid, sequences
// the kernel chains in the order of the reference's main loop (main.c:519-669) and of
// ion_radtransfer_3d (ionrad_3d.c:862-1047), and carries the scalar control flow (dt, sub-cycle
// termination) on the host exactly as the reference does.  No CPU arithmetic path exists here.
#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <string>
#include <thread>
#include <vector>
#include "api_internal.h"

using namespace aa;

static thread_local char g_err[1024] = "";
int aa_fail(int code, const char *fmt, ...)
{
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return code;
}
#define fail aa_fail

static void prof_drain(aa_grid *g)
{
  hipStreamSynchronize(g->st);
  for (auto &e : g->pe) {
    for (size_t i = 0; i + 1 < e.ev.size(); i += 2) {
      float ms = 0; hipEventElapsedTime(&ms, e.ev[i], e.ev[i + 1]); e.total_ms += ms;
      g->ev_pool.push_back(e.ev[i]); g->ev_pool.push_back(e.ev[i + 1]);
    }
    e.ev.clear();
  }
}

void aa_eval_grav_tables(const aa_params &p, const double dx[3], int N1, int N2, int N3, aa_gravpot_fn fn, std::vector<double> t[4])
{
  for (int w = 0; w < 4; w++) t[w].resize((size_t)N1*N2*N3);
  // the callback is a pure function of position (as in the reference, which calls it ~52 times
  // per cell per step): evaluate k-planes on all host cores
  unsigned nth = std::thread::hardware_concurrency(); if (nth < 1) nth = 1; if (nth > 64) nth = 64;
  if ((int)nth > N3) nth = N3;
  std::vector<std::thread> pool;
  for (unsigned w = 0; w < nth; w++) pool.emplace_back([&, w]() {
    for (int k = (int)w; k < N3; k += (int)nth) for (int j = 0; j < N2; j++) for (int i = 0; i < N1; i++) {
      // cc_pos.c:36-43
      const double x1 = p.MinX[0] + ((double)(i - AA_NGHOST) + 0.5)*dx[0];
      const double x2 = p.MinX[1] + ((double)(j - AA_NGHOST) + 0.5)*dx[1];
      const double x3 = p.MinX[2] + ((double)(k - AA_NGHOST) + 0.5)*dx[2];
      const size_t m = ((size_t)k*N2 + j)*N1 + i;
      t[0][m] = fn(x1, x2, x3);
      t[1][m] = fn(x1 - 0.5*dx[0], x2, x3);
      t[2][m] = fn(x1, x2 - 0.5*dx[1], x3);
      t[3][m] = fn(x1, x2, x3 - 0.5*dx[2]);
    }
  });
  for (auto &th : pool) th.join();
}

extern "C" {

const char *aa_last_error(void) { return g_err; }

int aa_create(const aa_params *p, aa_grid **out)
{
  if (!p || !out) return fail(-1, "[aa_create]: null argument");
  for (int d = 0; d < 3; d++)
    if (p->Nx[d] <= 1) return fail(-1, "[aa_create]: 3-D only, Nx%d=%d", d + 1, p->Nx[d]);
  if (p->cour_no > 0.5)   // integrate.c:66-68
    return fail(-1, "<time>cour_no was set to %g: must be <= 0.5 with 3D integrator", p->cour_no);
  if (p->ion && p->nscal != 1) return fail(-1, "[aa_create]: ion radiation needs NSCALARS=1");
  if (p->nscal != 0 && p->nscal != 1) return fail(-1, "[aa_create]: NSCALARS must be 0 or 1");
  { // one Grid of the caller on several GPUs: aa_params.nslab, or AA_NGPU in the environment (drop-in executables)
    int ns = p->nslab;
    if (ns == 0) { const char *e = getenv("AA_NGPU"); if (e) ns = atoi(e); }
    if (ns > 1) return slabs_create(p, ns, out);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(-3, "[aa_create]: no HIP device visible -- this library has no CPU path");
  HIPCHK(hipSetDevice(p->device));
  aa_grid *g = new aa_grid();
  g->p = *p;
  HostGrid &d = g->d;
  d = HostGrid();
  {   // the launch choices of this Grid (grid.h LaunchCfg): read here, once per Grid, never again
    LaunchCfg &c = d.cfg;
    auto env = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    c.strip = env("AA_STRIP", 64); c.xcd = env("AA_XCD", 1);
    c.x1_flat = env("AA_X1_FLAT", 1); c.slopes_march = env("AA_SLOPES_MARCH", 1);
    c.ca_kc = env("AA_CA_KC", 0); c.fu_kc = env("AA_FU_KC", 0);
    c.ion_pass_cap = env("AA_ION_PASS_BLOCKS", 4096); if (c.ion_pass_cap < 1) c.ion_pass_cap = 1;
    c.pitch_align = env("AA_PITCH_ALIGN", 1);
    c.mailbox = env("AA_MAILBOX", 1); c.mailbox_spin_us = env("AA_MAILBOX_SPIN_US", 300);
    c.bc_one = env("AA_BC_ONE", 1); c.fuse_pick = env("AA_ION_FUSE_PICK", 1); c.pin_one = env("AA_PIN_ONE", 1);
    c.edge_overlap = env("AA_EDGE_OVERLAP", 0);
  }
  d.Nx1 = p->Nx[0]; d.Nx2 = p->Nx[1]; d.Nx3 = p->Nx[2];
  d.N1 = d.Nx1 + 2*AA_NGHOST; d.N2 = d.Nx2 + 2*AA_NGHOST; d.N3 = d.Nx3 + 2*AA_NGHOST;
  d.is = d.js = d.ks = AA_NGHOST;
  d.ie = d.is + d.Nx1 - 1; d.je = d.js + d.Nx2 - 1; d.ke = d.ks + d.Nx3 - 1;
  // rows padded to a multiple of 16 doubles and every field shifted by 12 doubles: the first active zone of
  // every row (i = 4) then sits on a 128-byte line, and so does every wavefront of the kernels that walk the
  // active zones 64 at a time (-2.9 % of a 512^3 step; AA_PITCH_ALIGN=0: dense rows)
  const int pitch_align = d.cfg.pitch_align;
  d.sJ = pitch_align ? ((d.N1 + 15)/16)*16 : d.N1; d.sK = (long)d.sJ*d.N2; d.nc = d.sK*d.N3;
  if (p->level < 0 || p->level > 7) { delete g; return fail(-1, "[aa_create]: level %d out of range", p->level); }
  g->level = p->level;
  // the one-kernel correct pass (with both first passes on board since round 4) wins from about 70^3 zones on, the tile kernels
  // below (profiles/microbench/ca_threshold_r04.sh: 64^3 -11 %, 80^3 +20 %, 96^3 +21 %, 112^3 +21 %; until round 4 the
  // limit was 2^21 zones): same results bit for bit, so the choice follows the size unless AA_CORRECT_ALL forces it
  { const char *e = getenv("AA_CORRECT_ALL");
    g->correct_all = e ? atoi(e) != 0 : ((long long)p->Nx[0]*p->Nx[1]*p->Nx[2] >= 400000LL); }
  // (k_correct_all also does the x3 first pass: see x3_fused below)
  { const char *e = getenv("AA_CFL_FUSED"); g->cfl_step = e ? atoi(e) != 0 : true; g->cfl_force = e && atoi(e) == 2; }     // aa_step: new_dt's maxima from the update kernel
  { const char *e = getenv("AA_X3_FUSED"); g->x3_fused_mode = e ? (atoi(e) != 0) : -1; }    // (the potential arrives after aa_create)
  // rates inside the ray sweep: one block per 64 rays, so it needs many rays to fill the chip (512^2 rays:
  // -2.9 ms per step; 80^2 rays: +6 %); same results either way
  { const char *e = getenv("AA_FUSED_RATES"); g->fused_rates = e ? atoi(e) != 0 : ((long long)p->Nx[1]*p->Nx[2] >= (1LL << 17)); }
  { const char *e = getenv("AA_VL_PREDICT"); g->vl_predict = e ? atoi(e) != 0 : ((long long)p->Nx[0]*p->Nx[1]*p->Nx[2] >= (1LL << 18)); }
  { const char *e = getenv("AA_FUSED_UPDATE"); g->fused_update = e ? atoi(e) != 0 : true; }
  Real rootdx[3];
  for (int a = 0; a < 3; a++) {
    rootdx[a] = (p->xmax[a] - p->xmin[a])/(Real)(p->rootNx[a]);   // init_mesh.c:225
    d.dx[a] = rootdx[a]/(Real)(1 << p->level);                    // :245
  }
  d.Gamma = p->gamma; d.Gamma_1 = p->gamma - 1.0; d.rGamma_1 = 1.0/d.Gamma_1;
  // one pool: U 6 | LR 36 | F 18 | eta 3 | dhalf 1 | phi 4 | ion 5 + sign(1) | edgeflux
  const size_t nc = (size_t)d.nc;
  const size_t nef = (size_t)(d.Nx1 + 1)*(d.Nx2 + 1)*(d.Nx3 + 1);
  size_t n = nc*(6 + 36 + 18 + 5 + 1 + 4);       // eta: 3 + the two edge arrays of k_correct_all
  // the one-kernel sub-cycle wants whole wavefronts along the rays; AA_ION_FUSED forces either path
  { const char *e = getenv("AA_ION_BEGIN_FUSED"); g->ion_begin_fused = e ? atoi(e) != 0 : true; }
  { const char *e = getenv("AA_ION_SPECULATE"); g->ion_spec_on = e ? atoi(e) != 0 : true; }
  { const char *e = getenv("AA_ION_FUSED");
    // (rays of 48 zones or more: the 52^3 level of the reference's own deck takes 6 sub-cycles of 4 launches + a read-back the other
    //  way, 3 + a read-back this way -- there the launches count, not the 12 idle lanes of a wavefront; round 3: 64)
    g->ion_fused = p->ion && (p->ion_path ? p->ion_path == 1 : (e ? atoi(e) != 0 : p->Nx[0] >= 48)); }
  const size_t nrays = (size_t)d.Nx2*d.Nx3;
  if (p->ion) n += nc*6 + nef;
  if (g->ion_fused) n += 2*nc + 2*nrays;
  if (p->order != 0 && p->order != 2 && p->order != 3) { delete g; return fail(-1, "[aa_create]: order %d (2: PLM, 3: PPM)", p->order); }
  if (p->order == 3) n += nc*18;
  g->pool_doubles = n;
  n += 16;
  hipError_t e = hipMalloc(&g->pool, n*sizeof(Real));
  if (e != hipSuccess) { delete g; return fail(-2, "[aa_create]: hipMalloc of %.2f GB failed: %s", n*8e-9, hipGetErrorString(e)); }
  g->bytes = (long long)(n*sizeof(Real));
  hipMemset(g->pool, 0, n*sizeof(Real));
  Real *q = g->pool + (pitch_align ? 12 : 0);
  d.U = q; q += 6*nc; d.LR = q; q += 36*nc; d.F = q; q += 18*nc; d.eta = q; q += 5*nc; d.dhalf = q; q += nc;
  d.phi = q; q += 4*nc;
  if (p->order == 3) { d.slope = q; q += 18*nc; }
  if (p->ion) {
    d.ph_rate = q; q += nc; d.kin = q; q += nc; d.vmax = q; q += nc;
    d.e_init = q; q += nc; d.x_init = q; q += nc;
    d.sign = (int2*)q; q += nc; d.edgeflux = q; q += nef;
    d.sg16 = (unsigned short*)d.sign;                  // the fused path keeps its 2-byte sign words in the same storage
    d.fin[0] = d.ph_rate; d.fin[1] = d.ph_rate;
    if (g->ion_fused) { d.fin[1] = q; q += nc; d.s_init = q; q += nc; d.raylast = q; q += 2*nrays; }
    IonPar &ip = g->ion;
    ip.sigma_ph = p->sigma_ph; ip.m_H = p->m_H; ip.mu = p->mu; ip.e_gamma = p->e_gamma; ip.alpha_C = p->alpha_C;
    ip.k_B = p->k_B; ip.time_unit = p->time_unit;
    ip.max_de_iter = p->max_de_iter; ip.max_de_therm_iter = p->max_de_therm_iter; ip.max_dx_iter = p->max_dx_iter;
    ip.max_de_step = p->max_de_step; ip.max_de_therm_step = p->max_de_therm_step; ip.max_dx_step = p->max_dx_step;
    ip.tfloor = p->tfloor; ip.tceil = p->tceil; ip.cour_no = p->cour_no;
    // ionrad.c:112-131: smallest face area and the "low" neutral density, from the ROOT dx
    // (the reference's fallback at :129 is dx[1], reproduced as is)
    Real a1 = rootdx[0]*rootdx[1], a2 = rootdx[0]*rootdx[2], a3 = rootdx[1]*rootdx[2];
    if (a1 < a2) ip.min_area = (a1 < a3) ? a1 : a3; else ip.min_area = (a2 < a3) ? a2 : a3;
    Real maxdx = rootdx[0] > rootdx[1] ? rootdx[0] : rootdx[1];
    maxdx = maxdx > rootdx[2] ? maxdx : rootdx[1];
    ip.d_nlo = 1.0e-4 * p->m_H / (p->sigma_ph * maxdx);      // MINOPTDEPTH, ionrad.h:29
    ip.inv_mH = 1.0/p->m_H; ip.inv_kB = 1.0/p->k_B; ip.aC14 = p->alpha_C/(14.0*p->m_H);
    ip.rec_floor = 2.59e-13*pow(p->tfloor/1.0e4, -0.7);      // recomb_rate_coef(tfloor), ionrad_chemistry.c:111
    ip.cx1 = p->max_dx_iter/(1 + p->max_dx_iter);
    ip.ce1 = p->max_de_therm_iter/(1 + p->max_de_therm_iter); ip.ce2 = p->max_de_iter/(1 + p->max_de_iter);
    ip.ie1 = 1.0/(1.0 + p->max_de_therm_iter); ip.ie2 = 1.0/(1.0 + p->max_de_iter);
    for (int a = 0; a < 3; a++) ip.inv_dx[a] = 1.0/d.dx[a];
    ip.iso = (d.dx[0] == d.dx[1] && d.dx[1] == d.dx[2]);
  }
  if (hipMalloc(&g->sc, sizeof(DevScalars)) != hipSuccess ||
      hipHostMalloc(&g->mb, sizeof(Mailbox), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
    hipFree(g->pool); delete g; return fail(-2, "[aa_create]: scalar buffers");
  }
  memset(g->mb, 0, sizeof(Mailbox));
  g->sc_host = &g->mb->s;
  if (hipHostGetDevicePointer((void**)&g->mb_dev, g->mb, 0) != hipSuccess) { (void)hipGetLastError(); g->mb_dev = nullptr; d.cfg.mailbox = 0; }
  if (g->ion_fused) {
    if (hipMalloc(&g->ion_part, (size_t)ion_pass_blocks(d)*sizeof(IonPart)) != hipSuccess ||
        hipMalloc(&g->ion_words, AA_ION_WORDS*sizeof(Real)) != hipSuccess) {
      hipFree(g->pool); hipFree(g->sc); hipHostFree(g->mb); delete g; return fail(-2, "[aa_create]: ion reduction buffers");
    }
  }
  // (every word of these is written before it is read; zeroed all the same, so that no run ever depends on what a freed
  //  allocation of an earlier Grid left behind)
  (void)hipMemset(g->sc, 0, sizeof(DevScalars));
  if (g->ion_part) (void)hipMemset(g->ion_part, 0, (size_t)ion_pass_blocks(d)*sizeof(IonPart));
  if (g->ion_words) (void)hipMemset(g->ion_words, 0, AA_ION_WORDS*sizeof(Real));
  hipStreamCreate(&g->st); g->own_stream = true;
  *out = g;
  return 0;
}

void aa_destroy(aa_grid *g)
{
  if (!g) return;
  if (g->link) { slabs_destroy(g); return; }          // a composite handle, whether or not it has slabs yet
  hipStreamSynchronize(g->st);
  prof_drain(g);
  hipFree(g->pool); hipFree(g->sc); hipHostFree(g->mb);
  for (hipEvent_t ev : g->ev_pool) hipEventDestroy(ev);
  if (g->ion_part) hipFree(g->ion_part);
  if (g->ion_words) hipFree(g->ion_words);
  if (g->pin_idx) hipFree(g->pin_idx);
  if (g->pin_val) hipFree(g->pin_val);
  if (g->pin_mask) hipFree(g->pin_mask);
  if (g->cfl_part) hipFree(g->cfl_part);
  if (g->d.phalf) hipFree(g->d.phalf);
  if (g->side) hipStreamDestroy(g->side);
  if (g->ev_fork) hipEventDestroy(g->ev_fork);
  if (g->ev_join) hipEventDestroy(g->ev_join);
  if (g->own_stream) hipStreamDestroy(g->st);
  delete g;
}

int aa_set_stream(aa_grid *g, void *s)
{
  if (!g->slab.empty()) return fail(-1, "[aa_set_stream]: a Grid cut into slabs runs on its slabs' own streams");
  hipStreamSynchronize(g->st);
  if (g->own_stream) { hipStreamDestroy(g->st); g->own_stream = false; }
  g->st = (hipStream_t)s;
  return 0;
}
int aa_sync(aa_grid *g) { if (!g->slab.empty()) return slabs_sync(g); HIPCHK(hipStreamSynchronize(g->st)); return 0; }
long long aa_device_bytes(const aa_grid *g) { return g->slab.empty() ? g->bytes : slabs_device_bytes(g); }

// ---- state transfer: staging through the (idle) face-state area ---------------------------
int aa_upload_cons(aa_grid *g, const double *U)
{
  g->cfl_ready = false; g->active_dirty = false;      // (host and device agree on the active zones from here on)
  g->inner_swept = false;          // the state changes under a pending aa_integrate_begin: its sweeps are redone
  if (!g->slab.empty()) return slabs_upload_cons(g, U);
  const int nvar = 5 + g->p.nscal; const size_t n = (size_t)g->d.N1*g->d.N2*g->d.N3*nvar;
  HIPCHK(hipMemcpyAsync(g->d.LR, U, n*sizeof(Real), hipMemcpyHostToDevice, g->st));
  launch_aos_to_soa(g->d, nvar, g->d.LR, g->st);
  HIPCHK(hipStreamSynchronize(g->st));
  return 0;
}
int aa_download_cons(aa_grid *g, double *U)
{
  if (!g->slab.empty()) return slabs_download_cons(g, U);
  g->inner_swept = false;          // the face-state area is the staging buffer (see aa_integrate_begin in athena_amd.h)
  const int nvar = 5 + g->p.nscal; const size_t n = (size_t)g->d.N1*g->d.N2*g->d.N3*nvar;
  launch_soa_to_aos(g->d, nvar, g->d.LR, g->st);
  HIPCHK(hipMemcpyAsync(U, g->d.LR, n*sizeof(Real), hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  g->active_dirty = false;
  return 0;
}
// Only the GHOST zones of the host block [N3][N2][N1][nvar] (the caller knows its active zones are current: nothing but
// boundary calls ran on the device since the block last travelled).  The shell is 9 % of a 256^3 block: six strided copies
// out of the staging area instead of the whole block over PCIe.
int aa_download_ghost_zones(aa_grid *g, double *U)
{
  if (!g->slab.empty()) return slabs_download_cons(g, U);
  // the caller says its active zones are current; the library knows whether any call since the last full transfer wrote active zones
  // (the integrators, the ion step, pinned zones, restriction / flux correction): then the whole block travels
  if (g->active_dirty) return aa_download_cons(g, U);
  g->inner_swept = false;
  const int nvar = 5 + g->p.nscal, N1 = g->d.N1, N2 = g->d.N2, N3 = g->d.N3, ng = AA_NGHOST;
  const size_t row = (size_t)N1*nvar*sizeof(Real), plane = row*N2;
  launch_soa_to_aos(g->d, nvar, g->d.LR, g->st);
  const char *S = (const char*)g->d.LR; char *D = (char*)U;
  // x3: the four planes at either end, whole
  HIPCHK(hipMemcpyAsync(D, S, ng*plane, hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipMemcpyAsync(D + (size_t)(N3 - ng)*plane, S + (size_t)(N3 - ng)*plane, ng*plane, hipMemcpyDeviceToHost, g->st));
  const size_t mid = (size_t)ng*plane;      // the planes in between
  // x2: four rows at either end of every such plane
  HIPCHK(hipMemcpy2DAsync(D + mid, plane, S + mid, plane, ng*row, N3 - 2*ng, hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipMemcpy2DAsync(D + mid + (size_t)(N2 - ng)*row, plane, S + mid + (size_t)(N2 - ng)*row, plane, ng*row, N3 - 2*ng, hipMemcpyDeviceToHost, g->st));
  // x1: four zones at either end of every row of those planes (the rows are one constant stride apart across j and k)
  const size_t w = (size_t)ng*nvar*sizeof(Real), nrows = (size_t)(N3 - 2*ng)*N2;
  HIPCHK(hipMemcpy2DAsync(D + mid, row, S + mid, row, w, nrows, hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipMemcpy2DAsync(D + mid + row - w, row, S + mid + row - w, row, w, nrows, hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  return 0;
}
// planes [k_first, k_first + nplanes) of the host block [N3][N2][N1][nvar]
int aa_download_cons_planes(aa_grid *g, int k_first, int nplanes, double *dst)
{
  if (!g->slab.empty()) return fail(-1, "[aa_download_cons_planes]: one slab at a time");
  if (k_first < 0 || nplanes < 0 || k_first + nplanes > g->d.N3) return fail(-1, "[aa_download_cons_planes]: planes %d..%d", k_first, k_first + nplanes);
  g->inner_swept = false;
  const int nvar = 5 + g->p.nscal; const size_t pl = (size_t)g->d.N1*g->d.N2*nvar;
  launch_soa_to_aos(g->d, nvar, g->d.LR, g->st);
  HIPCHK(hipMemcpyAsync(dst, g->d.LR + (size_t)k_first*pl, (size_t)nplanes*pl*sizeof(Real), hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  return 0;
}
int aa_download_edgeflux_planes(aa_grid *g, int nplanes, double *dst)
{
  if (!g->slab.empty()) return fail(-1, "[aa_download_edgeflux_planes]: one slab at a time");
  if (!g->p.ion) return fail(-1, "[aa_download_edgeflux]: ion radiation is off");
  if (nplanes < 0 || nplanes > g->d.Nx3 + 1) return fail(-1, "[aa_download_edgeflux_planes]: %d planes of %d", nplanes, g->d.Nx3 + 1);
  { int rc = aa_edgeflux_ready(g); if (rc) return rc; }
  const size_t n = (size_t)(g->d.Nx1 + 1)*(g->d.Nx2 + 1)*(size_t)nplanes;
  HIPCHK(hipMemcpyAsync(dst, g->d.edgeflux, n*sizeof(Real), hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  return 0;
}
int aa_upload_edgeflux(aa_grid *g, const double *ef)
{
  if (!g->slab.empty()) return slabs_upload_edgeflux(g, ef);
  if (!g->p.ion) return fail(-1, "[aa_upload_edgeflux]: ion radiation is off");
  g->ef_stale = false;
  const size_t n = (size_t)(g->d.Nx1 + 1)*(g->d.Nx2 + 1)*(g->d.Nx3 + 1);
  HIPCHK(hipMemcpyAsync(g->d.edgeflux, ef, n*sizeof(Real), hipMemcpyHostToDevice, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  return 0;
}
int aa_download_edgeflux(aa_grid *g, double *ef)
{
  if (!g->slab.empty()) return slabs_download_edgeflux(g, ef);
  if (!g->p.ion) return fail(-1, "[aa_download_edgeflux]: ion radiation is off");
  { int rc = aa_edgeflux_ready(g); if (rc) return rc; }
  const size_t n = (size_t)(g->d.Nx1 + 1)*(g->d.Nx2 + 1)*(g->d.Nx3 + 1);
  HIPCHK(hipMemcpyAsync(ef, g->d.edgeflux, n*sizeof(Real), hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  return 0;
}
int aa_get_mesh_state(const aa_grid *g, double *time, double *dt, int *nstep)
{ if (time) *time = g->time; if (dt) *dt = g->dt; if (nstep) *nstep = g->nstep; return 0; }
int aa_set_mesh_state(aa_grid *g, double time, double dt, int nstep)
{ g->time = time; g->dt = dt; g->nstep = nstep; if (!g->slab.empty()) slabs_push_state(g); return 0; }

// ---- hooks --------------------------------------------------------------------------------
int aa_set_static_grav_tables(aa_grid *g, const double *pc, const double *p1, const double *p2, const double *p3)
{
  if (!g->slab.empty()) return slabs_set_grav_tables(g, pc, p1, p2, p3);
  if (!pc) { g->grav = false; return 0; }
  // host tables are dense [N3][N2][N1]; device rows are sJ apart
  const double *src[4] = {pc, p1, p2, p3};
  for (int w = 0; w < 4; w++)
    HIPCHK(hipMemcpy2DAsync(g->d.phi + (size_t)w*g->d.nc, (size_t)g->d.sJ*sizeof(Real), src[w], (size_t)g->d.N1*sizeof(Real),
                            (size_t)g->d.N1*sizeof(Real), (size_t)g->d.N2*g->d.N3, hipMemcpyHostToDevice, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  g->grav = true;
  return 0;
}

int aa_set_static_grav_pot(aa_grid *g, aa_gravpot_fn fn)
{
  if (!fn) return aa_set_static_grav_tables(g, nullptr, nullptr, nullptr, nullptr);
  // (a Grid cut into slabs evaluates the callback at the positions of the caller's ONE Grid: the slabs then hold
  //  the very numbers the undivided Grid would)
  const int N1 = g->p.Nx[0] + 2*AA_NGHOST, N2 = g->p.Nx[1] + 2*AA_NGHOST, N3 = g->p.Nx[2] + 2*AA_NGHOST;
  double dx[3];
  for (int a = 0; a < 3; a++) dx[a] = ((g->p.xmax[a] - g->p.xmin[a])/(Real)(g->p.rootNx[a]))/(Real)(1 << g->p.level);
  std::vector<double> t[4];
  aa_eval_grav_tables(g->p, dx, N1, N2, N3, fn, t);
  return aa_set_static_grav_tables(g, t[0].data(), t[1].data(), t[2].data(), t[3].data());
}

// globals.h:25 CoolingFunc.  The reference calls the enrolled function per state on the host; on the device the function has to
// be one the library carries: AA_COOL_KOYINUT = KoyInut (microphysics/cool.c:48), the one the reference ships.  The CTU
// integrator then runs the second compilation of its kernels (namespace aa_cool: hydro_kernels.hip with -DAA_COOLING=1) and
// keeps P^{n+1/2} beside d^{n+1/2}.  The van Leer integrator of the reference has no cooling terms (integrate_3d_vl.c).
int aa_set_cooling(aa_grid *g, int kind)
{
  if (kind != AA_COOL_NONE && kind != AA_COOL_KOYINUT) return fail(-1, "[aa_set_cooling]: kind=%d: only AA_COOL_NONE and AA_COOL_KOYINUT", kind);
  if (kind && g->p.integrator == 1) return fail(-1, "[aa_set_cooling]: the van Leer integrator has no cooling terms in the reference (integrate_3d_vl.c)");
  if (!g->slab.empty() || g->link) return slabs_set_cooling(g, kind);
  g->inner_swept = false;
  if (kind && !g->d.phalf) {
    HIPCHK(hipMalloc(&g->d.phalf, (size_t)g->d.nc*sizeof(Real)));
    HIPCHK(hipMemsetAsync(g->d.phalf, 0, (size_t)g->d.nc*sizeof(Real), g->st));
    g->bytes += (long long)g->d.nc*sizeof(Real);
  }
  g->cool = kind;
  return 0;
}

int aa_set_pinned_cells(aa_grid *g, long long n, const long long *index, const double *values)
{
  if (!g->slab.empty()) return slabs_set_pinned_cells(g, n, index, values);
  if (g->pin_idx) { hipFree(g->pin_idx); g->pin_idx = nullptr; }
  if (g->pin_val) { hipFree(g->pin_val); g->pin_val = nullptr; }
  if (g->pin_mask) { hipFree(g->pin_mask); g->pin_mask = nullptr; }
  g->npin = 0;
  g->cfl_ready = false; g->active_dirty = true;
  if (n <= 0) return 0;
  const int nvar = 5 + g->p.nscal;
  HIPCHK(hipMalloc(&g->pin_idx, (size_t)n*sizeof(long long)));
  HIPCHK(hipMalloc(&g->pin_val, (size_t)n*nvar*sizeof(Real)));
  HIPCHK(hipMemcpy(g->pin_idx, index, (size_t)n*sizeof(long long), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g->pin_val, values, (size_t)n*nvar*sizeof(Real), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&g->pin_mask, (size_t)g->d.nc));
  HIPCHK(hipMemsetAsync(g->pin_mask, 0, (size_t)g->d.nc, g->st));
  launch_pin_mask(g->d, n, g->pin_idx, g->pin_mask, g->st);
  g->npin = n;
  return 0;
}
int aa_apply_pinned_cells(aa_grid *g)
{
  if (!g->slab.empty()) return slabs_apply_pinned_cells(g);
  Scope s(g, "pinned_cells");
  // (cfl_ready: the zones were left out of the integrator's maxima and join them here, from the values written -- one launch;
  //  AA_PIN_ONE=0: k_pinned + k_pinned_cfl, which reads them back)
  if (g->cfl_ready && g->d.cfg.pin_one) launch_pinned(g->d, 5 + g->p.nscal, g->npin, g->pin_idx, g->pin_val, g->st, g->sc);
  else {
    launch_pinned(g->d, 5 + g->p.nscal, g->npin, g->pin_idx, g->pin_val, g->st);
    if (g->cfl_ready) launch_pinned_cfl(g->d, g->npin, g->pin_idx, g->sc, g->st);
  }
  return 0;
}

int aa_add_radplane_3d(aa_grid *g, int dir, double flux)
{
  if (!g->p.ion) return fail(-1, "[add_radplane_3d]: ion radiation is off");
  // Rays along +x1 (dir = -1) and along +x2 (dir = -2).  The reference has no working behaviour for the others:
  // right-to-left rays (dir > 0) never enter the loop `for (i=s; i<=e; i+=lr)` with s > e (ionradplane_3d.c:275, :335,
  // :367), and dir = -3 divides by a cell_len that is only assigned in the right-to-left branch (:136-145).
  if (dir != -1 && dir != -2)
    return fail(-1, "[add_radplane_3d]: dir=%d: only -1 (rays along +x1) and -2 (along +x2) have defined behaviour in the reference", dir);
  if (dir == -2) {
    if (g->p.level > 0) return fail(-1, "[add_radplane_3d]: rays along +x2 on a refined level are not supported");
    g->ion_fused = false;          // the two-kernel sub-cycle carries this sweep
  }
  if (!g->slab.empty()) { int rc = slabs_add_radplane(g, dir, flux); if (rc) return rc; }
  g->rad_dir = dir; g->flux_i = flux; g->nradplane = 1;
  return 0;
}

int aa_has_radplane(const aa_grid *g) { return (g->p.ion && g->nradplane > 0) ? 1 : 0; }   // main.c:546: the ion step runs iff nradplane > 0

// ---- per-step call sites ------------------------------------------------------------------
int aa_bvals_mhd(aa_grid *g)
{
  if (!g->slab.empty()) return slabs_bvals_mhd(g);
  Scope s(g, "bvals_mhd");
  if (g->d.cfg.bc_one) { launch_bc_shell(g->d, g->p.nscal, g->p.bc, g->st); return 0; }      // the three passes as one launch: same bits
  for (int d = 0; d < 3; d++)            // x1, x2, x3 so the corners fill (bvals_mhd.c:170)
    launch_bc_dir(g->d, g->p.nscal, d, g->p.bc[2*d], g->p.bc[2*d + 1], g->st);
  return 0;
}

int aa_bvals_mhd_side(aa_grid *g, int dir, int side)
{
  if (dir < 0 || dir > 2 || side < 0 || side > 1) return fail(-1, "[aa_bvals_mhd_side]: dir=%d side=%d", dir, side);
  if (!g->slab.empty()) return slabs_bvals_mhd_side(g, dir, side);
  const int flag = g->p.bc[2*dir + side];
  if (flag) { Scope s(g, "bvals_mhd"); launch_bc(g->d, g->p.nscal, dir, side, flag, g->st); }
  return 0;
}

int aa_bvals_ionrad(aa_grid *g)
{
  if (!g->slab.empty()) return slabs_bvals_ionrad(g);
  if (!g->p.ion) return 0;
  if (g->rad_dir == -1) launch_edgeflux_bc(g->d, g->flux_i, g->st);
  else if (g->rad_dir == -2) launch_edgeflux_bc_x2(g->d, g->flux_i, g->st);
  return 0;
}

// DevScalars device -> host.  The mailbox way (grid.h Mailbox; AA_MAILBOX=0: a copy + a stream wait): a one-wave kernel behind the
// producers writes the scalars into pinned host memory and stamps them; the host polls the stamp -- for a while: a wait that
// outlasts AA_MAILBOX_SPIN_US (a long kernel in front) ends in hipStreamSynchronize like the other way, so no core spins for
// milliseconds.  On the small Grids of the reference's own decks a step is a few dozen launches and 2 + N_sub read-backs, and a
// copy + wait costs ~25 us against ~8 for a launch + poll.
int aa_fetch_scalars(aa_grid *g)
{
  if (!g->slab.empty()) return fail(-1, "[aa_fetch_scalars]: not available on a Grid cut into slabs");
  g->host_syncs++;
  if (g->d.cfg.mailbox && g->mb_dev) {
    unsigned long long seq;
    if (g->mb_prepublished && g->mb_prepublished == g->mb_seq) { seq = g->mb_seq; g->mb_prepublished = 0; }      // (k_ion_reduce_pick has published them)
    else { seq = ++g->mb_seq; launch_publish(g->sc, g->mb_dev, seq, g->st); }
    volatile unsigned long long *stamp = &g->mb->seq;
    struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 1;; spins++) {
      if (__atomic_load_n(stamp, __ATOMIC_ACQUIRE) == seq) return 0;
      if ((spins & 127u) == 0) {
        struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec)*1000000L + (t1.tv_nsec - t0.tv_nsec)/1000L > g->d.cfg.mailbox_spin_us) break;
      }
      __builtin_ia32_pause();
    }
    HIPCHK(hipStreamSynchronize(g->st));          // (also where a kernel in front failed: the error comes out here)
    if (__atomic_load_n(stamp, __ATOMIC_ACQUIRE) != seq) return fail(-2, "[aa_fetch_scalars]: the mailbox kernel finished without its stamp");
    return 0;
  }
  HIPCHK(hipMemcpyAsync(g->sc_host, g->sc, sizeof(DevScalars), hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  return 0;
}
#define fetch_scalars aa_fetch_scalars

int aa_new_dt_local(aa_grid *g, double *dt_cfl)
{
  if (!g->slab.empty()) return slabs_new_dt_local(g, dt_cfl);
  if (g->cfl_ready) g->cfl_ready = false;        // the integrator (+ aa_apply_pinned_cells) has left the maxima in g->sc
  else { Scope s(g, "new_dt");
    HIPCHK(hipMemsetAsync(g->sc->max_v, 0, 3*sizeof(unsigned long long), g->st));
    launch_cfl(g->d, g->sc, g->st); }
  int rc = fetch_scalars(g); if (rc) return rc;
  double max_dti = 0.0;                   // new_dt.c:159-166
  for (int d = 0; d < 3; d++) { double v = bits_to_double(g->sc_host->max_v[d])/g->d.dx[d]; max_dti = (max_dti > v) ? max_dti : v; }
  *dt_cfl = g->p.cour_no/max_dti;
  return 0;
}

int aa_cfl_max_v(aa_grid *g, double *v)      // new_dt.c:72-140 of this Grid: max(|v_d| + a) per direction
{
  if (!g->slab.empty()) return slabs_cfl_max_v(g, v);
  if (g->cfl_ready) g->cfl_ready = false;        // the integrator (+ aa_apply_pinned_cells) has left the maxima in g->sc
  else { Scope s(g, "new_dt");
    HIPCHK(hipMemsetAsync(g->sc->max_v, 0, 3*sizeof(unsigned long long), g->st));
    launch_cfl(g->d, g->sc, g->st); }
  int rc = fetch_scalars(g); if (rc) return rc;
  for (int d = 0; d < 3; d++) v[d] = bits_to_double(g->sc_host->max_v[d]);
  return 0;
}

int aa_new_dt(aa_grid *g)
{
  double dtc; int rc = aa_new_dt_local(g, &dtc); if (rc) return rc;
  if (g->nstep == 0) g->dt = dtc; else g->dt = (2.0*g->dt < dtc) ? 2.0*g->dt : dtc;       // new_dt.c:169-173
  if ((g->time < g->p.tlim) && ((g->p.tlim - g->time) < g->dt)) g->dt = g->p.tlim - g->time;   // :183-185
  if (!g->slab.empty()) slabs_push_state(g);
  return 0;
}

// k_correct_all also does the x3 first pass (no k_sweep_march<2> launch, those fluxes never in HBM): -3.1 ms of a 512^3
// blast step, -4.2 ms ifront, -4.4 / -2.0 ms third order without / with gravity.  Second order + passive scalar + gravity
// (ioniz_sphere: 249 VGPRs at 2 waves per SIMD) was a draw in round 2 and left to the sweep kernel; with the hardware
// min / max in the reconstruction (AA_FD_MINMAX: 12 % fewer instructions in this kernel) it is ahead there too --
// 19.5 against 17.2 + 4.55 ms, hydro chain 45.5 -> 43.3 (round 3, same-box ABAB) -- and is the default everywhere.
// Same bits either way.
// aa_cfl_in_update: zero the maxima and let k_flux2_update fill them
static void cfl_arm(aa_grid *g)
{
  g->cfl_ready = g->cfl_in_update && (g->npin == 0 || g->pin_mask);
  if (g->cfl_ready) (void)hipMemsetAsync(g->sc->max_v, 0, 3*sizeof(unsigned long long), g->st);
}
int aa_cfl_in_update(aa_grid *g, int on)
{
  if (!g->slab.empty()) return 0;      // (composite Grids keep k_cfl)
  // (until round 4 the strict build kept k_cfl: with -ffp-contract=off its 6-variable update kernel spilled 15 registers with the
  //  CFL epilogue; with the lower x3 flux parked in LDS and dt/dx in scalar registers it holds 250 without scratch, and the strict
  //  step is 48.1 -> 47.1 ms at 512^3 with the maxima from the update kernel.  AA_CFL_FUSED=0: k_cfl; =2: as 1, kept for the tests)
  g->cfl_in_update = on != 0; g->cfl_ready = false;
  return 0;
}

static bool x3_fused(const aa_grid *g)      // default: always (round 3; until then not for second order + scalar + gravity)
{ return g->x3_fused_mode >= 0 ? g->x3_fused_mode != 0 : true; }

// The part of the step that needs none of the x3 neighbours' planes: the first-pass x1 and x2 sweeps of the k-planes
// ks .. ke (a pencil along x1 or x2 lies in one plane).  A multi-GPU caller posts the x3 halo, calls this, waits for the
// halo, unpacks it and calls aa_integrate_3d_ctu, which then sweeps only the four ghost planes in x1 / x2: the messages
// travel under ~10 ms of kernels at 512^3.  Same bits as without the call.  Does nothing (returns 0) where the split
// does not apply: composite Grids, third order (the slope arrays come first), the unfused correct / update chains, VL.
// the kernels of a run with a cooling function are the second compilation of hydro_kernels.hip (namespace aa_cool)
#define HL(f) (g->cool ? aa_cool::f : aa::f)
// ... and with it the x1 first pass (round 4: hydro_kernels.hip CA_X1F): no k_sweep_x1_flat launch before k_correct_all
static bool x1_fused(const aa_grid *g) { return g->correct_all && x3_fused(g) && HL(ca_x1_on_board)(); }
int aa_integrate_begin(aa_grid *g)
{
  if (!g->slab.empty() || g->p.integrator != 0 || g->d.slope || !g->correct_all || !g->fused_update || g->inner_swept) return 0;
  const HostGrid &d = g->d; const int ns = g->p.nscal; const Real dt = g->dt;
  const int nk = d.ke - d.ks + 1;
  { Scope s(g, "sweep_x2"); HL(launch_sweep)(d, ns, 1, dt, g->grav, g->st, 2, nk); }
  if (!x1_fused(g)) { Scope s(g, "sweep_x1"); HL(launch_sweep)(d, ns, 0, dt, g->grav, g->st, 2, nk); }
  g->inner_swept = true;
  g->inner_dt = dt;
  HIPCHK(hipGetLastError());
  return 0;
}

// aa_params.integrator 2: the CTU integrator of a reference built WITHOUT --enable-h-correction (its configure default).  Such a
// build has no eta arrays and roe.c uses |ev| where the H_CORRECTION build uses MAX(|ev|, etah) (roe.c:282-290): the numbers of
// etah = 0.  The correct kernels have just filled the etas; zero them before the second-pass fluxes read them (three fields:
// ~1 ms at 512^3 in this mode only; the kernels of the default mode carry no flag for it).
static void no_h_correction(aa_grid *g)
{
  if (g->p.integrator != 2) return;
  Scope s(g, "no_h_correction");
  (void)hipMemsetAsync(g->d.eta, 0, (size_t)3*g->d.nc*sizeof(Real), g->st);
}

int aa_integrate_3d_ctu(aa_grid *g)
{
  if (!g->slab.empty()) return slabs_integrate(g, 0);
  g->cfl_ready = false; g->active_dirty = true;
  const HostGrid &d = g->d; const int ns = g->p.nscal; const Real dt = g->dt;
  if (g->inner_swept) {      // aa_integrate_begin did the planes ks .. ke: the two ghost planes either side remain
    if (g->inner_dt != dt) return fail(-1, "[aa_integrate_3d_ctu]: dt changed after aa_integrate_begin");
    g->inner_swept = false;
    const int nk = d.ke - d.ks + 1;
    { Scope s(g, "sweep_x2"); HL(launch_sweep)(d, ns, 1, dt, g->grav, g->st, 0, 2); HL(launch_sweep)(d, ns, 1, dt, g->grav, g->st, 2 + nk, 2); }
    if (!x1_fused(g)) { Scope s(g, "sweep_x1"); HL(launch_sweep)(d, ns, 0, dt, g->grav, g->st, 0, 2); HL(launch_sweep)(d, ns, 0, dt, g->grav, g->st, 2 + nk, 2); }
    if (!x3_fused(g)) { Scope s(g, "sweep_x3"); HL(launch_sweep)(d, ns, 2, dt, g->grav, g->st, 0, -1); }
    { Scope s(g, "correct_all"); HL(launch_correct_all)(d, ns, dt, g->grav, x3_fused(g), g->st, false); }
    no_h_correction(g);
    Scope s(g, "flux2_update");
    cfl_arm(g);
    HL(launch_flux2_update)(d, ns, dt, g->grav, g->keep_flux ? &g->keep : nullptr, g->st, g->cfl_ready ? g->sc : nullptr, g->pin_mask);
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (d.slope) { Scope s(g, "ppm_slopes"); for (int dir = 0; dir < 3; dir++) HL(launch_slopes)(d, ns, dir, g->st, nullptr); }
  // AA_EDGE_OVERLAP=1: the tile-edge x1 fluxes k_correct_all needs (k_x1_edge_flux: a few bytes per zone, bound by its strided loads)
  // beside the x2 sweep (bound by its arithmetic) on a stream of their own; both read U only.  -0.27 ms of a 41.4 ms step at 512^3
  // (profiles/r04_x1f_ab.txt item 11) -- off by default: it is inside the box-to-box spread, and with two kernels in flight at once the
  // per-kernel durations a profiler reports are no longer each kernel's own
  bool edges_aside = false;
  if (x1_fused(g) && d.cfg.edge_overlap) {
    if (!g->side) {
      HIPCHK(hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(g->ev_fork, g->st));
    HIPCHK(hipStreamWaitEvent(g->side, g->ev_fork, 0));
    HL(launch_x1_edges)(d, ns, dt, g->grav, g->side);
    HIPCHK(hipEventRecord(g->ev_join, g->side));
    edges_aside = true;
  }
  // x2 and x3 first, so that the x1 sweep can do its first pass and its correct pass in one go
  { Scope s(g, "sweep_x2"); HL(launch_sweep)(d, ns, 1, dt, g->grav, g->st, 0, -1); }
  if (edges_aside) HIPCHK(hipStreamWaitEvent(g->st, g->ev_join, 0));
  if (!(g->correct_all && x3_fused(g))) { Scope s(g, "sweep_x3"); HL(launch_sweep)(d, ns, 2, dt, g->grav, g->st, 0, -1); }
  if (g->correct_all) {
    if (!x1_fused(g)) { Scope s(g, "sweep_x1"); HL(launch_sweep)(d, ns, 0, dt, g->grav, g->st, 0, -1); }
    { Scope s(g, "correct_all"); HL(launch_correct_all)(d, ns, dt, g->grav, x3_fused(g), g->st, edges_aside); }
  } else {
    { Scope s(g, "sweep_correct_x1"); HL(launch_sweep_correct_x1)(d, ns, dt, g->grav, g->st); }
    { Scope s(g, "correct_x2"); HL(launch_correct)(d, ns, 1, dt, g->grav, g->st); }
    { Scope s(g, "correct_x3"); HL(launch_correct)(d, ns, 2, dt, g->grav, g->st); }
  }
  no_h_correction(g);
  if (g->fused_update) {
    Scope s(g, "flux2_update");
    cfl_arm(g);
    HL(launch_flux2_update)(d, ns, dt, g->grav, g->keep_flux ? &g->keep : nullptr, g->st, g->cfl_ready ? g->sc : nullptr, g->pin_mask);
  } else {
    { Scope s(g, "flux2_x1"); HL(launch_flux2)(d, ns, 0, g->st); }
    { Scope s(g, "flux2_x2"); HL(launch_flux2)(d, ns, 1, g->st); }
    { Scope s(g, "flux2_x3"); HL(launch_flux2)(d, ns, 2, g->st); }
    { Scope s(g, "update");   HL(launch_update)(d, ns, d.dhalf, dt, g->grav, g->st, nullptr, nullptr, nullptr); }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

#undef HL
int aa_integrate_3d_vl(aa_grid *g)
{
  if (!g->slab.empty()) return slabs_integrate(g, 1);
  g->cfl_ready = false; g->active_dirty = true;
  // integrate_3d_vl.c:96-: donor-cell fluxes -> U^{n+1/2} -> PLM or PPM (no tracing) + Roe -> update
  const HostGrid &d = g->d; const int ns = g->p.nscal; const Real dt = g->dt;
  // donor-cell fluxes + U^{n+1/2} in one marching kernel from 2^18 zones (512^3: 16.8 -> 9.7 ms; same at 80^3;
  // 10 % slower at 32^3); AA_VL_PREDICT forces either, the results are the same bit for bit
  if (g->vl_predict) { Scope s(g, "vl_predict"); launch_vl_predict(d, ns, dt, g->grav, g->st); }
  else {
    { Scope s(g, "vl_flux1"); for (int dir = 0; dir < 3; dir++) launch_vl_flux1(d, ns, dir, g->st); }
    { Scope s(g, "vl_uhalf"); launch_vl_uhalf(d, ns, dt, g->grav, g->st); }
  }
  // --with-order=3: the corrector reconstructs U^{n+1/2} with parabolae (no tracing, lr_states_ppm.c:502-507): their slopes first
  if (d.slope) { Scope s(g, "ppm_slopes"); for (int dir = 0; dir < 3; dir++) launch_slopes(d, ns, dir, g->st, d.LR); }
  { Scope s(g, "vl_flux2_x1"); launch_vl_flux2(d, ns, 0, dt, g->st); }
  { Scope s(g, "vl_flux2_x2"); launch_vl_flux2(d, ns, 1, dt, g->st); }
  { Scope s(g, "vl_flux2_x3"); launch_vl_flux2(d, ns, 2, dt, g->st); }
  {
    Scope s(g, "update");
    cfl_arm(g);                      // aa_cfl_in_update: new_dt's maxima ride on the update, as in k_flux2_update
    if (g->cfl_ready) {
      const long nb = update_blocks(d);
      if (g->cfl_part_n < 3*nb) {
        if (g->cfl_part) { (void)hipFree(g->cfl_part); g->cfl_part = nullptr; g->cfl_part_n = 0; }
        if (hipMalloc(&g->cfl_part, (size_t)3*nb*sizeof(Real)) == hipSuccess) g->cfl_part_n = 3*nb; else g->cfl_ready = false; g->active_dirty = true;
      }
    }
    launch_update(d, ns, d.LR, dt, g->grav, g->st, g->cfl_ready ? g->sc : nullptr, g->cfl_part, g->pin_mask);   // d^{n+1/2} = Uhalf.d
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int aa_ion_begin(aa_grid *g)
{
  g->cfl_ready = false; g->active_dirty = true;
  if (!g->p.ion) return fail(-1, "[ion_radtransfer]: ion radiation is off");
  if (!g->slab.empty()) return slabs_ion_begin(g);
  if (g->ion_fused) {
    // (a sweep of the previous ion step that nobody asked for is dropped here: its buffers are about to be reused)
    g->ion_cur = 0; g->ion_pending = false; g->ef_stale = false;
    g->ion_spec_dt = -1.0; g->ion_spec_armed = false;
    if (g->ion_begin_fused) { g->ion_begin_due = true; return 0; }       // rides on the first pass
    Scope s(g, "ion_begin");
    launch_ion_begin16(g->d, g->ion, g->st);
    return 0;
  }
  Scope s(g, "ion_begin");
  launch_ion_begin(g->d, g->ion, g->st);
  return 0;
}

int aa_ion_rates(aa_grid *g, double *dt_chem, double *dt_therm)
{
  if (!g->slab.empty()) return slabs_ion_rates(g, dt_chem, dt_therm);
  if (g->ion_fused) return fail(-1, "[aa_ion_rates]: this Grid runs the one-kernel sub-cycle (aa_ion_pass / aa_ion_pick / aa_ion_fetch)");
  {
    DevScalars init; memset(&init, 0, sizeof init);
    init.dt_chem = double_to_bits(DBL_MAX); init.dt_therm = double_to_bits(DBL_MAX);
    *g->sc_host = init;
    HIPCHK(hipMemcpyAsync(g->sc, g->sc_host, sizeof(DevScalars), hipMemcpyHostToDevice, g->st));
  }
  if (g->nradplane > 0) {
    // ionradplane_3d.c:265: the hard-coded time ramp of the incident flux, evaluated once on the
    // host (it is the same for every ray of the root level)
    const Real flux0 = g->flux_i*(5.*(erf((g->time - 1.2e5)/8e4)+1)+0.1);
    // :264-271: a refined level starts every ray from the flux its parent left in EdgeFlux[..][..][0]
    if (g->rad_dir == -2) { Scope s(g, "ray_sweep"); launch_ray_sweep_x2(g->d, g->ion, g->flux_i, g->st); }
    else if (g->fused_rates) { Scope s(g, "ray_sweep_rates"); launch_ray_sweep_rates(g->d, g->ion, flux0, g->level > 0, g->sc, g->st); }
    else { Scope s(g, "ray_sweep"); launch_ray_sweep(g->d, g->ion, flux0, g->level > 0, g->st); }
  } else {
    HIPCHK(hipMemsetAsync(g->d.ph_rate, 0, (size_t)g->d.nc*sizeof(Real), g->st));
  }
  if (!(g->fused_rates && g->nradplane > 0 && g->rad_dir == -1)) { Scope s(g, "ion_rates"); launch_ion_rates(g->d, g->ion, g->sc, g->st); }
  int rc = fetch_scalars(g); if (rc) return rc;
  if (g->sc_host->neg_dt_chem) return fail(-4, "[compute_chem_rates]: negative dt_chem");   // ionrad_3d.c:389-391
  *dt_chem = bits_to_double(g->sc_host->dt_chem);
  *dt_therm = bits_to_double(g->sc_host->dt_therm);
  return 0;
}

int aa_ion_update(aa_grid *g, double dt, long long *cellcount, double *dt_hydro)
{
  g->cfl_ready = false; g->active_dirty = true;
  if (!g->slab.empty()) return slabs_ion_update(g, dt, cellcount, dt_hydro);
  if (g->ion_fused) return fail(-1, "[aa_ion_update]: this Grid runs the one-kernel sub-cycle (aa_ion_pass / aa_ion_pick / aa_ion_fetch)");
  HIPCHK(hipMemsetAsync(&g->sc->max_dti, 0, 2*sizeof(unsigned long long), g->st));
  { Scope s(g, "ion_update"); launch_ion_update(g->d, g->ion, dt, g->sc, g->st); }
  int rc = fetch_scalars(g); if (rc) return rc;
  if (cellcount) *cellcount = (long long)g->sc_host->cellcount;
  if (dt_hydro) *dt_hydro = g->p.cour_no/bits_to_double(g->sc_host->max_dti);
  return 0;
}

// One sub-cycle of ion_radtransfer_3d with the step chosen on the device (ionrad_3d.c:919-971): ray
// sweep, rates, k_ion_pick, update -- and ONE read-back.  `limit` is the hydro dt (root) or the coarse
// time (refined level).  The first call of an ion step must find the reductions armed (aa_ion_arm).
int aa_ion_arm(aa_grid *g)
{
  if (!g->slab.empty()) return fail(-1, "[aa_ion_arm]: not available on a Grid cut into slabs (aa_ion_run / aa_ion_rates + aa_ion_update)");
  DevScalars init; memset(&init, 0, sizeof init);
  init.dt_chem = double_to_bits(DBL_MAX); init.dt_therm = double_to_bits(DBL_MAX);
  *g->sc_host = init;
  HIPCHK(hipMemcpyAsync(g->sc, g->sc_host, sizeof(DevScalars), hipMemcpyHostToDevice, g->st));
  return 0;
}
int aa_ion_subcycle(aa_grid *g, double dt_done, double limit, double *dt, int *limit_hit, double *dt_chem,
                    double *dt_therm, long long *cellcount, double *dt_hydro)
{
  g->cfl_ready = false; g->active_dirty = true;
  if (!g->slab.empty()) return fail(-1, "[aa_ion_subcycle]: not available on a Grid cut into slabs (aa_ion_run / aa_ion_rates + aa_ion_update)");
  if (g->ion_fused) return fail(-1, "[aa_ion_subcycle]: this Grid runs the one-kernel sub-cycle");
  if (g->nradplane > 0) {
    const Real flux0 = g->flux_i*(5.*(erf((g->time - 1.2e5)/8e4)+1)+0.1);     // ionradplane_3d.c:265
    if (g->rad_dir == -2) { Scope s(g, "ray_sweep"); launch_ray_sweep_x2(g->d, g->ion, g->flux_i, g->st); }
    else if (g->fused_rates) { Scope s(g, "ray_sweep_rates"); launch_ray_sweep_rates(g->d, g->ion, flux0, g->level > 0, g->sc, g->st); }
    else { Scope s(g, "ray_sweep"); launch_ray_sweep(g->d, g->ion, flux0, g->level > 0, g->st); }
  } else {
    HIPCHK(hipMemsetAsync(g->d.ph_rate, 0, (size_t)g->d.nc*sizeof(Real), g->st));
  }
  if (!(g->fused_rates && g->nradplane > 0 && g->rad_dir == -1)) { Scope s(g, "ion_rates"); launch_ion_rates(g->d, g->ion, g->sc, g->st); }
  launch_ion_pick(g->sc, dt_done, limit, g->st);
  { Scope s(g, "ion_update"); launch_ion_update_sel(g->d, g->ion, g->sc, g->st); }
  int rc = fetch_scalars(g); if (rc) return rc;
  if (g->sc_host->neg_out) return fail(-4, "[compute_chem_rates]: negative dt_chem");   // ionrad_3d.c:389-391
  *dt = g->sc_host->dt_sel; *limit_hit = g->sc_host->limit_hit;
  *dt_chem = g->sc_host->dt_chem_out; *dt_therm = g->sc_host->dt_therm_out;
  *cellcount = (long long)g->sc_host->cellcount;
  *dt_hydro = g->p.cour_no/bits_to_double(g->sc_host->max_dti);
  return 0;
}

// ---- the one-kernel sub-cycle (ion_pass.hip), as phases: a driver that reduces over ranks puts ONE collective
// (an all-gather of AA_ION_WORDS doubles per rank, on this Grid's stream) between aa_ion_pass and aa_ion_pick
int aa_ion_is_fused(const aa_grid *g) { return g->ion_fused ? 1 : 0; }

// pass n: update(n-1) with the dt aa_ion_pick chose (if `update`), then sweep(n) + rates(n) (if `sweep`); this
// Grid's words of the reduction go to dev_words (DEVICE, AA_ION_WORDS doubles; NULL: kept in the Grid)
int aa_ion_pass(aa_grid *g, int update, int sweep, double *dev_words)
{
  g->cfl_ready = false; g->active_dirty = true;
  if (!g->ion_fused) return fail(-1, "[aa_ion_pass]: this Grid runs the two-kernel sub-cycle (aa_ion_rates / aa_ion_update)");
  if (!g->slab.empty()) {
    if (dev_words) return fail(-1, "[aa_ion_pass]: a Grid cut into slabs reduces over its slabs itself");
    return slabs_ion_pass(g, update, sweep);
  }
  if (!update && !sweep) return fail(-1, "[aa_ion_pass]: nothing to do");
  g->mb_prepublished = 0;
  if (update) {
    // relying on the rates of the previous sweep makes that sweep the one that counts
    if (!g->ion_pending) return fail(-1, "[aa_ion_pass]: update without a preceding sweep");
    g->ion_cur ^= 1; g->ion_pending = false;
  }
  // ionradplane_3d.c:265: the hard-coded time ramp of the incident flux (the same for every ray of the root
  // level); :264-271: a refined level starts every ray from the flux its parent left in EdgeFlux[..][..][0];
  // no radiation plane: flux 0 (every zone's ph_rate stays 0, as after ph_rate_init)
  const Real flux0 = (g->nradplane > 0) ? g->flux_i*(5.*(erf((g->time - 1.2e5)/8e4)+1)+0.1) : 0.0;
  const bool begin = g->ion_begin_due;
  if (begin && update) return fail(-1, "[aa_ion_pass]: the first pass of an ion step cannot apply an update");
  g->ion_begin_due = false;
  // aa_ion_speculate: the first pass also applies the first sub-cycle's update with the whole step (k_ion_pass)
  const Real spec_dt = (begin && g->ion_spec_dt >= 0.0) ? g->ion_spec_dt : -1.0;
  if (!update) g->ion_spec_armed = spec_dt >= 0.0;
  g->ion_spec_dt = -1.0;
  { Scope s(g, update ? (sweep ? "ion_pass" : "ion_pass_last") : (begin ? "ion_pass_begin" : "ion_pass_first"));
    launch_ion_pass(g->d, g->ion, update != 0, sweep != 0, begin, flux0, g->level > 0 && g->nradplane > 0, g->sc, g->ion_cur,
                    g->ion_part, dev_words ? dev_words : g->ion_words, g->st, spec_dt, !g->ion_fuse_pick); }
  if (sweep) g->ion_pending = true;
  HIPCHK(hipGetLastError());
  return 0;
}

// Before the first pass of an ion step (after aa_ion_begin): `limit` is the step the sub-cycles have to cover (the hydro
// step, or what is left of the coarse time on a refined level -- the value aa_ion_pick will be given).  The first pass then
// also applies the first sub-cycle's update with that whole step; where the reduction confirms it -- the stationary regime,
// ONE sub-cycle per step -- the closing update pass has nothing left to do (k_ion_pass).  Results are the same bit for bit.
int aa_ion_speculate(aa_grid *g, double limit)
{
  if (!g->ion_fused || !g->ion_spec_on) return 0;
  if (!g->slab.empty()) { for (aa_grid *c : g->slab) { c->ion_spec_dt = c->ion_begin_due ? limit : -1.0; c->ion_spec_limit = limit; } return 0; }
  g->ion_spec_dt = g->ion_begin_due ? limit : -1.0;      // (only the pass that also does the step's entry can speculate)
  g->ion_spec_limit = limit;
  return 0;
}

// ionrad_3d.c:941-967 from the words of all ranks (DEVICE, nranks x AA_ION_WORDS doubles; NULL: this Grid's own): books
// the update the pass just applied (its step, the time covered so far -- kept on the device) and picks the step of the
// next one, cut back to `limit`; `first` = the pass was the first of the ion step (no update applied yet)
int aa_ion_pick(aa_grid *g, const double *dev_words_all, int nranks, int first, double limit)
{
  if (!g->ion_fused) return fail(-1, "[aa_ion_pick]: this Grid runs the two-kernel sub-cycle");
  if (!g->slab.empty()) return slabs_ion_pick(g, first, limit);
  if (first && g->ion_spec_armed && limit != g->ion_spec_limit)
    return fail(-1, "[aa_ion_pick]: limit %.17g, but aa_ion_speculate was told %.17g", limit, g->ion_spec_limit);
  if (g->ion_fuse_pick) {      // (ion_run_fused, one rank: the pass left its records unfolded)
    if (dev_words_all) return fail(-1, "[aa_ion_pick]: internal: a fused pick with gathered words");
    // (not the first pick of an ion step: aa_ion_fetch follows at once -- the kernel publishes the scalars itself)
    const bool pub = !first && g->d.cfg.mailbox && g->mb_dev;
    if (pub) g->mb_prepublished = ++g->mb_seq;
    launch_ion_reduce_pick(g->d, g->ion_part, g->ion_words, g->sc, first, limit, g->st, first ? (g->ion_spec_armed ? 1 : 0) : 0,
                           pub ? g->mb_dev : nullptr, pub ? g->mb_seq : 0ULL);
  } else
  launch_ion_pick2(dev_words_all ? dev_words_all : g->ion_words, dev_words_all ? nranks : 1, g->sc, first, limit, g->st,
                   first ? (g->ion_spec_armed ? 1 : 0) : 0);
  if (!first) g->ion_spec_armed = false;
  HIPCHK(hipGetLastError());
  return 0;
}

// the ONE read-back of a sub-cycle, after pass + pick: of the update that pass applied its step `dt`, whether that step
// was cut back to the limit, the stop criteria's operands (cells out of range, dt_hydro) and `neg` = compute_chem_rates'
// negative-dt_chem flag of the rates the step came from; dt_chem / dt_therm are those of the NEXT step (diagnostics)
int aa_ion_fetch(aa_grid *g, double *dt, int *limit_hit, double *dt_chem, double *dt_therm, long long *cellcount, double *dt_hydro, int *neg)
{
  // (a Grid cut into slabs: every slab has picked from the gathered words; slab 0's scalars come back)
  { int rc = g->slab.empty() ? fetch_scalars(g) : slabs_fetch_scalars(g); if (rc) return rc; }
  *dt = g->sc_host->dt_applied; *limit_hit = g->sc_host->hit_applied;
  if (dt_chem) *dt_chem = g->sc_host->dt_chem_out;
  if (dt_therm) *dt_therm = g->sc_host->dt_therm_out;
  if (cellcount) *cellcount = (long long)g->sc_host->cellcount;
  if (dt_hydro) *dt_hydro = g->p.cour_no/bits_to_double(g->sc_host->max_dti);
  if (neg) *neg = g->sc_host->neg_applied;
  return 0;
}

// GridS.EdgeFlux <- the last sweep that counted (outputs, restart dumps, ionrad_prolong_snd read it).  Only marked here:
// the array is filled when somebody reads it (aa_download_edgeflux, the hand-off to a refined level) -- 16 B/zone that a
// step without output does not have to move
int aa_ion_finish(aa_grid *g)
{
  if (!g->ion_fused) return 0;
  if (!g->slab.empty()) return slabs_ion_finish(g);
  g->ion_pending = false;
  g->ef_stale = true;
  return 0;
}
} // extern "C"
int aa_edgeflux_ready(aa_grid *g)
{
  if (!g->ef_stale) return 0;
  { Scope s(g, "ion_finish"); launch_ion_finish(g->d, g->ion_cur, g->st); }
  g->ef_stale = false;
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" {

// the words of the pass just queued to all ranks (the driver's collective, on the Grid's stream): nothing to do with one rank
struct IonGather { double *words = nullptr; const double *all = nullptr; int nranks = 1; aa_gather_fn fn = nullptr; void *ctx = nullptr; };
static int ion_gather(const IonGather &G)
{
  if (!G.fn) return 0;
  const int rc = G.fn(G.ctx);
  return rc ? aa_fail(-5, "[ion_radtransfer_3d]: the driver's all-gather of the sub-cycle's words failed (%d)", rc) : 0;
}
static int ion_run_fused(aa_grid *g, bool fine, double limit, int *niter_out, double *dt_done_out, const IonGather &G = IonGather())
{
  double dt, dt_chem = 0, dt_therm = 0, dt_hydro = 0, dt_done = 0.0;
  long long cellcount = 0;
  int hit, neg, niter = 0, rc;
  const double *all = G.fn ? G.all : nullptr;
  const int nr = G.fn ? G.nranks : 1;
  // one rank: every pass of this loop is followed by its pick with nothing in between -- fold and pick in one launch
  struct FuseGuard { aa_grid *g; ~FuseGuard() { g->ion_fuse_pick = false; } } fuse_guard{g};
  g->ion_fuse_pick = !G.fn && g->d.cfg.fuse_pick != 0;
  if ((rc = aa_ion_begin(g))) return rc;
  if ((rc = aa_ion_speculate(g, limit))) return rc;                          // (the first pass may already apply update(0) with the whole step)
  if ((rc = aa_ion_pass(g, 0, 1, G.fn ? G.words : nullptr))) return rc;      // sweep(0) + rates(0)
  if ((rc = ion_gather(G))) return rc;
  if ((rc = aa_ion_pick(g, all, nr, 1, limit))) return rc;                   // -> dt_0 (stays on the device)
  for (;;) {
    // update(n) with the step picked on the device, then -- unless that step was cut back to the limit -- sweep(n+1) and
    // rates(n+1), speculatively: whether the loop goes on is only known from this pass's own reductions
    if ((rc = aa_ion_pass(g, 1, 1, G.fn ? G.words : nullptr))) return rc;
    if ((rc = ion_gather(G))) return rc;
    if ((rc = aa_ion_pick(g, all, nr, 0, limit))) return rc;
    if ((rc = aa_ion_fetch(g, &dt, &hit, &dt_chem, &dt_therm, &cellcount, &dt_hydro, &neg))) return rc;   // the one read-back
    if (neg) return fail(-4, "[compute_chem_rates]: negative dt_chem");      // ionrad_3d.c:389-391 (of the rates behind dt)
    dt_done += dt;
    niter++;
    if (!fine) {
      if (cellcount > MAXCELLCOUNT) { g->dt = dt_done; break; }
      if (hit) break;
      if (dt_hydro < dt_done) { g->dt = dt_done; break; }
    } else if (hit) { g->dt = dt_done; break; }
  }
  if ((rc = aa_ion_finish(g))) return rc;
  *niter_out = niter; *dt_done_out = dt_done;
  return 0;
}

static int ion_run_phased(aa_grid *g, bool fine, double limit, int *niter_out, double *dt_done_out)
{
  double dt_chem, dt_therm, dt_hydro = 0, dt, dt_done = 0.0;
  long long cellcount;
  int niter = 0, rc;
  if ((rc = aa_ion_begin(g))) return rc;
  if ((rc = aa_ion_arm(g))) return rc;
  for (;;) {
    int hit = 0;
    if ((rc = aa_ion_subcycle(g, dt_done, limit, &dt, &hit, &dt_chem, &dt_therm, &cellcount, &dt_hydro))) return rc;
    dt_done += dt;
    niter++;
    if (!fine) {
      if (cellcount > MAXCELLCOUNT) { g->dt = dt_done; break; }
      if (hit) break;
      if (dt_hydro < dt_done) { g->dt = dt_done; break; }
    } else if (hit) { g->dt = dt_done; break; }
  }
  *niter_out = niter; *dt_done_out = dt_done;
  return 0;
}

int aa_ion_run(aa_grid *g, int finegrid, double limit, int *niter_out, double *dt_done_out)
{
  g->cfl_ready = false; g->active_dirty = true;
  if (!g->slab.empty() && !g->ion_fused) return slabs_ion_run_phased(g, limit, niter_out, dt_done_out);
  return g->ion_fused ? ion_run_fused(g, finegrid != 0, limit, niter_out, dt_done_out)
                      : ion_run_phased(g, finegrid != 0, limit, niter_out, dt_done_out);
}

int aa_ion_radtransfer_3d(aa_grid *g, int *niter_out)
{
  // ionrad_3d.c:862-1047, root level
  int niter = 0, rc; double dt_done = 0.0;
  if ((rc = aa_ion_run(g, 0, g->dt, &niter, &dt_done))) return rc;
  if (niter == g->p.maxiter) g->dt = dt_done;
  if (g->dt < 0) return fail(-4, "[ion_radtransfer_3d]: dt = %e, dt_done = %e", g->dt, dt_done);
  if (niter_out) *niter_out = niter;
  return 0;
}

int aa_ion_radtransfer_3d_gather(aa_grid *g, double *dev_words, const double *dev_words_all, int nranks, aa_gather_fn gather,
                                 void *ctx, int *niter_out)
{
  g->cfl_ready = false; g->active_dirty = true;
  if (!g->slab.empty()) return fail(-1, "[aa_ion_radtransfer_3d_gather]: a Grid cut into slabs reduces over its slabs itself (aa_ion_radtransfer_3d)");
  if (!g->ion_fused) return fail(-1, "[aa_ion_radtransfer_3d_gather]: this Grid runs the two-kernel sub-cycle (aa_ion_rates / aa_ion_update)");
  if (gather && (!dev_words || !dev_words_all || nranks < 1)) return fail(-1, "[aa_ion_radtransfer_3d_gather]: word buffers");
  IonGather G; G.words = dev_words; G.all = dev_words_all; G.nranks = nranks; G.fn = gather; G.ctx = ctx;
  int niter = 0, rc; double dt_done = 0.0;
  if ((rc = ion_run_fused(g, false, g->dt, &niter, &dt_done, G))) return rc;
  if (niter == g->p.maxiter) g->dt = dt_done;
  if (g->dt < 0) return fail(-4, "[ion_radtransfer_3d]: dt = %e, dt_done = %e", g->dt, dt_done);
  if (niter_out) *niter_out = niter;
  return 0;
}

int aa_host_syncs(aa_grid *g, int reset)
{
  int n = g->host_syncs; if (reset) g->host_syncs = 0;
  for (aa_grid *c : g->slab) { n += c->host_syncs; if (reset) c->host_syncs = 0; }
  return n;
}

int aa_start(aa_grid *g)
{
  int rc;
  if ((rc = aa_bvals_mhd(g))) return rc;
  if ((rc = aa_bvals_ionrad(g))) return rc;
  return aa_new_dt(g);
}

int aa_step(aa_grid *g, int *niter_out)
{
  int rc, niter = 0;
  if (g->p.ion && g->nradplane > 0) {                       // main.c:546-556
    if ((rc = aa_ion_radtransfer_3d(g, &niter))) return rc;
    if ((rc = aa_bvals_mhd(g))) return rc;
  }
  // (between the integrator and new_dt this loop only pins zones: the integrator may leave new_dt's maxima behind)
  const bool keep_opt = g->cfl_in_update;
  if (g->slab.empty() && g->cfl_step) g->cfl_in_update = true;      // (both builds since round 4: aa_cfl_in_update)
  rc = (g->p.integrator == 1 ? aa_integrate_3d_vl(g) : aa_integrate_3d_ctu(g));                    // :572-585
  g->cfl_in_update = keep_opt;
  if (rc) return rc;
  if (g->npin > 0 && (rc = aa_apply_pinned_cells(g))) return rc;   // Userwork_in_loop :597
  g->nstep++; g->time += g->dt;                              // :618-626
  if ((rc = aa_new_dt(g))) return rc;                        // :629
  if ((rc = aa_bvals_mhd(g))) return rc;                     // :635-644
  if (niter_out) *niter_out = niter;
  return 0;
}

// ---- x3 halo ------------------------------------------------------------------------------
#define NO_SLABS(name) if (!g->slab.empty()) return fail(-1, "[" name "]: not available on a Grid cut into slabs")
long long aa_halo_doubles(const aa_grid *g)      // (a Grid cut into slabs exchanges its halos itself: 0)
{ return g->slab.empty() ? (long long)g->d.N1*g->d.N2*AA_NGHOST*(5 + g->p.nscal) : 0LL; }
int aa_pack_x3(aa_grid *g, int side, double *buf)
{
  NO_SLABS("aa_pack_x3");
  Scope s(g, "halo_pack");
  const int k0 = side == 0 ? g->d.ks : g->d.ke - AA_NGHOST + 1;    // pack_ix3 / pack_ox3
  launch_pack_x3(g->d, 5 + g->p.nscal, k0, buf, g->st);
  return 0;
}
int aa_unpack_x3(aa_grid *g, int side, const double *buf)
{
  NO_SLABS("aa_unpack_x3");
  Scope s(g, "halo_unpack");
  const int k0 = side == 0 ? g->d.ks - AA_NGHOST : g->d.ke + 1;    // unpack_ix3 / unpack_ox3
  launch_unpack_x3(g->d, 5 + g->p.nscal, k0, buf, g->st);
  return 0;
}

// x2 halo of an x2 x x3 pencil decomposition (init_mesh.c:526-620 allows any NGrid_x2 x NGrid_x3; never x1: the rays):
// bvals_mhd.c:2462-2560 pack_ix2 / pack_ox2, :2896-3000 unpack.  Order of a bvals_mhd call with pencils: x1 sides
// (aa_bvals_mhd_side), then the x2 exchange / the physical x2 sides, then x3 -- so that the corners travel (:170).
long long aa_halo_doubles_x2(const aa_grid *g)
{ return g->slab.empty() ? (long long)g->d.N1*AA_NGHOST*(g->d.ke - g->d.ks + 1)*(5 + g->p.nscal) : 0LL; }
int aa_pack_x2(aa_grid *g, int side, double *buf)
{
  NO_SLABS("aa_pack_x2");
  Scope s(g, "halo_pack");
  const int j0 = side == 0 ? g->d.js : g->d.je - AA_NGHOST + 1;    // pack_ix2 / pack_ox2
  launch_pack_x2(g->d, 5 + g->p.nscal, j0, buf, g->st);
  return 0;
}
int aa_unpack_x2(aa_grid *g, int side, const double *buf)
{
  NO_SLABS("aa_unpack_x2");
  g->inner_swept = false; g->cfl_ready = false;      // (ghost zones only: the host's active zones stay current)
  Scope s(g, "halo_unpack");
  const int j0 = side == 0 ? g->d.js - AA_NGHOST : g->d.je + 1;    // unpack_ix2 / unpack_ox2
  launch_unpack_x2(g->d, 5 + g->p.nscal, j0, buf, g->st);
  return 0;
}

// The same halos for a driver that moves them itself through HOST buffers (the reference's own MPI ranks on the shim,
// bvals_mhd.c:296-493 with MPI_Isend / MPI_Irecv): dir 1 = x2, 2 = x3; get = pack_i* / pack_o* into host_buf, put = the
// ghost planes from host_buf.  The packed block is staged in the face-state area (idle outside the integrator).
long long aa_halo_doubles_dir(const aa_grid *g, int dir) { return dir == 1 ? aa_halo_doubles_x2(g) : (dir == 2 ? aa_halo_doubles(g) : 0LL); }
int aa_halo_get(aa_grid *g, int dir, int side, double *host_buf)
{
  NO_SLABS("aa_halo_get");
  if ((dir != 1 && dir != 2) || side < 0 || side > 1 || !host_buf) return fail(-1, "[aa_halo_get]: dir=%d side=%d", dir, side);
  g->inner_swept = false;
  int rc = dir == 1 ? aa_pack_x2(g, side, g->d.LR) : aa_pack_x3(g, side, g->d.LR); if (rc) return rc;
  HIPCHK(hipMemcpyAsync(host_buf, g->d.LR, (size_t)aa_halo_doubles_dir(g, dir)*sizeof(Real), hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  return 0;
}
int aa_halo_put(aa_grid *g, int dir, int side, const double *host_buf)
{
  NO_SLABS("aa_halo_put");
  if ((dir != 1 && dir != 2) || side < 0 || side > 1 || !host_buf) return fail(-1, "[aa_halo_put]: dir=%d side=%d", dir, side);
  g->inner_swept = false; g->cfl_ready = false;      // (ghost zones only: the host's active zones stay current)
  HIPCHK(hipMemcpyAsync(g->d.LR, host_buf, (size_t)aa_halo_doubles_dir(g, dir)*sizeof(Real), hipMemcpyHostToDevice, g->st));
  int rc = dir == 1 ? aa_unpack_x2(g, side, g->d.LR) : aa_unpack_x3(g, side, g->d.LR); if (rc) return rc;
  HIPCHK(hipStreamSynchronize(g->st));      // (host_buf and the staging area are the caller's again)
  return 0;
}
int aa_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; } return n; }

// ---- function-level tests -------------------------------------------------------------------
int aa_test_fluxes(int nscal, double gamma, int n, const double *Ul, const double *Ur, const double *etah, double *F)
{
  const size_t nv = 5 + nscal, nb = (size_t)n*nv*sizeof(Real);
  Real *d = nullptr;
  HIPCHK(hipMalloc(&d, 3*nb + (size_t)n*sizeof(Real)));
  Real *dUl = d, *dUr = d + (size_t)n*nv, *dF = d + 2*(size_t)n*nv, *de = d + 3*(size_t)n*nv;
  HIPCHK(hipMemcpy(dUl, Ul, nb, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dUr, Ur, nb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(de, etah, (size_t)n*sizeof(Real), hipMemcpyHostToDevice));
  launch_test_fluxes(nscal, gamma, n, dUl, dUr, de, dF, 0);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(F, dF, nb, hipMemcpyDeviceToHost));
  hipFree(d);
  return 0;
}
static int test_lr_any(int order, int nscal, double gamma, int n, const double *W, double dt, double dx, int il, int iu, double *Wl, double *Wr)
{
  const size_t nv = 5 + nscal, nb = (size_t)n*nv*sizeof(Real);
  const int h = (order == 3) ? 3 : 2;
  if (il < h || iu > n - 1 - h) return fail(-1, "[aa_test_lr_states]: W must cover [il-%d, iu+%d]", h, h);
  Real *d = nullptr;
  HIPCHK(hipMalloc(&d, 3*nb));
  Real *dW = d, *dWl = d + (size_t)n*nv, *dWr = d + 2*(size_t)n*nv;
  HIPCHK(hipMemcpy(dW, W, nb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dWl, Wl, nb, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dWr, Wr, nb, hipMemcpyHostToDevice));
  if (order == 3) launch_test_lr_ppm(nscal, gamma, n, dW, dt, dx, il, iu, dWl, dWr, 0);
  else launch_test_lr(nscal, gamma, n, dW, dt, dx, il, iu, dWl, dWr, 0);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(Wl, dWl, nb, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(Wr, dWr, nb, hipMemcpyDeviceToHost));
  hipFree(d);
  return 0;
}
int aa_test_lr_states(int nscal, double gamma, int n, const double *W, double dt, double dx, int il, int iu, double *Wl, double *Wr)
{ return test_lr_any(2, nscal, gamma, n, W, dt, dx, il, iu, Wl, Wr); }
int aa_test_lr_states_ppm(int nscal, double gamma, int n, const double *W, double dt, double dx, int il, int iu, double *Wl, double *Wr)
{ return test_lr_any(3, nscal, gamma, n, W, dt, dx, il, iu, Wl, Wr); }

// the scaling-free quotient / square root of hydro_dev.h beside hipcc's own: out[5][n] = x_div(a,b), a/b, x_sqrt(a), sqrt(a),
// x_div_r(a, b, 1/b)
int aa_test_xdiv(int n, const double *a, const double *b, double *out)
{
  if (n <= 0) return fail(-1, "[aa_test_xdiv]: n = %d", n);
  Real *d = nullptr;
  HIPCHK(hipMalloc(&d, 7*(size_t)n*sizeof(Real)));
  HIPCHK(hipMemcpy(d, a, (size_t)n*sizeof(Real), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + n, b, (size_t)n*sizeof(Real), hipMemcpyHostToDevice));
  launch_test_xdiv(n, d, d + n, d + 2*(size_t)n, 0);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, d + 2*(size_t)n, 5*(size_t)n*sizeof(Real), hipMemcpyDeviceToHost));
  hipFree(d);
  return 0;
}

// exp / log of the one-kernel sub-cycle (ion_pass.hip) on n values (n a multiple of 4): ye = exp(x), yl = ln|x|
int aa_test_explog(int n, const double *x, double *ye, double *yl)
{
  if (n <= 0 || n % 4) return fail(-1, "[aa_test_explog]: n must be a positive multiple of 4");
  Real *d = nullptr;
  HIPCHK(hipMalloc(&d, 3*(size_t)n*sizeof(Real)));
  HIPCHK(hipMemcpy(d, x, (size_t)n*sizeof(Real), hipMemcpyHostToDevice));
  launch_test_explog(n, d, d + n, d + 2*(size_t)n, 0);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(ye, d + n, (size_t)n*sizeof(Real), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(yl, d + 2*(size_t)n, (size_t)n*sizeof(Real), hipMemcpyDeviceToHost));
  hipFree(d);
  return 0;
}

// ---- history sums (dump_history.c:157-200) ---------------------------------------------------
int aa_history(aa_grid *g, double *sums)
{
  if (!g->slab.empty()) return slabs_history(g, sums);
  g->inner_swept = false;
  // partial rows go through the face-state area, idle outside the integrator
  const int nb = launch_history(g->d, g->p.nscal, g->d.LR, g->st);
  std::vector<double> part((size_t)nb*9);
  HIPCHK(hipMemcpyAsync(part.data(), g->d.LR, part.size()*sizeof(double), hipMemcpyDeviceToHost, g->st));
  HIPCHK(hipStreamSynchronize(g->st));
  const double dVol = g->d.dx[0]*g->d.dx[1]*g->d.dx[2];
  for (int q = 0; q < 9; q++) {
    double s = 0.0;
    for (int b = 0; b < nb; b++) s += part[(size_t)b*9 + q];
    sums[q] = dVol*s;
  }
  return 0;
}

// ---- measurement -----------------------------------------------------------------------------
// (a Grid cut into slabs reports its first slab's kernels)
int aa_profile_enable(aa_grid *g, int on)
{
  if (!g->slab.empty()) return aa_profile_enable(g->slab[0], on);
  prof_drain(g); g->prof = on != 0;
  // events for a few dozen steps ahead: a timed region then creates none (hipEventCreate showed on the small Grids of the reference's decks)
  if (on) while (g->ev_pool.size() < 4096) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) break; g->ev_pool.push_back(ev); }
  return 0;
}
int aa_profile_reset(aa_grid *g)
{ if (!g->slab.empty()) return aa_profile_reset(g->slab[0]); prof_drain(g); for (auto &e : g->pe) { e.total_ms = 0; e.launches = 0; } return 0; }
int aa_profile_count(const aa_grid *g) { if (!g->slab.empty()) return aa_profile_count(g->slab[0]); return (int)g->pe.size(); }
const char *aa_profile_name(const aa_grid *g, int i)
{ if (!g->slab.empty()) return aa_profile_name(g->slab[0], i); return (i >= 0 && i < (int)g->pe.size()) ? g->pe[i].name.c_str() : ""; }
int aa_profile_get(aa_grid *g, int i, double *total_ms, long long *launches)
{
  if (!g->slab.empty()) return aa_profile_get(g->slab[0], i, total_ms, launches);
  if (i < 0 || i >= (int)g->pe.size()) return fail(-1, "[aa_profile_get]: bad index");
  prof_drain(g);
  if (total_ms) *total_ms = g->pe[i].total_ms;
  if (launches) *launches = g->pe[i].launches;
  return 0;
}

}  // extern "C"
