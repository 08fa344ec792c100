// hydro_kernels.hip -- the 3-D CTU integrator of integrators/integrate_3d_ctu.c as HIP kernels
// for gfx950 (HYDRO / ADIABATIC / CARTESIAN / PLM-characteristic / Roe + H-correction).
//
// Kernel chain of one integrate_3d_ctu() call (reference steps in brackets):
//   sweep<x2>, sweep<x3>, sweep<x1>   [1-3]  prim conversion + PLM/tracing + gravity kick +
//                                            first-pass Roe flux; writes only F of that direction
//   correct_all                       [5-8a,9a] per cell: L/R states re-derived instead of read
//                                            back, transverse flux-gradient corrections, stores the
//                                            corrected Ul,Ur, eta of the faces and d^{n+1/2}
//                                            (small Grids: one tile kernel per direction, x1 fused with
//                                            its first pass)
//   flux2_update                      [9b-d,11a,12] etah = max of 9 etas, second-pass Roe fluxes (kept in
//                                            registers / LDS), gravity source + conservative update of U
//                                            (AA_FUSED_UPDATE=0: flux2<x1..x3> + update through HBM)
// All arrays are struct-of-arrays with i fastest (grid.h); every global access of a wavefront
// is to consecutive i.  The x1 sweep exchanges neighbour cells through LDS; the x2/x3 sweeps
// march a register sliding window along the sweep direction so no cell is converted or
// reconstructed twice.  FP64 vector arithmetic throughout: there is no dense contraction, so
// MFMA is not used.
#include <stdlib.h>
#include "grid.h"
#include "hydro_dev.h"

// Compiled twice: as it stands into namespace aa, and with -DAA_COOLING=1 into namespace aa_cool -- the same kernels with the
// optically thin cooling terms of integrate_3d_ctu.c (Steps 1c-3c :359-368, :662-671, :846-855; 8b :2133-2266; 11c :2943-2953) for the
// cooling function the reference ships (KoyInut, microphysics/cool.c:48).  api.hip launches the second set when a cooling
// function is enrolled (aa_set_cooling); the kernels of every other run do not carry a register or an instruction of it.
#ifndef AA_COOLING
#define AA_COOLING 0
#endif
#if AA_COOLING
namespace aa_cool {
using namespace aa;
#else
namespace aa {
#endif

// ---- field accessors ------------------------------------------------------------------
AA_DEV Real *Uf(const DevGrid &g, int v) { return g.U + (long)v*g.nc; }
AA_DEV Real *LRf(const DevGrid &g, int d, int side, int v) { return g.LR + (long)((d*2 + side)*6 + v)*g.nc; }
AA_DEV Real *Ff(const DevGrid &g, int d, int v) { return g.F + (long)(d*6 + v)*g.nc; }
AA_DEV Real *Ef(const DevGrid &g, int d) { return g.eta + (long)d*g.nc; }
AA_DEV Real *Pf(const DevGrid &g, int which) { return g.phi + (long)which*g.nc; }   // 0 centre, 1+d face
// dt/dx_d and half of it, formed by the launcher (the same IEEE quotient the kernels form): as kernel arguments they live in scalar
// registers; formed in the kernel they took twelve vector registers through the whole march of k_correct_all, six in k_flux2_update
struct StepRatios { Real dtodx[3], q[3]; };
static inline StepRatios step_ratios(const DevGrid &g, Real dt)
{ StepRatios sr; for (int d = 0; d < 3; d++) { sr.dtodx[d] = dt/g.dx[d]; sr.q[d] = 0.5*(dt/g.dx[d]); } return sr; }

#if AA_COOLING
// microphysics/cool.c:48-86 KoyInut(): the cooling rate [erg cm^-3 s^-1] of the diffuse ISM (Koyama & Inutsuka 2002, eq. 4) from
// density and pressure in cgs units, limited so that the temperature stays above its equilibrium value over dt
AA_DEV Real cool_koyinut(Real dens, Real Press, Real dt, Real Gamma_1)
{
  const Real mbar = (1.37)*(1.6733e-24), kb = 1.380658e-16, HeatRate = 2.0e-26, Tmin = 10;
  const Real n = dens/mbar;
  const Real logn = log10(n);
  const Real T = rmax((Press/(n*kb)), Tmin);
  Real Teq = Tmin;
  const Real coolratepp = HeatRate*(n*(1.0e7*exp(-1.184e5/(T+1000.)) + 0.014*sqrt(T)*exp(-92.0/T)) - 1.0);
  const Real dT = coolratepp*dt*Gamma_1/kb;
  if ((T-dT) <= 185.0) {
    const Real lognT = 3.9247499 - 1.8479378*logn + 1.5335032*logn*logn
     -0.47665872*pow(logn,3.0) + 0.076789136*pow(logn,4.0)-0.0049052587*pow(logn,5.0);
    Teq = pow(10.0,lognT) / n;
  }
  const Real MaxdT = kb*(T-Teq)/(dt*Gamma_1);
  return n*rmin(coolratepp,MaxdT);
}
// Steps 1c / 2c / 3c (cont): the L/R primitive states lose pressure over half a step
AA_DEV void cool_states(const DevGrid &g, Real dt, Real wl[6], Real wr[6])
{
  const Real coolfl = cool_koyinut(wl[0], wl[4], (0.5*dt), g.Gamma_1);
  const Real coolfr = cool_koyinut(wr[0], wr[4], (0.5*dt), g.Gamma_1);
  wl[4] -= 0.5*dt*g.Gamma_1*coolfl;
  wr[4] -= 0.5*dt*g.Gamma_1*coolfr;
}
// Step 8b: P^{n+1/2} of zone m from U^n, the first-pass flux differences across the zone (global frame) and, with a static
// potential, q_e (phi_r - phi_l) d
AA_DEV Real p_half(const DevGrid &g, long m, const Real q[3], const Real dF[3][6], bool grav, const Real gm[3], Real dhalf)
{
  Real Mh[3];
#pragma unroll
  for (int e = 0; e < 3; e++) {
    Mh[e] = Uf(g, 1 + e)[m] - q[0]*dF[0][1 + e] - q[1]*dF[1][1 + e] - q[2]*dF[2][1 + e];
    if (grav) Mh[e] -= gm[e];
  }
  const Real Eh = Uf(g, 4)[m] - q[0]*dF[0][4] - q[1]*dF[1][4] - q[2]*dF[2][4];
  Real ph = Eh - 0.5*(Mh[0]*Mh[0] + Mh[1]*Mh[1] + Mh[2]*Mh[2])/dhalf;
  ph *= g.Gamma_1;
  return ph;
}
#endif

// sweep-frame component n of direction D lives in global field gv<D>(n):
// (Mx,My,Mz) = (M[D], M[D+1], M[D+2])   integrate_3d_ctu.c:206-208, :544-546, :727-729
template <int D> AA_DEV constexpr int gv(int n) { return (n >= 1 && n <= 3) ? 1 + ((D + n - 1) % 3) : n; }
template <int D> AA_DEV long stride(const DevGrid &g) { return D == 0 ? 1L : (D == 1 ? g.sJ : g.sK); }
AA_DEV long stride_rt(const DevGrid &g, int d) { return d == 0 ? 1L : (d == 1 ? g.sJ : g.sK); }

template <int D, int NS>
AA_DEV void load_sweep(const Real *fam, long nc, long m, Real out[6])
{
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) out[n] = fam[(long)gv<D>(n)*nc + m];
  if (!NS) out[5] = 0.0;
}
template <int D, int NS>
AA_DEV void store_sweep(Real *fam, long nc, long m, const Real in[6])
{
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) fam[(long)gv<D>(n)*nc + m] = in[n];
}

// L/R states of one cell: piecewise linear (ORD 2, lr_states_plm.c) or piecewise parabolic (ORD 3,
// lr_states_ppm.c; the slopes of the cell and of its two neighbours along D come from k_slopes)
AA_DEV Real *Sf(const DevGrid &g, int d, int n) { return g.slope + (long)(d*6 + n)*g.nc; }
template <int NS, bool TRACE, int ORD, int D>
AA_DEV void recon_cell(const DevGrid &g, long mcell, const Real wm[6], const Real w[6], const Real wp[6], Real dtodx,
                       Real wl_next[6], Real wr_here[6])
{
  if (ORD == 3) {
    const long s = stride<D>(g);
    Real Dm[6], D0[6], Dp[6];
#pragma unroll
    for (int n = 0; n < 5 + NS; n++) { const Real *q = Sf(g, D, n) + mcell; Dm[n] = q[-s]; D0[n] = q[0]; Dp[n] = q[s]; }
    if (!NS) { Dm[5] = 0.0; D0[5] = 0.0; Dp[5] = 0.0; }
    ppm_cell<NS, TRACE>(wm, w, wp, Dm, D0, Dp, dtodx, g.Gamma, wl_next, wr_here);
  } else plm_cell<NS, TRACE>(wm, w, wp, dtodx, g.Gamma, wl_next, wr_here);
}

// ---- order 3: monotonised slopes of every cell along D (Steps 1-5 of lr_states_ppm.c) -----------
// cells: along D [s-3, e+3] (a parabola of cell c in [s-2, e+2] needs c-1 and c+1), transverse
// [s-2, e+2] as the sweeps
template <int NS, int D>
__global__ void __launch_bounds__(256)
k_slopes(DevGrid g, const Real *src)
{
  const int lo[3] = {g.is, g.js, g.ks}, hi[3] = {g.ie, g.je, g.ke};
  int n3[3], o3[3];
#pragma unroll
  for (int d = 0; d < 3; d++) { o3[d] = lo[d] - (d == D ? 3 : 2); n3[d] = hi[d] - lo[d] + 1 + (d == D ? 6 : 4); }
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= (long)n3[0]*n3[1]*n3[2]) return;
  const int i = o3[0] + (int)(lin % n3[0]), j = o3[1] + (int)((lin / n3[0]) % n3[1]), k = o3[2] + (int)(lin / ((long)n3[0]*n3[1]));
  const long m = (long)k*g.sK + (long)j*g.sJ + i, s = stride<D>(g);
  Real u[6], wm[6], w[6], wp[6], dWm[6];
  load_sweep<D, NS>(src, g.nc, m - s, u); cons_to_prim<NS>(u, wm, g.Gamma_1);
  load_sweep<D, NS>(src, g.nc, m,     u); cons_to_prim<NS>(u, w,  g.Gamma_1);
  load_sweep<D, NS>(src, g.nc, m + s, u); cons_to_prim<NS>(u, wp, g.Gamma_1);
  limited_slopes<NS>(wm, w, wp, g.Gamma, dWm);
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) Sf(g, D, n)[m] = dWm[n];
}

// What a sweep does at each interface once the L/R primitive states are known.
//   MODE_FLUX1  CTU steps 1c-1d: gravity kick, conversion to conserved, first-pass flux (etah=0);
//               only the flux is stored
//   MODE_CORR   CTU steps 5-7 + 9a: the same L/R states are RE-derived (bit-identical: same code),
//               corrected with the transverse flux gradients (+ gravity), stored, and the
//               H-correction eta of the face is computed.  Recomputing PLM here instead of storing
//               the uncorrected face states in the first pass and reading them back saves
//               2 x 12 doubles of HBM traffic per zone and direction.
//   MODE_VL     integrate_3d_vl.c:751-795: no tracing (done in plm_cell), no kick, flux only
//   MODE_BOTH   FLUX1 and CORR in one pass: used for x1, which runs after the x2/x3 first passes
//               (their fluxes are all its correction needs), saving one x1 reconstruction sweep
enum { MODE_FLUX1 = 0, MODE_CORR = 1, MODE_VL = 2, MODE_BOTH = 3 };

template <int NS, int D, bool GRAV>
AA_DEV void face_correct(const DevGrid &g, long m, int i, int j, int k, Real dt, const Real ul_s[6], const Real ur_s[6])
{
  // ranges (integrate_3d_ctu.c:978, :1282, :1691): along D [s-1, e+2] (the sweep's own range),
  // transverse [s-1, e+1]
  const bool in = (D == 0 || (i >= g.is - 1 && i <= g.ie + 1)) && (D == 1 || (j >= g.js - 1 && j <= g.je + 1)) &&
                  (D == 2 || (k >= g.ks - 1 && k <= g.ke + 1));
  if (!in) return;
  constexpr int NV = 5 + NS;
  const long sD = stride<D>(g), ml = m - sD;
  Real ul[6], ur[6], q[3];
#pragma unroll
  for (int n = 0; n < 6; n++) { ul[gv<D>(n)] = ul_s[n]; ur[gv<D>(n)] = ur_s[n]; }   // to the global frame
#pragma unroll
  for (int d = 0; d < 3; d++) q[d] = 0.5*(dt/g.dx[d]);
#pragma unroll
  for (int e = 0; e < 3; e++) {
    if (e == D) continue;
    const long se = stride_rt(g, e);
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const Real *fe = Ff(g, e, v);
      ul[v] -= q[e]*(fe[ml + se] - fe[ml]);
      ur[v] -= q[e]*(fe[m + se] - fe[m]);
    }
  }
  if (GRAV) {   // :1167-1219, :1463-1526, :1873-1937
    const Real *pc = Pf(g, 0);
    const Real dR = Uf(g, 0)[m], dL = Uf(g, 0)[ml];
    Real phic = pc[m];
#pragma unroll
    for (int e = 0; e < 3; e++) {
      if (e == D) continue;
      const long se = stride_rt(g, e);
      const Real *pfe = Pf(g, 1 + e), *fd = Ff(g, e, 0);
      Real phir = pfe[m + se], phil = pfe[m];
      ur[1 + e] -= q[e]*(phir - phil)*dR;
      ur[4] -= q[e]*(fd[m]*(phic - phil) + fd[m + se]*(phir - phic));
    }
    phic = pc[ml];
#pragma unroll
    for (int e = 0; e < 3; e++) {
      if (e == D) continue;
      const long se = stride_rt(g, e);
      const Real *pfe = Pf(g, 1 + e), *fd = Ff(g, e, 0);
      Real phir = pfe[ml + se], phil = pfe[ml];
      ul[1 + e] -= q[e]*(phir - phil)*dL;
      ul[4] -= q[e]*(fd[ml]*(phic - phil) + fd[ml + se]*(phir - phic));
    }
  }
#pragma unroll
  for (int v = 0; v < NV; v++) { LRf(g, D, 0, v)[m] = ul[v]; LRf(g, D, 1, v)[m] = ur[v]; }
  // eta (integrate_3d_ctu.c:2300-2343), in the sweep frame of D
  Real sl[6], sr[6];
#pragma unroll
  for (int n = 0; n < 6; n++) { sl[n] = ul[gv<D>(n)]; sr[n] = ur[gv<D>(n)]; }
  Real lambdar = lambda_face(sr, g.Gamma, g.Gamma_1, 1.0), lambdal = lambda_face(sl, g.Gamma, g.Gamma_1, -1.0);
  Ef(g, D)[m] = 0.5*fabs(lambdar - lambdal);
  if ((GRAV || AA_COOLING) && D == 1 && j <= g.je + 1) {       // d^{n+1/2}, :2104-2125 (needs all first-pass fluxes:
                                               // done in the x2 correct pass, which runs after them)
    const Real dh = Uf(g, 0)[m]
      - q[0]*(Ff(g, 0, 0)[m + 1]    - Ff(g, 0, 0)[m])
      - q[1]*(Ff(g, 1, 0)[m + g.sJ] - Ff(g, 1, 0)[m])
      - q[2]*(Ff(g, 2, 0)[m + g.sK] - Ff(g, 2, 0)[m]);
    g.dhalf[m] = dh;
#if AA_COOLING
    {   // P^{n+1/2}, :2133-2266
      Real dF[3][6], gm[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int e = 0; e < 3; e++) {
        const long se = stride_rt(g, e);
#pragma unroll
        for (int v = 0; v < 6; v++) dF[e][v] = (v < NV) ? Ff(g, e, v)[m + se] - Ff(g, e, v)[m] : 0.0;
        if (GRAV) gm[e] = q[e]*(Pf(g, 1 + e)[m + se] - Pf(g, 1 + e)[m])*Uf(g, 0)[m];
      }
      g.phalf[m] = p_half(g, m, q, dF, GRAV, gm, dh);
    }
#endif
  }
}

template <int NS, int D, bool GRAV, int MODE>
AA_DEV void face_work(const DevGrid &g, long m, int i, int j, int k, Real dt, Real wl[6], Real wr[6])
{
  if (GRAV && MODE != MODE_VL) {   // integrate_3d_ctu.c:318-342 (x1), :611-628 (x2), :795-812 (x3)
    const long s = stride<D>(g);
    const Real dtodx = dt/g.dx[D];
    Real phicr = Pf(g, 0)[m], phicl = Pf(g, 0)[m - s], phifc = Pf(g, 1 + D)[m];
    wl[1] -= dtodx*(phifc - phicl);
    wr[1] -= dtodx*(phicr - phifc);
  }
#if AA_COOLING
  if (MODE != MODE_VL) cool_states(g, dt, wl, wr);       // :359-368, :662-671, :846-855 (the van Leer integrator has no such term)
#endif
  Real ul[6], ur[6];
  prim_to_cons<NS>(wl, ul, g.Gamma_1, g.rGamma_1);
  prim_to_cons<NS>(wr, ur, g.Gamma_1, g.rGamma_1);
  if (MODE == MODE_CORR) { face_correct<NS, D, GRAV>(g, m, i, j, k, dt, ul, ur); return; }
  Real f[6];
  flux_roe<NS>(ul, ur, wl, wr, 0.0, g.Gamma, g.Gamma_1, f);
  store_sweep<D, NS>(Ff(g, D, 0), g.nc, m, f);
  if (MODE == MODE_BOTH) face_correct<NS, D, GRAV>(g, m, i, j, k, dt, ul, ur);
}

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx b and b+8 share an XCD and its L2).
// Stencil kernels want NEIGHBOURING rows in the same L2, so give each XCD one contiguous 1/8 of
// the linear cell range: logical block = (b % 8)*per + b/8 with the grid rounded up to 8*per.
AA_DEV long xcd_block(unsigned per) { return (long)(blockIdx.x & 7u)*per + (blockIdx.x >> 3); }

// Zone ordering of the stencil kernels.  `strip` > 0 walks the (i,j,k) box strip-major: j is cut
// into strips of `strip` rows and each strip is traversed k-plane by k-plane, so the distance
// between a zone and its k+-1 neighbours is one strip-plane of all streamed fields (tens of MB:
// resident in the 256 MB Infinity Cache) instead of a full plane (~216 MB at 512^3).
struct Order { int strip; int xcd; };
AA_DEV bool decode_zone(const Order o, int ni, int nj, int nk, int &i, int &j, int &k)
{
  const long lin = (o.xcd ? xcd_block(gridDim.x >> 3) : (long)blockIdx.x)*blockDim.x + threadIdx.x;
  if (lin >= (long)ni*nj*nk) return false;
  if (o.strip <= 0 || o.strip >= nj) {
    i = (int)(lin % ni); j = (int)((lin / ni) % nj); k = (int)(lin / ((long)ni*nj));
    return true;
  }
  const long per_full = (long)ni*o.strip*nk;          // zones in a full strip
  const int s = (int)(lin / per_full);
  const int j0 = s*o.strip;
  const int sj = (j0 + o.strip <= nj) ? o.strip : nj - j0;   // last strip may be thinner
  const long r = lin - (long)s*per_full;
  i = (int)(r % ni); j = j0 + (int)((r / ni) % sj); k = (int)(r / ((long)ni*sj));
  return true;
}

// ---- steps 2,3: x2 / x3 sweeps, register sliding window along the sweep direction ---------
// One thread owns one (i, transverse) column and a chunk of `chunk` interfaces; lanes are
// consecutive in i.  Cells reconstructed: l..u = s-2..e+2; interfaces l+1..u (:179-184).
#ifndef SW_ALIGN
#define SW_ALIGN 1
#endif
// (the pool puts zone i = 4 of every row on a 128-byte line when the pitch is a multiple of 16 doubles: api.hip)
__host__ __device__ static inline int march_shift(const DevGrid &g) { return (g.sJ & 15) ? 0 : ((g.is - 2 + 12) & 15); }
__host__ __device__ static inline int march_slots(const DevGrid &g)
{
  const int n = march_shift(g) + g.ie - g.is + 5; return (g.sJ & 15) ? n : ((n + 15) & ~15);
}
__host__ __device__ static inline bool march_cell(const DevGrid &g, int q, int &i)      // slot q of a row -> zone i (false: idle lane)
{
  const int sh = march_shift(g);
  i = g.is - 2 - sh + q;
  return q >= sh && i <= g.ie + 2;
}
// The limited slopes along x2 / x3 as a march (SLOPES_MARCH, the default): k_slopes reads and converts three cells per zone, two of them
// in other rows, which at 512^3 come back from HBM for blocks on other XCDs; here a thread owns an (i, transverse) column and a
// chunk of cells along D with the three-cell window in registers (every cell read and converted once, lanes on whole 128-byte
// lines as in k_sweep_march).  ONE inlined instance of the conversion serves every cell, so the values do not depend on where
// the chunks start.  Cells [s-3, e+3] along D, [s-2, e+2] across, as k_slopes.
#ifndef SLOPES_MARCH
#define SLOPES_MARCH 1
#endif
template <int NS, int D>
__global__ void __launch_bounds__(64)
k_slopes_march(DevGrid g, const Real *src, int chunk)
{
  static_assert(D == 1 || D == 2, "march kernel is for the strided directions");
  const int nq = march_slots(g);
  const int nt = (D == 1 ? g.ke - g.ks : g.je - g.js) + 5;
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= (long)nq*nt) return;
  const int q = (int)(lin % nq);
  int i;
  const int t = (D == 1 ? g.ks : g.js) - 2 + (int)(lin / nq);
  if (!march_cell(g, q, i)) return;
  const int lo = (D == 1 ? g.js : g.ks) - 3, hi = (D == 1 ? g.je : g.ke) + 3;
  const int c0 = lo + blockIdx.y*chunk;
  int c1 = c0 + chunk - 1; if (c1 > hi) c1 = hi;
  if (c0 > c1) return;
  const long s = stride<D>(g);
  const long base = (D == 1) ? ((long)t*g.sK + i) : ((long)t*g.sJ + i);
  Real wm[6], w[6], wp[6], u[6], dW[6];
#pragma unroll
  for (int n = 0; n < 6; n++) { w[n] = 1.0; wp[n] = 1.0; }
#pragma nounroll
  for (int c = c0 - 2; c <= c1; c++) {
    long mc = base + (long)c*s;
    asm volatile("" : "+v"(mc));
#pragma unroll
    for (int n = 0; n < 6; n++) { wm[n] = w[n]; w[n] = wp[n]; }
    load_sweep<D, NS>(src, g.nc, mc + s, u); cons_to_prim<NS>(u, wp, g.Gamma_1);
    if (c >= c0) {
      limited_slopes<NS>(wm, w, wp, g.Gamma, dW);
#pragma unroll
      for (int n = 0; n < 5 + NS; n++) Sf(g, D, n)[mc] = dW[n];
    }
  }
}

#ifndef SW_OCC
#define SW_OCC 3
#endif
template <int NS, int D, bool GRAV, int MODE, int ORD>
__global__ void __launch_bounds__(256, SW_OCC)      // 3 waves per SIMD (168 VGPRs): the kernel is VALU-bound and sits right at that edge
k_sweep_march(DevGrid g, const Real *src, Real dt, int chunk, int toff, int tcnt)
{
  static_assert(D == 1 || D == 2, "march kernel is for the strided directions");
  // transverse lines toff .. toff+tcnt-1 of the (D == 1 ? ke - ks : je - js) + 5 (the launcher passes all of them, or
  // for x2 a range of k-planes: see launch_sweep)
  const int tlo = (D == 1 ? g.ks : g.js) - 2 + toff;
  const int nt  = tcnt;
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
#if SW_ALIGN
  // lanes on whole 128-byte lines: a row of slots starts on the line that holds is-2 and is a whole number of lines long (a
  // wavefront's 64 lanes are four whole lines of one or two rows; the lanes off the ends of [is-2, ie+2] leave)
  const int nq = march_slots(g);
  if (lin >= (long)nq*nt) return;
  const int q = (int)(lin % nq);
  int i;
  const int t = tlo + (int)(lin / nq);
  if (!march_cell(g, q, i)) return;
#else
  const int ni = g.ie - g.is + 5;                              // i in [is-2, ie+2]
  if (lin >= (long)ni*nt) return;
  const int i = g.is - 2 + (int)(lin % ni);
  const int t = tlo + (int)(lin / ni);
#endif
  const int lo = (D == 1 ? g.js : g.ks), hi = (D == 1 ? g.je : g.ke);
  const int f0 = lo - 1 + blockIdx.y*chunk;                    // first interface of this chunk
  int f1 = f0 + chunk - 1; if (f1 > hi + 2) f1 = hi + 2;
  if (f0 > f1) return;
  const long s = stride<D>(g);
  const long base = (D == 1) ? ((long)t*g.sK + i) : ((long)t*g.sJ + i);
  const Real dtodx = dt/g.dx[D];

  // ONE inlined instance of the reconstruction serves every cell, also the cell below the chunk's
  // first interface (iteration f0-1, whose face work is skipped): with fused multiply-adds allowed, two
  // instances may round differently, and the result must not depend on where the chunks start (the
  // chunk size follows the Grid size, hence the decomposition).
  Real wm[6], w[6], wp[6], wl_cur[6], wl_next[6], wr[6], u[6];
  load_sweep<D, NS>(src, g.nc, base + (long)(f0 - 2)*s, u); cons_to_prim<NS>(u, w,  g.Gamma_1);
  load_sweep<D, NS>(src, g.nc, base + (long)(f0 - 1)*s, u); cons_to_prim<NS>(u, wp, g.Gamma_1);
#pragma unroll
  for (int n = 0; n < 6; n++) wl_cur[n] = 0.0;
#pragma nounroll
  for (int f = f0 - 1; f <= f1; f++) {
    long mf = base + (long)f*s;
    asm volatile("" : "+v"(mf));     // one index for all fields instead of a strength-reduced pointer per field
#pragma unroll
    for (int n = 0; n < 6; n++) { wm[n] = w[n]; w[n] = wp[n]; }
    load_sweep<D, NS>(src, g.nc, mf + s, u); cons_to_prim<NS>(u, wp, g.Gamma_1);
    recon_cell<NS, MODE != MODE_VL, ORD, D>(g, mf, wm, w, wp, dtodx, wl_next, wr);  // cell f -> Wl[f+1], Wr[f]
    if (f >= f0) face_work<NS, D, GRAV, MODE>(g, mf, i, D == 1 ? f : t, D == 1 ? t : f, dt, wl_cur, wr);
#pragma unroll
    for (int n = 0; n < 6; n++) wl_cur[n] = wl_next[n];
  }
}

// ---- x2 / x3 sweeps on an LDS tile (used for the correct pass) ------------------------------------
// A block of 64 x BT threads owns 64 independent columns (consecutive i) and BT consecutive cells
// of each along the sweep direction; W goes through LDS once ([comp][BT+2][64]: a wavefront reads
// 64 consecutive doubles, conflict-free), neighbours and the Wl hand-off come from LDS, blocks
// overlap by one cell.  One cell per thread keeps the correct pass at ~130 VGPRs (3 waves/SIMD)
// where the marching form needs 255 (1 wave/SIMD, latency-bound: 32 ms instead of 9 at 512^3).
template <int NS, int D, bool GRAV, int MODE, int BT, int ORD>
__global__ void __launch_bounds__(64*BT, 4)
k_sweep_tile(DevGrid g, const Real *src, Real dt)
{
  static_assert(D == 1 || D == 2, "tile kernel is for the strided directions");
  extern __shared__ Real sm[];
  const int lane = threadIdx.x, t = threadIdx.y;
  const int ni = g.ie - g.is + 5;
  const int tlo = (D == 1 ? g.ks : g.js) - 2;
  const int nt  = (D == 1 ? g.ke - g.ks : g.je - g.js) + 5;
  const long colid = (long)blockIdx.x*64 + lane;
  const bool colok = colid < (long)ni*nt;
  const int i = g.is - 2 + (int)(colid % ni);
  const int tt = tlo + (int)(colid / ni);
  const int lo = (D == 1 ? g.js : g.ks), hi = (D == 1 ? g.je : g.ke);
  const int c0 = lo - 2 + blockIdx.y*(BT - 1);
  const int c = c0 + t;
  const long s = stride<D>(g);
  const long base = (D == 1) ? ((long)tt*g.sK + i) : ((long)tt*g.sJ + i);
  const Real dtodx = dt/g.dx[D];
  constexpr int P = (BT + 2)*64;                       // LDS pitch per component
  Real u[6], w[6];
  const bool have = colok && (c <= hi + 3);
  if (have) { load_sweep<D, NS>(src, g.nc, base + (long)c*s, u); cons_to_prim<NS>(u, w, g.Gamma_1); }
  else {
#pragma unroll
    for (int n = 0; n < 6; n++) w[n] = 1.0;
  }
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) sm[n*P + (t + 1)*64 + lane] = w[n];
  if (t == 0 || t == BT - 1) {                         // halo cells c0-1 and c0+BT
    const int ch = (t == 0) ? c0 - 1 : c + 1;
    Real uh[6], wh[6];
    if (colok && ch <= hi + 3) { load_sweep<D, NS>(src, g.nc, base + (long)ch*s, uh); cons_to_prim<NS>(uh, wh, g.Gamma_1); }
    else {
#pragma unroll
      for (int n = 0; n < 6; n++) wh[n] = 1.0;
    }
    const int row = (t == 0) ? 0 : BT + 1;
#pragma unroll
    for (int n = 0; n < 5 + NS; n++) sm[n*P + row*64 + lane] = wh[n];
  }
  __syncthreads();
  Real wm[6], wp[6], wl_next[6], wr[6];
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) { wm[n] = sm[n*P + t*64 + lane]; wp[n] = sm[n*P + (t + 2)*64 + lane]; }
  if (!NS) { wm[5] = 0.0; wp[5] = 0.0; }
  const bool recon = colok && (c <= hi + 2);
  if (recon) recon_cell<NS, MODE != MODE_VL, ORD, D>(g, base + (long)c*s, wm, w, wp, dtodx, wl_next, wr);
  else {
#pragma unroll
    for (int n = 0; n < 6; n++) { wl_next[n] = 1.0; wr[n] = 1.0; }
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) sm[n*P + t*64 + lane] = wl_next[n];
  __syncthreads();
  if (t >= 1 && recon) {
    Real wl[6];
#pragma unroll
    for (int n = 0; n < 5 + NS; n++) wl[n] = sm[n*P + (t - 1)*64 + lane];
    if (!NS) wl[5] = 0.0;
    face_work<NS, D, GRAV, MODE>(g, base + (long)c*s, i, D == 1 ? c : tt, D == 1 ? tt : c, dt, wl, wr);
  }
}

// ---- step 1: x1 sweep, neighbours through LDS ------------------------------------------
// A block of B threads reconstructs B consecutive cells of one (j,k) row and solves the B-1
// interfaces between them; blocks overlap by one cell.
#ifndef SW_X1_FLAT
#define SW_X1_FLAT 1
#endif
// The x1 sweep with the rows of a k-plane laid end to end (SW_X1_FLAT, the default): a plane's (je-js+5) rows of nc = ie-is+5
// cells l..u are ONE line of slots, cut into blocks of B-1 faces wherever they fall, so that only the plane's last block has
// idle lanes (a block per row piece left 576 lanes for 516 cells at 512^3, 128 for 68 at 64^3).  A row's first cell has no
// face to solve and its lower neighbour is not the slot before it: the lanes at a row's (or the block's) ends fetch and convert
// that one neighbour themselves, everybody else finds it in LDS.  slot -> (row, cell) by a multiply-shift with the host's
// reciprocal (exact below 2^30 slots per plane: launch_sweep checks).
template <int NS, bool GRAV, int MODE, int ORD>
__global__ void __launch_bounds__(256, 4)
k_sweep_x1_flat(DevGrid g, const Real *src, Real dt, int koff, unsigned nslots, unsigned rmul, int rsh)
{
  extern __shared__ Real sm[];
  const int B = blockDim.x, t = threadIdx.x;
  const int nc = g.ie - g.is + 5;                      // cells l..u of a row
  const unsigned s = blockIdx.x*(unsigned)(B - 1) + t;
  const bool have = (s < nslots);
  const unsigned r = have ? (unsigned)(((unsigned long long)s*rmul) >> rsh) : 0u;
  const int ci = have ? (int)(s - r*(unsigned)nc) : 1;
  const int j = g.js - 2 + (int)r, k = g.ks - 2 + koff + blockIdx.y;
  const int c = g.is - 2 + ci;
  const long row = (long)k*g.sK + (long)j*g.sJ;
  const Real dtodx = dt/g.dx[0];
  const bool lo_own = have && (t == 0 || ci == 0);          // the lower / upper neighbour is not in this block's LDS line
  const bool hi_own = have && (t == B - 1 || ci == nc - 1 || s + 1 == nslots);
  Real u[6], w[6], wm[6], wp[6], ulo[6], uhi[6];
  if (have) load_sweep<0, NS>(src, g.nc, row + c, u);
  if (lo_own) load_sweep<0, NS>(src, g.nc, row + c - 1, ulo);
  if (hi_own) load_sweep<0, NS>(src, g.nc, row + c + 1, uhi);
  if (have) cons_to_prim<NS>(u, w, g.Gamma_1);
  else {
#pragma unroll
    for (int n = 0; n < 6; n++) w[n] = 1.0;
  }
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) sm[n*B + t] = w[n];
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) { wm[n] = sm[n*B + (t > 0 ? t - 1 : 0)]; wp[n] = sm[n*B + (t < B - 1 ? t + 1 : t)]; }
  if (!NS) { wm[5] = 0.0; wp[5] = 0.0; }
  if (lo_own) cons_to_prim<NS>(ulo, wm, g.Gamma_1);
  if (hi_own) cons_to_prim<NS>(uhi, wp, g.Gamma_1);
  Real wl_next[6], wr[6];
  if (have) recon_cell<NS, MODE != MODE_VL, ORD, 0>(g, row + c, wm, w, wp, dtodx, wl_next, wr);
  else {
#pragma unroll
    for (int n = 0; n < 6; n++) { wl_next[n] = 1.0; wr[n] = 1.0; }
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) sm[n*B + t] = wl_next[n];
  __syncthreads();
  if (t >= 1 && ci >= 1 && have) {                     // interface c, between cells c-1 and c
    Real wl[6];
#pragma unroll
    for (int n = 0; n < 5 + NS; n++) wl[n] = sm[n*B + t - 1];
    if (!NS) wl[5] = 0.0;
    face_work<NS, 0, GRAV, MODE>(g, row + c, c, j, k, dt, wl, wr);
  }
}

template <int NS, bool GRAV, int MODE, int ORD>
__global__ void __launch_bounds__(256, 4)
k_sweep_x1(DevGrid g, const Real *src, Real dt, int koff)
{
  extern __shared__ Real sm[];
  const int B = blockDim.x, t = threadIdx.x;
  const int j = g.js - 2 + blockIdx.y, k = g.ks - 2 + koff + blockIdx.z;
  const int c0 = g.is - 2 + blockIdx.x*(B - 1);
  const int cend = c0 + B - 1;
  const int c = c0 + t;
  const long row = (long)k*g.sK + (long)j*g.sJ;
  const Real dtodx = dt/g.dx[0];
  const int P = B + 2;                                 // LDS pitch per component
  Real u[6], w[6];
  const bool have = (c <= g.ie + 3);                   // cells up to ie+3 feed the stencil
  if (have) { load_sweep<0, NS>(src, g.nc, row + c, u); cons_to_prim<NS>(u, w, g.Gamma_1); }
  else {
#pragma unroll
    for (int n = 0; n < 6; n++) w[n] = 1.0;
  }
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) sm[n*P + t + 1] = w[n];
  if (t == 0) {                                        // lower halo cell c0-1 (>= is-3)
    Real uh[6], wh[6];
    load_sweep<0, NS>(src, g.nc, row + c0 - 1, uh); cons_to_prim<NS>(uh, wh, g.Gamma_1);
#pragma unroll
    for (int n = 0; n < 5 + NS; n++) sm[n*P] = wh[n];
  }
  if (t == B - 1) {                                    // upper halo cell c0+B
    Real uh[6], wh[6];
    if (c + 1 <= g.ie + 3) { load_sweep<0, NS>(src, g.nc, row + c + 1, uh); cons_to_prim<NS>(uh, wh, g.Gamma_1); }
    else {
#pragma unroll
      for (int n = 0; n < 6; n++) wh[n] = 1.0;
    }
#pragma unroll
    for (int n = 0; n < 5 + NS; n++) sm[n*P + B + 1] = wh[n];
  }
  __syncthreads();
  Real wm[6], wp[6], wl_next[6], wr[6];
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) { wm[n] = sm[n*P + t]; wp[n] = sm[n*P + t + 2]; }
  if (!NS) { wm[5] = 0.0; wp[5] = 0.0; }
  const bool recon = (c <= g.ie + 2);                  // cells l..u
  if (recon) recon_cell<NS, MODE != MODE_VL, ORD, 0>(g, row + c, wm, w, wp, dtodx, wl_next, wr);
  else {
#pragma unroll
    for (int n = 0; n < 6; n++) { wl_next[n] = 1.0; wr[n] = 1.0; }
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) sm[n*P + t] = wl_next[n];
  __syncthreads();
  if (t >= 1 && recon && c <= cend) {                  // interface c, between cells c-1 and c
    Real wl[6];
#pragma unroll
    for (int n = 0; n < 5 + NS; n++) wl[n] = sm[n*P + t - 1];
    if (!NS) wl[5] = 0.0;
    face_work<NS, 0, GRAV, MODE>(g, row + c, c, j, k, dt, wl, wr);
  }
}

// ---- steps 5-8a, 9a for all three directions in one marching kernel ---------------------------------
// Per CELL instead of per face: the two face states a cell's reconstruction along D produces (left
// state of its upper face, right state of its lower face) take the SAME transverse correction, the
// first-pass flux differences across that cell (integrate_3d_ctu.c:978-1158, :1282-1452, :1691-1863),
// so a thread that owns a zone loads its six first-pass fluxes per variable once and corrects all six
// face states with them.  A thread owns one (i,j) column and marches along x3 with the primitive
// variables of three planes in registers (the x3 stencil) and the upper x3 flux carried over as the next
// zone's lower one; the x1 neighbours come by wavefront shuffle, the x2 neighbours through LDS.  Only
// eta couples neighbouring zones: eta of a face = 0.5 |lambda_r(right state) - lambda_l(left state)|
// needs one scalar from the zone below (shuffle / LDS / carried).  Against the three separate correct
// passes U and the first-pass fluxes are read once instead of three / two times and no flux plane is
// re-read; lane 0 and row 0 of a block repeat the last lane / row of the block before as providers.
// Every expression is face_correct's, in the same order: bit-identical under -ffp-contract=off.
#ifndef CA_TJ
#define CA_TJ 4
#endif
// Primitive variables of one zone for all three sweeps.  cons_to_prim sums the squared momenta in the
// order of the sweep frame, (Mx^2 + My^2) + Mz^2 with x = D, so the pressure differs in the last bit
// between the frames: w (global frame d, V1, V2, V3, P, r) carries the pressure of the x3 sweep, p0 and
// p1 are the pressures the x1 and x2 sweeps see.
template <int NS>
AA_DEV void load_prim3(const DevGrid &g, long m, Real w[6], Real &p0, Real &p1)
{
  Real u[6];
#pragma unroll
  for (int v = 0; v < 5 + NS; v++) u[v] = Uf(g, v)[m];
  if (!NS) u[5] = 0.0;
#if AA_FD_C2P
  const Real di = q_rcp(u[0]);          // as cons_to_prim: the same bits whichever kernel converts the zone
#else
  const Real di = 1.0/u[0];
#endif
  w[0] = u[0]; w[1] = u[1]*di; w[2] = u[2]*di; w[3] = u[3]*di;
  Real pa = u[4] - 0.5*(sqr(u[1]) + sqr(u[2]) + sqr(u[3]))*di;
  Real pb = u[4] - 0.5*(sqr(u[2]) + sqr(u[3]) + sqr(u[1]))*di;
  Real pc = u[4] - 0.5*(sqr(u[3]) + sqr(u[1]) + sqr(u[2]))*di;
  pa *= g.Gamma_1; pb *= g.Gamma_1; pc *= g.Gamma_1;
  p0 = rmax(pa, AA_TINY); p1 = rmax(pb, AA_TINY); w[4] = rmax(pc, AA_TINY);
  w[5] = NS ? u[5]*di : 0.0;
}
// sweep-frame primitives of a halo zone, exactly as the sweeps load them
template <int NS, int D>
AA_DEV void load_prim_sweep(const DevGrid &g, long m, Real w[6])
{
  Real u[6];
  load_sweep<D, NS>(g.U, g.nc, m, u);
  cons_to_prim<NS>(u, w, g.Gamma_1);
}
// global frame (d, V1, V2, V3, ., r) with the pressure of frame D -> sweep frame of D
template <int D> AA_DEV void to_sweep(const Real wg[6], Real p, Real ws[6])
{
#pragma unroll
  for (int n = 0; n < 6; n++) ws[n] = wg[gv<D>(n)];
  ws[4] = p;
}

#ifndef AA_NT_ST
#define AA_NT_ST 1
#endif
struct CellFlux {            // first-pass fluxes across one zone
  Real dF[3][6];             // F_e(upper face) - F_e(lower face), global frame
  Real gm[3], ge[3];         // gravity: q_e (phi_r - phi_l) d   and   q_e (F_lo (phi_c - phi_l) + F_hi (phi_r - phi_c))
  Real kl[3], kr[3];         // gravity kick of the zone's left / right state along e: dt/dx_e (phi_up - phi_c), dt/dx_e (phi_c - phi_lo)
};

// one zone along D, first half: reconstruction + gravity kick = the primitive states the FIRST pass solves its Riemann
// problems with (face_work): the left state belongs to the zone's upper face, the right state to its lower one
template <int NS, int D, bool GRAV, int ORD>
AA_DEV void cell_recon(const DevGrid &g, long m, Real dt, Real dtodx, const Real wm[6], const Real w[6], const Real wp[6], Real wl[6], Real wr[6])
{
  const long sD = stride<D>(g);
  recon_cell<NS, true, ORD, D>(g, m, wm, w, wp, dtodx, wl, wr);
  if (GRAV) {
    const Real phic = Pf(g, 0)[m], phi_up = Pf(g, 1 + D)[m + sD], phi_lo = Pf(g, 1 + D)[m];
    wl[1] -= dtodx*(phi_up - phic);
    wr[1] -= dtodx*(phic - phi_lo);
  }
#if AA_COOLING
  cool_states(g, dt, wl, wr);
#endif
}
template <int NS, int D, bool GRAV, int ORD>
AA_DEV void cell_recon(const DevGrid &g, long m, Real dt, const Real wm[6], const Real w[6], const Real wp[6], Real wl[6], Real wr[6])
{ cell_recon<NS, D, GRAV, ORD>(g, m, dt, dt/g.dx[D], wm, w, wp, wl, wr); }
// the same with the kicks of the zone taken from its CellFlux (formed at the head of the iteration from the potential
// values loaded there: a load inside the direction blocks would wait for the stores of the block before it)
template <int NS, int D, bool GRAV, int ORD>
AA_DEV void cell_recon(const DevGrid &g, long m, Real dt, Real dtodx, const CellFlux &cf, const Real wm[6], const Real w[6], const Real wp[6],
                       Real wl[6], Real wr[6])
{
  recon_cell<NS, true, ORD, D>(g, m, wm, w, wp, dtodx, wl, wr);
  if (GRAV) { wl[1] -= cf.kl[D]; wr[1] -= cf.kr[D]; }
#if AA_COOLING
  cool_states(g, dt, wl, wr);
#endif
}
// second half: conversion, transverse correction, face states stored; returns the wave speeds eta needs: lam_l of the
// left state the zone gave to its UPPER face, lam_r of the right state it gave to its LOWER face
template <int NS, int D, bool GRAV>
AA_DEV void cell_finish(const DevGrid &g, long m, const Real q[3], const CellFlux &cf, const Real wl[6], const Real wr[6],
                        bool store, Real &lam_l, Real &lam_r)
{
  constexpr int NV = 5 + NS;
  const long sD = stride<D>(g);
  Real sl[6], sr[6], ul[6], ur[6];
  prim_to_cons<NS>(wl, sl, g.Gamma_1, g.rGamma_1);
  prim_to_cons<NS>(wr, sr, g.Gamma_1, g.rGamma_1);
#pragma unroll
  for (int n = 0; n < 6; n++) { ul[gv<D>(n)] = sl[n]; ur[gv<D>(n)] = sr[n]; }   // to the global frame
#pragma unroll
  for (int e = 0; e < 3; e++) {
    if (e == D) continue;
#pragma unroll
    for (int v = 0; v < NV; v++) { ul[v] -= q[e]*cf.dF[e][v]; ur[v] -= q[e]*cf.dF[e][v]; }
  }
  if (GRAV) {
#pragma unroll
    for (int e = 0; e < 3; e++) {
      if (e == D) continue;
      ur[1 + e] -= cf.gm[e]; ur[4] -= cf.ge[e];
      ul[1 + e] -= cf.gm[e]; ul[4] -= cf.ge[e];
    }
  }
  if (store) {
#pragma unroll
    for (int v = 0; v < NV; v++) {      // written here, read once by the next kernel: non-temporal (AA_NT_ST=0: plain stores)
      if (AA_NT_ST) { __builtin_nontemporal_store(ul[v], LRf(g, D, 0, v) + m + sD); __builtin_nontemporal_store(ur[v], LRf(g, D, 1, v) + m); }
      else { LRf(g, D, 0, v)[m + sD] = ul[v]; LRf(g, D, 1, v)[m] = ur[v]; }
    }
  }
#pragma unroll
  for (int n = 0; n < 6; n++) { sl[n] = ul[gv<D>(n)]; sr[n] = ur[gv<D>(n)]; }
  lam_r = lambda_face(sr, g.Gamma, g.Gamma_1, 1.0);
  lam_l = lambda_face(sl, g.Gamma, g.Gamma_1, -1.0);
}
template <int NS, int D, bool GRAV, int ORD>
AA_DEV void cell_states(const DevGrid &g, long m, Real dt, Real dtodx, const Real q[3], const CellFlux &cf, const Real wm[6], const Real w[6],
                        const Real wp[6], bool store, Real &lam_l, Real &lam_r)
{
  Real wl[6], wr[6];
  cell_recon<NS, D, GRAV, ORD>(g, m, dt, dtodx, cf, wm, w, wp, wl, wr);
  cell_finish<NS, D, GRAV>(g, m, q, cf, wl, wr, store, lam_l, lam_r);
}

#ifndef CA_PARK
#define CA_PARK 1
#endif
// CA_X1F: with the x3 first pass on board (X3F), k_correct_all also does the x1 FIRST pass: a zone's reconstruction along x1 is the
// same for the first pass and for the correct pass, the right state of its upper face is one lane away, and the first-pass
// fluxes it needs are those of its own two faces.  k_sweep_x1_flat is not launched and its fluxes never reach HBM; what a tile
// of 64 zones cannot do alone, the flux of the face it shares with the next tile, comes from k_x1_edge_flux (1/64 of the faces),
// which keeps its result where the x1 flux array was: [variable][edge][k][j].  Same arithmetic, same bits.
#ifndef CA_X1F
#define CA_X1F 1
#endif
bool ca_x1_on_board() { return CA_X1F != 0; }
__host__ __device__ static inline int x1_edges(const DevGrid &g) { return (g.ie + 2 - (g.is - 16))/64; }   // edge b = 1 .. : face is-16+64b <= ie+2
AA_DEV long x1_edge_index(const DevGrid &g, int nbe, int v, int b, int j, int k) { return (((long)v*nbe + (b - 1))*g.N3 + k)*g.N2 + j; }
template <int NS, bool GRAV, int ORD>
__global__ void __launch_bounds__(64)
k_x1_edge_flux(DevGrid g, Real dt)
{
  // lanes along j: the stores on whole lines (the loads are one sector per zone pair either way)
  const int j = g.js - 1 + blockIdx.x*64 + threadIdx.x, k = g.ks - 1 + blockIdx.y, b = 1 + blockIdx.z;
  if (j > g.je + 1) return;
  const int nbe = x1_edges(g), i0 = g.is - 16 + 64*b;
  const long m = (long)k*g.sK + (long)j*g.sJ + i0;
  Real wa[6], wb[6], wc[6], wd[6], wl[6], wr[6], wx[6];
  load_prim_sweep<NS, 0>(g, m - 2, wa); load_prim_sweep<NS, 0>(g, m - 1, wb);
  load_prim_sweep<NS, 0>(g, m, wc);     load_prim_sweep<NS, 0>(g, m + 1, wd);
  cell_recon<NS, 0, GRAV, ORD>(g, m - 1, dt, wa, wb, wc, wl, wx);      // zone i0-1: its left state belongs to face i0
  cell_recon<NS, 0, GRAV, ORD>(g, m, dt, wb, wc, wd, wx, wr);          // zone i0: its right state
  Real ul[6], ur[6], f[6];
  prim_to_cons<NS>(wl, ul, g.Gamma_1, g.rGamma_1);
  prim_to_cons<NS>(wr, ur, g.Gamma_1, g.rGamma_1);
  flux_roe<NS>(ul, ur, wl, wr, 0.0, g.Gamma, g.Gamma_1, f);
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) g.F[x1_edge_index(g, nbe, gv<0>(n), b, j, k)] = f[n];
}
template <int NS, bool GRAV, int ORD, bool X3F>
__global__ void __launch_bounds__(64*CA_TJ, (X3F && NS && GRAV) ? 2 : 1)
k_correct_all(DevGrid g, Real dt, int kchunk, StepRatios sr)
{
  __shared__ Real s_w[CA_TJ][6][64];
  __shared__ Real s_l[CA_TJ][64];
  // X3F: the x3 states of zone k+1 wait here while zone k is corrected (each thread its own slots, no barrier): 24
  // registers that the 6-variable gravity kernel does not have at 2 waves per SIMD (it spilled 16 to scratch)
  // (CA_PARK3, with the x1 first pass on board: two sets of slots taken in turn, so that the states of zone k need no registers
  //  between the iterations nor under the x1 Riemann problem: they are read where they are used)
#ifndef CA_PARK3
#define CA_PARK3 1
#endif
  constexpr bool PARK3 = X3F && (CA_X1F != 0) && (CA_PARK != 0) && (CA_PARK3 != 0);
  __shared__ Real s_park[(X3F && CA_PARK) ? (PARK3 ? 24 : 12) : 1][CA_TJ][64];
  // ... and in the 6-variable gravity kernel, which spilled three registers, the x1 / x2 frame pressures of planes k+1 and k+2 wait
  // there too (each thread its own slots): no scratch, 18.53 -> 18.18 ms at 512^3 (same-box ABAB x 3; CA_PARK2=0: in registers)
#ifndef CA_PARK2
#define CA_PARK2 1
#endif
  constexpr bool PARK2 = (CA_PARK2 != 0) && X3F && NS && GRAV;
  __shared__ Real s_pp[PARK2 ? 4 : 1][CA_TJ][64];
  // The zones a tile needs from its neighbours (lane 0's lower / lane 63's upper x1 neighbour, row 0's lower / row CA_TJ-1's
  // upper x2 neighbour) are requested at the head of the iteration with everything else and wait here for their block: a load
  // issued inside a block would have to wait for the face-state stores of the block before it (loads and stores retire
  // through one counter).  s_h1: per wave, the 6 + 6 conserved variables of the two x1 neighbours, fetched by lanes 0..11.
  // X1F: and behind them the first-pass x1 fluxes of the tile's two edge faces (lanes 12..23)
  constexpr bool X1F = X3F && (CA_X1F != 0);
  __shared__ Real s_h1[CA_TJ][X1F ? 24 : 12];
  __shared__ Real s_h2[2][6][64];
  constexpr int NV = 5 + NS;
  const int lane = threadIdx.x, row = threadIdx.y;
  // Zones s-1 .. e+1 in every direction get their face states.  Tiles do NOT overlap in x1 / x2 (round 1's did by one
  // provider lane / row: 64x4 threads for 63x3 zones, 35 % more threads, loads and arithmetic than zones) and start on a
  // 128-byte line (zone is - 16: the rows' first active zone is line-aligned).  What a tile cannot do alone is the eta of
  // its own lowest x1 / x2 faces (lane 0, row 0), which needs lambda_l of the zone in the tile before: there the zone
  // stores its lambda_r in the face's eta slot, the zone before (lane 63 / row CA_TJ-1) its lambda_l in an edge array,
  // and k_eta_edges turns the pairs into etas afterwards (1/64 + 1/CA_TJ of the faces, < 10 B/zone).
  // (Tried and dropped: an XCD-aware blockIdx -> tile map that gives each XCD one contiguous run of tiles, so that the tiles
  //  either side of an edge share an L2.  Measured ABAB at 512^3: 21.3 vs 20.3 ms here, 15.1 vs 14.9 ms in k_flux2_update —
  //  the plain map, which spreads neighbouring tiles over all eight XCDs and memory channels, is the faster one.)
  const int i = g.is - 16 + blockIdx.x*64 + lane, j = g.js - 1 + blockIdx.y*CA_TJ + row;
  const int k0 = g.ks - 1 + blockIdx.z*kchunk;
  int k1 = k0 + kchunk - 1; if (k1 > g.ke + 1) k1 = g.ke + 1;
  const int kstart = (blockIdx.z == 0) ? k0 : k0 - 1;          // one provider plane below a later chunk
  // idle threads beyond the Grid keep valid addresses (their neighbours may read what they load)
  // (X1F: zone ie+2 gives its right state to the first-pass flux of face ie+2, so the lane of zone ie+3 holds that zone)
  const int imax = g.ie + (X1F ? 3 : 2);
  const int ic = (i <= imax) ? (i < 0 ? 0 : i) : imax, jc = (j <= g.je + 2) ? j : g.je + 2;
  const long mcol = (long)jc*g.sJ + ic;
  const int iW = i - lane, iE = i - lane + 63;                 // lane 0's and lane 63's zone, clamped like ic
  const int icW = (iW <= imax) ? (iW < 0 ? 0 : iW) : imax, icE = (iE <= imax) ? (iE < 0 ? 0 : iE) : imax;
  const int nbe = x1_edges(g);
  const bool in = (i >= g.is - 1) && (i <= g.ie + 1) && (j <= g.je + 1);
  const bool do1 = in, do2 = in, do3 = in;
  Real q[3];
#pragma unroll
  for (int d = 0; d < 3; d++) q[d] = sr.q[d];

  // X3F: the x3 FIRST pass rides on the march (k_sweep_march<2> is not launched, its fluxes never reach HBM): the
  // transverse differences a zone needs of the x3 first-pass flux are those of its own column, and the reconstruction
  // + gravity kick of a zone along x3 is the same for the first pass and for the correct pass.  Iteration k
  // reconstructs zone k+1 (A), solves face k+1 from the left state zone k kept and the right state of zone k+1 (B),
  // and then corrects zone k in all three directions (C).  A chunk starts two planes early with A only, then A + B: one
  // instance of every piece of arithmetic, whatever the chunking (see k_sweep_march).
  // The window: wc = zone k, wn = k+1 (and wm3 = k-1 without X3F, wn2 = k+2 with it).
  Real wm3[6], wc[6], wn[6], wn2[6], pc0 = 0.0, pc1 = 0.0, pn0, pn1, pn20 = 0.0, pn21 = 0.0, f3[6], lam3 = 0.0;
  Real wl3[6], wr3[6];                                          // X3F: kicked x3 states of zone k (left -> face k+1, right -> face k)
  const int kbeg = X3F ? kstart - 2 : kstart;
  if (X3F) {
    load_prim3<NS>(g, (long)kbeg*g.sK + mcol, wn, pn0, pn1);
    load_prim3<NS>(g, (long)(kbeg + 1)*g.sK + mcol, wn2, pn20, pn21);
    if (PARK2) { s_pp[0][row][lane] = pn0; s_pp[PARK2 ? 1 : 0][row][lane] = pn1; s_pp[PARK2 ? 2 : 0][row][lane] = pn20; s_pp[PARK2 ? 3 : 0][row][lane] = pn21; }
#pragma unroll
    for (int v = 0; v < 6; v++) { f3[v] = 0.0; wl3[v] = 1.0; wr3[v] = 1.0; wc[v] = 1.0; }
  } else {
    load_prim3<NS>(g, (long)(kstart - 1)*g.sK + mcol, wc, pn0, pn1);
    load_prim3<NS>(g, (long)kstart*g.sK + mcol, wn, pn0, pn1);
#pragma unroll
    for (int v = 0; v < 6; v++) f3[v] = (v < NV) ? Ff(g, 2, v)[(long)kstart*g.sK + mcol] : 0.0;
  }
#pragma nounroll
  for (int k = kbeg; k <= k1; k++) {
    long m = (long)k*g.sK + mcol;
    asm volatile("" : "+v"(m));                   // one index for all fields (see k_flux2_update)
    const bool full = (k >= k0);                  // block-uniform; the provider plane does x3 only
    const int pk = (k & 1) ? 12 : 0;              // PARK3: the slots of zone k (those of zone k+1: 12 - pk)
    const bool zone = (k >= kstart);              // block-uniform; X3F: the two planes before do the first pass only
    Real f3n[6], wlN[6], wrN[6];
#pragma unroll
    for (int v = 0; v < 6; v++) { f3n[v] = 0.0; wlN[v] = 1.0; wrN[v] = 1.0; }
    // first-pass fluxes across the zone
    CellFlux cf;
#pragma unroll
    for (int v = 0; v < 6; v++) { cf.dF[0][v] = 0.0; cf.dF[1][v] = 0.0; cf.dF[2][v] = 0.0; }
    Real mlo[3], mhi[3];
    Real hv1 = 0.0;
    // What the zone needs behind its x1 Riemann problem is requested in front of it where the registers allow: the wait would
    // otherwise come behind it with nothing left to do.  CA_X1F_EARLY=1 (the default): the x2 first-pass fluxes -- k_correct_all 22.6 ->
    // 20.2 ms at 512^3, three registers in scratch.  (The potentials and the x2 neighbour rows as well: tens of registers in scratch, a
    // loss -- also with the potentials parked in LDS; profiles/r04_x1f_ab.txt.)  The scalar kernel without gravity (ifront) has no room
    // at two waves per SIMD: CA_X1F_EARLY_NG, default 0.
#ifndef CA_X1F_EARLY
#define CA_X1F_EARLY 1
#endif
#ifndef CA_X1F_EARLY_NG
#define CA_X1F_EARLY_NG 0
#endif
    constexpr bool EARLY_F = X1F && (((NS && !GRAV) ? CA_X1F_EARLY_NG : CA_X1F_EARLY) != 0);
    Real hv2[6];
#pragma unroll
    for (int v = 0; v < 6; v++) hv2[v] = 0.0;
    const bool edge_row = (row == 0) || (row == CA_TJ - 1);
    // the x1 neighbour zones and edge fluxes of plane kk (lanes 0..23 of every wave)
    auto x1_halo = [&](int kk) -> Real {
      Real h = 0.0;
      if (lane < 12) {
        const int v = lane % 6, east = lane / 6;
        if (v < NV) h = Uf(g, v)[(long)kk*g.sK + (long)jc*g.sJ + (east ? icE + 1 : icW - 1)];
      } else if (lane < 24) {
        const int v = (lane - 12) % 6, b = (int)blockIdx.x + (lane - 12)/6;      // the tile's lower edge face, then its upper one
        if (v < NV && b >= 1 && b <= nbe) h = g.F[x1_edge_index(g, nbe, v, b, jc, kk)];
      }
      return h;
    };
    // requested here, used behind the x3 first pass (the provider plane too: its x3 states take x1 fluxes made here)
    if (X1F && zone) hv1 = x1_halo(k);
    if (X3F) {
#pragma unroll
      for (int n = 0; n < 6; n++) { wc[n] = wn[n]; wn[n] = wn2[n]; }
      if (PARK2) { pc0 = s_pp[0][row][lane]; pc1 = s_pp[1][row][lane]; s_pp[0][row][lane] = s_pp[PARK2 ? 2 : 0][row][lane]; s_pp[PARK2 ? 1 : 0][row][lane] = s_pp[PARK2 ? 3 : 0][row][lane]; }
      else { pc0 = pn0; pc1 = pn1; pn0 = pn20; pn1 = pn21; }
      load_prim3<NS>(g, m + 2*g.sK, wn2, pn20, pn21);
      if (PARK2) { s_pp[PARK2 ? 2 : 0][row][lane] = pn20; s_pp[PARK2 ? 3 : 0][row][lane] = pn21; }
      if (do3) {
        {   // (A) zone k+1 along x3
          Real wm[6], ws[6], wp[6];
          to_sweep<2>(wc, wc[4], wm); to_sweep<2>(wn, wn[4], ws); to_sweep<2>(wn2, wn2[4], wp);
          cell_recon<NS, 2, GRAV, ORD>(g, m + g.sK, dt, sr.dtodx[2], wm, ws, wp, wlN, wrN);
        }
#if CA_PARK
        if (PARK3) {      // (the states of zone k+1 go to their slots before (B): the left one is not needed in it)
#pragma unroll
          for (int n = 0; n < NV; n++) { s_park[(PARK3 ? 12 - pk : 0) + n][row][lane] = wlN[n]; s_park[(PARK3 ? 12 - pk : 0) + 6 + n][row][lane] = wrN[n]; }
        }
#endif
        if (k >= kstart - 1) {   // (B) first-pass flux of face k+1 (face_work<MODE_FLUX1>), sweep frame -> global variables
          Real ul[6], ur[6], f[6];
          if (PARK3) {
#pragma unroll
            for (int n = 0; n < 6; n++) wl3[n] = (n < NV) ? s_park[(PARK3 ? pk : 0) + n][row][lane] : 0.0;
          }
          prim_to_cons<NS>(wl3, ul, g.Gamma_1, g.rGamma_1);
          prim_to_cons<NS>(wrN, ur, g.Gamma_1, g.rGamma_1);
          flux_roe<NS>(ul, ur, wl3, wrN, 0.0, g.Gamma, g.Gamma_1, f);
#pragma unroll
          for (int n = 0; n < NV; n++) f3n[gv<2>(n)] = f[n];
        }
      }
#if CA_PARK
      if (!PARK3 || !do3) {
#pragma unroll
        for (int n = 0; n < NV; n++) { s_park[(PARK3 ? 12 - pk : 0) + n][row][lane] = wlN[n]; s_park[(PARK3 ? 12 - pk : 0) + 6 + n][row][lane] = wrN[n]; }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int n = 0; n < 6; n++) { wm3[n] = wc[n]; wc[n] = wn[n]; }
      pc0 = pn0; pc1 = pn1;
      load_prim3<NS>(g, m + g.sK, wn, pn0, pn1);
    }
    if (!X1F && full) {
      if (lane < 12) {
        const int v = lane % 6, east = lane / 6;
        if (v < NV) hv1 = Uf(g, v)[(long)k*g.sK + (long)jc*g.sJ + (east ? icE + 1 : icW - 1)];
      }
    }
    Real wl1[6], wr1[6];                          // X1F: the zone's kicked x1 states
#pragma unroll
    for (int v = 0; v < 6; v++) { wl1[v] = 1.0; wr1[v] = 1.0; }
    // X1F: what the zone needs behind its x1 Riemann problem (x2 first-pass fluxes, potentials, the x2 neighbour rows) is requested
    // in front of it -- the wait would otherwise come after it with nothing left to do (CA_X1F_EARLY=0: requested where it is used)
    Real eb0[6], eb1[6];
#pragma unroll
    for (int v = 0; v < 6; v++) { eb0[v] = 0.0; eb1[v] = 0.0; }
    if (EARLY_F && zone) {
#pragma unroll
      for (int v = 0; v < NV; v++) { eb0[v] = Ff(g, 1, v)[m]; eb1[v] = Ff(g, 1, v)[m + g.sJ]; }
    }
    if (X1F && zone) {   // ---- the x1 first pass, before anything else of the zone is formed: its fluxes correct the x2 and x3 states ----
      if (lane < 24) s_h1[row][lane] = hv1;
      __builtin_amdgcn_sched_barrier(0);
      Real wm[6], wp[6], ws[6];
      to_sweep<0>(wc, pc0, ws);
#pragma unroll
      for (int n = 0; n < 6; n++) { wm[n] = __shfl_up(ws[n], 1); wp[n] = __shfl_down(ws[n], 1); }
      if (lane == 0 || lane == 63) {      // the neighbour zone of the tile before / after, exactly as the sweeps convert it
        Real u[6];
#pragma unroll
        for (int v = 0; v < 6; v++) u[v] = (v < NV) ? s_h1[row][(lane ? 6 : 0) + v] : 0.0;
        Real wh[6];
        cons_to_prim<NS>(u, wh, g.Gamma_1);
#pragma unroll
        for (int v = 0; v < 6; v++) { if (lane == 0) wm[v] = wh[v]; else wp[v] = wh[v]; }
      }
      if (GRAV) {      // (the zone's kicks along x1; formed again with the others below)
        const Real phic = Pf(g, 0)[m], phir = Pf(g, 1)[m + stride_rt(g, 0)], phil = Pf(g, 1)[m], dtodx = sr.dtodx[0];
        cf.kl[0] = dtodx*(phir - phic); cf.kr[0] = dtodx*(phic - phil);
      }
      cell_recon<NS, 0, GRAV, ORD>(g, m, dt, sr.dtodx[0], cf, wm, ws, wp, wl1, wr1);
      {   // first-pass flux of the zone's upper face (face_work<MODE_FLUX1>): its right state is the lane above's
        Real wrE[6], ul[6], ur[6], f[6];
#pragma unroll
        for (int n = 0; n < 6; n++) wrE[n] = __shfl_down(wr1[n], 1);
        prim_to_cons<NS>(wl1, ul, g.Gamma_1, g.rGamma_1);
        prim_to_cons<NS>(wrE, ur, g.Gamma_1, g.rGamma_1);
        flux_roe<NS>(ul, ur, wl1, wrE, 0.0, g.Gamma, g.Gamma_1, f);
#pragma unroll
        for (int n = 0; n < NV; n++) {
          const Real fu = (lane == 63) ? s_h1[row][18 + gv<0>(n)] : f[n];      // the tile's edge faces: k_x1_edge_flux
          Real flo = __shfl_up(fu, 1);
          if (lane == 0) flo = s_h1[row][12 + gv<0>(n)];
          cf.dF[0][gv<0>(n)] = fu - flo;
          if (gv<0>(n) == 0) { mlo[0] = flo; mhi[0] = fu; }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (full) {
      if (edge_row) {
        const long m2 = (row == 0) ? m - g.sJ : m + g.sJ;
#pragma unroll
        for (int v = 0; v < NV; v++) hv2[v] = Uf(g, gv<1>(v))[m2];
      }
    }
    if (zone) {
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const Real a0 = X1F ? 0.0 : Ff(g, 0, v)[m], a1 = X1F ? 0.0 : Ff(g, 0, v)[m + 1];
      const Real b0 = EARLY_F ? eb0[v] : Ff(g, 1, v)[m], b1 = EARLY_F ? eb1[v] : Ff(g, 1, v)[m + g.sJ];
      const Real c0 = f3[v], c1 = X3F ? f3n[v] : Ff(g, 2, v)[m + g.sK];
      if (!X1F) cf.dF[0][v] = a1 - a0;
      cf.dF[1][v] = b1 - b0; cf.dF[2][v] = c1 - c0;
      if (v == 0) { if (!X1F) { mlo[0] = a0; mhi[0] = a1; } mlo[1] = b0; mhi[1] = b1; mlo[2] = c0; mhi[2] = c1; }
      if (!X3F) f3[v] = c1;
    }
    if (GRAV) {
      const Real dc = wc[0], phic = Pf(g, 0)[m];
#pragma unroll
      for (int e = 0; e < 3; e++) {
        const Real phir = Pf(g, 1 + e)[m + stride_rt(g, e)], phil = Pf(g, 1 + e)[m];
        cf.gm[e] = q[e]*(phir - phil)*dc;
        cf.ge[e] = q[e]*(mlo[e]*(phic - phil) + mhi[e]*(phir - phic));
        const Real dtodx = sr.dtodx[e];
        cf.kl[e] = dtodx*(phir - phic); cf.kr[e] = dtodx*(phic - phil);
      }
    }
    // the neighbour zones go to their (wave-private) LDS slots HERE, before the first face-state store of the iteration: the
    // wait for these loads would otherwise sit behind the x3 block's twelve stores and drain them (loads and stores retire
    // through ONE in-order counter: the ISA had an s_waitcnt vmcnt(0) in the middle of every iteration; 18.5 -> 18.3 ms).
    // (Tried on top and dropped, round 3: the x2 block's -- and the x1 block's -- twelve face states parked in LDS and stored at
    //  the head of the NEXT iteration behind its loads, so that no load wait sits behind fresh stores: 18.6-18.9 / 19.5-19.7
    //  against 18.3-18.5 ms.  The write latency is not what this kernel waits for; it moves its 92 GB at 5 TB/s.)
    if (full) {
      if (!X1F && lane < 12) s_h1[row][lane] = hv1;
      if (edge_row) {
#pragma unroll
        for (int v = 0; v < NV; v++) s_h2[row ? 1 : 0][v][lane] = hv2[v];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (X1F && full) {   // ---- x1: the states reconstructed above take their corrections ----
      Real ll = 0.0, lr = 0.0;
      if (do1) cell_finish<NS, 0, GRAV>(g, m, q, cf, wl1, wr1, true, ll, lr);
      const Real lprev = __shfl_up(ll, 1);
      if (do1) {
        if (lane > 0) { if (i > g.is - 1) Ef(g, 0)[m] = 0.5*fabs(lr - lprev); }
        else Ef(g, 0)[m] = lr;                        // tile edge: k_eta_edges finishes this face
        if (lane == 63) Ef(g, 3)[m] = ll;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (do3) {   // ---- x3 in front of x2 (and of x1 where the x1 first pass is not on board): what only this block needs (the plane below / the states zone k kept) is dead under the others ----
      Real ll, lr;
      if (PARK3) {
#pragma unroll
        for (int n = 0; n < 6; n++) { wl3[n] = (n < NV) ? s_park[(PARK3 ? pk : 0) + n][row][lane] : 0.0; wr3[n] = (n < NV) ? s_park[(PARK3 ? pk : 0) + 6 + n][row][lane] : 0.0; }
      }
      if (X3F) cell_finish<NS, 2, GRAV>(g, m, q, cf, wl3, wr3, full, ll, lr);
      else {
        Real wm[6], ws[6], wp[6];
        to_sweep<2>(wm3, wm3[4], wm); to_sweep<2>(wc, wc[4], ws); to_sweep<2>(wn, wn[4], wp);
        cell_states<NS, 2, GRAV, ORD>(g, m, dt, sr.dtodx[2], q, cf, wm, ws, wp, full, ll, lr);
      }
      if (full && k > g.ks - 1) Ef(g, 2)[m] = 0.5*fabs(lr - lam3);      // lam3 = lambda_l the zone below gave to this zone's lower face
      lam3 = ll;
      if ((GRAV || AA_COOLING) && full) {   // d^{n+1/2}, :2104-2125
        const Real dh = wc[0] - q[0]*cf.dF[0][0] - q[1]*cf.dF[1][0] - q[2]*cf.dF[2][0];
        g.dhalf[m] = dh;
#if AA_COOLING
        g.phalf[m] = p_half(g, m, q, cf.dF, GRAV, cf.gm, dh);      // P^{n+1/2}, :2133-2266
#endif
      }
    }
    if (full) {
      if (!X1F) {   // ---- x1: neighbours by shuffle ----
        Real wm[6], wp[6], ws[6], ll = 0.0, lr = 0.0;
        to_sweep<0>(wc, pc0, ws);
#pragma unroll
        for (int n = 0; n < 6; n++) { wm[n] = __shfl_up(ws[n], 1); wp[n] = __shfl_down(ws[n], 1); }
        if (lane == 0 || lane == 63) {      // the neighbour zone of the tile before / after, exactly as the sweeps convert it
          Real u[6];
#pragma unroll
          for (int v = 0; v < 6; v++) u[v] = (v < NV) ? s_h1[row][(lane ? 6 : 0) + v] : 0.0;
          Real wh[6];
          cons_to_prim<NS>(u, wh, g.Gamma_1);
#pragma unroll
          for (int v = 0; v < 6; v++) { if (lane == 0) wm[v] = wh[v]; else wp[v] = wh[v]; }
        }
        if (do1) cell_states<NS, 0, GRAV, ORD>(g, m, dt, sr.dtodx[0], q, cf, wm, ws, wp, true, ll, lr);
        const Real lprev = __shfl_up(ll, 1);
        if (do1) {
          if (lane > 0) { if (i > g.is - 1) Ef(g, 0)[m] = 0.5*fabs(lr - lprev); }
          else Ef(g, 0)[m] = lr;                        // tile edge: k_eta_edges finishes this face
          if (lane == 63) Ef(g, 3)[m] = ll;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      {   // ---- x2: neighbours through LDS ----
        Real wm[6], wp[6], ws[6], ll = 0.0, lr = 0.0;
        to_sweep<1>(wc, pc1, ws);
#pragma unroll
        for (int n = 0; n < NV; n++) s_w[row][n][lane] = ws[n];
        __syncthreads();
        if (row == 0) {
          Real u[6];
#pragma unroll
          for (int v = 0; v < 6; v++) u[v] = (v < NV) ? s_h2[0][v][lane] : 0.0;
          cons_to_prim<NS>(u, wm, g.Gamma_1);
        } else {
#pragma unroll
          for (int n = 0; n < NV; n++) wm[n] = s_w[row - 1][n][lane];
          if (!NS) wm[5] = 0.0;
        }
        if (row == CA_TJ - 1) {
          Real u[6];
#pragma unroll
          for (int v = 0; v < 6; v++) u[v] = (v < NV) ? s_h2[1][v][lane] : 0.0;
          cons_to_prim<NS>(u, wp, g.Gamma_1);
        } else {
#pragma unroll
          for (int n = 0; n < NV; n++) wp[n] = s_w[row + 1][n][lane];
          if (!NS) wp[5] = 0.0;
        }
        if (do2) cell_states<NS, 1, GRAV, ORD>(g, m, dt, sr.dtodx[1], q, cf, wm, ws, wp, true, ll, lr);
        s_l[row][lane] = ll;
        __syncthreads();
        if (do2) {
          if (row > 0) Ef(g, 1)[m] = 0.5*fabs(lr - s_l[row - 1][lane]);
          else if (j > g.js - 1) Ef(g, 1)[m] = lr;      // tile edge, as for x1
          if (row == CA_TJ - 1) Ef(g, 4)[m] = ll;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    if (X3F) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 6; n++) {
#if CA_PARK
        if (!PARK3) { wl3[n] = (n < NV) ? s_park[n][row][lane] : 0.0; wr3[n] = (n < NV) ? s_park[6 + n][row][lane] : 0.0; }
#else
        wl3[n] = wlN[n]; wr3[n] = wrN[n];
#endif
        f3[n] = f3n[n];
      }
    }
  }
}

// eta of the faces on the tile edges of k_correct_all: the face's slot holds lambda_r of the zone above it, the edge
// array lambda_l of the zone below (same expression as inside the tile: integrate_3d_ctu.c:2300-2343)
template <int D>
AA_DEV void eta_edges(const DevGrid &g)
{
  // faces along D at the tile origins after the first; all zones s-1 .. e+1 of the other two directions
  const int n1 = g.ie - g.is + 3, n2 = g.je - g.js + 3, n3 = g.ke - g.ks + 3;
  const int nb = (D == 0) ? (g.ie + 1 - (g.is - 16))/64 : (g.je + 1 - (g.js - 1))/CA_TJ;      // tile origins 1 .. nb
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  int i, j, k;
  if (D == 0) {
    if (lin >= (long)nb*n2*n3) return;
    i = g.is - 16 + 64*(1 + (int)(lin % nb)); j = g.js - 1 + (int)((lin / nb) % n2); k = g.ks - 1 + (int)(lin / ((long)nb*n2));
  } else {
    if (lin >= (long)n1*nb*n3) return;
    i = g.is - 1 + (int)(lin % n1); j = g.js - 1 + CA_TJ*(1 + (int)((lin / n1) % nb)); k = g.ks - 1 + (int)(lin / ((long)n1*nb));
  }
  if (i > g.ie + 1 || j > g.je + 1) return;
  const long m = (long)k*g.sK + (long)j*g.sJ + i, sD = stride<D>(g);
  Real *e = Ef(g, D);
  e[m] = 0.5*fabs(e[m] - Ef(g, 3 + D)[m - sD]);
}
// both directions in one launch (blockIdx.y; they touch different arrays)
__global__ void __launch_bounds__(256)
k_eta_edges(DevGrid g) { if (blockIdx.y == 0) eta_edges<0>(g); else eta_edges<1>(g); }

// ---- steps 9b-d: second-pass fluxes with the H-correction -----------------------------------
template <int NS, int D>
__global__ void __launch_bounds__(256)
k_flux2(DevGrid g, Order ord)
{
  // faces needed by the update: along D [s, e+1], transverse [s, e]
  const int ni = g.ie - g.is + 1 + (D == 0), nj = g.je - g.js + 1 + (D == 1), nk = g.ke - g.ks + 1 + (D == 2);
  int i, j, k;
  if (!decode_zone(ord, ni, nj, nk, i, j, k)) return;
  i += g.is; j += g.js; k += g.ks;
  const long m = (long)k*g.sK + (long)j*g.sJ + i;
  const long sD = stride<D>(g), ml = m - sD;
  // the reference's MAX chain (a > b ? a : b) visits the two transverse directions in
  // ASCENDING order, then the face's own eta (:2354-2363, :2383-2392, :2412-2421); the order
  // matters when an eta is NaN (negative face pressure)
  constexpr int E1 = (D == 0) ? 1 : 0, E2 = (D == 2) ? 1 : 2;
  const long s1 = stride<E1>(g), s2 = stride<E2>(g);
  const Real *e1 = Ef(g, E1), *e2 = Ef(g, E2);
  Real etah = rmax(e1[ml], e1[m]);
  etah = rmax(etah, e1[ml + s1]);
  etah = rmax(etah, e1[m + s1]);
  etah = rmax(etah, e2[ml]);
  etah = rmax(etah, e2[m]);
  etah = rmax(etah, e2[ml + s2]);
  etah = rmax(etah, e2[m + s2]);
  etah = rmax(etah, Ef(g, D)[m]);
  Real ul[6], ur[6], wl[6], wr[6], f[6];
  load_sweep<D, NS>(LRf(g, D, 0, 0), g.nc, m, ul);
  load_sweep<D, NS>(LRf(g, D, 1, 0), g.nc, m, ur);
  cons_to_prim<NS>(ul, wl, g.Gamma_1);
  cons_to_prim<NS>(ur, wr, g.Gamma_1);
  flux_roe<NS>(ul, ur, wl, wr, etah, g.Gamma, g.Gamma_1, f);
  store_sweep<D, NS>(Ff(g, D, 0), g.nc, m, f);
}

// ---- steps 11a, 12: gravity source and conservative update ---------------------------------
AA_DEV void cfl_zone(Real d, Real m1, Real m2, Real m3, Real e, Real Gamma, Real Gamma_1, Real mx[3]);
AA_DEV void atomic_max_pos(unsigned long long *addr, Real v);
template <int NS, bool GRAV, bool CFL>
__global__ void __launch_bounds__(256)
k_update(DevGrid g, const Real *dhalf, Real dt, Order ord, Real *cfl_part, const unsigned char *pinmask)
{
  // CFL (the van Leer integrator's update; the CTU path has it in k_flux2_update): the zone's contribution to new_dt's maxima from the
  // updated state while it is in registers; one zone per thread makes half a million blocks at 512^3, so a block leaves its three
  // maxima in cfl_part[d][block] (k_cfl_fold turns them into the three words) instead of 1.5 M same-address atomics
  __shared__ Real s_red[CFL ? 3 : 1][CFL ? 256 : 1];
  Real cmx[3] = {0.0, 0.0, 0.0};
  const int ni = g.ie - g.is + 1, nj = g.je - g.js + 1, nk = g.ke - g.ks + 1;
  int i, j, k;
  const bool ok = decode_zone(ord, ni, nj, nk, i, j, k);
  if (!CFL && !ok) return;
  if (ok) {
  i += g.is; j += g.js; k += g.ks;
  const long m = (long)k*g.sK + (long)j*g.sJ + i;
  constexpr int NV = 5 + NS;
  Real u[6];
#pragma unroll
  for (int v = 0; v < NV; v++) u[v] = Uf(g, v)[m];
  Real dtodx[3];
#pragma unroll
  for (int d = 0; d < 3; d++) dtodx[d] = dt/g.dx[d];
  if (GRAV) {   // :2741-2782
    const Real phic = Pf(g, 0)[m], dh = dhalf[m];
#pragma unroll
    for (int e = 0; e < 3; e++) {
      const long se = stride_rt(g, e);
      const Real phir = Pf(g, 1 + e)[m + se], phil = Pf(g, 1 + e)[m];
      const Real *fd = Ff(g, e, 0);
      u[1 + e] -= dtodx[e]*(phir - phil)*dh;
      u[4] -= dtodx[e]*(fd[m]*(phic - phil) + fd[m + se]*(phir - phic));
    }
  }
#if AA_COOLING
  u[4] -= dt*cool_koyinut(dhalf[m], g.phalf[m], dt, g.Gamma_1);      // Step 11c, :2943-2953
#endif
#pragma unroll
  for (int d = 0; d < 3; d++) {   // :2981-3050, x1 then x2 then x3
    const long sd = stride_rt(g, d);
#pragma unroll
    for (int v = 0; v < NV; v++) { const Real *f = Ff(g, d, v); u[v] -= dtodx[d]*(f[m + sd] - f[m]); }
  }
#pragma unroll
  for (int v = 0; v < NV; v++) Uf(g, v)[m] = u[v];
  if (CFL && !(pinmask && pinmask[m])) cfl_zone(u[0], u[1], u[2], u[3], u[4], g.Gamma, g.Gamma_1, cmx);
  }
  if (CFL) {
    const int t = threadIdx.x;
#pragma unroll
    for (int d = 0; d < 3; d++) s_red[CFL ? d : 0][CFL ? t : 0] = cmx[d];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (t < w) {
#pragma unroll
        for (int d = 0; d < 3; d++) s_red[CFL ? d : 0][CFL ? t : 0] = rmax(s_red[CFL ? d : 0][CFL ? t : 0], s_red[CFL ? d : 0][CFL ? t + w : 0]);
      }
      __syncthreads();
    }
    if (t < 3) cfl_part[(size_t)t*gridDim.x + blockIdx.x] = s_red[CFL ? t : 0][0];
  }
}

// the blocks' maxima of k_update<CFL> -> the three words of new_dt (MAX of non-negative doubles on their bit patterns: order-free)
__global__ void __launch_bounds__(256)
k_cfl_fold(const Real *part, int nb, DevScalars *sc)
{
  __shared__ Real red[3][256];
  Real mx[3] = {0.0, 0.0, 0.0};
  for (int b = blockIdx.x*256 + threadIdx.x; b < nb; b += gridDim.x*256) {
#pragma unroll
    for (int d = 0; d < 3; d++) mx[d] = rmax(mx[d], part[(size_t)d*nb + b]);
  }
#pragma unroll
  for (int d = 0; d < 3; d++) red[d][threadIdx.x] = mx[d];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int d = 0; d < 3; d++) red[d][threadIdx.x] = rmax(red[d][threadIdx.x], red[d][threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) atomic_max_pos(&sc->max_v[threadIdx.x], red[threadIdx.x][0]);
}

// ---- steps 9b-d + 11a + 12 in one marching kernel --------------------------------------------
// The second-pass fluxes never go to HBM: a thread owns one (i,j) column and marches along x3; at every
// zone it solves the Riemann problems of the zone's LOWER x1 and x2 faces and of its UPPER x3 face.
// The upper x1 (x2) face is the lower face of the next lane (row) and arrives by a wavefront shuffle
// (through LDS); the lower x3 face is the upper one of the previous step and stays in registers.  Lane 63
// and row FU_TJ-1 of a block only provide fluxes (63 x 7 zones per 64 x 8 threads: 16 % more Riemann
// solves than faces), and a chunk of `kchunk` zones starts with one extra x3 solve.  Against
// k_flux2 x3 + k_update this drops 18 stores and ~20 loads of doubles per zone.  Expressions and their
// order per zone are those of k_flux2 / k_update, so results are bit-identical to the unfused chain.
// Levels of a Mesh (KEEP): RestrictCorrect reads the second-pass fluxes on the level boundaries, so the
// faces on those planes (KeepPlanes: a few planes per direction) are stored as well.
#ifndef FU_TJ
#define FU_TJ 8
#endif
// operands of one second-pass Riemann problem: the corrected face states (sweep frame) and the 9 etas
struct FaceIn { Real ul[6], ur[6], eta[9]; };
template <int NS, int D>
AA_DEV void face_load(const DevGrid &g, long m, FaceIn &in)
{
  const long sD = stride<D>(g), ml = m - sD;
  constexpr int E1 = (D == 0) ? 1 : 0, E2 = (D == 2) ? 1 : 2;
  const long s1 = stride<E1>(g), s2 = stride<E2>(g);
  const Real *e1 = Ef(g, E1), *e2 = Ef(g, E2);
  in.eta[8] = Ef(g, D)[m];
  in.eta[0] = e1[ml]; in.eta[1] = e1[m]; in.eta[2] = e1[ml + s1]; in.eta[3] = e1[m + s1];
  in.eta[4] = e2[ml]; in.eta[5] = e2[m]; in.eta[6] = e2[ml + s2]; in.eta[7] = e2[m + s2];
  load_sweep<D, NS>(LRf(g, D, 0, 0), g.nc, m, in.ul);
  load_sweep<D, NS>(LRf(g, D, 1, 0), g.nc, m, in.ur);
}
template <int NS>
AA_DEV void face_solve(const DevGrid &g, const FaceIn &in, Real f[6])
{
  Real etah = rmax(in.eta[0], in.eta[1]);
#pragma unroll
  for (int n = 2; n < 9; n++) etah = rmax(etah, in.eta[n]);
  Real wl[6], wr[6];
  cons_to_prim<NS>(in.ul, wl, g.Gamma_1);
  cons_to_prim<NS>(in.ur, wr, g.Gamma_1);
#ifndef FU_XD
#define FU_XD 1
#endif
  flux_roe<NS, false, (FU_XD != 0)>(in.ul, in.ur, wl, wr, etah, g.Gamma, g.Gamma_1, f);      // (faster without the AA_FAST_DIV forms here; the AA_XDIV forms since the x3 flux is parked in LDS)
}
template <int NS, int D>
AA_DEV void face_flux2(const DevGrid &g, long m, Real f[6])
{
  FaceIn in;
  face_load<NS, D>(g, m, in);
  face_solve<NS>(g, in, f);
}

// new_dt.c:72-140 for one zone: max(|v_d| + a) per direction (the operands and their order as in k_cfl)
// (no multiply-add contraction in here, in either build: the same zone must give the same bits whether it is visited by
//  k_cfl, by k_pinned_cfl or inside the update kernel, so that AA_CFL_FUSED changes no bit of dt in the default build either)
AA_DEV void cfl_zone(Real d, Real m1, Real m2, Real m3, Real e, Real Gamma, Real Gamma_1, Real mx[3])
{
#pragma clang fp contract(off)
  const Real di = 1.0/d;
  const Real v1 = m1*di, v2 = m2*di, v3 = m3*di;
  const Real qsq = v1*v1 + v2*v2 + v3*v3;
  const Real p = rmax(Gamma_1*(e - 0.5*d*qsq), AA_TINY);
  const Real a = sqrt(Gamma*p*di);
  mx[0] = rmax(mx[0], fabs(v1) + a); mx[1] = rmax(mx[1], fabs(v2) + a); mx[2] = rmax(mx[2], fabs(v3) + a);
}
AA_DEV void atomic_max_pos(unsigned long long *addr, Real v)
{ if (v == v) atomicMax(addr, (unsigned long long)__double_as_longlong(v)); }   // v >= 0; NaN skipped

AA_DEV bool on_plane(const KeepPlanes &kp, int d, int x)
{
  return x == kp.p[d][0] || x == kp.p[d][1] || x == kp.p[d][2] || x == kp.p[d][3] ||
         x == kp.p[d][4] || x == kp.p[d][5] || x == kp.p[d][6] || x == kp.p[d][7];
}

// CFL: the zone's contribution to new_dt's maxima is taken from the updated state while it is in registers (k_cfl would
// read the five fields again: 0.9 ms at 512^3); zones marked in `pinmask` are left out, the caller adds them after it
// has overwritten them (k_pinned_cfl).
template <int NS, bool GRAV, bool KEEP, bool CFL>
__global__ void __launch_bounds__(64*FU_TJ)
k_flux2_update(DevGrid g, const Real *dhalf, Real dt, int kchunk, KeepPlanes kp, DevScalars *sc, const unsigned char *pinmask, StepRatios sr)
{
  // (two copies of the exchange arrays, used in turn: ONE barrier per plane instead of a second one that only kept the
  //  next plane's writers off this plane's readers -- 8 wavefronts, a whole CU, wait at every barrier of this block)
  __shared__ Real s_f2[2][FU_TJ][6][64];
  __shared__ Real s_f1e[2][FU_TJ - 1][6], s_f1s[2][FU_TJ - 1][6];
#ifndef FU_PARK
#define FU_PARK 1
#endif
  // The lower x3 flux of the zone -- live across the whole iteration in 12 registers -- waits in the thread's own LDS slots: the
  // 6-variable gravity kernel then holds 246 registers without scratch, and the scaling-free quotients / square roots of the
  // solver (FU_XD), which cost it three spilled registers and 4 % before, pay: 13.77 -> 13.45 ms at 512^3 (same-box ABAB x 3;
  // parked alone 13.70).  FU_PARK=0 / FU_XD=0: the round's earlier form.
  constexpr bool FPARK = (FU_PARK != 0);
  __shared__ Real s_f3[FPARK ? 6 : 1][FPARK ? FU_TJ : 1][64];
  const int lane = threadIdx.x, row = threadIdx.y;
  // Tiles of 64 x (FU_TJ - 1) zones; the rows start on a 128-byte line (zone is) and do not overlap in x1: with a stride
  // of 63 zones every 512-byte row request touched a fifth line and 512 zones took 9 tiles (rocprofv3: 559 B/zone fetched
  // for 400 needed).  Every lane solves the LOWER x1 face of its zone; the one face a row lacks, the upper face of lane
  // 63, is solved by the wavefront of the last thread row (which owns no zones: it solves the x2 faces above the tile and
  // would idle during the x1 phase), lane r for zone row r, with the same software pipeline, and handed over through LDS
  // behind the barrier the x2 phase has anyway.
  const int i0 = g.is + blockIdx.x*64;
  const int i = i0 + lane, j0 = g.js + blockIdx.y*(FU_TJ - 1), j = j0 + row;
  const int k0 = g.ks + blockIdx.z*kchunk;
  int k1 = k0 + kchunk - 1; if (k1 > g.ke) k1 = g.ke;
  constexpr int NV = 5 + NS;
  const bool cell = (row < FU_TJ - 1) && (i <= g.ie) && (j <= g.je);
  const bool edge = (row == FU_TJ - 1) && (lane < FU_TJ - 1) && (j0 + lane <= g.je) && (i0 + 64 <= g.ie + 1);   // x1 face i0+64 of row `lane`
  const bool need1 = ((row < FU_TJ - 1) && (j <= g.je) && (i <= g.ie + 1)) || edge;     // lower x1 face of (i,j) / the edge face
  const bool need2 = (i <= g.ie) && (j <= g.je + 1);                                     // lower x2 face of (i,j)
  const bool last = (row < FU_TJ - 1) && (lane == 63) && (i < g.ie + 1);                 // its upper x1 face comes from the edge wavefront
  // clamp the column of idle threads into the Grid so that shuffles / barriers stay uniform
  const int ic = (i <= g.ie + 1) ? i : g.ie + 1, jc = (j <= g.je + 1) ? j : g.je + 1;
  const long mcol = (long)jc*g.sJ + ic;
  // where this thread's x1 face is, relative to its zone: 0, or for the edge wavefront the hop to (i0 + 64, j0 + lane)
  const long m1off = edge ? ((long)(j0 + lane)*g.sJ + (i0 + 64)) - mcol : 0L;
  Real dtodx[3];
#pragma unroll
  for (int d = 0; d < 3; d++) dtodx[d] = sr.dtodx[d];
  // per direction: flux differences (hi - lo) of all components + the two mass fluxes (gravity)
  Real f3lo[6], d1[6], d2[6], d3[6], m1lo = 0.0, m1hi = 0.0, m2lo = 0.0, m2hi = 0.0, m3hi = 0.0;
  // CFL: the thread's running maxima live in LDS (its own three slots), not in registers carried round the loop: the
  // 6-variable gravity kernel has none to spare (6 spilled, 16.1 against 14.0 ms)
  __shared__ Real s_cmx[CFL ? 3 : 1][CFL ? FU_TJ : 1][CFL ? 64 : 1];
  if (CFL) { s_cmx[0][row][lane] = 0.0; s_cmx[1][row][lane] = 0.0; s_cmx[2][row][lane] = 0.0; }
#pragma unroll
  for (int n = 0; n < 6; n++) f3lo[n] = 0.0;
  const bool keep1 = KEEP && need1 && on_plane(kp, 0, edge ? i0 + 64 : i), keep2 = KEEP && need2 && on_plane(kp, 1, j);
  if (cell) {                                                                   // face k0
    face_flux2<NS, 2>(g, (long)k0*g.sK + mcol, f3lo);
    // (only the first chunk keeps its face k0: every later chunk's is face k1+1 of the chunk below, kept there by the loop's instance of
    //  the solver -- in the default build this instance may contract its multiply-adds differently, and two writers of one word with
    //  last-bit-different values made the flux a level's parent reads depend on which block finished last: round 4, found as a
    //  run-to-run difference of 1e-16 on the levels of a Mesh in the default build)
    if (KEEP && blockIdx.z == 0 && on_plane(kp, 2, k0)) store_sweep<2, NS>(Ff(g, 2, 0), g.nc, (long)k0*g.sK + mcol, f3lo);
  }
  if (FPARK) {
#pragma unroll
    for (int n = 0; n < NV; n++) s_f3[FPARK ? n : 0][FPARK ? row : 0][lane] = f3lo[n];
  }
  // Software pipeline: the operands of the NEXT Riemann problem are requested before the current one is
  // solved (x2's during the x1 solve, x3's during x2's, the next zone's x1's during x3's), so that with two
  // wavefronts per SIMD the memory pipe is not idle while a wavefront computes.
  FaceIn in1, in2;
  {
    long m0 = (long)k0*g.sK + mcol + m1off;
    asm volatile("" : "+v"(m0));
    if (need1) face_load<NS, 0>(g, m0, in1);
  }
  for (int k = k0; k <= k1; k++) {
    // (opaque to the optimiser: otherwise every one of the ~60 field pointers becomes its own
    //  strength-reduced 64-bit induction variable and the kernel spills)
    long m = (long)k*g.sK + mcol;
    asm volatile("" : "+v"(m));
    const int pb = k & 1;
    if (need2) face_load<NS, 1>(g, m, in2);
    __builtin_amdgcn_sched_barrier(0);
    {
      Real f[6];
#pragma unroll
      for (int n = 0; n < 6; n++) f[n] = 0.0;
      if (need1) face_solve<NS>(g, in1, f);
      if (keep1) store_sweep<0, NS>(Ff(g, 0, 0), g.nc, m + m1off, f);
#pragma unroll
      for (int n = 0; n < NV; n++) d1[n] = __shfl_down(f[n], 1) - f[n];
      m1lo = f[0]; m1hi = __shfl_down(f[0], 1);
      if (edge) {
#pragma unroll
        for (int n = 0; n < NV; n++) s_f1e[pb][lane][n] = f[n];
      }
      if (last) {
#pragma unroll
        for (int n = 0; n < NV; n++) s_f1s[pb][row][n] = f[n];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (cell) face_load<NS, 2>(g, m + g.sK, in1);           // (in1 is free: reused for the x3 face)
    __builtin_amdgcn_sched_barrier(0);
    {
      Real f[6];
#pragma unroll
      for (int n = 0; n < 6; n++) f[n] = 0.0;
      if (need2) face_solve<NS>(g, in2, f);
      if (keep2) store_sweep<1, NS>(Ff(g, 1, 0), g.nc, m, f);
#pragma unroll
      for (int n = 0; n < NV; n++) s_f2[pb][row][n][lane] = f[n];
      __syncthreads();
      if (row < FU_TJ - 1) {
#pragma unroll
        for (int n = 0; n < NV; n++) d2[n] = s_f2[pb][row + 1][n][lane] - f[n];
        m2hi = s_f2[pb][row + 1][0][lane];
      }
      m2lo = f[0];
      if (last) {           // the same subtraction as in the other lanes, operands through LDS
#pragma unroll
        for (int n = 0; n < NV; n++) d1[n] = s_f1e[pb][row][n] - s_f1s[pb][row][n];
        m1hi = s_f1e[pb][row][0];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // FU_EARLY=1 (the default): the potentials and d^{n+1/2} of the zone's update are requested in front of the x3 Riemann problem
    // instead of behind it (same-box, 512^3: 13.45 -> 13.15 ms; the conserved variables instead 13.26, both 14.57: four registers in
    // scratch -- profiles/r04_x1f_ab.txt item 7)
#ifndef FU_EARLY
#define FU_EARLY 1
#endif
    constexpr bool EP = GRAV && (FU_EARLY != 0);
    Real ep[7], edh = 0.0;
#pragma unroll
    for (int v = 0; v < 7; v++) ep[v] = 0.0;
    if (EP && cell) {
      ep[0] = Pf(g, 0)[m]; edh = dhalf[m];
      ep[1] = Pf(g, 1)[m + 1]; ep[2] = Pf(g, 1)[m]; ep[3] = Pf(g, 2)[m + g.sJ]; ep[4] = Pf(g, 2)[m]; ep[5] = Pf(g, 3)[m + g.sK]; ep[6] = Pf(g, 3)[m];
    }
    Real f3[6];
#pragma unroll
    for (int n = 0; n < 6; n++) f3[n] = 0.0;
    if (cell) face_solve<NS>(g, in1, f3);
    __builtin_amdgcn_sched_barrier(0);
    if (need1 && k < k1) face_load<NS, 0>(g, m + m1off + g.sK, in1);  // the next zone's x1 face
    __builtin_amdgcn_sched_barrier(0);
    if (cell) {
      if (KEEP && on_plane(kp, 2, k + 1)) store_sweep<2, NS>(Ff(g, 2, 0), g.nc, m + g.sK, f3);
#pragma unroll
      for (int n = 0; n < NV; n++) { if (FPARK) f3lo[n] = s_f3[FPARK ? n : 0][FPARK ? row : 0][lane]; d3[n] = f3[n] - f3lo[n]; }
      m3hi = f3[0];
      const Real m3lo = f3lo[0];
#pragma unroll
      for (int n = 0; n < NV; n++) { if (FPARK) s_f3[FPARK ? n : 0][FPARK ? row : 0][lane] = f3[n]; else f3lo[n] = f3[n]; }
      Real u[6];
#pragma unroll
      for (int v = 0; v < NV; v++) u[v] = Uf(g, v)[m];
      // (the mask byte with the zone's other operands, not behind the stores of U: a load that is consumed at once waits
      //  for everything issued before it, i.e. the wave sat out the six stores' round trip in every plane)
      unsigned char pinned = 0;
      if (CFL && pinmask) pinned = pinmask[m];
      if (GRAV) {   // :2741-2782, with the mass fluxes (sweep component 0) of the faces just solved
        const Real phic = EP ? ep[0] : Pf(g, 0)[m], dh = EP ? edh : dhalf[m];
        { const Real phir = EP ? ep[1] : Pf(g, 1)[m + 1], phil = EP ? ep[2] : Pf(g, 1)[m];
          u[1] -= dtodx[0]*(phir - phil)*dh;
          u[4] -= dtodx[0]*(m1lo*(phic - phil) + m1hi*(phir - phic)); }
        { const Real phir = EP ? ep[3] : Pf(g, 2)[m + g.sJ], phil = EP ? ep[4] : Pf(g, 2)[m];
          u[2] -= dtodx[1]*(phir - phil)*dh;
          u[4] -= dtodx[1]*(m2lo*(phic - phil) + m2hi*(phir - phic)); }
        { const Real phir = EP ? ep[5] : Pf(g, 3)[m + g.sK], phil = EP ? ep[6] : Pf(g, 3)[m];
          u[3] -= dtodx[2]*(phir - phil)*dh;
          u[4] -= dtodx[2]*(m3lo*(phic - phil) + m3hi*(phir - phic)); }
      }
#if AA_COOLING
      u[4] -= dt*cool_koyinut(dhalf[m], g.phalf[m], dt, g.Gamma_1);      // Step 11c, :2943-2953
#endif
      // :2981-3050, x1 then x2 then x3; sweep component n of direction D is global variable gv<D>(n)
#pragma unroll
      for (int n = 0; n < NV; n++) u[gv<0>(n)] -= dtodx[0]*d1[n];
#pragma unroll
      for (int n = 0; n < NV; n++) u[gv<1>(n)] -= dtodx[1]*d2[n];
#pragma unroll
      for (int n = 0; n < NV; n++) u[gv<2>(n)] -= dtodx[2]*d3[n];
#pragma unroll
      for (int v = 0; v < NV; v++) Uf(g, v)[m] = u[v];
      if (CFL) {
        if (!pinned) {
          Real cmx[3] = {s_cmx[0][row][lane], s_cmx[1][row][lane], s_cmx[2][row][lane]};
          cfl_zone(u[0], u[1], u[2], u[3], u[4], g.Gamma, g.Gamma_1, cmx);
          s_cmx[0][row][lane] = cmx[0]; s_cmx[1][row][lane] = cmx[1]; s_cmx[2][row][lane] = cmx[2];
        }
      }
    }
  }
  if (CFL) {       // block maximum -> 3 atomics (MAX of non-negative doubles on their bit patterns: order-free)
    Real *red = &s_cmx[0][0][0];           // [d][row][lane] = [d*64*FU_TJ + t]
    const int t = row*64 + lane;
    __syncthreads();
    for (int w = 32*FU_TJ; w > 0; w >>= 1) {
      if (t < w) {
#pragma unroll
        for (int d = 0; d < 3; d++) red[d*64*FU_TJ + t] = rmax(red[d*64*FU_TJ + t], red[d*64*FU_TJ + t + w]);
      }
      __syncthreads();
    }
    if (t == 0) for (int d = 0; d < 3; d++) atomic_max_pos(&sc->max_v[d], red[d*64*FU_TJ]);
  }
}

// the zones Userwork has just overwritten (k_pinned) join the maxima k_flux2_update<CFL> left them out of
__global__ void k_pinned_cfl(DevGrid g, long long n, const long long *idx, DevScalars *sc)
{
  // grid-stride: a few hundred same-address atomics per launch instead of one per 64 pinned zones (1.5e5 zones at 512^3 made
  // 7000 atomics on three words, ~0.1 ms of this kernel's 0.13; the MAX is order-free)
  Real mx[3] = {0.0, 0.0, 0.0};
  for (long lin = (long)blockIdx.x*blockDim.x + threadIdx.x; lin < n; lin += (long)gridDim.x*blockDim.x) {
    const long long c = idx[lin];
    const int i = (int)(c % g.N1), j = (int)((c / g.N1) % g.N2), k = (int)(c / ((long)g.N1*g.N2));
    if (i >= g.is && i <= g.ie && j >= g.js && j <= g.je && k >= g.ks && k <= g.ke) {      // new_dt looks at active zones only
      const long m = (long)k*g.sK + (long)j*g.sJ + i;
      cfl_zone(Uf(g, 0)[m], Uf(g, 1)[m], Uf(g, 2)[m], Uf(g, 3)[m], Uf(g, 4)[m], g.Gamma, g.Gamma_1, mx);
    }
  }
  // one atomic per wavefront and direction (NaN maxima are dropped by atomic_max_pos, as in k_cfl)
#pragma unroll
  for (int d = 0; d < 3; d++) {
    Real v = mx[d];
    for (int o = 32; o > 0; o >>= 1) { const Real w = __shfl_xor(v, o); v = (w > v || v != v) ? w : v; }
    if ((threadIdx.x & 63) == 0) atomic_max_pos(&sc->max_v[d], v);
  }
}
// byte mask of the pinned zones over the device array
__global__ void k_pin_mask(DevGrid g, long long n, const long long *idx, unsigned char *mask)
{
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= n) return;
  const long long c = idx[lin];
  const int i = (int)(c % g.N1), j = (int)((c / g.N1) % g.N2), k = (int)(c / ((long)g.N1*g.N2));
  mask[(long)k*g.sK + (long)j*g.sJ + i] = 1;
}

// ---- van Leer integrator (integrators/integrate_3d_vl.c, NO_H_CORRECTION) --------------------
// Predictor on small Grids: k_vl_flux1 x3 + k_vl_uhalf below; on big ones k_vl_predict further down does both.
// steps 1-3: first-order (donor-cell) fluxes, Wl = W[c-1], Wr = W[c], over the whole ghost range
template <int NS, int D>
__global__ void __launch_bounds__(256)
k_vl_flux1(DevGrid g)
{
  // faces along D: [s-3, e+4]; transverse: all zones incl. ghosts (:153-305)
  const int N[3] = {g.N1, g.N2, g.N3};
  int n[3] = {N[0], N[1], N[2]};
  n[D] = N[D] - 1;                                         // faces 1 .. N-1  (= s-3 .. e+4)
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= (long)n[0]*n[1]*n[2]) return;
  int idx[3];
  idx[0] = (int)(lin % n[0]); idx[1] = (int)((lin / n[0]) % n[1]); idx[2] = (int)(lin / ((long)n[0]*n[1]));
  idx[D] += 1;
  const long m = (long)idx[2]*g.sK + (long)idx[1]*g.sJ + idx[0];
  const long s = stride<D>(g);
  Real u[6], wl[6], wr[6], ul[6], ur[6], f[6];
  load_sweep<D, NS>(g.U, g.nc, m - s, u); cons_to_prim<NS>(u, wl, g.Gamma_1);
  load_sweep<D, NS>(g.U, g.nc, m, u);     cons_to_prim<NS>(u, wr, g.Gamma_1);
  prim_to_cons<NS>(wl, ul, g.Gamma_1, g.rGamma_1);                     // the reference round-trips through W (:166-169)
  prim_to_cons<NS>(wr, ur, g.Gamma_1, g.rGamma_1);
  flux_roe<NS>(ul, ur, wl, wr, 0.0, g.Gamma, g.Gamma_1, f);
  store_sweep<D, NS>(Ff(g, D, 0), g.nc, m, f);
}

// steps 5-6: U^{n+1/2} over [s-3, e+3]^3 (x1, x2, x3 flux differences in that order, then the
// gravity predictor); stored in the first six face-state arrays
// ---- van Leer predictor in one marching kernel: donor-cell fluxes + U^{n+1/2} (integrate_3d_vl.c:153-510) --
// The first-order fluxes only feed U^{n+1/2}, so they never go to HBM: thread = (i,j) column marching along
// x3 as in k_flux2_update.  The state of a zone is converted once for all three sweeps: d, V and r are the
// same in every frame, the pressure (cons_to_prim) and the round-tripped energy (prim_to_cons, the reference
// converts W back to U, :166-169) sum the squared velocities in sweep-frame order, so both come in three
// variants.  The lower x1 face takes its left state from the previous lane (shuffle), the lower x2 face
// from the row below (loaded), the lower x3 face from the previous step (registers); the upper fluxes come
// from the next lane / row / step.  Lane 63 and row VP_TJ-1 only provide fluxes.
#ifndef VP_TJ
#define VP_TJ 8
#endif
struct VlState { Real w[6], p[3], e[3]; };       // w: d, V1, V2, V3, (unused), r;  p, e per sweep frame
template <int NS>
AA_DEV void vl_state(const DevGrid &g, long m, VlState &q)
{
  Real u[6];
#pragma unroll
  for (int v = 0; v < 5 + NS; v++) u[v] = Uf(g, v)[m];
  if (!NS) u[5] = 0.0;
#pragma unroll
  for (int D = 0; D < 3; D++) {
    // exactly cons_to_prim / prim_to_cons on the state rotated into the frame of D
    Real us[6], ws[6], ub[6];
    us[0] = u[0]; us[1] = u[1 + D]; us[2] = u[1 + (D + 1) % 3]; us[3] = u[1 + (D + 2) % 3]; us[4] = u[4]; us[5] = u[5];
    cons_to_prim<NS>(us, ws, g.Gamma_1);
    prim_to_cons<NS>(ws, ub, g.Gamma_1, g.rGamma_1);
    q.p[D] = ws[4]; q.e[D] = ub[4];
    if (D == 0) { q.w[0] = ws[0]; q.w[1] = ws[1]; q.w[2] = ws[2]; q.w[3] = ws[3]; q.w[4] = 0.0; q.w[5] = ws[5]; }
  }
}
// sweep-frame primitive and (round-tripped) conserved state of direction D
template <int NS, int D>
AA_DEV void vl_rotate(const VlState &q, Real w[6], Real u[6])
{
  w[0] = q.w[0]; w[1] = q.w[1 + D]; w[2] = q.w[1 + (D + 1) % 3]; w[3] = q.w[1 + (D + 2) % 3]; w[4] = q.p[D]; w[5] = q.w[5];
  u[0] = w[0]; u[1] = w[0]*w[1]; u[2] = w[0]*w[2]; u[3] = w[0]*w[3]; u[4] = q.e[D]; u[5] = NS ? w[5]*w[0] : 0.0;
}
template <int NS, int D>
AA_DEV void vl_face(const DevGrid &g, const VlState &lo, const VlState &hi, Real f[6])
{
  Real wl[6], wr[6], ul[6], ur[6];
  vl_rotate<NS, D>(lo, wl, ul);
  vl_rotate<NS, D>(hi, wr, ur);
  flux_roe<NS>(ul, ur, wl, wr, 0.0, g.Gamma, g.Gamma_1, f);
}

template <int NS, bool GRAV>
__global__ void __launch_bounds__(64*VP_TJ)
k_vl_predict(DevGrid g, Real dt, int kchunk)
{
  __shared__ Real s_f2[VP_TJ][6][64];
  constexpr int NV = 5 + NS;
  const int lane = threadIdx.x, row = threadIdx.y;
  const int lo[3] = {g.is - 3, g.js - 3, g.ks - 3}, hi[3] = {g.ie + 3, g.je + 3, g.ke + 3};   // zones that get U^{n+1/2}
  const int i = lo[0] + blockIdx.x*63 + lane, j = lo[1] + blockIdx.y*(VP_TJ - 1) + row;
  const int k0 = lo[2] + blockIdx.z*kchunk;
  int k1 = k0 + kchunk - 1; if (k1 > hi[2]) k1 = hi[2];
  const bool cell = (lane < 63) && (row < VP_TJ - 1) && (i <= hi[0]) && (j <= hi[1]);
  const bool need1 = (row < VP_TJ - 1) && (j <= hi[1]) && (i <= hi[0] + 1);      // lower x1 face of (i,j)
  const bool need2 = (lane < 63) && (i <= hi[0]) && (j <= hi[1] + 1);            // lower x2 face of (i,j)
  const int ic = (i <= hi[0] + 1) ? i : hi[0] + 1, jc = (j <= hi[1] + 1) ? j : hi[1] + 1;
  const long mcol = (long)jc*g.sJ + ic;
  Real q[3];
#pragma unroll
  for (int d = 0; d < 3; d++) q[d] = 0.5*(dt/g.dx[d]);
  VlState below, here;
  Real f3lo[6];
#pragma unroll
  for (int n = 0; n < 6; n++) f3lo[n] = 0.0;
  vl_state<NS>(g, (long)(k0 - 1)*g.sK + mcol, below);
  vl_state<NS>(g, (long)k0*g.sK + mcol, here);
  if (cell) vl_face<NS, 2>(g, below, here, f3lo);                                // face k0
  for (int k = k0; k <= k1; k++) {
    long m = (long)k*g.sK + mcol;
    asm volatile("" : "+v"(m));                   // one index for all fields (see k_flux2_update)
    VlState above;
    vl_state<NS>(g, m + g.sK, above);
    Real d1[6], d2[6], d3[6], m1lo, m1hi, m2lo, m2hi = 0.0, m3lo, m3hi;
#pragma unroll
    for (int n = 0; n < 6; n++) d2[n] = 0.0;
    __builtin_amdgcn_sched_barrier(0);
    {   // ---- x1: left state from the previous lane ----
      VlState left;
#pragma unroll
      for (int n = 0; n < 6; n++) left.w[n] = __shfl_up(here.w[n], 1);
      left.p[0] = __shfl_up(here.p[0], 1); left.e[0] = __shfl_up(here.e[0], 1);
      left.p[1] = left.p[2] = left.e[1] = left.e[2] = 0.0;
      if (lane == 0) vl_state<NS>(g, m - 1, left);
      Real f[6];
#pragma unroll
      for (int n = 0; n < 6; n++) f[n] = 0.0;
      if (need1) vl_face<NS, 0>(g, left, here, f);
#pragma unroll
      for (int n = 0; n < NV; n++) d1[n] = __shfl_down(f[n], 1) - f[n];
      m1lo = f[0]; m1hi = __shfl_down(f[0], 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // ---- x2: left state = the zone of the row below (converted here: it is in L1/L2) ----
      VlState left;
      vl_state<NS>(g, m - g.sJ, left);
      Real f[6];
#pragma unroll
      for (int n = 0; n < 6; n++) f[n] = 0.0;
      if (need2) vl_face<NS, 1>(g, left, here, f);
#pragma unroll
      for (int n = 0; n < NV; n++) s_f2[row][n][lane] = f[n];
      __syncthreads();
      if (row < VP_TJ - 1) {
#pragma unroll
        for (int n = 0; n < NV; n++) d2[n] = s_f2[row + 1][n][lane] - f[n];
        m2hi = s_f2[row + 1][0][lane];
      }
      m2lo = f[0];
      __syncthreads();
    }
    __builtin_amdgcn_sched_barrier(0);
    if (cell) {
      Real f[6];
      vl_face<NS, 2>(g, here, above, f);                                         // face k+1
#pragma unroll
      for (int n = 0; n < NV; n++) d3[n] = f[n] - f3lo[n];
      m3lo = f3lo[0]; m3hi = f[0];
#pragma unroll
      for (int n = 0; n < NV; n++) f3lo[n] = f[n];
      Real u[6];
#pragma unroll
      for (int v = 0; v < NV; v++) u[v] = Uf(g, v)[m];
      const Real d0 = u[0];
      // :436-478 flux differences x1, x2, x3; sweep component n of direction D is global variable gv<D>(n)
#pragma unroll
      for (int n = 0; n < NV; n++) u[gv<0>(n)] -= q[0]*d1[n];
#pragma unroll
      for (int n = 0; n < NV; n++) u[gv<1>(n)] -= q[1]*d2[n];
#pragma unroll
      for (int n = 0; n < NV; n++) u[gv<2>(n)] -= q[2]*d3[n];
      if (GRAV) {   // :480-510
        const Real phic = Pf(g, 0)[m];
        { const Real phir = Pf(g, 1)[m + 1], phil = Pf(g, 1)[m];
          u[1] -= q[0]*(phir - phil)*d0;
          u[4] -= q[0]*(m1lo*(phic - phil) + m1hi*(phir - phic)); }
        { const Real phir = Pf(g, 2)[m + g.sJ], phil = Pf(g, 2)[m];
          u[2] -= q[1]*(phir - phil)*d0;
          u[4] -= q[1]*(m2lo*(phic - phil) + m2hi*(phir - phic)); }
        { const Real phir = Pf(g, 3)[m + g.sK], phil = Pf(g, 3)[m];
          u[3] -= q[2]*(phir - phil)*d0;
          u[4] -= q[2]*(m3lo*(phic - phil) + m3hi*(phir - phic)); }
      }
#pragma unroll
      for (int v = 0; v < NV; v++) LRf(g, 0, 0, v)[m] = u[v];
    }
    below = here; here = above;
  }
}

template <int NS, bool GRAV>
__global__ void __launch_bounds__(256)
k_vl_uhalf(DevGrid g, Real dt)
{
  const int ni = g.ie - g.is + 7, nj = g.je - g.js + 7, nk = g.ke - g.ks + 7;
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= (long)ni*nj*nk) return;
  const int i = g.is - 3 + (int)(lin % ni), j = g.js - 3 + (int)((lin / ni) % nj), k = g.ks - 3 + (int)(lin / ((long)ni*nj));
  const long m = (long)k*g.sK + (long)j*g.sJ + i;
  constexpr int NV = 5 + NS;
  Real u[6], q[3];
#pragma unroll
  for (int v = 0; v < NV; v++) u[v] = Uf(g, v)[m];
  const Real d0 = u[0];
#pragma unroll
  for (int d = 0; d < 3; d++) q[d] = 0.5*(dt/g.dx[d]);
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const long sd = stride_rt(g, d);
#pragma unroll
    for (int v = 0; v < NV; v++) { const Real *f = Ff(g, d, v); u[v] -= q[d]*(f[m + sd] - f[m]); }
  }
  if (GRAV) {   // :480-510
    const Real phic = Pf(g, 0)[m];
#pragma unroll
    for (int e = 0; e < 3; e++) {
      const long se = stride_rt(g, e);
      const Real phir = Pf(g, 1 + e)[m + se], phil = Pf(g, 1 + e)[m];
      const Real *fd = Ff(g, e, 0);
      u[1 + e] -= q[e]*(phir - phil)*d0;
      u[4] -= q[e]*(fd[m]*(phic - phil) + fd[m + se]*(phir - phic));
    }
  }
#pragma unroll
  for (int v = 0; v < NV; v++) LRf(g, 0, 0, v)[m] = u[v];
}

// ---- physical boundary conditions: bvals_mhd.c reflect :959, outflow :1319, periodic :1637 ---
// dir d, side 0/1; transverse extents: x1 -> active j,k; x2 -> all i, active k; x3 -> all i,j.
// side < 0: both sides in one launch, blockIdx.y = side, (flag, flag1) = the two flags (0 = leave alone)
__global__ void k_bc(DevGrid g, int nvar, int d, int side, int flag, int flag1)
{
  if (side < 0) { side = blockIdx.y; if (side) flag = flag1; if (!flag) return; }
  const int lo3[3] = {g.is, g.js, g.ks}, hi3[3] = {g.ie, g.je, g.ke}, N[3] = {g.N1, g.N2, g.N3};
  int r0[3], n[3];
  for (int a = 0; a < 3; a++) {
    if (a == d) { r0[a] = 0; n[a] = AA_NGHOST_; continue; }
    if (a < d) { r0[a] = 0; n[a] = N[a]; } else { r0[a] = lo3[a]; n[a] = hi3[a] - lo3[a] + 1; }
  }
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= (long)n[0]*n[1]*n[2]) return;
  int idx[3];
  idx[0] = r0[0] + (int)(lin % n[0]);
  idx[1] = r0[1] + (int)((lin / n[0]) % n[1]);
  idx[2] = r0[2] + (int)(lin / ((long)n[0]*n[1]));
  const int gl = idx[d] + 1;                                    // ghost layer 1..4
  int dst, src;
  if (side == 0) { dst = lo3[d] - gl; src = (flag == 1) ? lo3[d] + (gl - 1) : (flag == 2) ? lo3[d] : hi3[d] - (gl - 1); }
  else           { dst = hi3[d] + gl; src = (flag == 1) ? hi3[d] - (gl - 1) : (flag == 2) ? hi3[d] : lo3[d] + (gl - 1); }
  idx[d] = 0;
  const long base = (long)idx[2]*g.sK + (long)idx[1]*g.sJ + idx[0];
  const long sd = stride_rt(g, d);
  for (int v = 0; v < nvar; v++) {
    Real x = Uf(g, v)[base + src*sd];
    if (flag == 1 && v == 1 + d) x = -x;
    Uf(g, v)[base + dst*sd] = x;
  }
}

// bvals_mhd in ONE launch.  The three passes x1 -> x2 -> x3 are copies along one coordinate each (k_bc), so what a ghost zone ends up
// with is the value of ONE source zone, found by walking its ghost coordinates from x3 down: a pass d writes the zones whose
// coordinate d is a ghost one (flag != 0), for every value of the coordinates below d and ACTIVE coordinates above it, from the
// zone with coordinate d mapped into the active range -- whose lower ghost coordinates were filled by the passes before, the same
// way.  So: map x3 if it is a ghost coordinate with a boundary function, then x2, then x1; stop at the first ghost coordinate whose
// side has none (flag 0: a neighbour Grid's or a parent's data; the passes below never touch a zone with that coordinate outside
// the active range).  A zone is written iff its HIGHEST ghost coordinate has a boundary function; no source zone is itself a
// target, so the launch needs no ordering.  Reflecting sides negate their normal momentum once per mapped coordinate.  Same bits
// as the three passes (copies and sign flips only).
__global__ void __launch_bounds__(256)
k_bc_shell(DevGrid g, int nvar, int f0, int f1, int f2, int f3, int f4, int f5)
{
  const int flag[3][2] = {{f0, f1}, {f2, f3}, {f4, f5}};
  const int lo3[3] = {g.is, g.js, g.ks}, hi3[3] = {g.ie, g.je, g.ke}, N[3] = {g.N1, g.N2, g.N3};
  const int G2 = 2*AA_NGHOST_;
  // the shell in three pieces: ghost k (all i, j); ghost j, active k (all i); ghost i, active j and k
  const long n3 = (long)G2*N[1]*N[0], n2 = (long)G2*N[0]*(hi3[2] - lo3[2] + 1), n1 = (long)G2*(hi3[1] - lo3[1] + 1)*(hi3[2] - lo3[2] + 1);
  long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  int c[3];
  auto ghost = [&](int q, int d) { return q < AA_NGHOST_ ? q : hi3[d] + 1 + (q - AA_NGHOST_); };      // q-th ghost index of direction d
  if (lin < n3) {
    c[0] = (int)(lin % N[0]); c[1] = (int)((lin / N[0]) % N[1]); c[2] = ghost((int)(lin / ((long)N[0]*N[1])), 2);
  } else if ((lin -= n3) < n2) {
    c[0] = (int)(lin % N[0]); c[1] = ghost((int)((lin / N[0]) % G2), 1); c[2] = lo3[2] + (int)(lin / ((long)N[0]*G2));
  } else if ((lin -= n2) < n1) {
    c[0] = ghost((int)(lin % G2), 0); c[1] = lo3[1] + (int)((lin / G2) % (hi3[1] - lo3[1] + 1)); c[2] = lo3[2] + (int)(lin / ((long)G2*(hi3[1] - lo3[1] + 1)));
  } else return;
  const long mdst = (long)c[2]*g.sK + (long)c[1]*g.sJ + c[0];
  bool neg[3] = {false, false, false}, any = false;
#pragma unroll
  for (int d = 2; d >= 0; d--) {
    const int x = c[d];
    if (x >= lo3[d] && x <= hi3[d]) continue;
    const int side = (x < lo3[d]) ? 0 : 1, fl = flag[d][side];
    if (!fl) { if (!any) return; break; }              // (highest ghost coordinate without a boundary function: not ours; a lower one: stop)
    const int gl = side ? x - hi3[d] : lo3[d] - x;       // ghost layer 1..4
    if (side == 0) c[d] = (fl == 1) ? lo3[d] + (gl - 1) : (fl == 2) ? lo3[d] : hi3[d] - (gl - 1);
    else           c[d] = (fl == 1) ? hi3[d] - (gl - 1) : (fl == 2) ? hi3[d] : lo3[d] + (gl - 1);
    if (fl == 1) neg[d] = true;
    any = true;
  }
  const long msrc = (long)c[2]*g.sK + (long)c[1]*g.sJ + c[0];
  for (int v = 0; v < nvar; v++) {
    Real x = Uf(g, v)[msrc];
    if (v >= 1 && v <= 3 && neg[v - 1]) x = -x;
    Uf(g, v)[mdst] = x;
  }
}

// ---- CFL reduction: new_dt.c:72-170 -----------------------------------------------------
__global__ void __launch_bounds__(256)
k_cfl(DevGrid g, DevScalars *sc)
{
  const int ni = g.ie - g.is + 1, nj = g.je - g.js + 1, nk = g.ke - g.ks + 1;
  const long ntot = (long)ni*nj*nk;
  Real mx[3] = {0.0, 0.0, 0.0};
  for (long lin = (long)blockIdx.x*blockDim.x + threadIdx.x; lin < ntot; lin += (long)gridDim.x*blockDim.x) {
    const int i = g.is + (int)(lin % ni);
    const int j = g.js + (int)((lin / ni) % nj);
    const int k = g.ks + (int)(lin / ((long)ni*nj));
    const long m = (long)k*g.sK + (long)j*g.sJ + i;
    cfl_zone(Uf(g, 0)[m], Uf(g, 1)[m], Uf(g, 2)[m], Uf(g, 3)[m], Uf(g, 4)[m], g.Gamma, g.Gamma_1, mx);
  }
  __shared__ Real red[3][256];
  for (int d = 0; d < 3; d++) red[d][threadIdx.x] = mx[d];
  __syncthreads();
  for (int s = blockDim.x/2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) for (int d = 0; d < 3; d++) red[d][threadIdx.x] = rmax(red[d][threadIdx.x], red[d][threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) for (int d = 0; d < 3; d++) atomic_max_pos(&sc->max_v[d], red[d][0]);
}

// ---- history sums (dump_history.c:157-200): mass, E, momenta, kinetic energies, scalar; one
// partial row per block, added up on the host in block order (deterministic) ------------------
__global__ void __launch_bounds__(256)
k_history(DevGrid g, int nscal, Real *partial)
{
  const int ni = g.ie - g.is + 1, nj = g.je - g.js + 1, nk = g.ke - g.ks + 1;
  const long ntot = (long)ni*nj*nk;
  Real acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long lin = (long)blockIdx.x*blockDim.x + threadIdx.x; lin < ntot; lin += (long)gridDim.x*blockDim.x) {
    const int i = g.is + (int)(lin % ni);
    const int j = g.js + (int)((lin / ni) % nj);
    const int k = g.ks + (int)(lin / ((long)ni*nj));
    const long m = (long)k*g.sK + (long)j*g.sJ + i;
    const Real d = Uf(g, 0)[m], d1 = 1.0/d, M1 = Uf(g, 1)[m], M2 = Uf(g, 2)[m], M3 = Uf(g, 3)[m];
    acc[0] += d; acc[1] += Uf(g, 4)[m]; acc[2] += M1; acc[3] += M2; acc[4] += M3;
    acc[5] += 0.5*M1*M1*d1; acc[6] += 0.5*M2*M2*d1; acc[7] += 0.5*M3*M3*d1;
    if (nscal) acc[8] += Uf(g, 5)[m];
  }
  __shared__ Real red[9][256];
  for (int q = 0; q < 9; q++) red[q][threadIdx.x] = acc[q];
  __syncthreads();
  for (int s = blockDim.x/2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) for (int q = 0; q < 9; q++) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 9) partial[(long)blockIdx.x*9 + threadIdx.x] = red[threadIdx.x][0];
}

// ---- layout conversion, pinned cells, x3 halo pack/unpack ----------------------------------
__global__ void k_aos_to_soa(DevGrid g, int nvar, const Real *aos)
{
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  const long ntot = (long)g.N1*g.N2*g.N3;
  if (lin >= ntot) return;
  const int i = (int)(lin % g.N1), j = (int)((lin / g.N1) % g.N2), k = (int)(lin / ((long)g.N1*g.N2));
  const long m = (long)k*g.sK + (long)j*g.sJ + i;
  for (int v = 0; v < nvar; v++) Uf(g, v)[m] = aos[lin*nvar + v];
}
__global__ void k_soa_to_aos(DevGrid g, int nvar, Real *aos)
{
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  const long ntot = (long)g.N1*g.N2*g.N3;
  if (lin >= ntot) return;
  const int i = (int)(lin % g.N1), j = (int)((lin / g.N1) % g.N2), k = (int)(lin / ((long)g.N1*g.N2));
  const long m = (long)k*g.sK + (long)j*g.sJ + i;
  for (int v = 0; v < nvar; v++) aos[lin*nvar + v] = Uf(g, v)[m];
}
// the same with the zones' contribution to new_dt's maxima from the values just written (what k_pinned_cfl reads back): one launch
__global__ void k_pinned_with_cfl(DevGrid g, int nvar, long long n, const long long *idx, const Real *vals, DevScalars *sc)
{
  Real mx[3] = {0.0, 0.0, 0.0};
  for (long lin = (long)blockIdx.x*blockDim.x + threadIdx.x; lin < n; lin += (long)gridDim.x*blockDim.x) {
    const long long c = idx[lin];
    const int i = (int)(c % g.N1), j = (int)((c / g.N1) % g.N2), k = (int)(c / ((long)g.N1*g.N2));
    const long m = (long)k*g.sK + (long)j*g.sJ + i;
    const Real *q = vals + lin*nvar;
    for (int v = 0; v < nvar; v++) Uf(g, v)[m] = q[v];
    if (i >= g.is && i <= g.ie && j >= g.js && j <= g.je && k >= g.ks && k <= g.ke)      // new_dt looks at active zones only
      cfl_zone(q[0], q[1], q[2], q[3], q[4], g.Gamma, g.Gamma_1, mx);
  }
#pragma unroll
  for (int d = 0; d < 3; d++) {
    Real v = mx[d];
    for (int o = 32; o > 0; o >>= 1) { const Real w = __shfl_xor(v, o); v = (w > v || v != v) ? w : v; }
    if ((threadIdx.x & 63) == 0) atomic_max_pos(&sc->max_v[d], v);
  }
}
__global__ void k_pinned(DevGrid g, int nvar, long long n, const long long *idx, const Real *vals)
{
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= n) return;
  const long long c = idx[lin];                       // linear [k][j][i] index of the host block
  const int i = (int)(c % g.N1), j = (int)((c / g.N1) % g.N2), k = (int)(c / ((long)g.N1*g.N2));
  const long m = (long)k*g.sK + (long)j*g.sJ + i;
  for (int v = 0; v < nvar; v++) Uf(g, v)[m] = vals[lin*nvar + v];
}
// 4 k-planes starting at k0, all i and j (x3 is exchanged last, so corners travel: bvals_mhd.c:170)
// buffer layout [v][kk][j][i]
__global__ void k_pack_x3(DevGrid g, int nvar, int k0, Real *buf)
{
  const long plane = (long)g.N1*g.N2, ntot = plane*AA_NGHOST_;
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= ntot) return;
  const int i = (int)(lin % g.N1), j = (int)((lin / g.N1) % g.N2), kk = (int)(lin / plane);
  const long m = (long)(k0 + kk)*g.sK + (long)j*g.sJ + i;
  for (int v = 0; v < nvar; v++) buf[(long)v*ntot + lin] = Uf(g, v)[m];
}
// bvals_mhd.c:2462 pack_ix2 / pack_ox2 and :2896 unpack_ix2 / unpack_ox2: four rows j0 .. j0+3, all i incl. the x1 ghost
// zones (x1 comes first, so the x1-x2 corners travel), the active k-planes only (x3 comes after)
__global__ void k_pack_x2(DevGrid g, int nvar, int j0, Real *buf)
{
  const long nk = g.ke - g.ks + 1, ntot = (long)g.N1*AA_NGHOST_*nk;
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= ntot) return;
  const int i = (int)(lin % g.N1), jj = (int)((lin / g.N1) % AA_NGHOST_), kk = (int)(lin / ((long)g.N1*AA_NGHOST_));
  const long m = (long)(g.ks + kk)*g.sK + (long)(j0 + jj)*g.sJ + i;
  for (int v = 0; v < nvar; v++) buf[(long)v*ntot + lin] = Uf(g, v)[m];
}
__global__ void k_unpack_x2(DevGrid g, int nvar, int j0, const Real *buf)
{
  const long nk = g.ke - g.ks + 1, ntot = (long)g.N1*AA_NGHOST_*nk;
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= ntot) return;
  const int i = (int)(lin % g.N1), jj = (int)((lin / g.N1) % AA_NGHOST_), kk = (int)(lin / ((long)g.N1*AA_NGHOST_));
  const long m = (long)(g.ks + kk)*g.sK + (long)(j0 + jj)*g.sJ + i;
  for (int v = 0; v < nvar; v++) Uf(g, v)[m] = buf[(long)v*ntot + lin];
}
__global__ void k_unpack_x3(DevGrid g, int nvar, int k0, const Real *buf)
{
  const long plane = (long)g.N1*g.N2, ntot = plane*AA_NGHOST_;
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= ntot) return;
  const int i = (int)(lin % g.N1), j = (int)((lin / g.N1) % g.N2), kk = (int)(lin / plane);
  const long m = (long)(k0 + kk)*g.sK + (long)j*g.sJ + i;
  for (int v = 0; v < nvar; v++) Uf(g, v)[m] = buf[(long)v*ntot + lin];
}

// ---- function-level test kernels ------------------------------------------------------------
template <int NS>
__global__ void k_test_fluxes(Real Gamma, int n, const Real *Ul, const Real *Ur, const Real *eta, Real *F)
{
  const int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  constexpr int NV = 5 + NS;
  Real ul[6] = {0, 0, 0, 0, 0, 0}, ur[6] = {0, 0, 0, 0, 0, 0}, wl[6], wr[6], f[6];
  for (int v = 0; v < NV; v++) { ul[v] = Ul[(long)t*NV + v]; ur[v] = Ur[(long)t*NV + v]; }
  cons_to_prim<NS>(ul, wl, Gamma - 1.0); cons_to_prim<NS>(ur, wr, Gamma - 1.0);
  flux_roe<NS>(ul, ur, wl, wr, eta[t], Gamma, Gamma - 1.0, f);
  for (int v = 0; v < NV; v++) F[(long)t*NV + v] = f[v];
}
// x_div / x_sqrt / x_div_r (hydro_dev.h) beside hipcc's own a/b and sqrt(a): out = [5][n]
__global__ void k_test_xdiv(int n, const Real *a, const Real *b, Real *out)
{
  const int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Real x = a[t], y = b[t];
  out[t] = x_div(x, y); out[(long)n + t] = x/y;
  out[2L*n + t] = x_sqrt(x); out[3L*n + t] = sqrt(x);
  out[4L*n + t] = x_div_r(x, y, 1.0/y);                 // with the correctly rounded reciprocal (what the host supplies for gamma-1)
}
template <int NS>
__global__ void k_test_lr_ppm(Real Gamma, int n, const Real *W, Real dt, Real dx, int il, int iu, Real *Wl, Real *Wr)
{
  const int c = il - 1 + blockIdx.x*blockDim.x + threadIdx.x;     // cells il-1 .. iu+1
  if (c > iu + 1) return;
  constexpr int NV = 5 + NS;
  Real w5[5][6], D[3][6], a[6], b[6];
  for (int q = 0; q < 5; q++) for (int v = 0; v < 6; v++) w5[q][v] = (v < NV) ? W[(long)(c - 2 + q)*NV + v] : 0.0;
  for (int q = 0; q < 3; q++) limited_slopes<NS>(w5[q], w5[q + 1], w5[q + 2], Gamma, D[q]);
  ppm_cell<NS>(w5[1], w5[2], w5[3], D[0], D[1], D[2], dt/dx, Gamma, a, b);
  for (int v = 0; v < NV; v++) { Wl[(long)(c + 1)*NV + v] = a[v]; Wr[(long)c*NV + v] = b[v]; }
}
template <int NS>
__global__ void k_test_lr(Real Gamma, int n, const Real *W, Real dt, Real dx, int il, int iu, Real *Wl, Real *Wr)
{
  const int c = il - 1 + blockIdx.x*blockDim.x + threadIdx.x;     // cells il-1 .. iu+1
  if (c > iu + 1) return;
  constexpr int NV = 5 + NS;
  Real wm[6] = {0, 0, 0, 0, 0, 0}, w[6] = {0, 0, 0, 0, 0, 0}, wp[6] = {0, 0, 0, 0, 0, 0}, a[6], b[6];
  for (int v = 0; v < NV; v++) { wm[v] = W[(long)(c - 1)*NV + v]; w[v] = W[(long)c*NV + v]; wp[v] = W[(long)(c + 1)*NV + v]; }
  plm_cell<NS, true>(wm, w, wp, dt/dx, Gamma, a, b);
  for (int v = 0; v < NV; v++) { Wl[(long)(c + 1)*NV + v] = a[v]; Wr[(long)c*NV + v] = b[v]; }
}

// =============================================================================================
// host-side launchers
// =============================================================================================
void launch_test_xdiv(int n, const Real *a, const Real *b, Real *out, hipStream_t st)
{ hipLaunchKernelGGL(k_test_xdiv, dim3((n + 255)/256), dim3(256), 0, st, n, a, b, out); }
static inline unsigned nblk(long n, int b) { return (unsigned)((n + b - 1)/b); }
static Order zone_order(const HostGrid &g) { Order o = {g.cfg.strip, g.cfg.xcd}; return o; }      // (AA_STRIP / AA_XCD as aa_create found them: grid.h LaunchCfg)
// Grid-stride reduction kernels end in a few same-address atomics per block (one word sustains ~90
// atomics/us): at least 8 zones per thread, between 256 and 4096 blocks.
unsigned reduce_blocks(long n)
{ long nb = (n + 256L*8 - 1)/(256L*8); if (nb < 256) nb = 256; if (nb > 4096) nb = 4096; if (nb*256 > n) nb = (n + 255)/256; return (unsigned)nb; }
static inline unsigned nblk8(long n, int b) { unsigned x = nblk(n, b); return ((x + 7u)/8u)*8u; }   // for xcd_block()

template <int NS, bool GRAV, int MODE, int ORD>
static void sweep_impl_o(const HostGrid &g, const Real *src, int dir, Real dt, hipStream_t st, int koff = 0, int kcnt = -1)
{
  // (koff, kcnt): for the x1 and x2 sweeps, the k-planes ks-2+koff .. +kcnt-1 only (every pencil is on its own: any
  // partition of the planes gives the same bits)
  const int nk_all = g.ke - g.ks + 5;
  if (kcnt < 0) kcnt = nk_all - koff;
  if (kcnt <= 0) return;
  if (dir == 0) {
#if SW_X1_FLAT
    const int flat = g.cfg.x1_flat;
    const long nc = g.ie - g.is + 5, nslots = nc*(g.je - g.js + 5);
    if (flat && nslots < (1L << 30)) {
      int rsh = 31; while ((2L << (rsh - 31)) <= nc) rsh++;            // 31 + floor(log2 nc)
      const unsigned long long rmul = ((1ULL << rsh) + nc - 1)/nc;      // <= 2^31; exact quotients for slots < 2^30
      int Bf = (int)((nslots + 63)/64)*64; if (Bf > 256 || nslots > 256) Bf = 256;
      dim3 gridf((unsigned)((nslots - 1 + Bf - 2)/(Bf - 1)), kcnt);
      if (gridf.x < 1) gridf.x = 1;
      hipLaunchKernelGGL((k_sweep_x1_flat<NS, GRAV, MODE, ORD>), gridf, dim3(Bf), (size_t)(5 + NS)*Bf*sizeof(Real), st, g, src, dt,
                         koff, (unsigned)nslots, (unsigned)rmul, rsh);
      return;
    }
#endif
    const int nfaces = (g.ie - g.is + 1) + 3;          // interfaces l+1..u
    int nb = (nfaces + 254)/255;
    int B = (nfaces + nb - 1)/nb + 1;                  // B-1 interfaces per block
    B = ((B + 63)/64)*64; if (B > 256) B = 256;
    nb = (nfaces + (B - 1) - 1)/(B - 1);
    dim3 grid(nb, g.je - g.js + 5, kcnt);
    size_t lds = (size_t)(5 + NS)*(B + 2)*sizeof(Real);
    hipLaunchKernelGGL((k_sweep_x1<NS, GRAV, MODE, ORD>), grid, dim3(B), lds, st, g, src, dt, koff);
  } else if constexpr (MODE == MODE_CORR) {
    constexpr int BT = 8;     // 8 beats 16 (12.0 ms) despite the 10-rows-for-7-faces halo: more blocks in flight
    const long ni = g.ie - g.is + 5;
    const long nt = (dir == 1 ? g.ke - g.ks : g.je - g.js) + 5;
    const int nfaces = (dir == 1 ? g.je - g.js : g.ke - g.ks) + 1 + 3;
    dim3 grid(nblk(ni*nt, 64), (nfaces + BT - 2)/(BT - 1)), blk(64, BT);
    const size_t lds = (size_t)(5 + NS)*(BT + 2)*64*sizeof(Real);
    if (dir == 1) hipLaunchKernelGGL((k_sweep_tile<NS, 1, GRAV, MODE, BT, ORD>), grid, blk, lds, st, g, src, dt);
    else          hipLaunchKernelGGL((k_sweep_tile<NS, 2, GRAV, MODE, BT, ORD>), grid, blk, lds, st, g, src, dt);
  } else if constexpr (MODE == MODE_FLUX1 || MODE == MODE_VL) {
    const long ni = SW_ALIGN ? march_slots(g) : g.ie - g.is + 5;
    const long nt = (dir == 1 ? kcnt : g.je - g.js + 5);
    const int toff = (dir == 1 ? koff : 0);
    const int nfaces = (dir == 1 ? g.je - g.js : g.ke - g.ks) + 1 + 3;
    // 32 interfaces per thread amortise the 3-cell start-up of a chunk on big Grids; small Grids need the
    // parallelism more (80^3 has only 110 wavefronts of columns): halve the chunk until the launch has
    // a few wavefronts per SIMD (1024 SIMDs)
#ifndef SW_CHUNK
#define SW_CHUNK 32
#endif
    int chunk = SW_CHUNK;
    while (chunk > 4 && (long)nblk(ni*nt, 64)*((nfaces + chunk - 1)/chunk) < 4096) chunk >>= 1;
    dim3 grid(nblk(ni*nt, 64), (nfaces + chunk - 1)/chunk);
    if (dir == 1) hipLaunchKernelGGL((k_sweep_march<NS, 1, GRAV, MODE, ORD>), grid, dim3(64), 0, st, g, src, dt, chunk, toff, (int)nt);
    else          hipLaunchKernelGGL((k_sweep_march<NS, 2, GRAV, MODE, ORD>), grid, dim3(64), 0, st, g, src, dt, chunk, toff, (int)nt);
  }
}
template <int NS, bool GRAV, int MODE>
static void sweep_impl(const HostGrid &g, const Real *src, int dir, Real dt, hipStream_t st, int koff = 0, int kcnt = -1)
{
  // third order: the slope arrays were filled from the state the sweep reconstructs (U; U^{n+1/2} for the van Leer corrector)
  if (g.slope) sweep_impl_o<NS, GRAV, MODE, 3>(g, src, dir, dt, st, koff, kcnt);
  else sweep_impl_o<NS, GRAV, MODE, 2>(g, src, dir, dt, st, koff, kcnt);
}
template <int NS>
static void slopes_impl(const HostGrid &g, int dir, hipStream_t st, const Real *src)
{
  long n = 1;
  const int lo[3] = {g.is, g.js, g.ks}, hi[3] = {g.ie, g.je, g.ke};
  for (int d = 0; d < 3; d++) n *= hi[d] - lo[d] + 1 + (d == dir ? 6 : 4);
  dim3 grid(nblk(n, 256)), blk(256);
#if SLOPES_MARCH
  const int march = g.cfg.slopes_march;
  if (march && dir != 0) {
    const long nq = march_slots(g), nt = (dir == 1 ? g.ke - g.ks : g.je - g.js) + 5;
    const int ncell = (dir == 1 ? g.je - g.js : g.ke - g.ks) + 7;
    int chunk = 32;
    while (chunk > 4 && (long)nblk(nq*nt, 64)*((ncell + chunk - 1)/chunk) < 4096) chunk >>= 1;
    dim3 gm(nblk(nq*nt, 64), (ncell + chunk - 1)/chunk);
    if (dir == 1) hipLaunchKernelGGL((k_slopes_march<NS, 1>), gm, dim3(64), 0, st, g, src, chunk);
    else          hipLaunchKernelGGL((k_slopes_march<NS, 2>), gm, dim3(64), 0, st, g, src, chunk);
    return;
  }
#endif
  if (dir == 0) hipLaunchKernelGGL((k_slopes<NS, 0>), grid, blk, 0, st, g, src);
  else if (dir == 1) hipLaunchKernelGGL((k_slopes<NS, 1>), grid, blk, 0, st, g, src);
  else hipLaunchKernelGGL((k_slopes<NS, 2>), grid, blk, 0, st, g, src);
}
// the correct passes of all three directions in one kernel (after the three first passes)
// the first-pass x1 fluxes of the faces between the tiles of k_correct_all (needs U only: the caller may run it beside the x2 sweep)
template <int NS, bool GRAV>
static void x1_edges_impl(const HostGrid &g, Real dt, hipStream_t st)
{
  if (!CA_X1F || x1_edges(g) <= 0) return;
  const int nj = g.je - g.js + 3, nk = g.ke - g.ks + 3;
  dim3 ge(nblk(nj, 64), nk, x1_edges(g));
  if (g.slope) hipLaunchKernelGGL((k_x1_edge_flux<NS, GRAV, 3>), ge, dim3(64), 0, st, g, dt);
  else         hipLaunchKernelGGL((k_x1_edge_flux<NS, GRAV, 2>), ge, dim3(64), 0, st, g, dt);
}
void launch_x1_edges(const HostGrid &g, int nscal, Real dt, bool grav, hipStream_t st)
{
  if (nscal) { if (grav) x1_edges_impl<1, true>(g, dt, st); else x1_edges_impl<1, false>(g, dt, st); }
  else       { if (grav) x1_edges_impl<0, true>(g, dt, st); else x1_edges_impl<0, false>(g, dt, st); }
}
template <int NS, bool GRAV>
static void correct_all_impl(const HostGrid &g, Real dt, bool x3f, hipStream_t st, bool edges_done)
{
  const int ni = g.ie - g.is + 3, nj = g.je - g.js + 3, nk = g.ke - g.ks + 3;    // zones s-1 .. e+1
  // planes per block: with the x3 first pass on board a chunk starts three planes early (two of first-pass work only and the
  // provider plane), so longer chunks pay: 64 planes 19.7 against 20.0 - 20.6 ms for 32 at 512^3 (128: 20.4, 16: 20.2);
  // AA_CA_KC overrides
  const int kc_env = g.cfg.ca_kc;
  int kc = kc_env > 0 ? kc_env : (x3f ? 64 : 32);
  // ... on Grids with fewer columns shorter chunks pay more than their start-up planes cost: the launch should hold about
  // eight rounds of the 512 blocks that are resident at a time (measured, profiles/microbench/kc_small.sh: 128^3 and 192^3 best
  // with 8 planes, 1.30 against 1.65 ms at 192^3; 256^3 with 16, 2.80 against 3.09; 320^3 with 32; from 384^3 on with 64)
  if (kc_env <= 0) while (kc > 8 && (long)nblk(ni + 15, 64)*nblk(nj, CA_TJ)*((nk + kc - 1)/kc) < 4096) kc >>= 1;
  dim3 grid(nblk(ni + 15, 64), nblk(nj, CA_TJ), (nk + kc - 1)/kc), blk(64, CA_TJ);
  if (x3f && !edges_done) x1_edges_impl<NS, GRAV>(g, dt, st);
  const StepRatios sr = step_ratios(g, dt);
  if (x3f) {
    if (g.slope) hipLaunchKernelGGL((k_correct_all<NS, GRAV, 3, true>), grid, blk, 0, st, g, dt, kc, sr);
    else         hipLaunchKernelGGL((k_correct_all<NS, GRAV, 2, true>), grid, blk, 0, st, g, dt, kc, sr);
  } else {
    if (g.slope) hipLaunchKernelGGL((k_correct_all<NS, GRAV, 3, false>), grid, blk, 0, st, g, dt, kc, sr);
    else         hipLaunchKernelGGL((k_correct_all<NS, GRAV, 2, false>), grid, blk, 0, st, g, dt, kc, sr);
  }
  const long nb1 = (g.ie + 1 - (g.is - 16))/64, nb2 = (g.je + 1 - (g.js - 1))/CA_TJ;
  {
    const long c0 = nb1 > 0 ? nb1*nj*nk : 0, c1 = nb2 > 0 ? (long)ni*nb2*nk : 0;      // (a direction without inner tile edges returns at once)
    if (c0 > 0 || c1 > 0) hipLaunchKernelGGL(k_eta_edges, dim3(nblk(c0 > c1 ? c0 : c1, 256), 2), dim3(256), 0, st, g);
  }
}
void launch_correct_all(const HostGrid &g, int nscal, Real dt, bool grav, bool x3f, hipStream_t st, bool edges_done)
{
  if (nscal) { if (grav) correct_all_impl<1, true>(g, dt, x3f, st, edges_done); else correct_all_impl<1, false>(g, dt, x3f, st, edges_done); }
  else       { if (grav) correct_all_impl<0, true>(g, dt, x3f, st, edges_done); else correct_all_impl<0, false>(g, dt, x3f, st, edges_done); }
}

void launch_slopes(const HostGrid &g, int nscal, int dir, hipStream_t st, const Real *src)
{ if (!src) src = g.U; if (nscal) slopes_impl<1>(g, dir, st, src); else slopes_impl<0>(g, dir, st, src); }

void launch_sweep(const HostGrid &g, int nscal, int dir, Real dt, bool grav, hipStream_t st, int koff, int kcnt)
{
  if (nscal) { if (grav) sweep_impl<1, true, MODE_FLUX1>(g, g.U, dir, dt, st, koff, kcnt); else sweep_impl<1, false, MODE_FLUX1>(g, g.U, dir, dt, st, koff, kcnt); }
  else       { if (grav) sweep_impl<0, true, MODE_FLUX1>(g, g.U, dir, dt, st, koff, kcnt); else sweep_impl<0, false, MODE_FLUX1>(g, g.U, dir, dt, st, koff, kcnt); }
}
// x1 first pass + x1 correct pass in one sweep (must run after the x2 and x3 first passes)
void launch_sweep_correct_x1(const HostGrid &g, int nscal, Real dt, bool grav, hipStream_t st)
{
  if (nscal) { if (grav) sweep_impl<1, true, MODE_BOTH>(g, g.U, 0, dt, st); else sweep_impl<1, false, MODE_BOTH>(g, g.U, 0, dt, st); }
  else       { if (grav) sweep_impl<0, true, MODE_BOTH>(g, g.U, 0, dt, st); else sweep_impl<0, false, MODE_BOTH>(g, g.U, 0, dt, st); }
}
// CTU steps 5-7, 8a, 9a for the faces of one direction (after all three first-pass sweeps)
void launch_correct(const HostGrid &g, int nscal, int dir, Real dt, bool grav, hipStream_t st)
{
  if (nscal) { if (grav) sweep_impl<1, true, MODE_CORR>(g, g.U, dir, dt, st); else sweep_impl<1, false, MODE_CORR>(g, g.U, dir, dt, st); }
  else       { if (grav) sweep_impl<0, true, MODE_CORR>(g, g.U, dir, dt, st); else sweep_impl<0, false, MODE_CORR>(g, g.U, dir, dt, st); }
}
// VL second-order fluxes: PLM without tracing on U^{n+1/2} (kept in LR[0][L]), etah = 0
void launch_vl_flux2(const HostGrid &g, int nscal, int dir, Real dt, hipStream_t st)
{
  if (nscal) sweep_impl<1, false, MODE_VL>(g, g.LR, dir, dt, st); else sweep_impl<0, false, MODE_VL>(g, g.LR, dir, dt, st);
}
template <int NS>
static void vl_flux1_impl(const DevGrid &g, int dir, hipStream_t st)
{
  long n[3] = {g.N1, g.N2, g.N3}; n[dir] -= 1;
  dim3 grid(nblk(n[0]*n[1]*n[2], 256)), blk(256);
  if (dir == 0) hipLaunchKernelGGL((k_vl_flux1<NS, 0>), grid, blk, 0, st, g);
  else if (dir == 1) hipLaunchKernelGGL((k_vl_flux1<NS, 1>), grid, blk, 0, st, g);
  else hipLaunchKernelGGL((k_vl_flux1<NS, 2>), grid, blk, 0, st, g);
}
void launch_vl_flux1(const DevGrid &g, int nscal, int dir, hipStream_t st)
{ if (nscal) vl_flux1_impl<1>(g, dir, st); else vl_flux1_impl<0>(g, dir, st); }
// donor-cell fluxes + U^{n+1/2} in one kernel (what launch_vl_flux1 x3 + launch_vl_uhalf do)
void launch_vl_predict(const DevGrid &g, int nscal, Real dt, bool grav, hipStream_t st)
{
  const int ni = g.ie - g.is + 7, nj = g.je - g.js + 7, nk = g.ke - g.ks + 7;
  int kc = 32;
  while (kc > 4 && (long)nblk(ni, 63)*nblk(nj, VP_TJ - 1)*((nk + kc - 1)/kc) < 1024) kc >>= 1;
  dim3 grid(nblk(ni, 63), nblk(nj, VP_TJ - 1), (nk + kc - 1)/kc), blk(64, VP_TJ);
  if (nscal) { if (grav) hipLaunchKernelGGL((k_vl_predict<1, true>), grid, blk, 0, st, g, dt, kc);
               else      hipLaunchKernelGGL((k_vl_predict<1, false>), grid, blk, 0, st, g, dt, kc); }
  else       { if (grav) hipLaunchKernelGGL((k_vl_predict<0, true>), grid, blk, 0, st, g, dt, kc);
               else      hipLaunchKernelGGL((k_vl_predict<0, false>), grid, blk, 0, st, g, dt, kc); }
}
void launch_vl_uhalf(const DevGrid &g, int nscal, Real dt, bool grav, hipStream_t st)
{
  const long n = (long)(g.ie - g.is + 7)*(g.je - g.js + 7)*(g.ke - g.ks + 7);
  dim3 grid(nblk(n, 256)), blk(256);
  if (nscal) { if (grav) hipLaunchKernelGGL((k_vl_uhalf<1, true>), grid, blk, 0, st, g, dt);
               else      hipLaunchKernelGGL((k_vl_uhalf<1, false>), grid, blk, 0, st, g, dt); }
  else       { if (grav) hipLaunchKernelGGL((k_vl_uhalf<0, true>), grid, blk, 0, st, g, dt);
               else      hipLaunchKernelGGL((k_vl_uhalf<0, false>), grid, blk, 0, st, g, dt); }
}

template <int NS>
static void flux2_impl(const HostGrid &g, int dir, hipStream_t st)
{
  const long n = (long)(g.ie - g.is + 1 + (dir == 0))*(g.je - g.js + 1 + (dir == 1))*(g.ke - g.ks + 1 + (dir == 2));
  dim3 grid(nblk8(n, 256)), blk(256);
  if (dir == 0) hipLaunchKernelGGL((k_flux2<NS, 0>), grid, blk, 0, st, g, zone_order(g));
  else if (dir == 1) hipLaunchKernelGGL((k_flux2<NS, 1>), grid, blk, 0, st, g, zone_order(g));
  else hipLaunchKernelGGL((k_flux2<NS, 2>), grid, blk, 0, st, g, zone_order(g));
}
void launch_flux2(const HostGrid &g, int nscal, int dir, hipStream_t st)
{ if (nscal) flux2_impl<1>(g, dir, st); else flux2_impl<0>(g, dir, st); }

// fused second-pass fluxes + update (CTU); `keep`: the face planes whose fluxes are stored as well
template <int NS, bool GRAV>
static void launch_fu(const DevGrid &g, Real dt, int kc, dim3 grid, dim3 blk, const KeepPlanes *keep, DevScalars *sc, const unsigned char *pinmask,
                      hipStream_t st)
{
  KeepPlanes none = {0, {{0}}};
  if (sc) {
    if (keep && keep->n) hipLaunchKernelGGL((k_flux2_update<NS, GRAV, true, true>), grid, blk, 0, st, g, g.dhalf, dt, kc, *keep, sc, pinmask, step_ratios(g, dt));
    else                 hipLaunchKernelGGL((k_flux2_update<NS, GRAV, false, true>), grid, blk, 0, st, g, g.dhalf, dt, kc, none, sc, pinmask, step_ratios(g, dt));
  } else {
    if (keep && keep->n) hipLaunchKernelGGL((k_flux2_update<NS, GRAV, true, false>), grid, blk, 0, st, g, g.dhalf, dt, kc, *keep, sc, pinmask, step_ratios(g, dt));
    else                 hipLaunchKernelGGL((k_flux2_update<NS, GRAV, false, false>), grid, blk, 0, st, g, g.dhalf, dt, kc, none, sc, pinmask, step_ratios(g, dt));
  }
}
void launch_pinned_cfl(const DevGrid &g, long long n, const long long *idx, DevScalars *sc, hipStream_t st)
{ if (n > 0) { unsigned nb = nblk(n, 256*8); if (nb > 256) nb = 256; if (nb < 1) nb = 1;
               hipLaunchKernelGGL(k_pinned_cfl, dim3(nb), dim3(256), 0, st, g, n, idx, sc); } }
void launch_pin_mask(const DevGrid &g, long long n, const long long *idx, unsigned char *mask, hipStream_t st)
{ if (n > 0) hipLaunchKernelGGL(k_pin_mask, dim3(nblk(n, 256)), dim3(256), 0, st, g, n, idx, mask); }
void launch_flux2_update(const HostGrid &g, int nscal, Real dt, bool grav, const KeepPlanes *keep, hipStream_t st, DevScalars *sc, const unsigned char *pinmask)
{
  const int ni = g.ie - g.is + 1, nj = g.je - g.js + 1, nk = g.ke - g.ks + 1;
  const int kc_env = g.cfg.fu_kc;
  int kc = kc_env > 0 ? kc_env : 32;
  if (kc_env <= 0) while (kc > 4 && (long)nblk(ni, 64)*nblk(nj, FU_TJ - 1)*((nk + kc - 1)/kc) < 1024) kc >>= 1;
  dim3 grid(nblk(ni, 64), nblk(nj, FU_TJ - 1), (nk + kc - 1)/kc), blk(64, FU_TJ);
  if (nscal) { if (grav) launch_fu<1, true>(g, dt, kc, grid, blk, keep, sc, pinmask, st); else launch_fu<1, false>(g, dt, kc, grid, blk, keep, sc, pinmask, st); }
  else       { if (grav) launch_fu<0, true>(g, dt, kc, grid, blk, keep, sc, pinmask, st); else launch_fu<0, false>(g, dt, kc, grid, blk, keep, sc, pinmask, st); }
}

long update_blocks(const DevGrid &g)
{ return (long)nblk8((long)(g.ie - g.is + 1)*(g.je - g.js + 1)*(g.ke - g.ks + 1), 256); }
// sc != nullptr: new_dt's maxima ride on the update (cfl_part: 3 * update_blocks(g) doubles of scratch; pinmask: zones left out)
void launch_update(const HostGrid &g, int nscal, const Real *dhalf, Real dt, bool grav, hipStream_t st, DevScalars *sc, Real *cfl_part,
                   const unsigned char *pinmask)
{
  const long n = (long)(g.ie - g.is + 1)*(g.je - g.js + 1)*(g.ke - g.ks + 1);
  dim3 grid(nblk8(n, 256)), blk(256);
  if (sc && cfl_part) {
    if (nscal) { if (grav) hipLaunchKernelGGL((k_update<1, true, true>), grid, blk, 0, st, g, dhalf, dt, zone_order(g), cfl_part, pinmask);
                 else      hipLaunchKernelGGL((k_update<1, false, true>), grid, blk, 0, st, g, dhalf, dt, zone_order(g), cfl_part, pinmask); }
    else       { if (grav) hipLaunchKernelGGL((k_update<0, true, true>), grid, blk, 0, st, g, dhalf, dt, zone_order(g), cfl_part, pinmask);
                 else      hipLaunchKernelGGL((k_update<0, false, true>), grid, blk, 0, st, g, dhalf, dt, zone_order(g), cfl_part, pinmask); }
    unsigned nf = nblk(grid.x, 256*8); if (nf > 256) nf = 256;
    hipLaunchKernelGGL(k_cfl_fold, dim3(nf), dim3(256), 0, st, cfl_part, (int)grid.x, sc);
    return;
  }
  if (nscal) { if (grav) hipLaunchKernelGGL((k_update<1, true, false>), grid, blk, 0, st, g, dhalf, dt, zone_order(g), nullptr, nullptr);
               else      hipLaunchKernelGGL((k_update<1, false, false>), grid, blk, 0, st, g, dhalf, dt, zone_order(g), nullptr, nullptr); }
  else       { if (grav) hipLaunchKernelGGL((k_update<0, true, false>), grid, blk, 0, st, g, dhalf, dt, zone_order(g), nullptr, nullptr);
               else      hipLaunchKernelGGL((k_update<0, false, false>), grid, blk, 0, st, g, dhalf, dt, zone_order(g), nullptr, nullptr); }
}

void launch_bc(const DevGrid &g, int nscal, int dir, int side, int flag, hipStream_t st)
{
  const int lo3[3] = {g.is, g.js, g.ks}, hi3[3] = {g.ie, g.je, g.ke}, N[3] = {g.N1, g.N2, g.N3};
  long n = 1;
  for (int a = 0; a < 3; a++) n *= (a == dir) ? 4 : (a < dir ? N[a] : hi3[a] - lo3[a] + 1);
  hipLaunchKernelGGL(k_bc, dim3(nblk(n, 256)), dim3(256), 0, st, g, 5 + nscal, dir, side, flag, 0);
}
// the two sides of one direction never read each other's ghost zones (sources are active zones along
// `dir`, bvals_mhd.c:959-2317): one launch for both
void launch_bc_dir(const DevGrid &g, int nscal, int dir, int flag_in, int flag_out, hipStream_t st)
{
  if (!flag_in && !flag_out) return;
  const int lo3[3] = {g.is, g.js, g.ks}, hi3[3] = {g.ie, g.je, g.ke}, N[3] = {g.N1, g.N2, g.N3};
  long n = 1;
  for (int a = 0; a < 3; a++) n *= (a == dir) ? 4 : (a < dir ? N[a] : hi3[a] - lo3[a] + 1);
  hipLaunchKernelGGL(k_bc, dim3(nblk(n, 256), 2), dim3(256), 0, st, g, 5 + nscal, dir, -1, flag_in, flag_out);
}

void launch_bc_shell(const DevGrid &g, int nscal, const int flags[6], hipStream_t st)
{
  const long G2 = 2*AA_NGHOST_, nk = g.ke - g.ks + 1, nj = g.je - g.js + 1;
  const long n = G2*g.N2*g.N1 + G2*g.N1*nk + G2*nj*nk;
  hipLaunchKernelGGL(k_bc_shell, dim3(nblk(n, 256)), dim3(256), 0, st, g, 5 + nscal, flags[0], flags[1], flags[2], flags[3], flags[4], flags[5]);
}
void launch_cfl(const DevGrid &g, DevScalars *sc, hipStream_t st)
{
  const long n = (long)(g.ie - g.is + 1)*(g.je - g.js + 1)*(g.ke - g.ks + 1);
  const unsigned nb = reduce_blocks(n);
  hipLaunchKernelGGL(k_cfl, dim3(nb), dim3(256), 0, st, g, sc);
}

int launch_history(const DevGrid &g, int nscal, Real *partial, hipStream_t st)
{
  const long n = (long)(g.ie - g.is + 1)*(g.je - g.js + 1)*(g.ke - g.ks + 1);
  unsigned nb = nblk(n, 256); if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_history, dim3(nb), dim3(256), 0, st, g, nscal, partial);
  return (int)nb;
}

void launch_aos_to_soa(const DevGrid &g, int nvar, const Real *aos, hipStream_t st)
{ const long n = (long)g.N1*g.N2*g.N3; hipLaunchKernelGGL(k_aos_to_soa, dim3(nblk(n, 256)), dim3(256), 0, st, g, nvar, aos); }
void launch_soa_to_aos(const DevGrid &g, int nvar, Real *aos, hipStream_t st)
{ const long n = (long)g.N1*g.N2*g.N3; hipLaunchKernelGGL(k_soa_to_aos, dim3(nblk(n, 256)), dim3(256), 0, st, g, nvar, aos); }
void launch_pinned(const DevGrid &g, int nvar, long long n, const long long *idx, const Real *vals, hipStream_t st, DevScalars *sc)
{
  if (n <= 0) return;
  if (sc) { unsigned nb = nblk(n, 256*8); if (nb > 256) nb = 256; if (nb < 1) nb = 1;
            hipLaunchKernelGGL(k_pinned_with_cfl, dim3(nb), dim3(256), 0, st, g, nvar, n, idx, vals, sc); }
  else hipLaunchKernelGGL(k_pinned, dim3(nblk(n, 256)), dim3(256), 0, st, g, nvar, n, idx, vals);
}
void launch_pack_x3(const DevGrid &g, int nvar, int k0, Real *buf, hipStream_t st)
{ const long n = (long)g.N1*g.N2*4; hipLaunchKernelGGL(k_pack_x3, dim3(nblk(n, 256)), dim3(256), 0, st, g, nvar, k0, buf); }
void launch_unpack_x3(const DevGrid &g, int nvar, int k0, const Real *buf, hipStream_t st)
{ const long n = (long)g.N1*g.N2*4; hipLaunchKernelGGL(k_unpack_x3, dim3(nblk(n, 256)), dim3(256), 0, st, g, nvar, k0, buf); }
void launch_pack_x2(const DevGrid &g, int nvar, int j0, Real *buf, hipStream_t st)
{ const long n = (long)g.N1*4*(g.ke - g.ks + 1); hipLaunchKernelGGL(k_pack_x2, dim3(nblk(n, 256)), dim3(256), 0, st, g, nvar, j0, buf); }
void launch_unpack_x2(const DevGrid &g, int nvar, int j0, const Real *buf, hipStream_t st)
{ const long n = (long)g.N1*4*(g.ke - g.ks + 1); hipLaunchKernelGGL(k_unpack_x2, dim3(nblk(n, 256)), dim3(256), 0, st, g, nvar, j0, buf); }

void launch_test_fluxes(int nscal, Real gamma, int n, const Real *Ul, const Real *Ur, const Real *eta, Real *F, hipStream_t st)
{
  if (nscal) hipLaunchKernelGGL((k_test_fluxes<1>), dim3(nblk(n, 128)), dim3(128), 0, st, gamma, n, Ul, Ur, eta, F);
  else       hipLaunchKernelGGL((k_test_fluxes<0>), dim3(nblk(n, 128)), dim3(128), 0, st, gamma, n, Ul, Ur, eta, F);
}
void launch_test_lr_ppm(int nscal, Real gamma, int n, const Real *W, Real dt, Real dx, int il, int iu, Real *Wl, Real *Wr, hipStream_t st)
{
  const int nc = iu - il + 3;
  if (nscal) hipLaunchKernelGGL((k_test_lr_ppm<1>), dim3(nblk(nc, 128)), dim3(128), 0, st, gamma, n, W, dt, dx, il, iu, Wl, Wr);
  else       hipLaunchKernelGGL((k_test_lr_ppm<0>), dim3(nblk(nc, 128)), dim3(128), 0, st, gamma, n, W, dt, dx, il, iu, Wl, Wr);
}
void launch_test_lr(int nscal, Real gamma, int n, const Real *W, Real dt, Real dx, int il, int iu, Real *Wl, Real *Wr, hipStream_t st)
{
  (void)n;
  const int nc = iu - il + 3;
  if (nscal) hipLaunchKernelGGL((k_test_lr<1>), dim3(nblk(nc, 128)), dim3(128), 0, st, gamma, n, W, dt, dx, il, iu, Wl, Wr);
  else       hipLaunchKernelGGL((k_test_lr<0>), dim3(nblk(nc, 128)), dim3(128), 0, st, gamma, n, W, dt, dx, il, iu, Wl, Wr);
}

}  // namespace aa / aa_cool
