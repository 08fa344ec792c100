// ion_kernels.hip -- plane-parallel ionizing-radiation step (ionradiation/ionrad_3d.c,
// ionradplane_3d.c, ionrad_chemistry.c) as HIP kernels for gfx950.
//
// During the ion step only E and s0 change; d and the momenta (hence the kinetic energy and the
// speeds) are frozen.  ion_begin therefore stores ke and max_d|v_d| once, and one radiation
// sub-cycle is three kernels that touch 1, 5 and 9 fields per zone (the reference's loops touch
// 6 + scratch each):
//
//   ray_sweep   get_ph_rate_plane (ionradplane_3d.c:88).  Every (j,k) ray is an exclusive prefix
//               product of exp(-tau) along x1 with a data-dependent cut-off.  A block stages a
//               64-ray x RS_CH-cell tile: all 256 threads evaluate exp(-tau) (the expensive part)
//               with coalesced loads; one wavefront -- one lane per ray -- carries the product
//               serially through LDS in exactly the reference's multiplication order; then all
//               threads turn the staged incoming fluxes into ph_rate / EdgeFlux with coalesced
//               stores.  The flux is carried across tiles in LDS; a block whose rays are all
//               extinguished only streams zeros.
//   ion_rates   compute_chem_rates + compute_therm_rates (ionrad_3d.c:288, :414): the two
//               time-step limits (block MIN -> 2 atomics per block).  The rates themselves are NOT
//               stored.
//   ion_update  ionization_update + apply_temp_floor + apply_neutral_floor + check_range +
//               compute_dt_hydro (:565, :70, :140, :206, :593): re-evaluates the two rates from
//               (state, ph_rate) -- a log and three exp per cell are cheaper than 32 B/zone of
//               HBM round trip -- applies them with the globally reduced dt, and reduces the
//               out-of-range count and the hydro CFL limit.
// HBM-bound FP64 streaming; no MFMA.
#include <float.h>
#include "grid.h"
#include "hydro_dev.h"

namespace aa {

#define MINFLUXFRAC 1.0e-3    /* ionrad.h:26 */
#define IONFRACFLOOR 1.0e-4   /* :31 */
#define CION 8.0e5            /* :36 */
#define MAXSIGNCOUNT 4        /* ionrad_3d.c:286 */
#define DAMPFACTOR 0.5        /* :287 */
#define KB_CHEM 1.38e-16      /* ionrad_chemistry.c:43 */

AA_DEV Real *Uq(const DevGrid &g, int v) { return g.U + (long)v*g.nc; }

struct Cell { Real d, ke, E, s; };                       // what the ion step needs of a zone
struct IonQ { Real n_H, n_Hplus, n_e, x, e_th, T, di, muq; };

// ionrad_3d.c:82-101 (same expressions are repeated at :313-331 and :438-456).  The ion step is
// not bit-reproducible against the CPU anyway (device exp/log vs glibc), so the seven divisions
// of the reference are folded into two plus multiplications by host-computed reciprocals:
// FP64 division is ~10x the cost of a multiply on CDNA4.
AA_DEV IonQ ion_q(const Cell &c, const IonPar &p, Real Gamma_1)
{
  IonQ q;
  q.n_H = c.s * p.inv_mH;
  q.n_Hplus = (c.d - c.s) * p.inv_mH;
  q.n_e = q.n_Hplus + c.d * p.aC14;
  q.x = q.n_e / (q.n_H + q.n_Hplus);
  q.di = 1.0 / c.d;
  q.e_th = c.E - c.ke;
  q.muq = q.x*0.5*p.m_H+(1.0-q.x)*p.mu;
  q.T = Gamma_1 * (q.e_th * q.di) * q.muq * p.inv_kB;
  return q;
}

AA_DEV Real neutral_lim(Real d, const IonPar &p)   // ionrad_3d.c:147-148
{ Real d_nlim = d*IONFRACFLOOR; return d_nlim < p.d_nlo ? d_nlim : p.d_nlo; }

// apply_temp_floor (:70-131) then apply_neutral_floor (:140-156) on one cell; `q` returns the derived
// quantities of the cell as it entered, `changed` whether E or s was touched (then q is stale)
AA_DEV void floors(Cell &c, const IonPar &p, Real Gamma_1, IonQ &q, bool &changed)
{
  const Real E0 = c.E, s0 = c.s;
  q = ion_q(c, p, Gamma_1);
  if (q.T < p.tfloor) {
    Real e_sp = p.tfloor * p.k_B / (q.muq * Gamma_1);
    c.E = c.ke + e_sp * c.d;
  }
  if ((q.T > p.tceil) && (p.tceil > 0)) {
    Real e_sp = p.tceil * p.k_B / (q.muq * Gamma_1);
    c.E = c.ke + e_sp * c.d;
  }
  Real d_nlim = neutral_lim(c.d, p);
  if (c.s < d_nlim) c.s = d_nlim; else if (c.s > c.d) c.s = c.d;
  changed = (c.E != E0) || (c.s != s0);
}
AA_DEV void floors(Cell &c, const IonPar &p, Real Gamma_1) { IonQ q; bool ch; floors(c, p, Gamma_1, q, ch); }

// Undamped rate of change of the neutral density (compute_chem_rates, ionrad_3d.c:334-341).
// recomb_rate_coef = 2.59e-13 (T/1e4)^-0.7 and recomb_cool_rate_coef = 6.11e-10 T^-0.89 k_B T
// (ionrad_chemistry.c:111,:137) share ONE log: T^y = exp(y ln T) (rel. error ~|y ln T| eps ~1e-15);
// the floored temperature uses the host-computed coefficient.
AA_DEV Real chem_rate(const IonQ &q, Real ph, const IonPar &p, Real &lnT, bool &cold)
{
  cold = (q.T < p.tfloor);
  Real rec;
  if (cold) { lnT = 0.0; rec = p.rec_floor; }
  else { lnT = log(q.T); rec = 2.59e-13*exp(-0.7*(lnT - 9.210340371976184)); }   // ln(1e4)
  return rec * p.time_unit * q.n_e * q.n_Hplus - ph * q.n_H;
}

// edot of compute_therm_rates (ionrad_3d.c:460-490); `skip` cells get 0
AA_DEV Real therm_rate(const IonQ &q, Real ph, Real lnT, const IonPar &p)
{
  const Real Tt = q.T;
  const Real rcool = (Tt < 100.0) ? 0.0 : 6.11e-10*exp(-0.89*lnT)*KB_CHEM*Tt;          // chemistry :137
  const Real arg = 118348/Tt;
  const Real lya = (arg > 745.2) ? 0.0 : -7.5e-19*q.n_e*q.n_H*exp(-arg);              // :350, call at ionrad_3d.c:484
  return ph * p.e_gamma * q.n_H - rcool * p.time_unit * q.n_Hplus * q.n_e + lya * p.time_unit;
}

AA_DEV Real damp(Real nHdot, int sign_count)          // ionrad_3d.c:360-363
{ for (int n = MAXSIGNCOUNT; n < sign_count; n++) nHdot *= DAMPFACTOR; return nHdot; }

AA_DEV bool ratio_ge(Real a, Real b, Real L)
{ return (a > 0.0 && b > 0.0) ? (a >= L*b) : (a / b >= L); }

AA_DEV bool active_cell(const DevGrid &g, long lin, long &m)
{
  const int ni = g.Nx1, nj = g.Nx2;
  if (lin >= (long)ni*nj*g.Nx3) return false;
  const int i = g.is + (int)(lin % ni), j = g.js + (int)((lin / ni) % nj), k = g.ks + (int)(lin / ((long)ni*nj));
  m = (long)k*g.sK + (long)j*g.sJ + i;
  return true;
}

// ---- entry of ion_radtransfer_3d: floors + save_energy_and_x (:896-905, :162-196) -------------
// also freezes ke and max_d |v_d|
__global__ void __launch_bounds__(256)
k_ion_begin(DevGrid g, IonPar p)
{
  long m;
  if (!active_cell(g, (long)blockIdx.x*blockDim.x + threadIdx.x, m)) return;
  const Real d = Uq(g,0)[m], M1 = Uq(g,1)[m], M2 = Uq(g,2)[m], M3 = Uq(g,3)[m];
  const Real di = 1.0/d;
  Cell c; c.d = d; c.ke = 0.5 * (M1*M1 + M2*M2 + M3*M3) * di; c.E = Uq(g,4)[m]; c.s = Uq(g,5)[m];
  const Real E0 = c.E, s0 = c.s;
  IonQ q; bool floored;
  floors(c, p, g.Gamma_1, q, floored);
  if (c.E != E0) Uq(g,4)[m] = c.E;
  if (c.s != s0) Uq(g,5)[m] = c.s;
  if (floored) q = ion_q(c, p, g.Gamma_1);
  g.e_init[m] = c.E;
  g.x_init[m] = q.x;
  g.sign[m] = make_int2(0, 0);
  g.kin[m] = c.ke;
  // compute_dt_hydro takes max_d (|v_d| + a)/dx_d with one sound speed a: for dx1=dx2=dx3 that is
  // (max_d|v_d| + a)/dx exactly, so one frozen number per zone replaces the three momenta
  g.vmax[m] = rmax(rmax(fabs(M1*di), fabs(M2*di)), fabs(M3*di));
}

// ---- block reductions --------------------------------------------------------------------------
AA_DEV void block_min_to(unsigned long long *addr, Real v, Real *red)
{
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = blockDim.x/2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = rmin(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMin(addr, (unsigned long long)__double_as_longlong(red[0]));
  __syncthreads();
}

// compute_chem_rates :334-394 + compute_therm_rates :460-557 of one zone: the sign bookkeeping `sg`
// (last_sign, sign_count) is updated in place, the two time-step limits of the zone are returned.  The
// rates themselves are not stored (k_ion_update re-evaluates them).
AA_DEV void rates_cell(const Cell &c, Real ph, int2 &sg, const IonPar &p, Real Gamma_1, DevScalars *sc, Real &dt_chem, Real &dt_therm)
{
  const IonQ iq = ion_q(c, p, Gamma_1);
  Real lnT; bool cold;
  Real nHdot = chem_rate(iq, ph, p, lnT, cold);
  if (nHdot < 0.0) {
    if (sg.x == 1) sg.y++; else if (sg.y > 0) sg.y--;
    sg.x = -1;
  } else if (nHdot > 0.0) {
    if (sg.x == -1) sg.y++; else if (sg.y > 0) sg.y--;
    sg.x = 1;
  } else { sg.x = 0; sg.y = 0; }
  nHdot = damp(nHdot, sg.y);
  const Real d_nlim = neutral_lim(c.d, p);
  const Real inv_n = 1.0/nHdot;
  Real dt1, dt2;
  if (nHdot == 0.0) { dt1 = dt2 = DBL_MAX; }
  else if (nHdot > 0.0) {
    dt1 = p.cx1 * iq.n_e * inv_n;               // max_dx_iter/(1+max_dx_iter) * n_e / nHdot
    dt2 = p.max_dx_iter * iq.n_H * inv_n;
  } else if (c.s > 1.0001*d_nlim) {
    dt1 = -p.max_dx_iter * iq.n_e * inv_n;
    dt2 = -p.cx1 * iq.n_H * inv_n;
  } else { dt1 = dt2 = DBL_MAX; }
  dt_chem = (dt1 < dt2) ? dt1 : dt2;
  if (dt_chem < 0) { atomicExch(&sc->neg_dt_chem, 1); dt_chem = DBL_MAX; }
  dt_therm = DBL_MAX;
  const bool skip = cold || ((nHdot < 0) && (c.s < 1.0001*d_nlim));
  if (!skip) {
    const Real edot = therm_rate(iq, ph, lnT, p);
    Real t1, t2; bool have = true;
    const Real inv_e = 1.0/edot;
    if (edot == 0.0) { t1 = t2 = DBL_MAX; }
    else if (edot > 0.0) {
      t1 = p.max_de_iter * c.E * inv_e;
      t2 = p.max_de_therm_iter * iq.e_th * inv_e;
    } else {
      const Real e_sp_min = p.tfloor * p.k_B / (iq.muq * Gamma_1);
      const Real e_th_min = e_sp_min * c.d;
      const Real e_min = c.ke + e_th_min;
      if ((iq.e_th*p.ie1 < e_th_min) && (c.E*p.ie2 < e_min)) have = false;   // e/(1+max_de*_iter)
      t1 = -p.ce2 * c.E * inv_e;
      t2 = -p.ce1 * iq.e_th * inv_e;
    }
    if (have) dt_therm = (t1 < t2) ? t1 : t2;
    if (!(dt_therm == dt_therm) || dt_therm < 0) dt_therm = DBL_MAX;
  }
}

// ---- ray sweep --------------------------------------------------------------------------------
#ifndef RS_CH
#define RS_CH 32
#endif
#define RS_RAYS 64
#define RS_PASS (RS_RAYS*RS_CH/256)      /* cells per thread per tile */
// RATES: the chemistry / thermal rates of every zone (compute_chem_rates + compute_therm_rates) are
// evaluated right where its ph_rate is produced, and the two time-step limits reduced here: the separate
// k_ion_rates pass (which re-reads s0 and ph_rate and is bound by its log + 3 exp per zone) then overlaps
// with the loads / barriers / serial product this kernel is bound by.
template <bool RATES>
__global__ void __launch_bounds__(256, RATES ? 3 : 4)
k_ray_sweep(DevGrid g, IonPar p, Real flux0, int from_edgeflux, DevScalars *sc)
{
  __shared__ Real s_etau[RS_RAYS][RS_CH + 1];
  __shared__ Real s_fin[RS_RAYS][RS_CH + 1];
  __shared__ Real s_flux[RS_RAYS];
  __shared__ Real s_f0[RS_RAYS];                              // flux entering the ray (denominator of :299)
  __shared__ int  s_dead[RS_RAYS];
  __shared__ int  s_nalive;
  __shared__ Real red[RATES ? 256 : 1];
  const int tid = threadIdx.x;
  const int j0 = g.js + blockIdx.x*RS_RAYS;                 // rays: 64 consecutive j at one k
  const int k = g.ks + blockIdx.y;
  const int nrays = min(RS_RAYS, g.je - j0 + 1);
  const int col = tid % RS_CH, rsub = tid / RS_CH;           // 256/RS_CH ray-rows per pass
  constexpr int RSTEP = 256/RS_CH;
  const long efp = (long)(g.Nx1 + 1), efrow = (long)(g.Nx2 + 1)*efp;
  Real dt_chem_min = DBL_MAX, dt_therm_min = DBL_MAX;
  if (tid < RS_RAYS) {
    Real f0 = flux0;
    if (from_edgeflux && tid < nrays)                          // :271 refined level: the parent's flux
      f0 = g.edgeflux[(long)(k - g.ks)*efrow + (long)(j0 + tid - g.js)*efp];
    s_flux[tid] = f0; s_f0[tid] = f0; s_dead[tid] = (tid < nrays) ? 0 : 1;
  }
  if (tid == 0) s_nalive = nrays;
  __syncthreads();
  // the neutral densities of a tile are loaded one tile ahead (during the serial product of the tile
  // before), so the chain: loads -> exp -> barrier -> serial product -> barrier -> stores does not leave the
  // memory pipe idle
  Real sv[RS_PASS], sn[RS_PASS];
#pragma unroll
  for (int q = 0; q < RS_PASS; q++) {
    const int r = rsub + RSTEP*q;
    sv[q] = 0.0; sn[q] = 0.0;
    if (g.is + col <= g.ie && r < nrays) sv[q] = Uq(g,5)[(long)k*g.sK + (long)(j0 + r)*g.sJ + g.is + col];
  }
  for (int c0 = g.is; c0 <= g.ie; c0 += RS_CH) {
    const int i = c0 + col;
    const bool incol = (i <= g.ie);
    const int ncol = min(RS_CH, g.ie - c0 + 1);
    const bool alive = (s_nalive > 0);                        // block-uniform
    Real nH[RS_PASS];
    if (alive) {
#pragma unroll
      for (int q = 0; q < RS_PASS; q++) {
        const int r = rsub + RSTEP*q;
        nH[q] = 1.0;
        if (incol && r < nrays) {
          const Real n_H = sv[q] * p.inv_mH;                   // ionradplane_3d.c:281
          const Real tau = p.sigma_ph * n_H * g.dx[0];        // :294
          nH[q] = n_H;
          s_etau[r][col] = exp(-tau);
        }
      }
    }
    if ((alive || RATES) && c0 + RS_CH <= g.ie) {             // next tile (without RATES wasted once, when every ray dies in this one)
#pragma unroll
      for (int q = 0; q < RS_PASS; q++) {
        const int r = rsub + RSTEP*q;
        if (i + RS_CH <= g.ie && r < nrays) sn[q] = Uq(g,5)[(long)k*g.sK + (long)(j0 + r)*g.sJ + i + RS_CH];
      }
    }
    __syncthreads();
    if (alive && tid < nrays) {                               // one lane per ray: serial product
      const int r = tid;
      Real flux = s_flux[r]; int dead = s_dead[r];
      // flux/(f0 + 1e-12) < MINFLUXFRAC (:299-300) as flux < MINFLUXFRAC*(f0 + 1e-12): no division in the
      // serial chain (a ray whose ratio rounds exactly onto the threshold may be cut one zone apart)
      const Real cut = MINFLUXFRAC*(s_f0[r] + 1e-12);
      for (int cc = 0; cc < ncol; cc++) {
        if (dead) { s_fin[r][cc] = 0.0; continue; }
        s_fin[r][cc] = flux;                                  // EdgeFlux[..][i-s] = flux  (:279)
        flux *= s_etau[r][cc];                                // :298
        if (flux < cut) { dead = 1; flux = 0.0; atomicSub(&s_nalive, 1); }   // :300-306
      }
      s_flux[r] = flux; s_dead[r] = dead;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RS_PASS; q++) {
      const int r = rsub + RSTEP*q;
      if (incol && r < nrays) {
        const long m = (long)k*g.sK + (long)(j0 + r)*g.sJ + i;
        Real kph = 0.0, fin = 0.0;
        if (alive) {
          fin = s_fin[r][col];
          kph = fin * (1.0 - s_etau[r][col]) / (nH[q]*g.dx[0]);   // :296
        }
        g.ph_rate[m] = kph;                                   // ph_rate_init + "+=" (:55, :297)
        g.edgeflux[(long)(k - g.ks)*efrow + (long)(j0 + r - g.js)*efp + (i - g.is)] = fin;
        if (RATES) {
          Cell c; c.d = Uq(g,0)[m]; c.ke = g.kin[m]; c.E = Uq(g,4)[m]; c.s = sv[q];
          int2 sg = g.sign[m];
          const int2 sg0 = sg;
          Real dt_chem, dt_therm;
          rates_cell(c, kph, sg, p, g.Gamma_1, sc, dt_chem, dt_therm);
          if (sg.x != sg0.x || sg.y != sg0.y) g.sign[m] = sg;
          dt_chem_min = rmin(dt_chem_min, dt_chem);
          dt_therm_min = rmin(dt_therm_min, dt_therm);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RS_PASS; q++) sv[q] = sn[q];
    __syncthreads();
  }
  if (tid < nrays)                                            // :308
    g.edgeflux[(long)(k - g.ks)*efrow + (long)(j0 + tid - g.js)*efp + g.Nx1] = s_dead[tid] ? 0.0 : s_flux[tid];
  if (RATES) {
    block_min_to(&sc->dt_chem, dt_chem_min, red);
    block_min_to(&sc->dt_therm, dt_therm_min, red);
  }
}

// ---- rates: compute_chem_rates :334-394 + compute_therm_rates :460-557.  Streams d, ke, E, s0,
// ph_rate (+ the int2 sign bookkeeping), stores nothing but the sign changes, and reduces the two
// time-step limits (grid-stride; 2 atomics per block) --------------------------------------------
__global__ void __launch_bounds__(256)
k_ion_rates(DevGrid g, IonPar p, DevScalars *sc)
{
  __shared__ Real red[256];
  long m;
  Real dt_chem_min = DBL_MAX, dt_therm_min = DBL_MAX;
  // software-pipelined like k_ion_update: the next zone's operands are in flight during this zone's rates
  const long stride = (long)gridDim.x*blockDim.x;
  long lin = (long)blockIdx.x*blockDim.x + threadIdx.x, m_n = 0;
  bool have = active_cell(g, lin, m);
  Real n_d = 0, n_ke = 0, n_E = 0, n_s = 0, n_ph = 0; int2 n_sg = make_int2(0, 0);
  if (have) { n_d = Uq(g,0)[m]; n_ke = g.kin[m]; n_E = Uq(g,4)[m]; n_s = Uq(g,5)[m]; n_ph = g.ph_rate[m]; n_sg = g.sign[m]; }
  for (; have; lin += stride, m = m_n) {
    Cell c; c.d = n_d; c.ke = n_ke; c.E = n_E; c.s = n_s;
    const Real ph = n_ph;
    int2 sg = n_sg;
    have = active_cell(g, lin + stride, m_n);
    if (have) { n_d = Uq(g,0)[m_n]; n_ke = g.kin[m_n]; n_E = Uq(g,4)[m_n]; n_s = Uq(g,5)[m_n]; n_ph = g.ph_rate[m_n]; n_sg = g.sign[m_n]; }
    const int2 sg0 = sg;
    Real dt_chem, dt_therm;
    rates_cell(c, ph, sg, p, g.Gamma_1, sc, dt_chem, dt_therm);
    if (sg.x != sg0.x || sg.y != sg0.y) g.sign[m] = sg;
    dt_chem_min = rmin(dt_chem_min, dt_chem);
    dt_therm_min = rmin(dt_therm_min, dt_therm);
  }
  // dt values are > 0 (or +DBL_MAX): their bit patterns order like the values
  block_min_to(&sc->dt_chem, dt_chem_min, red);
  block_min_to(&sc->dt_therm, dt_therm_min, red);
}

// bvals_ionrad.c:63 / outflow_flux_ix1 :308: EdgeFlux[k][j][0] = flux_i for k<=Nx3, j<=Nx2
__global__ void k_edgeflux_bc(DevGrid g, Real flux_i)
{
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  const long n = (long)(g.Nx2 + 1)*(g.Nx3 + 1);
  if (lin >= n) return;
  g.edgeflux[lin*(long)(g.Nx1 + 1)] = flux_i;
}

// ---- update + floors + range check + hydro CFL --------------------------------------------------
__global__ void __launch_bounds__(256)
k_ion_update(DevGrid g, IonPar p, Real dt_arg, DevScalars *sc, int dt_from_sc)
{
  const Real dt = dt_from_sc ? sc->dt_sel : dt_arg;
  __shared__ Real red[256];
  __shared__ unsigned int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  long m;
  Real dti = 0.0;
  unsigned int mycnt = 0;
  const Real *ke_f = g.kin, *vmx_f = g.vmax;
  // grid-stride: a capped grid keeps the number of same-address atomics at 2 per block (one
  // word sustains only ~90 atomics/us on MI355X)
  // software-pipelined: the next zone's operands are in flight while this zone's rates are evaluated
  // (a log and three exp between the loads and the stores otherwise leave the memory pipe idle)
  const long stride = (long)gridDim.x*blockDim.x;
  long lin = (long)blockIdx.x*blockDim.x + threadIdx.x, m_n = 0;
  bool have = active_cell(g, lin, m);
  Real n_d = 0, n_ke = 0, n_E = 0, n_s = 0, n_ph = 0, n_e0 = 0; int n_sy = 0;
  if (have) { n_d = Uq(g,0)[m]; n_ke = ke_f[m]; n_E = Uq(g,4)[m]; n_s = Uq(g,5)[m]; n_ph = g.ph_rate[m]; n_sy = g.sign[m].y; n_e0 = g.e_init[m]; }
  for (; have; lin += stride, m = m_n) {
    Cell c; c.d = n_d; c.ke = n_ke; c.E = n_E; c.s = n_s;
    const Real E0 = c.E, s0 = c.s;
    const Real ph = n_ph, e_init0 = n_e0;
    const int sign_y = n_sy;
    have = active_cell(g, lin + stride, m_n);
    if (have) { n_d = Uq(g,0)[m_n]; n_ke = ke_f[m_n]; n_E = Uq(g,4)[m_n]; n_s = Uq(g,5)[m_n]; n_ph = g.ph_rate[m_n]; n_sy = g.sign[m_n].y; n_e0 = g.e_init[m_n]; }
    {   // the rates ion_rates derived its time-step limits from, re-evaluated (same code path)
      const IonQ q0 = ion_q(c, p, g.Gamma_1);
      Real lnT; bool cold;
      const Real nHdot = damp(chem_rate(q0, ph, p, lnT, cold), sign_y);
      const Real d_nlim = neutral_lim(c.d, p);
      const bool skip = cold || ((nHdot < 0) && (c.s < 1.0001*d_nlim));
      const Real edot = skip ? 0.0 : therm_rate(q0, ph, lnT, p);
      if ((nHdot > 0) || (c.s > 1.0001*d_nlim)) {          // ionization_update :577-585
        c.E += edot * dt;
        c.s += nHdot * dt * p.m_H;
      }
    }
    IonQ q; bool floored;
    floors(c, p, g.Gamma_1, q, floored);
    if (c.E != E0) Uq(g,4)[m] = c.E;
    if (c.s != s0) Uq(g,5)[m] = c.s;
    if (floored) q = ion_q(c, p, g.Gamma_1);               // otherwise what floors() derived still holds
    // check_range :223-264.  a/b >= L is tested as a >= L*b when both are positive (the common
    // case; ratios sit near 1, limits at 11), by division otherwise.
    {
      bool counted = false;
      const bool dtype = (q.n_H > 0.0) ? (ph > 2.0*CION*p.min_area*q.n_H) : (ph / (p.min_area * q.n_H) > 2.0*CION);
      if (!dtype) {
        // e_th_init (ionrad_3d.c:176) = e_init - ke with ke frozen over the ion step: not stored
        const Real e0 = e_init0, eth0 = e0 - c.ke;
        const Real L1 = 1 + p.max_de_therm_step, L2 = 1 + p.max_de_step, L3 = 1 + p.max_dx_step;
        if (ratio_ge(q.e_th, eth0, L1) || ratio_ge(eth0, q.e_th, L1)) counted = true;
        else if ((p.max_de_step > 0) && (ratio_ge(c.E, e0, L2) || ratio_ge(e0, c.E, L2))) counted = true;
        else if (p.max_dx_step > 0) {
          const Real x0 = g.x_init[m];
          if (ratio_ge(q.x, x0, L3) || ratio_ge(x0, q.x, L3)) counted = true;
        }
      }
      if (counted) mycnt++;
    }
    // compute_dt_hydro :609-660 (only compared against dt_done, never used as a time step)
    {
      const Real pp = rmax(g.Gamma_1*(c.E - c.ke), AA_TINY);
      const Real a = sqrt(g.Gamma*pp*q.di);
      Real t3;
      if (p.iso) t3 = (vmx_f[m] + a)*p.inv_dx[0];
      else {                                               // anisotropic zones: per-direction speeds
        const Real v1 = fabs(Uq(g,1)[m]*q.di), v2 = fabs(Uq(g,2)[m]*q.di), v3 = fabs(Uq(g,3)[m]*q.di);
        t3 = rmax(rmax((v1 + a)*p.inv_dx[0], (v2 + a)*p.inv_dx[1]), (v3 + a)*p.inv_dx[2]);
      }
      if (t3 == t3) dti = rmax(dti, t3);
    }
  }
  if (mycnt) atomicAdd(&cnt, mycnt);
  red[threadIdx.x] = dti;
  __syncthreads();
  for (int s = blockDim.x/2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = rmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicMax(&sc->max_dti, (unsigned long long)__double_as_longlong(red[0]));
    if (cnt) atomicAdd(&sc->cellcount, (unsigned long long)cnt);
  }
}

// =============================================================================================
static inline unsigned nblk(long n, int b) { return (unsigned)((n + b - 1)/b); }

void launch_ion_begin(const DevGrid &g, const IonPar &p, hipStream_t st)
{ const long n = (long)g.Nx1*g.Nx2*g.Nx3; hipLaunchKernelGGL(k_ion_begin, dim3(nblk(n, 256)), dim3(256), 0, st, g, p); }
void launch_ray_sweep(const DevGrid &g, const IonPar &p, Real flux0, bool from_edgeflux, hipStream_t st)
{ hipLaunchKernelGGL(k_ray_sweep<false>, dim3((g.Nx2 + RS_RAYS - 1)/RS_RAYS, g.Nx3), dim3(256), 0, st, g, p, flux0, from_edgeflux ? 1 : 0, (DevScalars*)nullptr); }
// ray sweep + the rates of every zone + the two time-step limits (what launch_ray_sweep + launch_ion_rates do)
void launch_ray_sweep_rates(const DevGrid &g, const IonPar &p, Real flux0, bool from_edgeflux, DevScalars *sc, hipStream_t st)
{ hipLaunchKernelGGL(k_ray_sweep<true>, dim3((g.Nx2 + RS_RAYS - 1)/RS_RAYS, g.Nx3), dim3(256), 0, st, g, p, flux0, from_edgeflux ? 1 : 0, sc); }
void launch_ion_rates(const DevGrid &g, const IonPar &p, DevScalars *sc, hipStream_t st)
{ const long n = (long)g.Nx1*g.Nx2*g.Nx3; const unsigned nb = reduce_blocks(n);
  hipLaunchKernelGGL(k_ion_rates, dim3(nb), dim3(256), 0, st, g, p, sc); }
void launch_ion_update(const DevGrid &g, const IonPar &p, Real dt, DevScalars *sc, hipStream_t st)
{ const long n = (long)g.Nx1*g.Nx2*g.Nx3; const unsigned nb = reduce_blocks(n);
  hipLaunchKernelGGL(k_ion_update, dim3(nb), dim3(256), 0, st, g, p, dt, sc, 0); }
void launch_ion_update_sel(const DevGrid &g, const IonPar &p, DevScalars *sc, hipStream_t st)
{ const long n = (long)g.Nx1*g.Nx2*g.Nx3; const unsigned nb = reduce_blocks(n);
  hipLaunchKernelGGL(k_ion_update, dim3(nb), dim3(256), 0, st, g, p, 0.0, sc, 1); }

// ionrad_3d.c:941-963 for one sub-cycle, on the device: dt = MIN(dt_therm, dt_chem), cut back to what is
// left of the hydro step (root) or of the coarse time (refined level).  Also re-arms the reductions of the
// next round, so that a sub-cycle costs one read-back of the scalars instead of two plus two resets.
__global__ void k_ion_pick(DevScalars *sc, Real dt_done, Real dt_limit)
{
  const Real dt_chem = __longlong_as_double((long long)sc->dt_chem), dt_therm = __longlong_as_double((long long)sc->dt_therm);
  Real dt = (dt_therm < dt_chem) ? dt_therm : dt_chem;
  int hit = 0;
  if (dt_done + dt > dt_limit) { dt = dt_limit - dt_done; hit = 1; }
  sc->dt_sel = dt; sc->limit_hit = hit;
  sc->dt_chem_out = dt_chem; sc->dt_therm_out = dt_therm; sc->neg_out = sc->neg_dt_chem;
  sc->dt_chem = (unsigned long long)__double_as_longlong(DBL_MAX); sc->dt_therm = sc->dt_chem; sc->neg_dt_chem = 0;
  sc->max_dti = 0; sc->cellcount = 0;
}
void launch_ion_pick(DevScalars *sc, Real dt_done, Real dt_limit, hipStream_t st)
{ hipLaunchKernelGGL(k_ion_pick, dim3(1), dim3(1), 0, st, sc, dt_done, dt_limit); }
void launch_edgeflux_bc(const DevGrid &g, Real flux_i, hipStream_t st)
{ const long n = (long)(g.Nx2 + 1)*(g.Nx3 + 1); hipLaunchKernelGGL(k_edgeflux_bc, dim3(nblk(n, 256)), dim3(256), 0, st, g, flux_i); }

}  // namespace aa
