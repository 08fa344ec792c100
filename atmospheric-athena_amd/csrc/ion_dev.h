// ion_dev.h -- per-zone device functions of the ion-radiation step (ionradiation/ionrad_3d.c,
// ionrad_chemistry.c), shared by the kernels of ion_kernels.hip (tile sweep + separate update) and of
// ion_pass.hip (the one-kernel sub-cycle).
#pragma once
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
AA_DEV void rates_cell(const Cell &c, Real ph, int2 &sg, const IonPar &p, Real Gamma_1, bool &neg, Real &dt_chem, Real &dt_therm)
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
  if (dt_chem < 0) { neg = true; dt_chem = DBL_MAX; }
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

// ---- one zone of ionization_update (:577-585): the two rates compute_chem_rates / compute_therm_rates derived
// the time-step limits from are re-evaluated from (state, ph_rate) -- same code path, same bits -- and applied
AA_DEV void update_cell(Cell &c, Real ph, int sign_count, Real dt, const IonPar &p, Real Gamma_1)
{
  const IonQ q0 = ion_q(c, p, Gamma_1);
  Real lnT; bool cold;
  const Real nHdot = damp(chem_rate(q0, ph, p, lnT, cold), sign_count);
  const Real d_nlim = neutral_lim(c.d, p);
  const bool skip = cold || ((nHdot < 0) && (c.s < 1.0001*d_nlim));
  const Real edot = skip ? 0.0 : therm_rate(q0, ph, lnT, p);
  if ((nHdot > 0) || (c.s > 1.0001*d_nlim)) {          // :577-585
    c.E += edot * dt;
    c.s += nHdot * dt * p.m_H;
  }
}

// ---- one zone of check_range (:223-264); q = derived quantities of the updated, floored zone.  a/b >= L is
// tested as a >= L*b when both are positive (the common case; ratios sit near 1, limits at 11), by division
// otherwise.  e_th_init (ionrad_3d.c:176) = e_init - ke with ke frozen over the ion step: not stored.
AA_DEV bool out_of_range(const Cell &c, const IonQ &q, Real ph, Real e0, Real x0, const IonPar &p)
{
  const bool dtype = (q.n_H > 0.0) ? (ph > 2.0*CION*p.min_area*q.n_H) : (ph / (p.min_area * q.n_H) > 2.0*CION);
  if (dtype) return false;
  const Real eth0 = e0 - c.ke;
  const Real L1 = 1 + p.max_de_therm_step, L2 = 1 + p.max_de_step, L3 = 1 + p.max_dx_step;
  if (ratio_ge(q.e_th, eth0, L1) || ratio_ge(eth0, q.e_th, L1)) return true;
  if ((p.max_de_step > 0) && (ratio_ge(c.E, e0, L2) || ratio_ge(e0, c.E, L2))) return true;
  if (p.max_dx_step > 0) return ratio_ge(q.x, x0, L3) || ratio_ge(x0, q.x, L3);
  return false;
}

// ---- one zone of get_ph_rate_plane (ionradplane_3d.c:281-297): photoionization rate from the flux that
// enters the zone; zones behind the cut-off (flux 0) keep the 0 of ph_rate_init
AA_DEV Real ph_from_flux(Real fin, Real etau, Real n_H, Real cell_len)
{ return (fin == 0.0) ? 0.0 : fin * (1.0 - etau) / (n_H*cell_len); }

AA_DEV void rates_cell(const Cell &c, Real ph, int2 &sg, const IonPar &p, Real Gamma_1, DevScalars *sc, Real &dt_chem, Real &dt_therm)
{
  bool neg = false;
  rates_cell(c, ph, sg, p, Gamma_1, neg, dt_chem, dt_therm);
  if (neg) atomicExch(&sc->neg_dt_chem, 1);                  // ionrad_3d.c:389-391: fatal on the host
}

}  // namespace aa
