// ion_pass.hip -- one radiation sub-cycle of ion_radtransfer_3d (ionrad_3d.c:919-1012) as ONE streaming
// kernel for gfx950, with the optical depth accumulated along the rays by a wavefront prefix scan.
//
// The reference's sub-cycle n is   sweep(n) -> rates(n) -> [global MIN -> dt_n] -> update(n) -> floors ->
// [global SUM / MIN -> stop?].  The only true barrier is the reduction that yields dt_n, so the loop is cut
// THERE: pass n of this kernel applies update(n-1) (with dt_{n-1}, read from device memory) and, on the
// updated zones still in registers, runs sweep(n) and rates(n).  A zone's state is then read and written
// once per sub-cycle (~90 B/zone with the frozen fields) instead of once per phase (133 B in the two-kernel
// form of ion_kernels.hip; SURVEY 8d counts 64 B of compulsory traffic).
//
//   wave   = one ray (j,k) at a time; lane = zone along x1, 64 zones per tile, tiles marched in ray order
//   loads  = 512 contiguous bytes per field and wave, the next tile's operands in flight during this
//            tile's arithmetic (also across the end of a ray)
//   sweep  = exp(-tau) per lane, inclusive prefix PRODUCT across the wave (6 shuffle steps), times the flux
//            carried from the previous tile (a wave-uniform scalar); the cut-off "flux/(f0+1e-12) <
//            MINFLUXFRAC" (ionradplane_3d.c:299-306) is a ballot + find-first-set, after which the ray is
//            dead for the rest of the row (wave-uniform branch: no exp, no scan, zeros)
//   rates  = compute_chem_rates + compute_therm_rates of the zone, from registers
//   stores = E, s0 (where changed), the flux that entered the zone, the 2-byte sign word (where changed)
//   reductions: per wave by shuffles, per block through LDS, one record per block in HBM (no atomics);
//            k_ion_reduce folds the records, k_ion_pick2 turns the folded words of all ranks into dt_n.
//
// The sweep after a data-dependent stop (cell count out of range / dt_hydro < dt_done, known only after the
// update's reductions) has already been done when the host learns of the stop: it is speculative, so the
// incoming fluxes are double-buffered and GridS.EdgeFlux is filled from the buffer of the last sweep that
// counted (k_ion_finish).  ph_rate is not stored at all: the update re-derives it from the stored incoming
// flux and the zone's own neutral density (one more exp per zone instead of 16 B of traffic).
//
// Tried and dropped here: the per-zone logic without divergent branches (selects for the sign cases of the rates, product
// tests for check_range's quotients): 602 instead of 562 vector instructions per zone and 3.37 instead of 3.02 ms per
// pass -- the selects cost more than the skipped branches, and the extra lane masks spill scalar registers to scratch.
//
// Multiplication order: the reference multiplies the flux zone by zone; the scan multiplies in a tree
// (differences ~1e-16, below those of device exp vs glibc exp, which no implementation can avoid).  The
// tile kernel of ion_kernels.hip keeps the serial order and remains the path for short rays (Nx1 < 64).
#include "ion_dev.h"
#include "ion_tables.h"

namespace aa {

AA_DEV unsigned short sg_pack(int2 sg) { const int c = sg.y > 16383 ? 16383 : sg.y; return (unsigned short)((sg.x + 1) | (c << 2)); }
AA_DEV int2 sg_unpack(unsigned short v) { return make_int2((int)(v & 3) - 1, (int)(v >> 2)); }

// inclusive prefix product over the 64 lanes of a wave
AA_DEV Real wave_incl_prod(Real v, int lane)
{
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const Real t = __shfl_up(v, off);
    if (lane >= off) v *= t;
  }
  return v;
}


// ---- per-zone arithmetic of this kernel.  Same expressions as ion_dev.h (ionrad_3d.c:82-101, :334-394, :460-557) with
// the FP64 divisions -- 13 per zone and pass, ~15 instructions each, a third of them quarter rate -- cut down: 1/d
// once per zone (d is frozen during the ion step), x = n_e/(n_H + n_H+) as n_e * m_H/d, the remaining quotients
// through a reciprocal refined to ~1 ulp (v_rcp_f64 + two Newton steps; true division outside the safe exponent
// range).  The ion step is not bit-comparable with the CPU anyway (device exp/log); these stay at 1e-16.
AA_DEV Real frcp(Real x)
{
#ifdef STUB_RCP
  return x;
#endif
  const Real ax = fabs(x);
  if (ax > 1.0e-280 && ax < 1.0e280) {
    Real r = __builtin_amdgcn_rcp(x);
    Real e = fma(-x, r, 1.0); r = fma(e, r, r);
    e = fma(-x, r, 1.0); r = fma(e, r, r);
    return r;
  }
  return 1.0/x;
}

// 1/x for operands that are positive normal numbers by construction (densities, number densities, mean particle
// mass); for a temperature: every T below the floor -- zero, negative -- only ever selects the floor branches
AA_DEV Real frcp_n(Real x)
{
#ifdef STUB_RCP
  return x;
#endif
  Real r = __builtin_amdgcn_rcp(x);
  Real e = fma(-x, r, 1.0); r = fma(e, r, r);
  e = fma(-x, r, 1.0); r = fma(e, r, r);
  return r;
}
// sqrt(x) for x >= AA_TINY: x * rsqrt(x) with two Newton steps on the reciprocal root (v_sqrt_f64's expansion is 22 instructions)
AA_DEV Real fsqrt_n(Real x)
{
  Real r = __builtin_amdgcn_rsq(x);
  Real h = 0.5*r, e = fma(-x*r, h, 0.5);        // e = (1 - x r^2)/2
  r = fma(r, e, r);
  h = 0.5*r; e = fma(-x*r, h, 0.5);
  r = fma(r, e, r);
  const Real y = x*r;
  return fma(fma(-y, y, x), 0.5*r, y);          // one correction of the root itself
}

// exp of N arguments at once, table-driven.  OCML's exp(double) is 42 vector instructions of which 19 only move the
// polynomial's coefficients into registers (FP64 operands cannot be literals), and a zone needs four exponentials per
// phase: here the N Horner chains run side by side and share each coefficient, and a 32-entry table of 2^(j/32) shortens
// them: k = rint(32 x / ln2) = 32 n + j, r = x - k ln2/32 (two-part constant: exact product for |k| < 2^21), |r| <= ln2/64,
// exp(x) = 2^n T[j] (1 + (e^r - 1)) with e^r - 1 by a degree-6 series (truncation 3e-18), one fma onto T[j], ldexp.
// 16 instructions + one LDS read per exponential.  |error| <= 1 ulp; NaN -> NaN; x <= -746 -> 0 (v_ldexp_f64 underflows cleanly).
template <int N>
AA_DEV void exp_n(const Real (&x)[N], Real (&y)[N], const Real *tE)
{
#ifdef STUB_EXP
  for (int k = 0; k < N; k++) y[k] = x[k]; return;
#endif
  Real kf[N], r[N], q[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    kf[k] = __builtin_rint(x[k]*46.166241308446828384);
    r[k] = fma(-kf[k], 6.93147180369123816490e-01/32.0, x[k]);
    r[k] = fma(-kf[k], 1.90821492927058770002e-10/32.0, r[k]);
    q[k] = fma(r[k], 1.0/720.0, 1.0/120.0);
  }
  const Real cf[4] = {1.0/24.0, 1.0/6.0, 0.5, 1.0};
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int k = 0; k < N; k++) q[k] = fma(q[k], r[k], cf[c]);
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int ki = (int)kf[k];
    const Real t = tE[ki & 31];
    y[k] = __builtin_ldexp(fma(t, q[k]*r[k], t), ki >> 5);
  }
}

// ln x for positive normal x (temperatures), table-driven: x = m 2^e with m in [1,2); the top 7 mantissa bits pick
// one of 128 intervals with centre c_j; r = (m - c_j)/c_j (the difference is exact, 1/c_j from the table), |r| <= 1/256;
// ln m = ln c_j + log1p(r) with a degree-7 series (truncation 2e-18 r); ln x = e ln2 + ln m with the two-part ln2.
// 24 vector instructions + two LDS reads (OCML's log: 98, double-double throughout; a 20-term atanh series without
// tables: 56).  |error| <= 2 ulp for |ln x| >= 0.5 (every temperature above the floor), <= 3e-16 absolute below that
// (x just under a power of two: e ln2 and ln m cancel); NaN -> NaN.
AA_DEV Real log_pos(Real x, const Real *tR, const Real *tL)
{
#ifdef STUB_LOG
  return x;
#endif
  const Real m = 2.0*__builtin_amdgcn_frexp_mant(x);
  const int e = __builtin_amdgcn_frexp_exp(x) - 1;
  int j = (int)((m - 1.0)*128.0);
  j = j < 0 ? 0 : (j > 127 ? 127 : j);
  const Real c = fma((Real)j + 0.5, 1.0/128.0, 1.0);
  const Real r = (m - c)*tR[j];
  Real q = fma(r, 1.0/7.0, -1.0/6.0);
  q = fma(q, r, 0.2); q = fma(q, r, -0.25); q = fma(q, r, 1.0/3.0); q = fma(q, r, -0.5);
  const Real lm = tL[j] + fma(r*r, q, r);
  const Real ef = (Real)e;
  return fma(ef, 6.93147180369123816490e-01, fma(ef, 1.90821492927058770002e-10, lm));
}

// everything of a zone that needs exp / log, in one batch: exp(-tau) of its neutral column (ionradplane_3d.c:294-295),
// the recombination coefficient 2.59e-13 (T/1e4)^-0.7 (or its floor value; ionrad_chemistry.c:111), and the factors
// T^-0.89 and exp(-118348/T) of the cooling rates (:137, :350); T^y = exp(y ln T) with one shared log
struct Therm { Real etau, rec, e89, elya, arg; bool cold; };
AA_DEV Therm zone_therm(const IonQ &q, Real tau, const IonPar &p, const Real *tR, const Real *tL, const Real *tE)
{
  Therm th;
  th.cold = (q.T < p.tfloor);
  const Real lnT = th.cold ? 0.0 : log_pos(q.T, tR, tL);
  th.arg = 118348*frcp_n(q.T);
  const Real x[4] = {-tau, -0.7*(lnT - 9.210340371976184), -0.89*lnT, -th.arg};
  Real y[4];
  exp_n<4>(x, y, tE);
  th.etau = y[0];
  th.rec = th.cold ? p.rec_floor : 2.59e-13*y[1];
  th.e89 = y[2]; th.elya = y[3];
  return th;
}
AA_DEV Real zone_chem(const IonQ &q, const Therm &th, Real ph, const IonPar &p)      // ionrad_3d.c:334-341, undamped
{ return th.rec * p.time_unit * q.n_e * q.n_Hplus - ph * q.n_H; }
AA_DEV Real zone_edot(const IonQ &q, const Therm &th, Real ph, const IonPar &p)      // :460-490
{
  const Real rcool = (q.T < 100.0) ? 0.0 : 6.11e-10*th.e89*KB_CHEM*q.T;
  const Real lya = (th.arg > 745.2) ? 0.0 : -7.5e-19*q.n_e*q.n_H*th.elya;
  return ph * p.e_gamma * q.n_H - rcool * p.time_unit * q.n_Hplus * q.n_e + lya * p.time_unit;
}

AA_DEV IonQ zone_q(const Cell &c, Real di, const IonPar &p, Real Gamma_1)
{
  IonQ q;
  q.n_H = c.s * p.inv_mH;
  q.n_Hplus = (c.d - c.s) * p.inv_mH;
  q.n_e = q.n_Hplus + c.d * p.aC14;
  q.x = q.n_e * (p.m_H * di);
  q.di = di;
  q.e_th = c.E - c.ke;
  q.muq = q.x*0.5*p.m_H+(1.0-q.x)*p.mu;
  q.T = Gamma_1 * (q.e_th * di) * q.muq * p.inv_kB;
  return q;
}

// apply_temp_floor + apply_neutral_floor (:70-156) given the derived quantities q of the zone as it is; returns
// whether E or s was touched (then q is stale)
AA_DEV bool zone_floors(Cell &c, const IonQ &q, const IonPar &p, Real Gamma_1)
{
  const Real E0 = c.E, s0 = c.s;
  if (q.T < p.tfloor) c.E = c.ke + (p.tfloor * p.k_B * frcp_n(q.muq * Gamma_1)) * c.d;
  if ((q.T > p.tceil) && (p.tceil > 0)) c.E = c.ke + (p.tceil * p.k_B * frcp_n(q.muq * Gamma_1)) * c.d;
  const Real d_nlim = neutral_lim(c.d, p);
  if (c.s < d_nlim) c.s = d_nlim; else if (c.s > c.d) c.s = c.d;
  return (c.E != E0) || (c.s != s0);
}

// ionization_update (:577-585) with the rates re-evaluated from (state, ph_rate); q0, th0 = zone_q / zone_therm of the state
AA_DEV void zone_update(Cell &c, const IonQ &q0, const Therm &th0, Real ph, int sign_count, Real dt, const IonPar &p)
{
  const Real nHdot = damp(zone_chem(q0, th0, ph, p), sign_count);
  const Real d_nlim = neutral_lim(c.d, p);
  const bool skip = th0.cold || ((nHdot < 0) && (c.s < 1.0001*d_nlim));
  const Real edot = skip ? 0.0 : zone_edot(q0, th0, ph, p);
  if ((nHdot > 0) || (c.s > 1.0001*d_nlim)) {
    c.E += edot * dt;
    c.s += nHdot * dt * p.m_H;
  }
}

// compute_chem_rates :334-394 + compute_therm_rates :460-557 of one zone whose derived quantities are iq, th
AA_DEV void zone_rates(const Cell &c, const IonQ &iq, const Therm &th, Real ph, int2 &sg, const IonPar &p, Real Gamma_1, bool &neg,
                       Real &dt_chem, Real &dt_therm)
{
  Real nHdot = zone_chem(iq, th, ph, p);
  if (nHdot < 0.0) {
    if (sg.x == 1) sg.y++; else if (sg.y > 0) sg.y--;
    sg.x = -1;
  } else if (nHdot > 0.0) {
    if (sg.x == -1) sg.y++; else if (sg.y > 0) sg.y--;
    sg.x = 1;
  } else { sg.x = 0; sg.y = 0; }
  nHdot = damp(nHdot, sg.y);
  const Real d_nlim = neutral_lim(c.d, p);
  Real dt1, dt2;
  if (nHdot == 0.0) { dt1 = dt2 = DBL_MAX; }
  else {
    const Real inv_n = frcp(nHdot);
    if (nHdot > 0.0) {
      dt1 = p.cx1 * iq.n_e * inv_n;               // max_dx_iter/(1+max_dx_iter) * n_e / nHdot
      dt2 = p.max_dx_iter * iq.n_H * inv_n;
    } else if (c.s > 1.0001*d_nlim) {
      dt1 = -p.max_dx_iter * iq.n_e * inv_n;
      dt2 = -p.cx1 * iq.n_H * inv_n;
    } else { dt1 = dt2 = DBL_MAX; }
  }
  dt_chem = (dt1 < dt2) ? dt1 : dt2;
  if (dt_chem < 0) { neg = true; dt_chem = DBL_MAX; }
  dt_therm = DBL_MAX;
  const bool skip = th.cold || ((nHdot < 0) && (c.s < 1.0001*d_nlim));
  if (!skip) {
    const Real edot = zone_edot(iq, th, ph, p);
    Real t1, t2; bool have = true;
    if (edot == 0.0) { t1 = t2 = DBL_MAX; }
    else {
      const Real inv_e = frcp(edot);
      if (edot > 0.0) {
        t1 = p.max_de_iter * c.E * inv_e;
        t2 = p.max_de_therm_iter * iq.e_th * inv_e;
      } else {
        const Real e_th_min = (p.tfloor * p.k_B * frcp_n(iq.muq * Gamma_1)) * c.d;
        const Real e_min = c.ke + e_th_min;
        if ((iq.e_th*p.ie1 < e_th_min) && (c.E*p.ie2 < e_min)) have = false;   // e/(1+max_de*_iter)
        t1 = -p.ce2 * c.E * inv_e;
        t2 = -p.ce1 * iq.e_th * inv_e;
      }
    }
    if (have) dt_therm = (t1 < t2) ? t1 : t2;
    if (!(dt_therm == dt_therm) || dt_therm < 0) dt_therm = DBL_MAX;
  }
}

AA_DEV Real zone_ph(Real fin, Real etau, Real n_H, Real inv_len)      // ionradplane_3d.c:296; 0 behind the cut-off
{ return (fin == 0.0) ? 0.0 : fin * (1.0 - etau) * frcp_n(n_H) * inv_len; }

struct Ops { Real d, ke, E, s, fp, e0, x0, vm; unsigned short sg; };

#ifndef AA_ION_PREFETCH
#define AA_ION_PREFETCH 0            /* 1: next tile's operands in flight during this tile's arithmetic (17 more VGPRs) */
#endif
#ifndef AA_ION_PREFETCH_HALF
#define AA_ION_PREFETCH_HALF 0       /* the same for the first / closing passes (which have the registers to spare) */
#endif
#ifndef AA_ION_PAR_LDS
#define AA_ION_PAR_LDS 1
#endif
// waves per SIMD the register budget is cut for.  With OCML's exp / log and IEEE divisions the full pass needed 3 waves
// plus a software prefetch of the next tile (157 VGPRs); with the table-driven exp / log it fits 4 waves without the
// prefetch (127 VGPRs, no spills) and the fourth wave hides more latency than the prefetch did: 2.63 against 2.98 ms
#ifndef AA_ION_PASS_WAVES
#define AA_ION_PASS_WAVES 4
#endif
#ifndef AA_ION_PASS_WAVES_HALF
#define AA_ION_PASS_WAVES_HALF 4     /* the first / closing passes (sweep + rates only, update only) */
#endif
// BEG: the first pass of an ion step also does the step's entry -- floors, save_energy_and_x, the frozen kinetic
// energy and max|v| (ionrad_3d.c:896-905; k_ion_begin16's arithmetic, bit for bit) -- on the zones it reads anyway
template <bool UPD, bool SWP, bool BEG>
__global__ void __launch_bounds__(256, (UPD && SWP) ? AA_ION_PASS_WAVES : AA_ION_PASS_WAVES_HALF)
k_ion_pass(DevGrid g, IonPar p_arg, Real flux0, int from_edgeflux, const DevScalars *sc, int cur, IonPart *part, int only_if_hit, Real spec_dt)
{
  // an updating pass is launched in both forms -- with and without the sweep of a further sub-cycle -- and the one
  // that does not match what the device picked (sc->limit_hit; the host has not read it yet) leaves at once
  if (UPD && only_if_hit >= 0 && (sc->limit_hit != 0) != (only_if_hit != 0)) return;
  // Speculation (spec_dt >= 0, the first pass of an ion step only): where the ion step does not cut the hydro step -- the
  // stationary regime of a long run -- the first sub-cycle's update uses the whole step, which is known before the
  // reduction that would confirm it.  The first pass therefore applies that update right behind its sweep and rates
  // (same operands, same expressions as the updating pass would use: the zone is still in registers) and counts its
  // check_range / dt_hydro operands; k_ion_pick2 then finds out whether the step was the one.  If so (spec_state 1)
  // the closing update pass finds nothing to do and leaves its predecessor's records in place, which are the ones
  // k_ion_pick2 books; if not (2), the next pass starts from the state the entry saved (e_init, s_init) and writes
  // every zone back.  One pass per step instead of two where the radiation is quiet.
  const bool spec = BEG && spec_dt >= 0.0;
  const int spec_state = UPD ? sc->spec_state : 0;
  if (UPD && !SWP && spec_state == 1) return;
  const bool from_init = UPD && spec_state == 2;
  // ~35 FP64 parameters on top of a dozen field pointers do not fit the 102 scalar registers of a wave (the first
  // build spilled 58 of them, partly to scratch): the parameter block lives in LDS and is re-read per tile through a
  // pointer the compiler cannot see through, i.e. as short-lived vector registers next to their uses
  __shared__ IonPar s_par;
  __shared__ Real s_logR[128], s_logL[128], s_expT[32];          // tables of log_pos and exp_n
  if (threadIdx.x == 0) s_par = p_arg;
  if (threadIdx.x < 128) { s_logR[threadIdx.x] = c_log_R[threadIdx.x]; s_logL[threadIdx.x] = c_log_L[threadIdx.x]; }
  if (threadIdx.x < 32) s_expT[threadIdx.x] = c_exp_T[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long nrays = (long)g.Nx2*g.Nx3;
  const long nwaves = (long)gridDim.x*4;        // (rays per Grid stay below 2^31)
  const int ntile = (g.Nx1 + 63) >> 6;
  const Real dt = UPD ? sc->dt_sel : 0.0;
  constexpr bool sweep = SWP;     // (whether a sweep is wanted after an update is settled by which of the two launches stays)
  const Real *fprev = g.fin[cur];
  Real *fnext = g.fin[cur ^ 1];
  const long efp = (long)(g.Nx1 + 1), efrow = (long)(g.Nx2 + 1)*efp;
  const Real dx1 = g.dx[0];
  const Real iso_inv_dx = p_arg.inv_dx[0];
  Real dt_chem_min = DBL_MAX, dt_therm_min = DBL_MAX, dti = 0.0;
  unsigned int cnt = 0;
  bool neg = false;

  // rows: ray -> (j,k) once per ray (wave-uniform 32-bit arithmetic)
  auto row_of = [&](unsigned ray_) -> long {
    const unsigned kk = ray_ / (unsigned)g.Nx2, jj = ray_ - kk*(unsigned)g.Nx2;
    return (long)(g.ks + (int)kk)*g.sK + (long)(g.js + (int)jj)*g.sJ + g.is;
  };
  auto load = [&](long row_, int t_, Ops &o, long &m_, bool &in_) {
    const int ii = 64*t_ + lane;
    in_ = (ii < g.Nx1);
    m_ = row_ + ii;
    // lanes past the end of the row take part in the shuffles with harmless operands
    o.d = 1.0; o.ke = 0.0; o.E = 1.0; o.s = 1.0; o.fp = 0.0; o.e0 = 1.0; o.x0 = 0.5; o.vm = 0.0; o.sg = 1;
    if (in_) {
      o.d = Uq(g,0)[m_]; o.E = Uq(g,4)[m_]; o.s = Uq(g,5)[m_];
      if (BEG) { o.fp = Uq(g,1)[m_]; o.e0 = Uq(g,2)[m_]; o.x0 = Uq(g,3)[m_]; }      // the momenta, in the slots UPD would use
      else { o.ke = g.kin[m_]; o.sg = g.sg16[m_]; }
      if (UPD) { o.fp = fprev[m_]; o.e0 = g.e_init[m_]; o.x0 = g.x_init[m_]; o.vm = g.vmax[m_]; }
      if (from_init) { o.E = o.e0; o.s = g.s_init[m_]; }      // (U holds the update that turned out too long)
    }
  };

  unsigned ray = blockIdx.x*4u + (unsigned)wv;
  const unsigned nrays_u = (unsigned)nrays, nwaves_u = (unsigned)nwaves;
  int t = 0;
  bool have = ray < nrays_u;
  Ops o; long m = 0, row = 0; bool in = false;
  if (have) { row = row_of(ray); load(row, 0, o, m, in); }
  Real carry = 0.0, cut = 0.0;
  bool dead = false;
  while (have) {
    unsigned nray = ray; int nt = t + 1; long nrow = row;
    bool nhave = true;
    if (nt == ntile) { nt = 0; nray = ray + nwaves_u; nhave = nray < nrays_u; if (nhave) nrow = row_of(nray); }
    Ops no; long nm = 0; bool nin = false;
    constexpr bool PF = (UPD && SWP) ? (AA_ION_PREFETCH != 0) : (AA_ION_PREFETCH_HALF != 0);
    if (PF && nhave) load(nrow, nt, no, nm, nin);            // in flight during this tile's arithmetic
#if AA_ION_PAR_LDS
#define PAR_HERE(name) int name##_off = 0; asm volatile("" : "+v"(name##_off)); \
                       const IonPar &name = *(const IonPar*)((const char*)&s_par + name##_off)
#else
#define PAR_HERE(name) const IonPar &name = p_arg
#endif

    if (sweep && t == 0) {                                    // ionradplane_3d.c:262-271: flux entering the ray
      Real f0 = flux0;
      if (from_edgeflux) {
        const unsigned kk = ray / (unsigned)g.Nx2, jj = ray - kk*(unsigned)g.Nx2;
        f0 = g.edgeflux[(long)kk*efrow + (long)jj*efp];
      }
      carry = f0; cut = MINFLUXFRAC*(f0 + 1e-12); dead = false;
    }
    Cell c; c.d = o.d; c.ke = o.ke; c.E = o.E; c.s = o.s;
    const Real E0 = c.E, s0 = c.s;
    int2 sg = sg_unpack(o.sg);
    bool sg_new = false;
    Real e_here = 0.0, x_here = 0.0, vm_here = 0.0;          // BEG: what the entry stores as e_init, x_init, vmax
    if (BEG) {
      PAR_HERE(p);
      const Real M1 = o.fp, M2 = o.e0, M3 = o.x0, die = 1.0/c.d;
      c.ke = 0.5 * (M1*M1 + M2*M2 + M3*M3) * die;
      IonQ qb; bool floored;
      floors(c, p, g.Gamma_1, qb, floored);
      if (floored) qb = ion_q(c, p, g.Gamma_1);
      if (in) {
        if (!spec) {
          if (c.E != E0) Uq(g,4)[m] = c.E;
          if (c.s != s0) Uq(g,5)[m] = c.s;
        } else g.s_init[m] = c.s;
        g.e_init[m] = c.E; g.x_init[m] = qb.x; g.kin[m] = c.ke;
        g.vmax[m] = rmax(rmax(fabs(M1*die), fabs(M2*die)), fabs(M3*die));
      }
      e_here = c.E; x_here = qb.x; vm_here = rmax(rmax(fabs(M1*die), fabs(M2*die)), fabs(M3*die));
      sg = make_int2(0, 0); sg_new = true;
    }
    const Real di = frcp_n(c.d);
    IonQ q; Therm th;
    { PAR_HERE(p); q = zone_q(c, di, p, g.Gamma_1); th = zone_therm(q, p.sigma_ph * q.n_H * dx1, p, s_logR, s_logL, s_expT); }
    if (UPD) {
      // the photoionization rate the previous sweep gave this zone: same expression, same operands
      Real php;
      { PAR_HERE(p);
        php = zone_ph(o.fp, th.etau, q.n_H, iso_inv_dx);
        zone_update(c, q, th, php, sg.y, dt, p); }
      { PAR_HERE(p);
        q = zone_q(c, di, p, g.Gamma_1);
        if (zone_floors(c, q, p, g.Gamma_1)) q = zone_q(c, di, p, g.Gamma_1);
        if (sweep) th = zone_therm(q, p.sigma_ph * q.n_H * dx1, p, s_logR, s_logL, s_expT); }
      PAR_HERE(p);
      if (in) {
        if (out_of_range(c, q, php, o.e0, o.x0, p)) cnt++;
        // compute_dt_hydro :609-660 (only compared against dt_done, never used as a time step)
        const Real pp = rmax(g.Gamma_1*(c.E - c.ke), AA_TINY);
        const Real a = fsqrt_n(g.Gamma*pp*di);
        Real t3;
        if (p.iso) t3 = (o.vm + a)*p.inv_dx[0];
        else {
          const Real v1 = fabs(Uq(g,1)[m]*di), v2 = fabs(Uq(g,2)[m]*di), v3 = fabs(Uq(g,3)[m]*di);
          t3 = rmax(rmax((v1 + a)*p.inv_dx[0], (v2 + a)*p.inv_dx[1]), (v3 + a)*p.inv_dx[2]);
        }
        if (t3 == t3) dti = rmax(dti, t3);
      }
    }
    Real fin = 0.0;
    if (sweep) {
      Real et = 1.0, ph = 0.0;
      PAR_HERE(p);
      if (!dead) {                                            // wave-uniform
        if (in) et = th.etau;                                 // ionradplane_3d.c:281, :294-295
        const Real P = wave_incl_prod(et, lane);
        const Real Fout = carry*P;                            // flux leaving the zone (:298)
        Real Fin = __shfl_up(Fout, 1);
        if (lane == 0) Fin = carry;
        const unsigned long long cm = __ballot(in && (Fout < cut));   // :299-300
        const int first = cm ? (__ffsll((long long)cm) - 1) : 64;
        if (lane <= first) fin = Fin;                         // EdgeFlux[..][i-s] = flux (:279); 0 behind the cut (:303)
        ph = zone_ph(fin, et, q.n_H, iso_inv_dx);             // :296
        if (cm) { dead = true; carry = 0.0; } else carry = __shfl(Fout, 63);
      }
      if (in) {
        const int2 sg0 = sg;
        Real dtc, dtt;
        zone_rates(c, q, th, ph, sg, p, g.Gamma_1, neg, dtc, dtt);
        dt_chem_min = rmin(dt_chem_min, dtc);
        dt_therm_min = rmin(dt_therm_min, dtt);
        fnext[m] = fin;
        if (sg_new || sg.x != sg0.x || sg.y != sg0.y) g.sg16[m] = sg_pack(sg);
      }
      if (BEG && spec) {        // the update of the first sub-cycle with the whole step, as the UPD branch above would do it next pass
        const Real e0 = e_here, x0 = x_here, vm = vm_here;    // = e_init, x_init, vmax of this zone (BEG keeps the momenta in o.fp, o.e0, o.x0)
        zone_update(c, q, th, ph, sg.y, spec_dt, p);
        q = zone_q(c, di, p, g.Gamma_1);
        if (zone_floors(c, q, p, g.Gamma_1)) q = zone_q(c, di, p, g.Gamma_1);
        if (in) {
          if (out_of_range(c, q, ph, e0, x0, p)) cnt++;
          const Real pp = rmax(g.Gamma_1*(c.E - c.ke), AA_TINY);
          const Real a = fsqrt_n(g.Gamma*pp*di);
          Real t3;
          if (p.iso) t3 = (vm + a)*p.inv_dx[0];
          else {
            const Real v1 = fabs(o.fp*di), v2 = fabs(o.e0*di), v3 = fabs(o.x0*di);
            t3 = rmax(rmax((v1 + a)*p.inv_dx[0], (v2 + a)*p.inv_dx[1]), (v3 + a)*p.inv_dx[2]);
          }
          if (t3 == t3) dti = rmax(dti, t3);
          if (c.E != E0) Uq(g,4)[m] = c.E;
          if (c.s != s0) Uq(g,5)[m] = c.s;
        }
      }
      if (t == ntile - 1 && lane == 0) g.raylast[(long)(cur ^ 1)*nrays + ray] = dead ? 0.0 : carry;   // :308
    }
    if (UPD && in) {
      if (from_init || c.E != E0) Uq(g,4)[m] = c.E;
      if (from_init || c.s != s0) Uq(g,5)[m] = c.s;
    }
    if (!PF && nhave) load(nrow, nt, no, nm, nin);
    ray = nray; t = nt; have = nhave; row = nrow; o = no; m = nm; in = nin;
  }

  // ---- reductions: wave (shuffles) -> block (LDS) -> one record per block ----
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dt_chem_min = rmin(dt_chem_min, __shfl_xor(dt_chem_min, off));
    dt_therm_min = rmin(dt_therm_min, __shfl_xor(dt_therm_min, off));
    dti = rmax(dti, __shfl_xor(dti, off));
    cnt += __shfl_xor(cnt, off);
  }
  const unsigned long long negm = __ballot(neg);
  __shared__ Real r_c[4], r_t[4], r_d[4];
  __shared__ unsigned int r_n[4];
  __shared__ int r_g[4];
  if (lane == 0) { r_c[wv] = dt_chem_min; r_t[wv] = dt_therm_min; r_d[wv] = dti; r_n[wv] = cnt; r_g[wv] = negm ? 1 : 0; }
  __syncthreads();
  if (threadIdx.x == 0) {
    IonPart r;
    r.dt_chem = rmin(rmin(r_c[0], r_c[1]), rmin(r_c[2], r_c[3]));
    r.dt_therm = rmin(rmin(r_t[0], r_t[1]), rmin(r_t[2], r_t[3]));
    r.max_dti = rmax(rmax(r_d[0], r_d[1]), rmax(r_d[2], r_d[3]));
    r.cellcount = (Real)(r_n[0] + r_n[1] + r_n[2] + r_n[3]);
    r.neg = (r_g[0] | r_g[1] | r_g[2] | r_g[3]) ? 1.0 : 0.0;
    part[blockIdx.x] = r;
  }
}

// folds the per-block records of one pass into the five words a rank contributes to the sub-cycle's
// reduction: MIN dt_chem, MIN dt_therm (ionrad_3d.c:399, :554), MAX (|v|+a)/dx (:672), SUM cell count (:275),
// OR of the negative-dt_chem flag (:389)
__global__ void __launch_bounds__(256)
k_ion_reduce(const IonPart *part, int n, Real *words)
{
  __shared__ Real red[5][256];
  Real a = DBL_MAX, b = DBL_MAX, c = 0.0, d = 0.0, e = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const IonPart r = part[i];
    a = rmin(a, r.dt_chem); b = rmin(b, r.dt_therm); c = rmax(c, r.max_dti); d += r.cellcount; e = rmax(e, r.neg);
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c; red[3][threadIdx.x] = d; red[4][threadIdx.x] = e;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] = rmin(red[0][threadIdx.x], red[0][threadIdx.x + s]);
      red[1][threadIdx.x] = rmin(red[1][threadIdx.x], red[1][threadIdx.x + s]);
      red[2][threadIdx.x] = rmax(red[2][threadIdx.x], red[2][threadIdx.x + s]);
      red[3][threadIdx.x] += red[3][threadIdx.x + s];          // integers below 2^53: exact in any order
      red[4][threadIdx.x] = rmax(red[4][threadIdx.x], red[4][threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 8) words[threadIdx.x] = (threadIdx.x < 5) ? red[threadIdx.x][0] : 0.0;
}
AA_DEV void ion_pick_body(Real dt_chem, Real dt_therm, Real max_dti, Real count, Real neg, DevScalars *sc, int first, Real dt_limit, int spec_armed);
// ... and, where ONE rank reduces alone, the pick of k_ion_pick2 by the same launch (the same fold, the same arithmetic): a sub-cycle
// on a small Grid is a handful of launches and one read-back, and every launch less is ~5 us of 40
__global__ void __launch_bounds__(256)
k_ion_reduce_pick(const IonPart *part, int n, Real *words, DevScalars *sc, int first, Real dt_limit, int spec_armed, Mailbox *mb, unsigned long long seq)
{
  __shared__ Real red[5][256];
  Real a = DBL_MAX, b = DBL_MAX, c = 0.0, d = 0.0, e = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const IonPart r = part[i];
    a = rmin(a, r.dt_chem); b = rmin(b, r.dt_therm); c = rmax(c, r.max_dti); d += r.cellcount; e = rmax(e, r.neg);
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c; red[3][threadIdx.x] = d; red[4][threadIdx.x] = e;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] = rmin(red[0][threadIdx.x], red[0][threadIdx.x + s]);
      red[1][threadIdx.x] = rmin(red[1][threadIdx.x], red[1][threadIdx.x + s]);
      red[2][threadIdx.x] = rmax(red[2][threadIdx.x], red[2][threadIdx.x + s]);
      red[3][threadIdx.x] += red[3][threadIdx.x + s];
      red[4][threadIdx.x] = rmax(red[4][threadIdx.x], red[4][threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 8) words[threadIdx.x] = (threadIdx.x < 5) ? red[threadIdx.x][0] : 0.0;
  // (one rank: k_ion_pick2's fold over ranks is MIN / MAX / + of one operand with its neutral element: the operand itself)
  if (threadIdx.x == 0)
    ion_pick_body(rmin(DBL_MAX, red[0][0]), rmin(DBL_MAX, red[1][0]), rmax(0.0, red[2][0]), 0.0 + red[3][0], rmax(0.0, red[4][0]), sc, first, dt_limit, spec_armed);
  if (mb) {      // ... and the scalars straight into the host's mailbox (grid.h Mailbox, k_publish): the read-back that follows needs no launch
    __threadfence(); __syncthreads();
    constexpr int nw = (int)(sizeof(DevScalars)/sizeof(unsigned long long));
    const unsigned long long *src = (const unsigned long long*)sc;
    unsigned long long *dst = (unsigned long long*)&mb->s;
    if ((int)threadIdx.x < nw) dst[threadIdx.x] = __hip_atomic_load(src + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system(); __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&mb->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ionrad_3d.c:941-967 on the device, over the words of all ranks (AA_ION_WORDS doubles each; one rank: the
// Grid's own): dt = MIN(dt_therm, dt_chem) cut back to what is left of the hydro step (root) or of the
// coarse time (refined level) -- the time covered so far is kept here, on the device -- and, for the host's ONE
// read-back per sub-cycle, the step and the stop criteria's operands of the update the last pass applied
__global__ void k_ion_pick2(const Real *words, int nranks, DevScalars *sc, int first, Real dt_limit, int spec_armed)
{
  Real dt_chem = DBL_MAX, dt_therm = DBL_MAX, max_dti = 0.0, count = 0.0, neg = 0.0;
  for (int r = 0; r < nranks; r++) {
    const Real *w = words + (long)r*AA_ION_WORDS;
    dt_chem = rmin(dt_chem, w[0]); dt_therm = rmin(dt_therm, w[1]); max_dti = rmax(max_dti, w[2]); count += w[3]; neg = rmax(neg, w[4]);
  }
  ion_pick_body(dt_chem, dt_therm, max_dti, count, neg, sc, first, dt_limit, spec_armed);
}
AA_DEV void ion_pick_body(Real dt_chem, Real dt_therm, Real max_dti, Real count, Real neg, DevScalars *sc, int first, Real dt_limit, int spec_armed)
{
  // what belongs to the update the pass before this kernel applied (none after the first pass of an ion step)
  Real dt_done = 0.0;
  if (!first) {
    sc->dt_applied = sc->dt_sel; sc->hit_applied = sc->limit_hit; sc->neg_applied = sc->neg_out;
    dt_done = sc->dt_done + sc->dt_sel;                     // dt_done += dt (:967)
  }
  sc->dt_done = dt_done;
  sc->max_dti = (unsigned long long)__double_as_longlong(max_dti);
  sc->cellcount = (unsigned long long)count;
  // the step of the next update (:941-963)
  Real dt = (dt_therm < dt_chem) ? dt_therm : dt_chem;
  int hit = 0;
  if (dt_done + dt > dt_limit) { dt = dt_limit - dt_done; hit = 1; }
  sc->dt_sel = dt; sc->limit_hit = hit;
  sc->spec_state = (first && spec_armed) ? (hit ? 1 : 2) : 0;      // (see k_ion_pass)
  sc->dt_chem_out = dt_chem; sc->dt_therm_out = dt_therm; sc->neg_out = (neg != 0.0);
}

// GridS.EdgeFlux of the last sweep that counted: [k][j][0..Nx1-1] = the flux that entered each zone,
// [k][j][Nx1] = what left the row (ionradplane_3d.c:279, :308)
__global__ void __launch_bounds__(256)
k_ion_finish(DevGrid g, int cur)
{
  const long n1 = g.Nx1 + 1;
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  const long nrays = (long)g.Nx2*g.Nx3;
  if (lin >= nrays*n1) return;
  const long ray = lin / n1; const int ii = (int)(lin % n1);
  const int j = (int)(ray % g.Nx2), k = (int)(ray / g.Nx2);
  const long efrow = (long)(g.Nx2 + 1)*n1;
  Real v;
  if (ii < g.Nx1) v = g.fin[cur][(long)(k + g.ks)*g.sK + (long)(j + g.js)*g.sJ + g.is + ii];
  else v = g.raylast[(long)cur*nrays + ray];
  g.edgeflux[(long)k*efrow + (long)j*n1 + ii] = v;
}

// entry of the ion step for this path: floors + save_energy_and_x (:896-905, :162-196); freezes ke and
// max_d|v_d|; the sign word of every zone starts at (last_sign 0, sign_count 0)
__global__ void __launch_bounds__(256)
k_ion_begin16(DevGrid g, IonPar p)
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
  g.sg16[m] = 1;
  g.kin[m] = c.ke;
  g.vmax[m] = rmax(rmax(fabs(M1*di), fabs(M2*di)), fabs(M3*di));
}

// function-level test of this file's exp / log (tests/test_gpu_ion_pass.py): y = exp_n<4> of x (lanes of four), l = log_pos(|x|)
__global__ void k_test_explog(int n, const Real *x, Real *ye, Real *yl)
{
  __shared__ Real s_logR[128], s_logL[128], s_expT[32];
  if (threadIdx.x < 128) { s_logR[threadIdx.x] = c_log_R[threadIdx.x]; s_logL[threadIdx.x] = c_log_L[threadIdx.x]; }
  if (threadIdx.x < 32) s_expT[threadIdx.x] = c_exp_T[threadIdx.x];
  __syncthreads();
  const int i = (blockIdx.x*blockDim.x + threadIdx.x)*4;
  if (i + 3 >= n) return;
  const Real a[4] = {x[i], x[i + 1], x[i + 2], x[i + 3]};
  Real y[4];
  exp_n<4>(a, y, s_expT);
  for (int k = 0; k < 4; k++) { ye[i + k] = y[k]; yl[i + k] = log_pos(fabs(a[k]), s_logR, s_logL); }
}
void launch_test_explog(int n, const Real *x, Real *ye, Real *yl, hipStream_t st)
{ hipLaunchKernelGGL(k_test_explog, dim3((n/4 + 255)/256), dim3(256), 0, st, n, x, ye, yl); }

// =============================================================================================
static inline unsigned nblk(long n, int b) { return (unsigned)((n + b - 1)/b); }

int ion_pass_blocks(const HostGrid &g)
{
  const long nrays = (long)g.Nx2*g.Nx3;
  const long nb = (nrays + 3)/4, cap = g.cfg.ion_pass_cap;      // (AA_ION_PASS_BLOCKS as aa_create found it: grid.h LaunchCfg)
  return (int)(nb < cap ? nb : cap);
}

void launch_ion_begin16(const DevGrid &g, const IonPar &p, hipStream_t st)
{ const long n = (long)g.Nx1*g.Nx2*g.Nx3; hipLaunchKernelGGL(k_ion_begin16, dim3(nblk(n, 256)), dim3(256), 0, st, g, p); }

void launch_ion_reduce_pick(const HostGrid &g, const IonPart *part, Real *words, DevScalars *sc, int first, Real dt_limit, hipStream_t st, int spec_armed,
                            Mailbox *mb_dev, unsigned long long seq)
{ hipLaunchKernelGGL(k_ion_reduce_pick, dim3(1), dim3(256), 0, st, part, ion_pass_blocks(g), words, sc, first, dt_limit, spec_armed, mb_dev, seq); }
void launch_ion_pass(const HostGrid &g, const IonPar &p, bool update, bool sweep, bool begin, Real flux0, bool from_edgeflux,
                     const DevScalars *sc, int cur, IonPart *part, Real *words, hipStream_t st, Real spec_dt, bool reduce)
{
  const int nb = ion_pass_blocks(g);
  const dim3 grid(nb), blk(256);
  const int fe = from_edgeflux ? 1 : 0;
  if (update && sweep) {
    hipLaunchKernelGGL((k_ion_pass<true, true, false>), grid, blk, 0, st, g, p, flux0, fe, sc, cur, part, 0, -1.0);
    hipLaunchKernelGGL((k_ion_pass<true, false, false>), grid, blk, 0, st, g, p, flux0, fe, sc, cur, part, 1, -1.0);
  }
  else if (sweep && begin) hipLaunchKernelGGL((k_ion_pass<false, true, true>), grid, blk, 0, st, g, p, flux0, fe, sc, cur, part, -1, spec_dt);
  else if (sweep)      hipLaunchKernelGGL((k_ion_pass<false, true, false>), grid, blk, 0, st, g, p, flux0, fe, sc, cur, part, -1, -1.0);
  else                 hipLaunchKernelGGL((k_ion_pass<true, false, false>), grid, blk, 0, st, g, p, flux0, fe, sc, cur, part, -1, -1.0);
  if (reduce) hipLaunchKernelGGL(k_ion_reduce, dim3(1), dim3(256), 0, st, part, nb, words);
}
void launch_ion_pick2(const Real *words, int nranks, DevScalars *sc, int first, Real dt_limit, hipStream_t st, int spec_armed)
{ hipLaunchKernelGGL(k_ion_pick2, dim3(1), dim3(1), 0, st, words, nranks, sc, first, dt_limit, spec_armed); }
void launch_ion_finish(const DevGrid &g, int cur, hipStream_t st)
{ const long n = (long)g.Nx2*g.Nx3*(g.Nx1 + 1); hipLaunchKernelGGL(k_ion_finish, dim3(nblk(n, 256)), dim3(256), 0, st, g, cur); }

}  // namespace aa
