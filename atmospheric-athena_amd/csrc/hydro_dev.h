// hydro_dev.h -- per-cell / per-interface device arithmetic of the hydro hot path (gfx950).
//
// States are 6 doubles in the SWEEP frame: conserved (d,Mx,My,Mz,E,s0), primitive
// (d,Vx,Vy,Vz,P,r0) -- the reference's Cons1DS/Prim1DS field order (athena.h:148-188).
// Operation order follows the reference so that a build with -ffp-contract=off reproduces
// the CPU results bit for bit; the default build lets the compiler fuse multiply-adds.
// The dense 5x5 eigen-matrix products of the reference are written in their sparse form
// (esystem_prim.c:138-198, esystem_roe.c:151-214): every partial sum is unchanged.
#pragma once
#include <hip/hip_runtime.h>

namespace aa {

typedef double Real;
#define AA_TINY 1.0e-20
#define AA_DEV __device__ __forceinline__

// The reference's MAX/MIN macros (defs.h.in:146-151).  NOT fmax/fmin: with a NaN operand the
// ternary form is order-sensitive and the H-correction chain depends on that behaviour.
AA_DEV Real rmax(Real a, Real b) { return (a > b) ? a : b; }
AA_DEV Real rmin(Real a, Real b) { return (a < b) ? a : b; }
AA_DEV Real sqr(Real x) { return x*x; }

// ---- the default build (AA_FAST_DIV; the Makefile sets it for libathena_amd.so only).  An FP64 division is ~12
// instructions on CDNA4, five of them quarter rate, a square root ~22, and a first-pass face costs 18 + 4 of them --
// a third of the instructions of the VALU-bound sweeps.  Where AA_FAST_DIV is set, quotients with a positive normal
// denominator (densities, sound speeds and their squares) become products with a reciprocal / reciprocal square root
// refined by two Newton steps (~1 ulp), and several quotients by the same denominator share it.  Results move by a
// few 1e-16 per operation; libathena_amd_strict.so (-ffp-contract=off, no AA_FAST_DIV) keeps the reference's
// operations and stays bit-identical to the CPU.
#ifndef AA_FAST_DIV
#define AA_FAST_DIV 0
#endif
#ifndef AA_FD_C2P
#define AA_FD_C2P AA_FAST_DIV
#endif
#ifndef AA_FD_PLM
#define AA_FD_PLM AA_FAST_DIV
#endif
// P/(gamma - 1) in prim_to_cons as a product with the (loop-invariant) reciprocal: two divisions less per face; x2 / x3
// first passes -6 %, hydro chain 45.7 -> 45.1 ms (round 3, same-box ABAB at 512^3) -- and OFF: the conserved face states
// feed the Riemann solver's Roe -> HLLE switch (see AA_FD_ROE below), and with the energy rounded differently the
// 3-level blast of the drop-in test left the reference by 7.6e-7 in time after six steps (one flipped switch)
#ifndef AA_FD_P2C
#define AA_FD_P2C 0
#endif
// The Riemann solver keeps the reference's operations in every build: its switch to HLLE (negative density or pressure of
// an intermediate state, roe.c:256-286) is the one DISCONTINUOUS decision of the hydro step, and symmetric flows sit
// exactly on it -- with the reciprocal forms in flux_roe a 3-level blast differed from the reference by 2.6e-3 in 116
// zones after two steps (one flipped switch; 5e-16 without them).  Reconstruction and the variable conversion are
// continuous in their inputs: there a changed last bit stays a changed last bit.
#ifndef AA_FD_ROE
#define AA_FD_ROE 0
#endif
AA_DEV Real q_rcp(Real x)
{
  Real r = __builtin_amdgcn_rcp(x);
  Real e = fma(-x, r, 1.0); r = fma(e, r, r);
  e = fma(-x, r, 1.0); r = fma(e, r, r);
  return r;
}
AA_DEV Real q_rsqrt(Real x)
{
  Real r = __builtin_amdgcn_rsq(x);
  Real h = 0.5*r, e = fma(-x*r, h, 0.5);          // e = (1 - x r^2)/2
  r = fma(r, e, r);
  h = 0.5*r; e = fma(-x*r, h, 0.5);
  r = fma(r, e, r);
  return r;
}
// a/b for a denominator that may leave the range in which the Newton steps are safe
AA_DEV Real q_div_checked(Real a, Real b)
{
  const Real ab = fabs(b);
  return (ab > 1.0e-280 && ab < 1.0e280) ? a*q_rcp(b) : a/b;
}

// ---- both builds (AA_XDIV): IEEE quotients and square roots without the range scaling of hipcc's expansions.  a/b compiles to
// v_div_scale x2, v_rcp, four Newton fmas, q = n r, rem = fma(-d, q, n), v_div_fmas, v_div_fixup (11 instructions); sqrt(x) to a
// compare + select + v_ldexp in front of and v_ldexp + class test + 3 selects behind v_rsq and nine mul / fma (20).  The scaling
// only acts on denormal / huge denominators, quotients near the ends of the range, numerators below 2^-969 and x < 2^-767; the
// special-case selects only on zero / infinite operands.  For the operands of the Riemann solver (densities, sound speeds,
// total enthalpies, momenta squared: positive normal numbers in any unit system a run survives in, or exactly zero as a
// numerator) the SAME instruction sequence without those parts gives the SAME bits -- correctly rounded, so the strict build
// stays bit-identical to the CPU and no Roe -> HLLE switch can move (tests: x_div / x_sqrt against the compiler's forms on
// random and on edge operands; every bitwise parity test of the strict build runs on them).  A NaN operand gives NaN either way.
// With a loop-invariant denominator the refined reciprocal is the correctly rounded one the host supplies (g.rGamma_1 =
// 1/(gamma-1) in IEEE arithmetic) and a quotient is three instructions: q = a y, rem = fma(-b, q, a), fma(rem, y, q).  That is
// the IEEE quotient on every operand tested (tests/test_gpu_parity.py: 3e6 pairs over 400 decades, and every bitwise run of the strict
// build), NOT a proven one: Markstein's theorem (1990) wants q = RN(a y) to be a faithful rounding of a/b as well, and a RN(1/b) can be
// off by ~1.5 ulp, so a quotient within ~2^-51 ulp of a rounding boundary could come out one ulp apart -- no test can rule that out.
#ifndef AA_XDIV
#define AA_XDIV 1
#endif
AA_DEV Real x_rcp_ref(Real b)                    // the reciprocal hipcc's division refines (two Newton steps on v_rcp_f64)
{
  Real r = __builtin_amdgcn_rcp(b);
  Real e = fma(-b, r, 1.0); r = fma(r, e, r);
  e = fma(-b, r, 1.0); r = fma(r, e, r);
  return r;
}
AA_DEV Real x_div_r(Real a, Real b, Real r)      // a/b with r = x_rcp_ref(b) or the correctly rounded 1/b
{
  const Real q = a*r;
  const Real rem = fma(-b, q, a);
  return fma(rem, r, q);
}
template <bool XD = true> AA_DEV Real x_div(Real a, Real b)
{
#if AA_XDIV
  if (XD) return x_div_r(a, b, x_rcp_ref(b));
#endif
  return a/b;
}
template <bool XD = true> AA_DEV Real x_sqrt(Real x)
{
#if AA_XDIV
  if (!XD) return sqrt(x);
  const Real y = __builtin_amdgcn_rsq(x);
  Real g = x*y, h = y*0.5;
  const Real r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  Real d = fma(-g, g, x); g = fma(d, h, g);
  d = fma(-g, g, x); g = fma(d, h, g);
  return g;
#else
  return sqrt(x);
#endif
}

// convert_var.c:389 Cons1D_to_Prim1D
template <int NS>
AA_DEV void cons_to_prim(const Real u[6], Real w[6], Real Gamma_1)
{
#if AA_FD_C2P
  Real di = q_rcp(u[0]);
#else
  Real di = 1.0/u[0];
#endif
  w[0] = u[0]; w[1] = u[1]*di; w[2] = u[2]*di; w[3] = u[3]*di;
  Real p = u[4] - 0.5*(sqr(u[1]) + sqr(u[2]) + sqr(u[3]))*di;
  p *= Gamma_1;
  w[4] = rmax(p, AA_TINY);
  w[5] = NS ? u[5]*di : 0.0;
}

// convert_var.c:432 Prim1D_to_Cons1D
// rG_1: the correctly rounded 1/(gamma-1) (DevGrid.rGamma_1): the quotient P/(gamma-1) in three instructions, same bits
template <int NS>
AA_DEV void prim_to_cons(const Real w[6], Real u[6], Real Gamma_1, Real rG_1)
{
  u[0] = w[0]; u[1] = w[0]*w[1]; u[2] = w[0]*w[2]; u[3] = w[0]*w[3];
#if AA_XDIV
  u[4] = x_div_r(w[4], Gamma_1, rG_1) + 0.5*w[0]*(sqr(w[1]) + sqr(w[2]) + sqr(w[3]));
#elif AA_FD_P2C
  u[4] = w[4]*(1.0/Gamma_1) + 0.5*w[0]*(sqr(w[1]) + sqr(w[2]) + sqr(w[3]));      // (the quotient is loop-invariant: one division per thread)
#else
  u[4] = w[4]/Gamma_1 + 0.5*w[0]*(sqr(w[1]) + sqr(w[2]) + sqr(w[3]));
#endif
  u[5] = NS ? w[5]*w[0] : 0.0;
}

// convert_var.c:470 cfast (hydro)
AA_DEV Real cfast(const Real u[6], Real Gamma, Real Gamma_1)
{
  Real p = Gamma_1*(u[4] - 0.0 - 0.5*(sqr(u[1]) + sqr(u[2]) + sqr(u[3]))/u[0]);
  Real asq = Gamma*p/u[0];
  return sqrt(asq);
}

// The wave speeds the H-correction's eta is made of (integrate_3d_ctu.c:2300-2343): v - cfast of a face's left state,
// v + cfast of its right state.  Default build (AA_FD_ETA): one reciprocal of the density serves the velocity, the pressure
// and the sound speed, the square root comes from v_rsq_f64 -- 3 divisions + 1 square root (~60 instructions, most of them
// quarter rate) become ~25.  eta enters the second-pass fluxes only through max(|lambda|, etah): continuous, no switch
// hangs on its last bit.  A non-positive a^2 (negative face pressure beside the planet's density jump: the NaN etas the
// reference produces there) takes the reference's own expression, so the NaN pattern is the reference's.
#ifndef AA_FD_ETA
#define AA_FD_ETA AA_FAST_DIV
#endif
AA_DEV Real lambda_face(const Real u[6], Real Gamma, Real Gamma_1, Real sign)
{
#if AA_FD_ETA
  const Real d = u[0];
  if (d > 1.0e-280 && d < 1.0e280) {
    const Real di = q_rcp(d);
    const Real p = Gamma_1*(u[4] - 0.0 - 0.5*(sqr(u[1]) + sqr(u[2]) + sqr(u[3]))*di);
    const Real asq = Gamma*p*di;
    if (asq > 1.0e-280 && asq < 1.0e280) return u[1]*di + sign*(asq*q_rsqrt(asq));
  }
#endif
  return u[1]/u[0] + sign*cfast(u, Gamma, Gamma_1);
}

// rsolvers/hlle.c:62 (compiled into roe.c as flux_hlle, roe.c:339-341)
template <int NS, bool XD = true>
AA_DEV void flux_hlle(const Real ul[6], const Real ur[6], const Real wl[6], const Real wr[6],
                      Real Gamma, Real Gamma_1, Real f[6])
{
  // (the same scaling-free forms as in flux_roe, whose values these are: one evaluation serves both)
  Real sqrtdl = x_sqrt<XD>(wl[0]), sqrtdr = x_sqrt<XD>(wr[0]);
  Real isdlpdr = x_div<XD>(1.0, sqrtdl + sqrtdr);
  Real v1 = (sqrtdl*wl[1] + sqrtdr*wr[1])*isdlpdr;
  Real v2 = (sqrtdl*wl[2] + sqrtdr*wr[2])*isdlpdr;
  Real v3 = (sqrtdl*wl[3] + sqrtdr*wr[3])*isdlpdr;
  Real h  = (x_div<XD>(ul[4] + wl[4] + 0.0, sqrtdl) + x_div<XD>(ur[4] + wr[4] + 0.0, sqrtdr))*isdlpdr;
  Real vsq = v1*v1 + v2*v2 + v3*v3;
  Real a = x_sqrt<XD>(Gamma_1*rmax((h - 0.5*vsq), AA_TINY));
  Real ev0 = v1 - a, ev4 = v1 + a;

  Real asq = Gamma*wl[4]/wl[0];
  Real qsq = 0.0 + 0.0 + asq, tmp = 0.0 + 0.0 - asq;
  Real cfl = sqrt(0.5*(qsq + sqrt(tmp*tmp + 4.0*asq*0.0)));
  asq = Gamma*wr[4]/wr[0];
  qsq = 0.0 + 0.0 + asq; tmp = 0.0 + 0.0 - asq;
  Real cfr = sqrt(0.5*(qsq + sqrt(tmp*tmp + 4.0*asq*0.0)));

  Real ar = rmax(ev4, (wr[1] + cfr));
  Real al = rmin(ev0, (wl[1] - cfl));
  Real bp = rmax(ar, 0.0), bm = rmin(al, 0.0);

  Real Fl[6], Fr[6];
  Fl[0] = ul[1] - bm*ul[0];          Fr[0] = ur[1] - bp*ur[0];
  Fl[1] = ul[1]*(wl[1] - bm);        Fr[1] = ur[1]*(wr[1] - bp);
  Fl[2] = ul[2]*(wl[1] - bm);        Fr[2] = ur[2]*(wr[1] - bp);
  Fl[3] = ul[3]*(wl[1] - bm);        Fr[3] = ur[3]*(wr[1] - bp);
  Fl[1] += wl[4];                    Fr[1] += wr[4];
  Fl[4] = ul[4]*(wl[1] - bm) + wl[4]*wl[1];
  Fr[4] = ur[4]*(wr[1] - bp) + wr[4]*wr[1];
  Fl[5] = Fl[0]*wl[5];               Fr[5] = Fr[0]*wr[5];
  tmp = 0.5*(bp + bm)/(bp - bm);
#pragma unroll
  for (int n = 0; n < 5 + NS; n++) f[n] = 0.5*(Fl[n] + Fr[n]) + (Fl[n] - Fr[n])*tmp;
  if (!NS) f[5] = 0.0;
}

// rsolvers/roe.c:59 fluxes() with the H-correction etah and the HLLE fallback;
// eigensystem rsolvers/esystem_roe.c:132
// FAST: the reciprocal forms (AA_FD_ROE, off: see above; k_flux2_update, at its register limit, was also slower with them)
// XD: the scaling-free quotients / square roots (AA_XDIV).  k_flux2_update, which is not bound by its instruction count, lost 4 %
// with them while it was at its register limit (13.9 -> 14.4 ms at 512^3, three more registers spilled) and gains 2 % since its
// carried x3 flux waits in LDS (FU_PARK, hydro_kernels.hip)
template <int NS, bool FAST = (AA_FD_ROE != 0), bool XD = true>
AA_DEV void flux_roe(const Real ul[6], const Real ur[6], const Real wl[6], const Real wr[6],
                     Real etah, Real Gamma, Real Gamma_1, Real f[6])
{
  Real sqrtdl, sqrtdr, isdlpdr, v1, v2, v3, h, vsq, asq, a, iasq = 0.0;
  if (FAST) {
  const Real isl = q_rsqrt(wl[0]), isr = q_rsqrt(wr[0]);
    sqrtdl = wl[0]*isl; sqrtdr = wr[0]*isr;
    isdlpdr = q_rcp(sqrtdl + sqrtdr);
    v1 = (sqrtdl*wl[1] + sqrtdr*wr[1])*isdlpdr;
    v2 = (sqrtdl*wl[2] + sqrtdr*wr[2])*isdlpdr;
    v3 = (sqrtdl*wl[3] + sqrtdr*wr[3])*isdlpdr;
    h  = ((ul[4] + wl[4] + 0.0)*isl + (ur[4] + wr[4] + 0.0)*isr)*isdlpdr;
    vsq = v1*v1 + v2*v2 + v3*v3;
    asq = Gamma_1*rmax((h - 0.5*vsq), AA_TINY);
    const Real ia = q_rsqrt(asq);
    iasq = ia*ia;
    a = asq*ia;
  } else {
    sqrtdl = x_sqrt<XD>(wl[0]); sqrtdr = x_sqrt<XD>(wr[0]);
    isdlpdr = x_div<XD>(1.0, sqrtdl + sqrtdr);
    v1 = (sqrtdl*wl[1] + sqrtdr*wr[1])*isdlpdr;
    v2 = (sqrtdl*wl[2] + sqrtdr*wr[2])*isdlpdr;
    v3 = (sqrtdl*wl[3] + sqrtdr*wr[3])*isdlpdr;
    h  = (x_div<XD>(ul[4] + wl[4] + 0.0, sqrtdl) + x_div<XD>(ur[4] + wr[4] + 0.0, sqrtdr))*isdlpdr;
    vsq = v1*v1 + v2*v2 + v3*v3;
    asq = Gamma_1*rmax((h - 0.5*vsq), AA_TINY);
    a = x_sqrt<XD>(asq);
  }
  Real ev0 = v1 - a, ev4 = v1 + a;

  Real Fl[6], Fr[6];
  Fl[0] = ul[1];            Fr[0] = ur[1];
  Fl[1] = ul[1]*wl[1];      Fr[1] = ur[1]*wr[1];
  Fl[2] = ul[1]*wl[2];      Fr[2] = ur[1]*wr[2];
  Fl[3] = ul[1]*wl[3];      Fr[3] = ur[1]*wr[3];
  Fl[1] += wl[4];           Fr[1] += wr[4];
  Fl[4] = (ul[4] + wl[4])*wl[1];
  Fr[4] = (ur[4] + wr[4])*wr[1];
  Fl[5] = NS ? Fl[0]*wl[5] : 0.0;
  Fr[5] = NS ? Fr[0]*wr[5] : 0.0;

  if (ev0 >= 0.0) {                       // roe.c:215-235 supersonic: upwind flux
#pragma unroll
    for (int n = 0; n < 6; n++) f[n] = Fl[n];
    return;
  }
  if (ev4 <= 0.0) {
#pragma unroll
    for (int n = 0; n < 6; n++) f[n] = Fr[n];
    return;
  }

  const Real rasq = (FAST || !XD || !AA_XDIV) ? 0.0 : x_rcp_ref(asq);      // one refined reciprocal for the two quotients by a^2
  Real na = FAST ? 0.5*iasq : ((XD && AA_XDIV) ? x_div_r(0.5, asq, rasq) : 0.5/asq);
  Real l00 = na*(0.5*Gamma_1*vsq + v1*a);
  Real l01 = -na*(Gamma_1*v1 + a);
  Real l02 = -na*Gamma_1*v2;
  Real l03 = -na*Gamma_1*v3;
  Real l04 = na*Gamma_1;
  Real qa = FAST ? Gamma_1*iasq : ((XD && AA_XDIV) ? x_div_r(Gamma_1, asq, rasq) : Gamma_1/asq);
  Real l30 = 1.0 - na*Gamma_1*vsq;
  Real l31 = qa*v1, l32 = qa*v2, l33 = qa*v3, l34 = -qa;
  Real l40 = na*(0.5*Gamma_1*vsq - v1*a);
  Real l41 = -na*(Gamma_1*v1 - a);

  Real dU[5], aa_[5];
#pragma unroll
  for (int n = 0; n < 5; n++) dU[n] = ur[n] - ul[n];
  aa_[0] = l00*dU[0]; aa_[0] += l01*dU[1]; aa_[0] += l02*dU[2]; aa_[0] += l03*dU[3]; aa_[0] += l04*dU[4];
  aa_[1] = (-v2)*dU[0]; aa_[1] += dU[2];
  aa_[2] = (-v3)*dU[0]; aa_[2] += dU[3];
  aa_[3] = l30*dU[0]; aa_[3] += l31*dU[1]; aa_[3] += l32*dU[2]; aa_[3] += l33*dU[3]; aa_[3] += l34*dU[4];
  aa_[4] = l40*dU[0]; aa_[4] += l41*dU[1]; aa_[4] += l02*dU[2]; aa_[4] += l03*dU[3]; aa_[4] += l04*dU[4];

  // roe.c:256-286: positivity of the intermediate states (tests only where ev[n+1] > ev[n])
  Real u0 = ul[0], u1 = ul[1], u2 = ul[2], u3 = ul[3], u4 = ul[4];
  bool hlle = false;
  u0 += aa_[0]; u1 += aa_[0]*(v1 - a); u2 += aa_[0]*v2; u3 += aa_[0]*v3; u4 += aa_[0]*(h - v1*a);
  if (v1 > ev0) {
    if (u0 <= 0.0) hlle = true;
    else {
      Real p_inter = FAST ? u4 - q_div_checked(0.5*(sqr(u1) + sqr(u2) + sqr(u3)), u0) : u4 - x_div<XD>(0.5*(sqr(u1) + sqr(u2) + sqr(u3)), u0);
      if (p_inter < 0.0) hlle = true;
    }
  }
  if (!hlle) {
    u2 += aa_[1]; u4 += aa_[1]*v2;
    u3 += aa_[2]; u4 += aa_[2]*v3;
    u0 += aa_[3]; u1 += aa_[3]*v1; u2 += aa_[3]*v2; u3 += aa_[3]*v3; u4 += aa_[3]*(0.5*vsq);
    if (ev4 > v1) {
      if (u0 <= 0.0) hlle = true;
      else {
      Real p_inter = FAST ? u4 - q_div_checked(0.5*(sqr(u1) + sqr(u2) + sqr(u3)), u0) : u4 - x_div<XD>(0.5*(sqr(u1) + sqr(u2) + sqr(u3)), u0);
      if (p_inter < 0.0) hlle = true;
    }
    }
  }
  if (hlle) { flux_hlle<NS, XD>(ul, ur, wl, wr, Gamma, Gamma_1, f); return; }

  Real c0 = 0.5*rmax(fabs(ev0), etah)*aa_[0];
  Real c1 = 0.5*rmax(fabs(v1), etah)*aa_[1];
  Real c2 = 0.5*rmax(fabs(v1), etah)*aa_[2];
  Real c3 = 0.5*rmax(fabs(v1), etah)*aa_[3];
  Real c4 = 0.5*rmax(fabs(ev4), etah)*aa_[4];
  f[0] = 0.5*(Fl[0] + Fr[0]); f[0] -= c0; f[0] -= c3; f[0] -= c4;
  f[1] = 0.5*(Fl[1] + Fr[1]); f[1] -= c0*(v1 - a); f[1] -= c3*v1; f[1] -= c4*(v1 + a);
  f[2] = 0.5*(Fl[2] + Fr[2]); f[2] -= c0*v2; f[2] -= c1; f[2] -= c3*v2; f[2] -= c4*v2;
  f[3] = 0.5*(Fl[3] + Fr[3]); f[3] -= c0*v3; f[3] -= c2; f[3] -= c3*v3; f[3] -= c4*v3;
  f[4] = 0.5*(Fl[4] + Fr[4]); f[4] -= c0*(h - v1*a); f[4] -= c1*v2; f[4] -= c2*v3;
  f[4] -= c3*(0.5*vsq); f[4] -= c4*(h + v1*a);
  f[5] = 0.0;
  if (NS) f[5] = (f[0] >= 0.0) ? f[0]*wl[5] : f[0]*wr[5];
}

// reconstruction/lr_states_plm.c:62, one cell: from W[i-1], W[i], W[i+1] produce
// wl_next = Wl[i+1] (left state of the upper interface) and wr_here = Wr[i] (right state of
// the lower interface), PLM in characteristic variables + CTU characteristic tracing.
// TRACE=false is the VL_INTEGRATOR branch (lr_states_plm.c:250-255).
// AA_FD_MINMAX (default build): v_min_f64 / v_max_f64 for the 66 MIN / MAX of a reconstructed cell instead of the
// reference's compare-and-select macro (3 instructions each): -5 % of the first-pass sweeps, -3 % of k_correct_all
// (round 3, same-box ABAB at 512^3: hydro chain 48.0 -> 46.7 ms).  Equal on finite operands up to the sign of a zero; a NaN
// operand is dropped by the hardware min / max where the macro's result depends on the operand order -- the strict build
// keeps the macro, and so does every MAX of the H-correction's eta chain in both builds (NaN etas are part of the
// reference's behaviour beside the planet's density jump).
#ifndef AA_FD_MINMAX
#define AA_FD_MINMAX AA_FAST_DIV
#endif
#if AA_FD_MINMAX
AA_DEV Real pmax(Real a, Real b) { return __builtin_fmax(a, b); }
AA_DEV Real pmin(Real a, Real b) { return __builtin_fmin(a, b); }
#else
AA_DEV Real pmax(Real a, Real b) { return rmax(a, b); }
AA_DEV Real pmin(Real a, Real b) { return rmin(a, b); }
#endif
template <int NS, bool TRACE>
AA_DEV void plm_cell(const Real wm[6], const Real w[6], const Real wp[6], Real dtodx, Real Gamma,
                     Real wl_next[6], Real wr_here[6])
{
  constexpr int NV = 5 + NS;
  Real d = w[0], vx = w[1];
#if AA_FD_PLM
  const Real id = q_rcp(d);
  Real asq = (Gamma*w[4])*id;
  const Real ia = q_rsqrt(asq), iasq = ia*ia;
  Real a = asq*ia;
  Real ev0 = vx - a, ev4 = vx + a;
  Real r10 = -a*id, r14 = -r10;
  Real l01 = -0.5*d*ia, l04 = 0.5*iasq, l14 = -iasq, l41 = -l01;
#else
  Real asq = (Gamma*w[4])/d, a = sqrt(asq);
  Real ev0 = vx - a, ev4 = vx + a;
  Real r10 = -a/d, r14 = -r10;
  Real l01 = -0.5*d/a, l04 = 0.5/asq, l14 = -1.0/asq, l41 = -l01;
#endif

  Real dWc[6], dWl[6], dWr[6], dWg[6];
#pragma unroll
  for (int n = 0; n < NV; n++) {
    dWc[n] = wp[n] - wm[n]; dWl[n] = w[n] - wm[n]; dWr[n] = wp[n] - w[n];
#if AA_FD_PLM
    // (where the product is positive the sum is a normal number: two operands small enough for a subnormal sum have
    //  a product that underflows to zero)
    dWg[n] = (dWl[n]*dWr[n] > 0.0) ? 2.0*dWl[n]*dWr[n]*q_rcp(dWl[n] + dWr[n]) : 0.0;
#else
    dWg[n] = (dWl[n]*dWr[n] > 0.0) ? 2.0*dWl[n]*dWr[n]/(dWl[n] + dWr[n]) : 0.0;
#endif
  }
  Real dac[6], dal[6], dar[6], dag[6];
#define AA_PROJ(o, x) { o[0] = l01*x[1]; o[0] += l04*x[4]; o[1] = x[0]; o[1] += l14*x[4]; \
                        o[2] = x[2]; o[3] = x[3]; o[4] = l41*x[1]; o[4] += l04*x[4]; \
                        if (NS) o[5] = x[5]; }
  AA_PROJ(dac, dWc) AA_PROJ(dal, dWl) AA_PROJ(dar, dWr) AA_PROJ(dag, dWg)
#undef AA_PROJ
  Real da[6];
#pragma unroll
  for (int n = 0; n < NV; n++) {
    da[n] = 0.0;
    if (dal[n]*dar[n] > 0.0) {
      Real lim1 = pmin(fabs(dal[n]), fabs(dar[n]));
      Real lim2 = pmin(0.5*fabs(dac[n]), fabs(dag[n]));
      da[n] = ((dac[n] < 0.) ? -1. : 1.)*pmin(2.0*lim1, lim2);
    }
  }
  Real dWm[6];
  dWm[0] = da[0]; dWm[0] += da[1]; dWm[0] += da[4];
  dWm[1] = da[0]*r10; dWm[1] += da[4]*r14;
  dWm[2] = da[2]; dWm[3] = da[3];
  dWm[4] = da[0]*asq; dWm[4] += da[4]*asq;
  if (NS) dWm[5] = da[5];

  Real Wlv[6], Wrv[6], dW[6];
#pragma unroll
  for (int n = 0; n < NV; n++) {
    Wlv[n] = w[n] - 0.5*dWm[n];
    Wrv[n] = w[n] + 0.5*dWm[n];
    Real C = Wrv[n] + Wlv[n];
    Wlv[n] = pmax(pmin(w[n], wm[n]), Wlv[n]);
    Wlv[n] = pmin(pmax(w[n], wm[n]), Wlv[n]);
    Wrv[n] = C - Wlv[n];
    Wrv[n] = pmax(pmin(w[n], wp[n]), Wrv[n]);
    Wrv[n] = pmin(pmax(w[n], wp[n]), Wrv[n]);
    Wlv[n] = (C - Wrv[n]);
    dW[n] = Wrv[n] - Wlv[n];
  }
  if (!NS) { wl_next[5] = 0.0; wr_here[5] = 0.0; }
  if (!TRACE) {
#pragma unroll
    for (int n = 0; n < NV; n++) { wl_next[n] = Wrv[n]; wr_here[n] = Wlv[n]; }
    return;
  }
  Real qx = 0.5*pmax(ev4, 0.0)*dtodx;
#pragma unroll
  for (int n = 0; n < NV; n++) wl_next[n] = Wrv[n] - qx*dW[n];
  qx = -0.5*pmin(ev0, 0.0)*dtodx;
#pragma unroll
  for (int n = 0; n < NV; n++) wr_here[n] = Wlv[n] + qx*dW[n];

  Real qa, qx1, qx2;
  qx1 = 0.5*dtodx*ev4;
  if (ev0 >= 0.0) {
    qx2 = 0.5*dtodx*ev0; qx = qx1 - qx2;
    qa = 0.0; qa += l01*qx*dW[1]; qa += l04*qx*dW[4];
    wl_next[0] += qa; wl_next[1] += qa*r10; wl_next[4] += qa*asq;
  }
  if (vx >= 0.0) {
    qx2 = 0.5*dtodx*vx; qx = qx1 - qx2;
    qa = 0.0; qa += qx*dW[0]; qa += l14*qx*dW[4];  wl_next[0] += qa;
    qa = 0.0; qa += qx*dW[2];                      wl_next[2] += qa;
    qa = 0.0; qa += qx*dW[3];                      wl_next[3] += qa;
  }
  qx1 = -0.5*dtodx*ev0;
  if (vx <= 0.0) {
    qx2 = -0.5*dtodx*vx; qx = -qx1 + qx2;
    qa = 0.0; qa += qx*dW[0]; qa += l14*qx*dW[4];  wr_here[0] += qa;
    qa = 0.0; qa += qx*dW[2];                      wr_here[2] += qa;
    qa = 0.0; qa += qx*dW[3];                      wr_here[3] += qa;
  }
  if (ev4 <= 0.0) {
    qx2 = -0.5*dtodx*ev4; qx = -qx1 + qx2;
    qa = 0.0; qa += l41*qx*dW[1]; qa += l04*qx*dW[4];
    wr_here[0] += qa; wr_here[1] += qa*r14; wr_here[4] += qa*asq;
  }
  if (NS) {
    if (vx > 0.)      wl_next[5] += 0.5*dtodx*(ev4 - vx)*dW[5];
    else if (vx < 0.) wr_here[5] += 0.5*dtodx*(ev0 - vx)*dW[5];
  }
}

// ---- reconstruction/lr_states_ppm.c:91-610 (THIRD_ORDER_CHAR, --with-order=3) -----------------
// Monotonised characteristic slope of one cell (Steps 1-5 / 8-12; identical to lr_states_plm.c:
// 131-202).  Evaluated once per cell and direction by k_slopes and kept in HBM: a cell's parabola
// needs the slopes of both neighbours as well.
template <int NS>
AA_DEV void limited_slopes(const Real wm[6], const Real w[6], const Real wp[6], Real Gamma, Real dWm[6])
{
  constexpr int NV = 5 + NS;
  Real d = w[0];
#if AA_FD_PLM      // (default build: the reciprocal forms of plm_cell)
  const Real id = q_rcp(d);
  Real asq = (Gamma*w[4])*id;
  const Real ia = q_rsqrt(asq), iasq = ia*ia;
  Real a = asq*ia;
  Real r10 = -a*id, r14 = -r10;
  Real l01 = -0.5*d*ia, l04 = 0.5*iasq, l14 = -iasq, l41 = -l01;
#else
  Real asq = (Gamma*w[4])/d, a = sqrt(asq);
  Real r10 = -a/d, r14 = -r10;
  Real l01 = -0.5*d/a, l04 = 0.5/asq, l14 = -1.0/asq, l41 = -l01;
#endif
  Real dWc[6], dWl[6], dWr[6], dWg[6];
#pragma unroll
  for (int n = 0; n < NV; n++) {
    dWc[n] = wp[n] - wm[n]; dWl[n] = w[n] - wm[n]; dWr[n] = wp[n] - w[n];
#if AA_FD_PLM
    dWg[n] = (dWl[n]*dWr[n] > 0.0) ? 2.0*dWl[n]*dWr[n]*q_rcp(dWl[n] + dWr[n]) : 0.0;
#else
    dWg[n] = (dWl[n]*dWr[n] > 0.0) ? 2.0*dWl[n]*dWr[n]/(dWl[n] + dWr[n]) : 0.0;
#endif
  }
  Real dac[6], dal[6], dar[6], dag[6];
#define AA_PROJ(o, x) { o[0] = l01*x[1]; o[0] += l04*x[4]; o[1] = x[0]; o[1] += l14*x[4]; \
                        o[2] = x[2]; o[3] = x[3]; o[4] = l41*x[1]; o[4] += l04*x[4]; \
                        if (NS) o[5] = x[5]; }
  AA_PROJ(dac, dWc) AA_PROJ(dal, dWl) AA_PROJ(dar, dWr) AA_PROJ(dag, dWg)
#undef AA_PROJ
  Real da[6];
#pragma unroll
  for (int n = 0; n < NV; n++) {
    da[n] = 0.0;
    if (dal[n]*dar[n] > 0.0) {
      Real lim1 = pmin(fabs(dal[n]), fabs(dar[n]));
      Real lim2 = pmin(0.5*fabs(dac[n]), fabs(dag[n]));
      da[n] = ((dac[n] < 0.) ? -1. : 1.)*pmin(2.0*lim1, lim2);
    }
  }
  dWm[0] = da[0]; dWm[0] += da[1]; dWm[0] += da[4];
  dWm[1] = da[0]*r10; dWm[1] += da[4]*r14;
  dWm[2] = da[2]; dWm[3] = da[3];
  dWm[4] = da[0]*asq; dWm[4] += da[4]*asq;
  dWm[5] = NS ? da[5] : 0.0;
}

// Parabola of one cell from its own and its neighbours' slopes (Steps 14-19), traced over the
// domain of dependence: Wl of the upper interface, Wr of the lower one.  With a passive scalar the
// reference's work arrays overlap (NWAVE columns allocated, NWAVE+NSCALARS indexed, :689-692): the
// scalar's left parabola edge is the density interface value and its right edge mixes the scalar
// and density slopes of cell i+1; reproduced as is (see oracle/athena_oracle.c lr_states_ppm).
// TRACE=false: the branch for integrators other than CTU (van Leer), lr_states_ppm.c:502-507: the parabola's edges as they are.
template <int NS, bool TRACE = true>
AA_DEV void ppm_cell(const Real wm[6], const Real w[6], const Real wp[6], const Real Dm[6], const Real D0[6],
                     const Real Dp[6], Real dtodx, Real Gamma, Real wl_next[6], Real wr_here[6])
{
  constexpr int NV = 5 + NS;
  constexpr Real FOUR_3RDS = 1.333333333333333, TWO_3RDS = 0.6666666666666667;     // defs.h.in:158-159
  const Real gamma_curv = 0.0, qxx1 = 0.0, qxx2 = 0.0;
  Real d = w[0], vx = w[1];
#if AA_FD_PLM      // (default build: the reciprocal forms of plm_cell; the sixths as products)
  const Real id = q_rcp(d);
  Real asq = (Gamma*w[4])*id;
  const Real ia = q_rsqrt(asq), iasq = ia*ia;
  Real a = asq*ia;
  Real ev0 = vx - a, ev4 = vx + a;
  Real r10 = -a*id, r14 = -r10;
  Real l01 = -0.5*d*ia, l04 = 0.5*iasq, l14 = -iasq, l41 = -l01;
#define AA_SIXTH(x) ((x)*(1.0/6.0))
#else
  Real asq = (Gamma*w[4])/d, a = sqrt(asq);
  Real ev0 = vx - a, ev4 = vx + a;
  Real r10 = -a/d, r14 = -r10;
  Real l01 = -0.5*d/a, l04 = 0.5/asq, l14 = -1.0/asq, l41 = -l01;
#define AA_SIXTH(x) ((x)/6.0)
#endif
  Real Wlv[6], Wrv[6], dW[6], W6[6];
#pragma unroll
  for (int n = 0; n < 5; n++) {
    Wlv[n] = 0.5*(w[n] + wm[n]) - AA_SIXTH(D0[n] - Dm[n]);
    Wrv[n] = 0.5*(wp[n] + w[n]) - AA_SIXTH(Dp[n] - D0[n]);
  }
  if (NS) {
    Wlv[5] = Wrv[0];
    Wrv[5] = 0.5*(wp[5] + w[5]) - AA_SIXTH(Dp[5] - Dp[0]);
  }
#undef AA_SIXTH
#pragma unroll
  for (int n = 0; n < NV; n++) {
    Real qa = (Wrv[n] - w[n])*(w[n] - Wlv[n]);
    Real qb = Wrv[n] - Wlv[n];
    Real qc = 6.0*(w[n] - 0.5*(Wlv[n]*(1.0 - gamma_curv) + Wrv[n]*(1.0 + gamma_curv)));
    if (qa <= 0.0) { Wlv[n] = w[n]; Wrv[n] = w[n]; }
    else if ((qb*qc) > (qb*qb)) Wlv[n] = (6.0*w[n] - Wrv[n]*(4.0 + 3.0*gamma_curv))/(2.0 - 3.0*gamma_curv);
    else if ((qb*qc) < -(qb*qb)) Wrv[n] = (6.0*w[n] - Wlv[n]*(4.0 - 3.0*gamma_curv))/(2.0 + 3.0*gamma_curv);
  }
#pragma unroll
  for (int n = 0; n < NV; n++) {
    Wlv[n] = pmax(pmin(w[n], wm[n]), Wlv[n]);
    Wlv[n] = pmin(pmax(w[n], wm[n]), Wlv[n]);
    Wrv[n] = pmax(pmin(w[n], wp[n]), Wrv[n]);
    Wrv[n] = pmin(pmax(w[n], wp[n]), Wrv[n]);
  }
  if (!TRACE) {
#pragma unroll
    for (int n = 0; n < NV; n++) { wl_next[n] = Wrv[n]; wr_here[n] = Wlv[n]; }
    if (!NS) { wl_next[5] = 0.0; wr_here[5] = 0.0; }
    return;
  }
#pragma unroll
  for (int n = 0; n < NV; n++) {
    dW[n] = Wrv[n] - Wlv[n];
    W6[n] = 6.0*(w[n] - 0.5*(Wlv[n]*(1.0 - gamma_curv) + Wrv[n]*(1.0 + gamma_curv)));
  }
  Real qx1 = 0.5*pmax(ev4, 0.0)*dtodx;
#pragma unroll
  for (int n = 0; n < NV; n++)
    wl_next[n] = Wrv[n] - qx1 *(dW[n] - (1.0 - FOUR_3RDS*qx1)*W6[n])
                        + qxx1*(dW[n] - (1.0 -       2.0*qx1)*W6[n]);
  Real qx2 = -0.5*pmin(ev0, 0.0)*dtodx;
#pragma unroll
  for (int n = 0; n < NV; n++)
    wr_here[n] = Wlv[n] + qx2 *(dW[n] + (1.0 - FOUR_3RDS*qx2)*W6[n])
                        + qxx2*(dW[n] + (1.0 -       2.0*qx2)*W6[n]);
  if (!NS) { wl_next[5] = 0.0; wr_here[5] = 0.0; }
  Real qa, qb, qc;
#define AA_TL(m) (qb*(dW[m] - W6[m]) + qc*W6[m])
#define AA_TR(m) (qb*(dW[m] + W6[m]) + qc*W6[m])
  qx1 = 0.5*dtodx*ev4;
  if (ev0 >= 0.0) {
    qx2 = 0.5*dtodx*ev0; qb = qx1 - qx2; qc = FOUR_3RDS*(qx1*qx1 - qx2*qx2);
    qa = 0.0; qa += l01*AA_TL(1); qa += l04*AA_TL(4);
    wl_next[0] += qa; wl_next[1] += qa*r10; wl_next[4] += qa*asq;
  }
  if (vx >= 0.0) {
    qx2 = 0.5*dtodx*vx; qb = qx1 - qx2; qc = FOUR_3RDS*(qx1*qx1 - qx2*qx2);
    qa = 0.0; qa += 1.0*AA_TL(0); qa += l14*AA_TL(4);  wl_next[0] += qa;
    qa = 0.0; qa += 1.0*AA_TL(2);                      wl_next[2] += qa;
    qa = 0.0; qa += 1.0*AA_TL(3);                      wl_next[3] += qa;
  }
  qx1 = 0.5*dtodx*ev0;
  if (vx <= 0.0) {
    qx2 = 0.5*dtodx*vx; qb = qx1 - qx2; qc = FOUR_3RDS*(qx1*qx1 - qx2*qx2);
    qa = 0.0; qa += 1.0*AA_TR(0); qa += l14*AA_TR(4);  wr_here[0] += qa;
    qa = 0.0; qa += 1.0*AA_TR(2);                      wr_here[2] += qa;
    qa = 0.0; qa += 1.0*AA_TR(3);                      wr_here[3] += qa;
  }
  if (ev4 <= 0.0) {
    qx2 = 0.5*dtodx*ev4; qb = qx1 - qx2; qc = FOUR_3RDS*(qx1*qx1 - qx2*qx2);
    qa = 0.0; qa += l41*AA_TR(1); qa += l04*AA_TR(4);
    wr_here[0] += qa; wr_here[1] += qa*r14; wr_here[4] += qa*asq;
  }
#undef AA_TL
#undef AA_TR
  if (NS) {
    if (vx > 0.) {
      qb = 0.5*dtodx*(ev4 - vx);
      qc = 0.5*dtodx*dtodx*TWO_3RDS*(ev4*ev4 - vx*vx);
      wl_next[5] += qb*(dW[5] - W6[5]) + qc*W6[5];
    } else if (vx < 0.) {
      qb = 0.5*dtodx*(ev0 - vx);
      qc = 0.5*dtodx*dtodx*TWO_3RDS*(ev0*ev0 - vx*vx);
      wr_here[5] += qb*(dW[5] + W6[5]) + qc*W6[5];
    }
  }
}

}  // namespace aa
