/* oracle/athena_oracle.c -- TEST INFRASTRUCTURE ONLY (see athena_oracle.h).
 *
 * Scalar CPU restatement of the reference's per-timestep hot path.  Every routine cites the
 * reference lines it follows (paths relative to /root/reference/src).  Floating-point
 * expression shapes and summation orders follow the reference so that results agree with
 * the reference binary to rounding (the dense 5x5 eigen-matrix loops are written out in
 * their sparse form, which leaves every partial sum unchanged).
 *
 * Layout differences from the reference (deliberate): face states and fluxes are kept in
 * the GLOBAL momentum frame (M1,M2,M3) and rotated on entry to the 1-D kernels, so the
 * transverse corrections need no component permutation tables; the ion step is split in
 * phases so a slab-decomposed driver can interleave global reductions.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include <math.h>
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "athena_oracle.h"

#define NGHOST 4
#define NW 5
#define TINY_NUMBER 1.0e-20
#define MINFLUXFRAC 1.0e-3   /* ionradiation/ionrad.h:26 */
#define MINOPTDEPTH 1.0e-4   /* :29 */
#define IONFRACFLOOR 1.0e-4  /* :31 */
#define CION 8.0e5           /* :36 */
#define MAXCELLCOUNT 20      /* :38 */
#define MAXSIGNCOUNT 4       /* ionradiation/ionrad_3d.c:286 */
#define DAMPFACTOR 0.5       /* :287 */
#define LARGE DBL_MAX        /* ionrad.h:16 */
#define KB_CHEM 1.38e-16     /* ionradiation/ionrad_chemistry.c:43 */

#define MAXR(a,b) (((a) > (b)) ? (a) : (b))
#define MINR(a,b) (((a) < (b)) ? (a) : (b))
#define SQR(x) ((x)*(x))

typedef double Real;
typedef struct { Real d, M[3], E, s; } Cons;          /* global frame */
typedef struct { Real d, Mx, My, Mz, E, s; } C1;      /* sweep frame, athena.h:148 */
typedef struct { Real d, Vx, Vy, Vz, P, r; } P1;      /* sweep frame, athena.h:171 */

/* diagnostics: how often the Roe solver fell back to HLLE / returned an upwind flux */
long orc_dbg_hlle = 0, orc_dbg_supersonic = 0;

struct OrcSim {
  OrcParams p;
  int cool;              /* 0: CoolingFunc == NULL, 1: KoyInut (microphysics/cool.c:48) */
  Real *phalf;
  int N[3], is, ie, js, je, ks, ke;
  Real dx[3], rootdx[3];
  Real Gamma, Gamma_1;
  Real time, dt; int nstep;
  Cons *U; Real *EdgeFlux;
  Cons *Ul[3], *Ur[3], *F[3];
  Real *eta[3], *dhalf;
  /* ion */
  int rad_dir, nradplane; Real flux_i;
  Real min_area, d_nlo;
  Real *ph_rate, *edot, *nHdot, *e_init, *e_th_init, *x_init;
  int *last_sign, *sign_count;
  int niter_last, nchem_last, ntherm_last;
  int level;            /* DomainS.Level: 0 unless the sim is a level of an OrcMesh */
};

#define IDX(s,k,j,i) (((size_t)(k)*(s)->N[1] + (j))*(s)->N[0] + (i))

/* ------------------------------------------------------------------------------------ */
/* conversions: convert_var.c:389 (Cons1D_to_Prim1D), :432 (Prim1D_to_Cons1D), :470 (cfast) */

static P1 cons_to_prim(const C1 *u, Real Gamma_1, int nscal)
{
  P1 w; Real di = 1.0/u->d;
  w.d = u->d; w.Vx = u->Mx*di; w.Vy = u->My*di; w.Vz = u->Mz*di;
  w.P = u->E - 0.5*(SQR(u->Mx)+SQR(u->My)+SQR(u->Mz))*di;
  w.P *= Gamma_1;
  w.P = MAXR(w.P, TINY_NUMBER);
  w.r = nscal ? u->s*di : 0.0;
  return w;
}

static C1 prim_to_cons(const P1 *w, Real Gamma_1, int nscal)
{
  C1 u;
  u.d = w->d; u.Mx = w->d*w->Vx; u.My = w->d*w->Vy; u.Mz = w->d*w->Vz;
  u.E = w->P/Gamma_1 + 0.5*w->d*(SQR(w->Vx)+SQR(w->Vy)+SQR(w->Vz));
  u.s = nscal ? w->r*w->d : 0.0;
  return u;
}

static Real cfast_c1(const C1 *u, Real Gamma, Real Gamma_1)
{
  Real p = Gamma_1*(u->E - 0.0 - 0.5*(SQR(u->Mx)+SQR(u->My)+SQR(u->Mz))/u->d);
  Real asq = Gamma*p/u->d;
  return sqrt(asq);
}

/* ------------------------------------------------------------------------------------ */
/* HLLE flux: rsolvers/hlle.c:62-263 (hydro, adiabatic) */

static void flux_hlle(const C1 *Ul, const C1 *Ur, const P1 *Wl, const P1 *Wr,
                      Real Gamma, Real Gamma_1, int nscal, C1 *pF)
{
  Real sqrtdl = sqrt(Wl->d), sqrtdr = sqrt(Wr->d);
  Real isdlpdr = 1.0/(sqrtdl + sqrtdr);
  Real v1roe = (sqrtdl*Wl->Vx + sqrtdr*Wr->Vx)*isdlpdr;
  Real v2roe = (sqrtdl*Wl->Vy + sqrtdr*Wr->Vy)*isdlpdr;
  Real v3roe = (sqrtdl*Wl->Vz + sqrtdr*Wr->Vz)*isdlpdr;
  Real hroe  = ((Ul->E + Wl->P + 0.0)/sqrtdl + (Ur->E + Wr->P + 0.0)/sqrtdr)*isdlpdr;
  /* esys_roe_adb_hyd eigenvalues only (esystem_roe.c:137-148) */
  Real vsq = v1roe*v1roe + v2roe*v2roe + v3roe*v3roe;
  Real asqr = Gamma_1*MAXR((hroe-0.5*vsq), TINY_NUMBER);
  Real ar_ = sqrt(asqr);
  Real ev0 = v1roe - ar_, ev4 = v1roe + ar_;
  Real asq, qsq, tmp, cfsq, cfl, cfr, ar, al, bp, bm;
  Real Fl[6], Fr[6], *F = (Real*)pF; int n;

  asq = Gamma*Wl->P/Wl->d;           /* hlle.c:146-158, MHD terms are 0.0 */
  qsq = 0.0 + 0.0 + asq; tmp = 0.0 + 0.0 - asq;
  cfsq = 0.5*(qsq + sqrt(tmp*tmp + 4.0*asq*0.0));
  cfl = sqrt(cfsq);
  asq = Gamma*Wr->P/Wr->d;
  qsq = 0.0 + 0.0 + asq; tmp = 0.0 + 0.0 - asq;
  cfsq = 0.5*(qsq + sqrt(tmp*tmp + 4.0*asq*0.0));
  cfr = sqrt(cfsq);

  ar = MAXR(ev4, (Wr->Vx + cfr));
  al = MINR(ev0, (Wl->Vx - cfl));
  bp = MAXR(ar, 0.0);
  bm = MINR(al, 0.0);

  Fl[0] = Ul->Mx - bm*Ul->d;           Fr[0] = Ur->Mx - bp*Ur->d;
  Fl[1] = Ul->Mx*(Wl->Vx - bm);        Fr[1] = Ur->Mx*(Wr->Vx - bp);
  Fl[2] = Ul->My*(Wl->Vx - bm);        Fr[2] = Ur->My*(Wr->Vx - bp);
  Fl[3] = Ul->Mz*(Wl->Vx - bm);        Fr[3] = Ur->Mz*(Wr->Vx - bp);
  Fl[1] += Wl->P;                      Fr[1] += Wr->P;
  Fl[4] = Ul->E*(Wl->Vx - bm) + Wl->P*Wl->Vx;
  Fr[4] = Ur->E*(Wr->Vx - bp) + Wr->P*Wr->Vx;
  Fl[5] = Fl[0]*Wl->r;                 Fr[5] = Fr[0]*Wr->r;

  tmp = 0.5*(bp + bm)/(bp - bm);
  for (n = 0; n < NW + nscal; n++) F[n] = 0.5*(Fl[n] + Fr[n]) + (Fl[n] - Fr[n])*tmp;
  if (!nscal) F[5] = 0.0;
}

/* ------------------------------------------------------------------------------------ */
/* Roe flux with H-correction and HLLE fallback: rsolvers/roe.c:59-336,
 * eigensystem rsolvers/esystem_roe.c:132-215 */

static void flux_roe(const C1 *Ul, const C1 *Ur, const P1 *Wl, const P1 *Wr, Real etah,
                     Real Gamma, Real Gamma_1, int nscal, C1 *pF)
{
  Real sqrtdl = sqrt(Wl->d), sqrtdr = sqrt(Wr->d);
  Real isdlpdr = 1.0/(sqrtdl + sqrtdr);
  Real v1 = (sqrtdl*Wl->Vx + sqrtdr*Wr->Vx)*isdlpdr;
  Real v2 = (sqrtdl*Wl->Vy + sqrtdr*Wr->Vy)*isdlpdr;
  Real v3 = (sqrtdl*Wl->Vz + sqrtdr*Wr->Vz)*isdlpdr;
  Real h  = ((Ul->E + Wl->P + 0.0)/sqrtdl + (Ur->E + Wr->P + 0.0)/sqrtdr)*isdlpdr;
  Real vsq = v1*v1 + v2*v2 + v3*v3;
  Real asq = Gamma_1*MAXR((h-0.5*vsq), TINY_NUMBER);
  Real a = sqrt(asq);
  Real ev[5], Fl[6], Fr[6], dU[5], aa[5], u[5], coeff[5];
  Real na, qa, l00,l01,l02,l03,l04,l30,l31,l32,l33,l34,l40,l41, p_inter;
  Real *F = (Real*)pF;
  const Real *pUl = (const Real*)Ul, *pUr = (const Real*)Ur;
  int n, hlle_flag = 0;

  ev[0] = v1 - a; ev[1] = v1; ev[2] = v1; ev[3] = v1; ev[4] = v1 + a;

  Fl[0] = Ul->Mx;            Fr[0] = Ur->Mx;            /* roe.c:159-208 */
  Fl[1] = Ul->Mx*Wl->Vx;     Fr[1] = Ur->Mx*Wr->Vx;
  Fl[2] = Ul->Mx*Wl->Vy;     Fr[2] = Ur->Mx*Wr->Vy;
  Fl[3] = Ul->Mx*Wl->Vz;     Fr[3] = Ur->Mx*Wr->Vz;
  Fl[1] += Wl->P;            Fr[1] += Wr->P;
  Fl[4] = (Ul->E + Wl->P)*Wl->Vx;
  Fr[4] = (Ur->E + Wr->P)*Wr->Vx;
  Fl[5] = nscal ? Fl[0]*Wl->r : 0.0;
  Fr[5] = nscal ? Fr[0]*Wr->r : 0.0;

  if (ev[0] >= 0.0) { orc_dbg_supersonic++; for (n = 0; n < 6; n++) F[n] = Fl[n]; return; }   /* :215-235 */
  if (ev[4] <= 0.0) { orc_dbg_supersonic++; for (n = 0; n < 6; n++) F[n] = Fr[n]; return; }

  /* left eigenvectors (rows), esystem_roe.c:183-214 */
  na = 0.5/asq;
  l00 = na*(0.5*Gamma_1*vsq + v1*a);
  l01 = -na*(Gamma_1*v1 + a);
  l02 = -na*Gamma_1*v2;
  l03 = -na*Gamma_1*v3;
  l04 = na*Gamma_1;
  qa = Gamma_1/asq;
  l30 = 1.0 - na*Gamma_1*vsq;
  l31 = qa*v1; l32 = qa*v2; l33 = qa*v3; l34 = -qa;
  l40 = na*(0.5*Gamma_1*vsq - v1*a);
  l41 = -na*(Gamma_1*v1 - a);

  for (n = 0; n < NW; n++) dU[n] = pUr[n] - pUl[n];     /* :241-248 */
  aa[0] = l00*dU[0]; aa[0] += l01*dU[1]; aa[0] += l02*dU[2]; aa[0] += l03*dU[3]; aa[0] += l04*dU[4];
  aa[1] = (-v2)*dU[0]; aa[1] += dU[2];
  aa[2] = (-v3)*dU[0]; aa[2] += dU[3];
  aa[3] = l30*dU[0]; aa[3] += l31*dU[1]; aa[3] += l32*dU[2]; aa[3] += l33*dU[3]; aa[3] += l34*dU[4];
  aa[4] = l40*dU[0]; aa[4] += l41*dU[1]; aa[4] += l02*dU[2]; aa[4] += l03*dU[3]; aa[4] += l04*dU[4];

  /* intermediate states, :256-286 (right eigenvectors are the columns at esystem_roe.c:151-181) */
  for (n = 0; n < NW; n++) u[n] = pUl[n];
  /* wave 0 */
  u[0] += aa[0]; u[1] += aa[0]*(v1 - a); u[2] += aa[0]*v2; u[3] += aa[0]*v3; u[4] += aa[0]*(h - v1*a);
  if (ev[1] > ev[0]) {
    if (u[0] <= 0.0) hlle_flag = 1;
    else { p_inter = u[4] - 0.5*(SQR(u[1])+SQR(u[2])+SQR(u[3]))/u[0]; if (p_inter < 0.0) hlle_flag = 2; }
  }
  if (!hlle_flag) {
    /* waves 1,2 (ev equal: no test), wave 3 */
    u[2] += aa[1]; u[4] += aa[1]*v2;
    u[3] += aa[2]; u[4] += aa[2]*v3;
    u[0] += aa[3]; u[1] += aa[3]*v1; u[2] += aa[3]*v2; u[3] += aa[3]*v3; u[4] += aa[3]*(0.5*vsq);
    if (ev[4] > ev[3]) {
      if (u[0] <= 0.0) hlle_flag = 1;
      else { p_inter = u[4] - 0.5*(SQR(u[1])+SQR(u[2])+SQR(u[3]))/u[0]; if (p_inter < 0.0) hlle_flag = 2; }
    }
  }
  if (hlle_flag) { orc_dbg_hlle++; flux_hlle(Ul, Ur, Wl, Wr, Gamma, Gamma_1, nscal, pF); return; }

  for (n = 0; n < NW; n++) coeff[n] = 0.5*MAXR(fabs(ev[n]), etah)*aa[n];   /* :291-312 */
  F[0] = 0.5*(Fl[0] + Fr[0]); F[0] -= coeff[0]; F[0] -= coeff[3]; F[0] -= coeff[4];
  F[1] = 0.5*(Fl[1] + Fr[1]); F[1] -= coeff[0]*(v1 - a); F[1] -= coeff[3]*v1; F[1] -= coeff[4]*(v1 + a);
  F[2] = 0.5*(Fl[2] + Fr[2]); F[2] -= coeff[0]*v2; F[2] -= coeff[1]; F[2] -= coeff[3]*v2; F[2] -= coeff[4]*v2;
  F[3] = 0.5*(Fl[3] + Fr[3]); F[3] -= coeff[0]*v3; F[3] -= coeff[2]; F[3] -= coeff[3]*v3; F[3] -= coeff[4]*v3;
  F[4] = 0.5*(Fl[4] + Fr[4]); F[4] -= coeff[0]*(h - v1*a); F[4] -= coeff[1]*v2; F[4] -= coeff[2]*v3;
  F[4] -= coeff[3]*(0.5*vsq); F[4] -= coeff[4]*(h + v1*a);
  F[5] = 0.0;
  if (nscal) F[5] = (F[0] >= 0.0) ? F[0]*Wl->r : F[0]*Wr->r;                /* :316-320 */
}

/* ------------------------------------------------------------------------------------ */
/* PLM + characteristic tracing: reconstruction/lr_states_plm.c:62-374,
 * eigensystem reconstruction/esystem_prim.c:120-199.  Reconstructs cells il-1..iu+1,
 * writing Wl[i+1] and Wr[i] (so interfaces il..iu+1 are complete). */

static void lr_states_x(const P1 *W, Real dt, Real dx, int il, int iu, P1 *Wl, P1 *Wr,
                        Real Gamma, int nscal, int trace);
static void lr_states_ppm(const P1 *W, Real dt, Real dx, int il, int iu, P1 *Wl, P1 *Wr, Real Gamma, int nscal, int trace);
static void lr_states(const P1 *W, Real dt, Real dx, int il, int iu, P1 *Wl, P1 *Wr,
                      Real Gamma, int nscal, int order)
{
  if (order == 3) lr_states_ppm(W, dt, dx, il, iu, Wl, Wr, Gamma, nscal, 1);
  else lr_states_x(W, dt, dx, il, iu, Wl, Wr, Gamma, nscal, 1);
}

static void lr_states_x(const P1 *W, Real dt, Real dx, int il, int iu, P1 *Wl, P1 *Wr,
                        Real Gamma, int nscal, int trace)
{
  const Real dtodx = dt/dx;
  const int nv = NW + nscal;
  int i, n;
  for (i = il-1; i <= iu+1; i++) {
    const Real *w = (const Real*)&W[i], *wm = (const Real*)&W[i-1], *wp = (const Real*)&W[i+1];
    Real d = W[i].d, vx = W[i].Vx, asq = (Gamma*W[i].P)/d, a = sqrt(asq);
    Real ev0 = vx - a, ev4 = vx + a;
    Real r10 = -a/d, r14 = -r10;                 /* rem[1][0], rem[1][4] */
    Real l01 = -0.5*d/a, l04 = 0.5/asq, l14 = -1.0/asq, l41 = -l01;
    Real dWc[6], dWl[6], dWr[6], dWg[6], dac[6], dal[6], dar[6], dag[6], da[6], dWm[6];
    Real Wlv[6], Wrv[6], dW[6], *pWl = (Real*)&Wl[i+1], *pWr = (Real*)&Wr[i];
    Real lim1, lim2, C, qx, qa, qx1, qx2;

    for (n = 0; n < nv; n++) {                                          /* :131-147 */
      dWc[n] = wp[n] - wm[n]; dWl[n] = w[n] - wm[n]; dWr[n] = wp[n] - w[n];
      dWg[n] = (dWl[n]*dWr[n] > 0.0) ? 2.0*dWl[n]*dWr[n]/(dWl[n]+dWr[n]) : 0.0;
    }
#define PROJ(o,x) do { o[0] = l01*x[1]; o[0] += l04*x[4]; o[1] = x[0]; o[1] += l14*x[4]; \
                       o[2] = x[2]; o[3] = x[3]; o[4] = l41*x[1]; o[4] += l04*x[4]; \
                       if (nscal) o[5] = x[5]; } while (0)
    PROJ(dac,dWc); PROJ(dal,dWl); PROJ(dar,dWr); PROJ(dag,dWg);         /* :152-174 */
#undef PROJ
    for (n = 0; n < nv; n++) {                                          /* :180-187 */
      da[n] = 0.0;
      if (dal[n]*dar[n] > 0.0) {
        lim1 = MINR(fabs(dal[n]), fabs(dar[n]));
        lim2 = MINR(0.5*fabs(dac[n]), fabs(dag[n]));
        da[n] = ((dac[n] < 0.) ? -1. : 1.)*MINR(2.0*lim1, lim2);
      }
    }
    dWm[0] = da[0]; dWm[0] += da[1]; dWm[0] += da[4];                   /* :192-202 */
    dWm[1] = da[0]*r10; dWm[1] += da[4]*r14;
    dWm[2] = da[2]; dWm[3] = da[3];
    dWm[4] = da[0]*asq; dWm[4] += da[4]*asq;
    if (nscal) dWm[5] = da[5];

    for (n = 0; n < nv; n++) {                                          /* :213-240, beta=1 */
      Wlv[n] = w[n] - 0.5*dWm[n]*1.0;
      Wrv[n] = w[n] + 0.5*dWm[n]*1.0;
    }
    for (n = 0; n < nv; n++) {
      C = Wrv[n] + 1.0*Wlv[n];
      Wlv[n] = MAXR(MINR(w[n], wm[n]), Wlv[n]);
      Wlv[n] = MINR(MAXR(w[n], wm[n]), Wlv[n]);
      Wrv[n] = C - 1.0*Wlv[n];
      Wrv[n] = MAXR(MINR(w[n], wp[n]), Wrv[n]);
      Wrv[n] = MINR(MAXR(w[n], wp[n]), Wrv[n]);
      Wlv[n] = (C - Wrv[n])*1.0;
    }
    for (n = 0; n < nv; n++) dW[n] = Wrv[n] - Wlv[n];

    if (!trace) {                                                       /* VL_INTEGRATOR, :250-255 */
      for (n = 0; n < nv; n++) { pWl[n] = Wrv[n]; pWr[n] = Wlv[n]; }
      if (!nscal) { pWl[5] = 0.0; pWr[5] = 0.0; }
      continue;
    }
    qx = 0.5*MAXR(ev4, 0.0)*dtodx;                                      /* :296-313 */
    for (n = 0; n < nv; n++) pWl[n] = Wrv[n] - qx*dW[n];
    qx = -0.5*MINR(ev0, 0.0)*dtodx;
    for (n = 0; n < nv; n++) pWr[n] = Wlv[n] + qx*dW[n];
    if (!nscal) { pWl[5] = 0.0; pWr[5] = 0.0; }

    /* :322-339: waves with ev>=0 that do not reach the right interface */
    qx1 = 0.5*dtodx*ev4;
    if (ev0 >= 0.0) {
      qx2 = 0.5*dtodx*ev0; qx = qx1 - qx2;
      qa = 0.0; qa += l01*qx*dW[1]; qa += l04*qx*dW[4];
      pWl[0] += qa; pWl[1] += qa*r10; pWl[4] += qa*asq;
    }
    if (vx >= 0.0) {
      qx2 = 0.5*dtodx*vx; qx = qx1 - qx2;
      qa = 0.0; qa += 1.0*qx*dW[0]; qa += l14*qx*dW[4];  pWl[0] += qa;
      qa = 0.0; qa += 1.0*qx*dW[2];                      pWl[2] += qa;
      qa = 0.0; qa += 1.0*qx*dW[3];                      pWl[3] += qa;
    }
    /* n=4: qx = qx1-qx1 = 0 adds exact zeros */
    /* :341-358: waves with ev<=0 that do not reach the left interface */
    qx1 = -0.5*dtodx*ev0;
    /* n=0: qx = 0 adds exact zeros */
    if (vx <= 0.0) {
      qx2 = -0.5*dtodx*vx; qx = -qx1 + qx2;
      qa = 0.0; qa += 1.0*qx*dW[0]; qa += l14*qx*dW[4];  pWr[0] += qa;
      qa = 0.0; qa += 1.0*qx*dW[2];                      pWr[2] += qa;
      qa = 0.0; qa += 1.0*qx*dW[3];                      pWr[3] += qa;
    }
    if (ev4 <= 0.0) {
      qx2 = -0.5*dtodx*ev4; qx = -qx1 + qx2;
      qa = 0.0; qa += l41*qx*dW[1]; qa += l04*qx*dW[4];
      pWr[0] += qa; pWr[1] += qa*r14; pWr[4] += qa*asq;
    }
    if (nscal) {                                                        /* :361-367 */
      if (vx > 0.)      pWl[5] += 0.5*dtodx*(ev4 - vx)*dW[5];
      else if (vx < 0.) pWr[5] += 0.5*dtodx*(ev0 - vx)*dW[5];
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* reconstruction/lr_states_ppm.c:91-610 (THIRD_ORDER_CHAR, --with-order=3): piecewise parabolic
 * interpolation with the same characteristic slope limiting as PLM (Steps 1-6 / 8-13 are
 * lr_states_plm.c:131-202 word for word), parabola monotonisation (CW84 eqn 1.10) and
 * characteristic tracing (eqn 3.5ff).  Needs W over [il-3, iu+3].
 *
 * With passive scalars the reference indexes its work arrays dWm[][] and Wim1h[][] with
 * n < NWAVE+NSCALARS although lr_states_init allocates NWAVE columns (:689-692), so column
 * NWAVE of row i IS column 0 of row i+1.  Followed through the loop order, the scalar sees
 *   left  parabola edge  = the DENSITY interface value W_{i+1/2}[0] just computed,
 *   right parabola edge  = (r_i + r_{i+1})/2 - (dr_{i+1} - d(rho)_{i+1})/6,
 * both then clamped between the neighbouring scalar values by Step 16.  Restated as such. */
#define FOUR_3RDS 1.333333333333333     /* defs.h.in:159 */
#define TWO_3RDS  0.6666666666666667    /* :158 */

static void limited_slopes(const P1 *W, int i, Real Gamma, int nscal, Real dWm[6])
{
  const Real *w = (const Real*)&W[i], *wm = (const Real*)&W[i-1], *wp = (const Real*)&W[i+1];
  const int nv = NW + nscal; int n;
  Real d = W[i].d, asq = (Gamma*W[i].P)/d, a = sqrt(asq);
  Real r10 = -a/d, r14 = -r10, l01 = -0.5*d/a, l04 = 0.5/asq, l14 = -1.0/asq, l41 = -l01;
  Real dWc[6], dWl[6], dWr[6], dWg[6], dac[6], dal[6], dar[6], dag[6], da[6], lim1, lim2;
  for (n = 0; n < nv; n++) {
    dWc[n] = wp[n] - wm[n]; dWl[n] = w[n] - wm[n]; dWr[n] = wp[n] - w[n];
    dWg[n] = (dWl[n]*dWr[n] > 0.0) ? 2.0*dWl[n]*dWr[n]/(dWl[n]+dWr[n]) : 0.0;
  }
#define PROJ(o,x) do { o[0] = l01*x[1]; o[0] += l04*x[4]; o[1] = x[0]; o[1] += l14*x[4]; \
                       o[2] = x[2]; o[3] = x[3]; o[4] = l41*x[1]; o[4] += l04*x[4]; \
                       if (nscal) o[5] = x[5]; } while (0)
  PROJ(dac,dWc); PROJ(dal,dWl); PROJ(dar,dWr); PROJ(dag,dWg);
#undef PROJ
  for (n = 0; n < nv; n++) {
    da[n] = 0.0;
    if (dal[n]*dar[n] > 0.0) {
      lim1 = MINR(fabs(dal[n]), fabs(dar[n]));
      lim2 = MINR(0.5*fabs(dac[n]), fabs(dag[n]));
      da[n] = ((dac[n] < 0.) ? -1. : 1.)*MINR(2.0*lim1, lim2);
    }
  }
  dWm[0] = da[0]; dWm[0] += da[1]; dWm[0] += da[4];
  dWm[1] = da[0]*r10; dWm[1] += da[4]*r14;
  dWm[2] = da[2]; dWm[3] = da[3];
  dWm[4] = da[0]*asq; dWm[4] += da[4]*asq;
  dWm[5] = nscal ? da[5] : 0.0;
}

static void lr_states_ppm(const P1 *W, Real dt, Real dx, int il, int iu, P1 *Wl, P1 *Wr, Real Gamma, int nscal, int trace)
{
  const Real dtodx = dt/dx;
  const int nv = NW + nscal;
  int i, n;
  for (i = il-1; i <= iu+1; i++) {
    const Real *w = (const Real*)&W[i], *wm = (const Real*)&W[i-1], *wp = (const Real*)&W[i+1];
    Real d = W[i].d, vx = W[i].Vx, asq = (Gamma*W[i].P)/d, a = sqrt(asq);
    Real ev0 = vx - a, ev4 = vx + a;
    Real r10 = -a/d, r14 = -r10, l01 = -0.5*d/a, l04 = 0.5/asq, l14 = -1.0/asq, l41 = -l01;
    Real Dm[6], D0[6], Dp[6], Wlv[6], Wrv[6], dW[6], W6[6], *pWl = (Real*)&Wl[i+1], *pWr = (Real*)&Wr[i];
    Real qa, qb, qc, qx1, qx2, qxx1 = 0.0, qxx2 = 0.0;
    const Real gamma_curv = 0.0;
    limited_slopes(W, i-1, Gamma, nscal, Dm); limited_slopes(W, i, Gamma, nscal, D0); limited_slopes(W, i+1, Gamma, nscal, Dp);
    for (n = 0; n < NW; n++) {                                          /* Steps 7/14 :258-263, :398-408 */
      Wlv[n] = 0.5*(w[n] + wm[n]) - (D0[n] - Dm[n])/6.0;
      Wrv[n] = 0.5*(wp[n] + w[n]) - (Dp[n] - D0[n])/6.0;
    }
    if (nscal) {                                                        /* the aliased column, see above */
      Wlv[5] = Wrv[0];
      Wrv[5] = 0.5*(wp[5] + w[5]) - (Dp[5] - Dp[0])/6.0;
    }
    for (n = 0; n < nv; n++) {                                          /* Step 16 :421-446 */
      qa = (Wrv[n]-w[n])*(w[n]-Wlv[n]);
      qb = Wrv[n]-Wlv[n];
      qc = 6.0*(w[n] - 0.5*(Wlv[n]*(1.0-gamma_curv) + Wrv[n]*(1.0+gamma_curv)));
      if (qa <= 0.0) { Wlv[n] = w[n]; Wrv[n] = w[n]; }
      else if ((qb*qc) > (qb*qb)) Wlv[n] = (6.0*w[n] - Wrv[n]*(4.0+3.0*gamma_curv))/(2.0-3.0*gamma_curv);
      else if ((qb*qc) < -(qb*qb)) Wrv[n] = (6.0*w[n] - Wlv[n]*(4.0-3.0*gamma_curv))/(2.0+3.0*gamma_curv);
    }
    for (n = 0; n < nv; n++) {
      Wlv[n] = MAXR(MINR(w[n],wm[n]),Wlv[n]);
      Wlv[n] = MINR(MAXR(w[n],wm[n]),Wlv[n]);
      Wrv[n] = MAXR(MINR(w[n],wp[n]),Wrv[n]);
      Wrv[n] = MINR(MAXR(w[n],wp[n]),Wrv[n]);
    }
    for (n = 0; n < nv; n++) {                                          /* Step 17 :451-455 */
      dW[n] = Wrv[n] - Wlv[n];
      W6[n] = 6.0*(w[n] - 0.5*(Wlv[n]*(1.0-gamma_curv) + Wrv[n]*(1.0+gamma_curv)));
    }
    if (!trace) {                                                       /* integrators other than CTU (VL), :502-507 */
      for (n = 0; n < nv; n++) { pWl[n] = Wrv[n]; pWr[n] = Wlv[n]; }
      if (!nscal) { pWl[5] = 0.0; pWr[5] = 0.0; }
      continue;
    }
    qx1 = 0.5*MAXR(ev4,0.0)*dtodx;                                      /* Step 18 :461-500 */
    for (n = 0; n < nv; n++)
      pWl[n] = Wrv[n] - qx1 *(dW[n] - (1.0-FOUR_3RDS*qx1)*W6[n])
                      + qxx1*(dW[n] - (1.0-      2.0*qx1)*W6[n]);
    qx2 = -0.5*MINR(ev0,0.0)*dtodx;
    for (n = 0; n < nv; n++)
      pWr[n] = Wlv[n] + qx2 *(dW[n] + (1.0-FOUR_3RDS*qx2)*W6[n])
                      + qxx2*(dW[n] + (1.0-      2.0*qx2)*W6[n]);
    if (!nscal) { pWl[5] = 0.0; pWr[5] = 0.0; }
    /* Step 19 :508-560, sparse form of the loops over the eigenmatrices (wave n=4 / n=0 of the first /
     * second block have qb = qc = 0 and add exact zeros) */
#define TERML(m) (qb*(dW[m]-W6[m]) + qc*W6[m])
#define TERMR(m) (qb*(dW[m]+W6[m]) + qc*W6[m])
    qx1 = 0.5*dtodx*ev4;
    if (ev0 >= 0.0) {
      qx2 = 0.5*dtodx*ev0; qb = qx1 - qx2; qc = FOUR_3RDS*(SQR(qx1) - SQR(qx2));
      qa = 0.0; qa += l01*TERML(1); qa += l04*TERML(4);
      pWl[0] += qa; pWl[1] += qa*r10; pWl[4] += qa*asq;
    }
    if (vx >= 0.0) {
      qx2 = 0.5*dtodx*vx; qb = qx1 - qx2; qc = FOUR_3RDS*(SQR(qx1) - SQR(qx2));
      qa = 0.0; qa += 1.0*TERML(0); qa += l14*TERML(4);  pWl[0] += qa;
      qa = 0.0; qa += 1.0*TERML(2);                      pWl[2] += qa;
      qa = 0.0; qa += 1.0*TERML(3);                      pWl[3] += qa;
    }
    qx1 = 0.5*dtodx*ev0;
    if (vx <= 0.0) {
      qx2 = 0.5*dtodx*vx; qb = qx1 - qx2; qc = FOUR_3RDS*(SQR(qx1) - SQR(qx2));
      qa = 0.0; qa += 1.0*TERMR(0); qa += l14*TERMR(4);  pWr[0] += qa;
      qa = 0.0; qa += 1.0*TERMR(2);                      pWr[2] += qa;
      qa = 0.0; qa += 1.0*TERMR(3);                      pWr[3] += qa;
    }
    if (ev4 <= 0.0) {
      qx2 = 0.5*dtodx*ev4; qb = qx1 - qx2; qc = FOUR_3RDS*(SQR(qx1) - SQR(qx2));
      qa = 0.0; qa += l41*TERMR(1); qa += l04*TERMR(4);
      pWr[0] += qa; pWr[1] += qa*r14; pWr[4] += qa*asq;
    }
    if (nscal) {                                                        /* :563-575 (dW[m], W6[m] with m = NWAVE) */
      if (vx > 0.) {
        qb = 0.5*dtodx*(ev4-vx);
        qc = 0.5*dtodx*dtodx*TWO_3RDS*(SQR(ev4) - SQR(vx));
        pWl[5] += qb*(dW[5]-W6[5]) + qc*W6[5];
      } else if (vx < 0.) {
        qb = 0.5*dtodx*(ev0-vx);
        qc = 0.5*dtodx*dtodx*TWO_3RDS*(ev0*ev0 - vx*vx);
        pWr[5] += qb*(dW[5]+W6[5]) + qc*W6[5];
      }
    }
#undef TERML
#undef TERMR
  }
}

/* ------------------------------------------------------------------------------------ */
/* microphysics/cool.c:34-86 KoyInut(): analytic fit to the cooling of the diffuse ISM (Koyama & Inutsuka 2002, eq. 4); the
 * only cooling function the reference ships (enrolled as CoolingFunc = KoyInut, prob/ti.c:474).  cgs constants :35-38. */
static Real cool_koyinut(Real dens, Real Press, Real dt, Real Gamma_1)
{
  const Real mbar = (1.37)*(1.6733e-24), kb = 1.380658e-16, HeatRate = 2.0e-26, Tmin = 10;
  Real n, coolrate = 0.0, T, coolratepp, MaxdT, dT, Teq, logn, lognT;
  n = dens/mbar;
  logn = log10(n);
  T = MAXR((Press/(n*kb)), Tmin);
  Teq = Tmin;
  coolratepp = HeatRate*
   (n*(1.0e7*exp(-1.184e5/(T+1000.)) + 0.014*sqrt(T)*exp(-92.0/T)) - 1.0);
  dT = coolratepp*dt*Gamma_1/kb;
  if ((T-dT) <= 185.0){
    lognT = 3.9247499 - 1.8479378*logn + 1.5335032*logn*logn
     -0.47665872*pow(logn,3) + 0.076789136*pow(logn,4)-0.0049052587*pow(logn,5);
    Teq = pow(10.0,lognT) / n;
  }
  MaxdT = kb*(T-Teq)/(dt*Gamma_1);
  coolrate = MINR(coolratepp,MaxdT);
  return n*coolrate;
}
void orc_set_cooling(OrcSim *s, int kind) { s->cool = kind; }

/* static gravitational potential: prob/ioniz_sphere.c:316-330 (PlanetPot, non-shearing-box) */

static Real potential(const OrcSim *s, Real x1, Real x2, Real x3)
{
  Real rad = sqrt(SQR(x1)+SQR(x2)+SQR(x3));
  Real adist = 7.48e11;
  Real GMstar = 6.67e-8 * 1.99e33;
  Real omega = sqrt(GMstar / (pow(adist,3)));
  Real radstar = sqrt(SQR(x1+adist) + SQR(x2) + SQR(x3));
  Real rcentrif = sqrt(SQR(x1+adist) + SQR(x2));
  return -s->p.pot_GM/(rad+s->p.pot_Rsoft)-GMstar/radstar -.5*SQR(omega*rcentrif);
}

static void cc_pos(const OrcSim *s, int i, int j, int k, Real x[3])   /* cc_pos.c:36-43 */
{
  x[0] = s->p.MinX[0] + ((Real)(i - s->is) + 0.5)*s->dx[0];
  x[1] = s->p.MinX[1] + ((Real)(j - s->js) + 0.5)*s->dx[1];
  x[2] = s->p.MinX[2] + ((Real)(k - s->ks) + 0.5)*s->dx[2];
}

/* Phi at cell centre of (i,j,k) displaced by a[] cell widths (a in {-1,-0.5,0,0.5}),
 * evaluated as the reference does: (x - dx), (x - 0.5*dx), (x + 0.5*dx). */
static Real phi_at(const OrcSim *s, const Real x[3], int d0, Real a0, int d1, Real a1)
{
  Real y[3]; y[0] = x[0]; y[1] = x[1]; y[2] = x[2];
  if (a0 == -1.0) y[d0] = x[d0] - s->dx[d0];
  else if (a0 == -0.5) y[d0] = x[d0] - 0.5*s->dx[d0];
  else if (a0 == 0.5) y[d0] = x[d0] + 0.5*s->dx[d0];
  if (a1 == -1.0) y[d1] = x[d1] - s->dx[d1];
  else if (a1 == -0.5) y[d1] = x[d1] - 0.5*s->dx[d1];
  else if (a1 == 0.5) y[d1] = x[d1] + 0.5*s->dx[d1];
  return potential(s, y[0], y[1], y[2]);
}

/* ------------------------------------------------------------------------------------ */
/* rotations between the global frame and the sweep frame of direction d:
 * (Mx,My,Mz) = (M[d], M[d+1], M[d+2])  -- integrate_3d_ctu.c:206-208, :544-546, :727-729 */

static C1 to_sweep(const Cons *u, int d)
{ C1 c; c.d = u->d; c.Mx = u->M[d]; c.My = u->M[(d+1)%3]; c.Mz = u->M[(d+2)%3]; c.E = u->E; c.s = u->s; return c; }
static Cons from_sweep(const C1 *c, int d)
{ Cons u; u.d = c->d; u.M[d] = c->Mx; u.M[(d+1)%3] = c->My; u.M[(d+2)%3] = c->Mz; u.E = c->E; u.s = c->s; return u; }

/* ------------------------------------------------------------------------------------ */
/* 3-D CTU integrator: integrators/integrate_3d_ctu.c:110-3368 */

static void integrate_vl(OrcSim *s);

void orc_integrate(OrcSim *s)
{
  const int nscal = s->p.nscal, nv = NW + nscal;
  const Real Gamma = s->Gamma, Gamma_1 = s->Gamma_1, dt = s->dt;
  const int N0 = s->N[0], N1 = s->N[1];
  const size_t str[3] = {1, (size_t)N0, (size_t)N0*N1};
  const int lo[3] = {s->is, s->js, s->ks}, hi[3] = {s->ie, s->je, s->ke};
  int l[3], u[3];      /* il..iu etc. = s-2 .. e+2 (:179-184) */
  Real dtodx[3], q[3];
  int nmax = MAXR(MAXR(s->N[0], s->N[1]), s->N[2]);
  P1 *W = (P1*)malloc(nmax*sizeof(P1)), *Wl = (P1*)malloc(nmax*sizeof(P1)), *Wr = (P1*)malloc(nmax*sizeof(P1));
  int d, e, n, i, j, k;
  const int grav = (s->p.pot != 0);

  if (s->p.integrator == 1) { free(W); free(Wl); free(Wr); integrate_vl(s); return; }
  for (d = 0; d < 3; d++) {
    l[d] = lo[d] - 2; u[d] = hi[d] + 2;
    dtodx[d] = dt/s->dx[d]; q[d] = 0.5*dtodx[d];
  }

  /* === Steps 1-3: L/R states and first-pass fluxes in each direction (:202-897) === */
  for (d = 0; d < 3; d++) {
    const int d1 = (d+1)%3, d2 = (d+2)%3;   /* the two transverse directions */
    int a, b, c, idx[3];
    /* transverse loop extents: x1 sweep: k,j in [l,u]; x2 sweep: k in [kl,ku], i in [il,iu];
       x3 sweep: j,i in [l,u] -- i.e. always [l,u] in both transverse directions */
    for (a = l[d2]; a <= u[d2]; a++) for (b = l[d1]; b <= u[d1]; b++) {
      size_t base;
      idx[d1] = b; idx[d2] = a; idx[d] = 0;
      base = IDX(s, idx[2], idx[1], idx[0]);
      for (c = lo[d]-NGHOST; c <= hi[d]+NGHOST; c++) {
        C1 u1 = to_sweep(&s->U[base + c*str[d]], d);
        W[c] = cons_to_prim(&u1, Gamma_1, nscal);
      }
      lr_states(W, dt, s->dx[d], l[d]+1, u[d]-1, Wl, Wr, Gamma, nscal, s->p.order);
      if (grav) {                                                      /* :318-342 etc. */
        for (c = l[d]+1; c <= u[d]; c++) {
          Real x[3], phicr, phicl, phifc;
          idx[d] = c; cc_pos(s, idx[0], idx[1], idx[2], x);
          phicr = phi_at(s, x, d, 0.0, d, 0.0);
          phicl = phi_at(s, x, d, -1.0, d, 0.0);
          phifc = phi_at(s, x, d, -0.5, d, 0.0);
          Wl[c].Vx -= dtodx[d]*(phifc - phicl);
          Wr[c].Vx -= dtodx[d]*(phicr - phifc);
        }
      }
      if (s->cool) {                                                   /* :359-368, :662-671, :846-855 */
        for (c = l[d]+1; c <= u[d]; c++) {
          Real coolfl = cool_koyinut(Wl[c].d, Wl[c].P, (0.5*dt), Gamma_1);
          Real coolfr = cool_koyinut(Wr[c].d, Wr[c].P, (0.5*dt), Gamma_1);
          Wl[c].P -= 0.5*dt*Gamma_1*coolfl;
          Wr[c].P -= 0.5*dt*Gamma_1*coolfr;
        }
      }
      for (c = l[d]+1; c <= u[d]; c++) {                               /* :515-524 */
        C1 ul = prim_to_cons(&Wl[c], Gamma_1, nscal), ur = prim_to_cons(&Wr[c], Gamma_1, nscal), f;
        size_t m = base + c*str[d];
        flux_roe(&ul, &ur, &Wl[c], &Wr[c], 0.0, Gamma, Gamma_1, nscal, &f);
        s->Ul[d][m] = from_sweep(&ul, d); s->Ur[d][m] = from_sweep(&ur, d); s->F[d][m] = from_sweep(&f, d);
      }
    }
  }

  /* === Steps 5-7: transverse flux-gradient corrections of the face states (:978-1937) === */
  for (d = 0; d < 3; d++) {
    int rl[3], ru[3];
    for (e = 0; e < 3; e++) { rl[e] = l[e]+1; ru[e] = (e == d) ? u[e] : u[e]-1; }
    for (k = rl[2]; k <= ru[2]; k++) for (j = rl[1]; j <= ru[1]; j++) for (i = rl[0]; i <= ru[0]; i++) {
      size_t m = IDX(s,k,j,i), ml = m - str[d];
      Real *pl = (Real*)&s->Ul[d][m], *pr = (Real*)&s->Ur[d][m];
      for (e = 0; e < 3; e++) if (e != d) {                    /* ascending e, as the reference */
        const Real *fl0 = (const Real*)&s->F[e][ml], *fl1 = (const Real*)&s->F[e][ml + str[e]];
        const Real *fr0 = (const Real*)&s->F[e][m],  *fr1 = (const Real*)&s->F[e][m + str[e]];
        for (n = 0; n < NW; n++) pl[n] -= q[e]*(fl1[n] - fl0[n]);
        for (n = 0; n < NW; n++) pr[n] -= q[e]*(fr1[n] - fr0[n]);
        if (nscal) { pl[5] -= q[e]*(fl1[5] - fl0[5]); pr[5] -= q[e]*(fr1[5] - fr0[5]); }
      }
    }
    if (grav) {                                                /* :1167-1219, :1463-1526, :1873-1937 */
      for (k = rl[2]; k <= ru[2]; k++) for (j = rl[1]; j <= ru[1]; j++) for (i = rl[0]; i <= ru[0]; i++) {
        size_t m = IDX(s,k,j,i), ml = m - str[d];
        Real x[3], phic, phir, phil;
        cc_pos(s, i, j, k, x);
        /* right state: cell m */
        phic = phi_at(s, x, d, 0.0, d, 0.0);
        for (e = 0; e < 3; e++) if (e != d) {
          phir = phi_at(s, x, e, 0.5, e, 0.0);
          phil = phi_at(s, x, e, -0.5, e, 0.0);
          s->Ur[d][m].M[e] -= q[e]*(phir-phil)*s->U[m].d;
          s->Ur[d][m].E -= q[e]*(s->F[e][m].d*(phic - phil) + s->F[e][m + str[e]].d*(phir - phic));
        }
        /* left state: cell m-1 along d, potentials at (x_d - dx_d) */
        phic = phi_at(s, x, d, -1.0, d, 0.0);
        for (e = 0; e < 3; e++) if (e != d) {
          phir = phi_at(s, x, d, -1.0, e, 0.5);
          phil = phi_at(s, x, d, -1.0, e, -0.5);
          s->Ul[d][m].M[e] -= q[e]*(phir-phil)*s->U[ml].d;
          s->Ul[d][m].E -= q[e]*(s->F[e][ml].d*(phic - phil) + s->F[e][ml + str[e]].d*(phir - phic));
        }
      }
    }
  }

  /* === Step 8a: d^{n+1/2} (:2104-2125) === */
  if (grav || s->cool) {
    for (k = l[2]+1; k <= u[2]-1; k++) for (j = l[1]+1; j <= u[1]-1; j++) for (i = l[0]+1; i <= u[0]-1; i++) {
      size_t m = IDX(s,k,j,i);
      s->dhalf[m] = s->U[m].d
        - q[0]*(s->F[0][m + str[0]].d - s->F[0][m].d)
        - q[1]*(s->F[1][m + str[1]].d - s->F[1][m].d)
        - q[2]*(s->F[2][m + str[2]].d - s->F[2][m].d);
    }
  }

  /* === Step 8b: P^{n+1/2}, needed with cooling (:2133-2266) === */
  if (s->cool) {
    for (k = l[2]+1; k <= u[2]-1; k++) for (j = l[1]+1; j <= u[1]-1; j++) for (i = l[0]+1; i <= u[0]-1; i++) {
      size_t m = IDX(s,k,j,i);
      Real Mh[3], Eh;
      for (e = 0; e < 3; e++)
        Mh[e] = s->U[m].M[e]
          - q[0]*(s->F[0][m + str[0]].M[e] - s->F[0][m].M[e])
          - q[1]*(s->F[1][m + str[1]].M[e] - s->F[1][m].M[e])
          - q[2]*(s->F[2][m + str[2]].M[e] - s->F[2][m].M[e]);
      Eh = s->U[m].E
        - q[0]*(s->F[0][m + str[0]].E - s->F[0][m].E)
        - q[1]*(s->F[1][m + str[1]].E - s->F[1][m].E)
        - q[2]*(s->F[2][m + str[2]].E - s->F[2][m].E);
      if (grav) {
        Real x[3], phir, phil;
        cc_pos(s, i, j, k, x);
        for (e = 0; e < 3; e++) {
          phir = phi_at(s, x, e, 0.5, e, 0.0);
          phil = phi_at(s, x, e, -0.5, e, 0.0);
          Mh[e] -= q[e]*(phir-phil)*s->U[m].d;
        }
      }
      s->phalf[m] = Eh - 0.5*(Mh[0]*Mh[0] + Mh[1]*Mh[1] + Mh[2]*Mh[2])/s->dhalf[m];
      s->phalf[m] *= Gamma_1;
    }
  }

  /* === Step 9a: eta for the H-correction (:2300-2343) === */
  /* (a build without --enable-h-correction, the configure default, has neither the eta arrays nor the etah argument: roe.c
   *  :282-290 then uses |ev| where the H_CORRECTION build uses MAX(|ev|, etah) -- the same numbers as etah = 0) */
  if (s->p.integrator == 2) {
    const size_t nc0 = (size_t)s->N[0]*s->N[1]*s->N[2];
    for (d = 0; d < 3; d++) memset(s->eta[d], 0, nc0*sizeof(Real));
  } else
  for (d = 0; d < 3; d++) {
    int rl[3], ru[3];
    for (e = 0; e < 3; e++) { rl[e] = lo[e]-1; ru[e] = hi[e] + ((e == d) ? 2 : 1); }
    for (k = rl[2]; k <= ru[2]; k++) for (j = rl[1]; j <= ru[1]; j++) for (i = rl[0]; i <= ru[0]; i++) {
      size_t m = IDX(s,k,j,i);
      C1 ur = to_sweep(&s->Ur[d][m], d), ul = to_sweep(&s->Ul[d][m], d);
      Real cfr = cfast_c1(&ur, Gamma, Gamma_1), cfl = cfast_c1(&ul, Gamma, Gamma_1);
      Real lambdar = ur.Mx/ur.d + cfr, lambdal = ul.Mx/ul.d - cfl;
      s->eta[d][m] = 0.5*fabs(lambdar - lambdal);
    }
  }

  /* === Steps 9b-d: second-pass fluxes with etah (:2350-2437) === */
  for (d = 0; d < 3; d++) {
    /* the two transverse directions in ASCENDING order: the reference's MAX chain
       (a > b ? a : b) is order-sensitive when an eta is NaN (negative face pressure) */
    const int d1 = (d == 0) ? 1 : 0, d2 = (d == 2) ? 1 : 2;
    int rl[3], ru[3];
    for (e = 0; e < 3; e++) { rl[e] = (e == d) ? lo[e] : lo[e]-1; ru[e] = hi[e]+1; }
    for (k = rl[2]; k <= ru[2]; k++) for (j = rl[1]; j <= ru[1]; j++) for (i = rl[0]; i <= ru[0]; i++) {
      size_t m = IDX(s,k,j,i), ml = m - str[d];
      Real etah;
      C1 ul, ur, f; P1 wl, wr;
      etah = MAXR(s->eta[d1][ml], s->eta[d1][m]);
      etah = MAXR(etah, s->eta[d1][ml + str[d1]]);
      etah = MAXR(etah, s->eta[d1][m  + str[d1]]);
      etah = MAXR(etah, s->eta[d2][ml]);
      etah = MAXR(etah, s->eta[d2][m]);
      etah = MAXR(etah, s->eta[d2][ml + str[d2]]);
      etah = MAXR(etah, s->eta[d2][m  + str[d2]]);
      etah = MAXR(etah, s->eta[d][m]);
      ul = to_sweep(&s->Ul[d][m], d); ur = to_sweep(&s->Ur[d][m], d);
      wl = cons_to_prim(&ul, Gamma_1, nscal); wr = cons_to_prim(&ur, Gamma_1, nscal);
      flux_roe(&ul, &ur, &wl, &wr, etah, Gamma, Gamma_1, nscal, &f);
      s->F[d][m] = from_sweep(&f, d);
    }
  }

  /* === Step 11a: gravity source terms for the full step (:2741-2782) === */
  if (grav) {
    for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
      size_t m = IDX(s,k,j,i);
      Real x[3], phic, phir, phil;
      cc_pos(s, i, j, k, x);
      phic = phi_at(s, x, 0, 0.0, 0, 0.0);
      for (e = 0; e < 3; e++) {
        phir = phi_at(s, x, e, 0.5, e, 0.0);
        phil = phi_at(s, x, e, -0.5, e, 0.0);
        s->U[m].M[e] -= dtodx[e]*(phir-phil)*s->dhalf[m];
        s->U[m].E -= dtodx[e]*(s->F[e][m].d*(phic - phil) + s->F[e][m + str[e]].d*(phir - phic));
      }
    }
  }

  /* === Step 11c: optically thin cooling for the full step (:2943-2953) === */
  if (s->cool) {
    for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
      size_t m = IDX(s,k,j,i);
      Real coolf = cool_koyinut(s->dhalf[m], s->phalf[m], dt, Gamma_1);
      s->U[m].E -= dt*coolf;
    }
  }

  /* === Step 12: conservative update, one direction after the other (:2981-3050) === */
  for (d = 0; d < 3; d++) {
    for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
      size_t m = IDX(s,k,j,i);
      Real *pu = (Real*)&s->U[m];
      const Real *f0 = (const Real*)&s->F[d][m], *f1 = (const Real*)&s->F[d][m + str[d]];
      for (n = 0; n < nv; n++) pu[n] -= dtodx[d]*(f1[n] - f0[n]);
    }
  }
  free(W); free(Wl); free(Wr);
}


/* ------------------------------------------------------------------------------------ */
/* 3-D van Leer integrator: integrators/integrate_3d_vl.c:96-, NO_H_CORRECTION */

static void integrate_vl(OrcSim *s)
{
  const int nscal = s->p.nscal, nv = NW + nscal;
  const Real Gamma = s->Gamma, Gamma_1 = s->Gamma_1, dt = s->dt;
  const int N0 = s->N[0], N1 = s->N[1];
  const size_t str[3] = {1, (size_t)N0, (size_t)N0*N1};
  const int lo[3] = {s->is, s->js, s->ks}, hi[3] = {s->ie, s->je, s->ke};
  const size_t nc = (size_t)s->N[0]*s->N[1]*s->N[2];
  Cons *Uh = s->Ul[0];                  /* Uhalf lives in an unused face array */
  Real dtodx[3], q[3];
  int nmax = MAXR(MAXR(s->N[0], s->N[1]), s->N[2]);
  P1 *W = (P1*)malloc(nmax*sizeof(P1)), *Wl = (P1*)malloc(nmax*sizeof(P1)), *Wr = (P1*)malloc(nmax*sizeof(P1));
  int d, e, n, i, j, k;
  const int grav = (s->p.pot != 0);
  for (d = 0; d < 3; d++) { dtodx[d] = dt/s->dx[d]; q[d] = 0.5*dtodx[d]; }
  memcpy(Uh, s->U, nc*sizeof(Cons));                                   /* :132-143 */

  /* steps 1-3: first-order (donor-cell) fluxes over the whole ghost range (:153-305) */
  for (d = 0; d < 3; d++) {
    const int d1 = (d+1)%3, d2 = (d+2)%3;
    int a, b, c, idx[3];
    for (a = lo[d2]-NGHOST; a <= hi[d2]+NGHOST; a++) for (b = lo[d1]-NGHOST; b <= hi[d1]+NGHOST; b++) {
      size_t base;
      idx[d1] = b; idx[d2] = a; idx[d] = 0; base = IDX(s, idx[2], idx[1], idx[0]);
      for (c = lo[d]-NGHOST; c <= hi[d]+NGHOST; c++) {
        C1 u1 = to_sweep(&s->U[base + c*str[d]], d);
        W[c] = cons_to_prim(&u1, Gamma_1, nscal);
      }
      for (c = lo[d]-(NGHOST-1); c <= hi[d]+NGHOST; c++) {
        C1 ul, ur, f;
        Wl[c] = W[c-1]; Wr[c] = W[c];
        ul = prim_to_cons(&Wl[c], Gamma_1, nscal); ur = prim_to_cons(&Wr[c], Gamma_1, nscal);
        flux_roe(&ul, &ur, &Wl[c], &Wr[c], 0.0, Gamma, Gamma_1, nscal, &f);
        s->F[d][base + c*str[d]] = from_sweep(&f, d);
      }
    }
  }
  /* step 5: half-step update over [s-3, e+3]^3, x1 then x2 then x3 (:392-470) */
  for (d = 0; d < 3; d++)
    for (k = lo[2]-3; k <= hi[2]+3; k++) for (j = lo[1]-3; j <= hi[1]+3; j++) for (i = lo[0]-3; i <= hi[0]+3; i++) {
      size_t m = IDX(s,k,j,i);
      Real *pu = (Real*)&Uh[m];
      const Real *f0 = (const Real*)&s->F[d][m], *f1 = (const Real*)&s->F[d][m + str[d]];
      for (n = 0; n < nv; n++) pu[n] -= q[d]*(f1[n] - f0[n]);
    }
  /* step 6a: gravity predictor (:480-510) */
  if (grav)
    for (k = lo[2]-3; k <= hi[2]+3; k++) for (j = lo[1]-3; j <= hi[1]+3; j++) for (i = lo[0]-3; i <= hi[0]+3; i++) {
      size_t m = IDX(s,k,j,i);
      Real x[3], phic, phir, phil;
      cc_pos(s, i, j, k, x);
      phic = phi_at(s, x, 0, 0.0, 0, 0.0);
      for (e = 0; e < 3; e++) {
        phir = phi_at(s, x, e, 0.5, e, 0.0);
        phil = phi_at(s, x, e, -0.5, e, 0.0);
        Uh[m].M[e] -= q[e]*(phir-phil)*s->U[m].d;
        Uh[m].E -= q[e]*(s->F[e][m].d*(phic - phil) + s->F[e][m + str[e]].d*(phir - phic));
      }
    }
  /* steps 7-10: PLM (no tracing) on Uhalf and second-order fluxes, etah = 0 (:560-795) */
  for (d = 0; d < 3; d++) {
    const int d1 = (d+1)%3, d2 = (d+2)%3;
    int a, b, c, idx[3];
    for (a = lo[d2]-1; a <= hi[d2]+1; a++) for (b = lo[d1]-1; b <= hi[d1]+1; b++) {
      size_t base;
      idx[d1] = b; idx[d2] = a; idx[d] = 0; base = IDX(s, idx[2], idx[1], idx[0]);
      for (c = lo[d]-3; c <= hi[d]+3; c++) {
        C1 u1 = to_sweep(&Uh[base + c*str[d]], d);
        W[c] = cons_to_prim(&u1, Gamma_1, nscal);
      }
      if (s->p.order == 3) lr_states_ppm(W, dt, s->dx[d], lo[d], hi[d], Wl, Wr, Gamma, nscal, 0);   /* --with-order=3 */
      else lr_states_x(W, dt, s->dx[d], lo[d], hi[d], Wl, Wr, Gamma, nscal, 0);
      for (c = lo[d]; c <= hi[d]+1; c++) {
        C1 ul = prim_to_cons(&Wl[c], Gamma_1, nscal), ur = prim_to_cons(&Wr[c], Gamma_1, nscal), f;
        flux_roe(&ul, &ur, &Wl[c], &Wr[c], 0.0, Gamma, Gamma_1, nscal, &f);
        s->F[d][base + c*str[d]] = from_sweep(&f, d);       /* second-order flux replaces the first-order one */
      }
    }
  }
  /* step 12a: gravity with d^{n+1/2} = Uhalf.d (:832-860) */
  if (grav)
    for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
      size_t m = IDX(s,k,j,i);
      Real x[3], phic, phir, phil;
      cc_pos(s, i, j, k, x);
      phic = phi_at(s, x, 0, 0.0, 0, 0.0);
      for (e = 0; e < 3; e++) {
        phir = phi_at(s, x, e, 0.5, e, 0.0);
        phil = phi_at(s, x, e, -0.5, e, 0.0);
        s->U[m].M[e] -= dtodx[e]*(phir-phil)*Uh[m].d;
        s->U[m].E -= dtodx[e]*(s->F[e][m].d*(phic - phil) + s->F[e][m + str[e]].d*(phir - phic));
      }
    }
  /* step 13: update (:880-940) */
  for (d = 0; d < 3; d++)
    for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
      size_t m = IDX(s,k,j,i);
      Real *pu = (Real*)&s->U[m];
      const Real *f0 = (const Real*)&s->F[d][m], *f1 = (const Real*)&s->F[d][m + str[d]];
      for (n = 0; n < nv; n++) pu[n] -= dtodx[d]*(f1[n] - f0[n]);
    }
  free(W); free(Wl); free(Wr);
}

/* ------------------------------------------------------------------------------------ */
/* ghost zones: bvals_mhd.c:174 (order x1, x2, x3), reflect :959-1290, outflow :1319-1570,
 * periodic :1637-1790 */

static void bc_fill(OrcSim *s, int d, int side, int flag)
{
  const int N0 = s->N[0], N1 = s->N[1];
  const size_t str[3] = {1, (size_t)N0, (size_t)N0*N1};
  const int lo[3] = {s->is, s->js, s->ks}, hi[3] = {s->ie, s->je, s->ke};
  int rl[3], ru[3], g, a, b, idx[3];
  /* transverse ranges: x1: active j,k; x2: i incl ghosts, active k; x3: i,j incl ghosts */
  for (a = 0; a < 3; a++) {
    if (a < d) { rl[a] = lo[a]-NGHOST; ru[a] = hi[a]+NGHOST; } else { rl[a] = lo[a]; ru[a] = hi[a]; }
  }
  {
    const int d1 = (d+1)%3, d2 = (d+2)%3;
    for (a = rl[d2]; a <= ru[d2]; a++) for (b = rl[d1]; b <= ru[d1]; b++) {
      size_t base;
      idx[d1] = b; idx[d2] = a; idx[d] = 0; base = IDX(s, idx[2], idx[1], idx[0]);
      for (g = 1; g <= NGHOST; g++) {
        int dst, src;
        if (side == 0) {
          dst = lo[d]-g;
          src = (flag == 1) ? lo[d]+(g-1) : (flag == 2) ? lo[d] : hi[d]-(g-1);
        } else {
          dst = hi[d]+g;
          src = (flag == 1) ? hi[d]-(g-1) : (flag == 2) ? hi[d] : lo[d]+(g-1);
        }
        s->U[base + dst*str[d]] = s->U[base + src*str[d]];
        if (flag == 1) s->U[base + dst*str[d]].M[d] = -s->U[base + dst*str[d]].M[d];
      }
    }
  }
}

void orc_bvals(OrcSim *s)
{
  int d;
  for (d = 0; d < 3; d++) {
    if (s->p.Nx[d] > 1) {
      if (s->p.bc[2*d])   bc_fill(s, d, 0, s->p.bc[2*d]);
      if (s->p.bc[2*d+1]) bc_fill(s, d, 1, s->p.bc[2*d+1]);
    }
  }
}

/* one (*BCFun)(pGrid) call of bvals_mhd.c:196-420: for drivers that put an exchange between the directions (pencils) */
void orc_bvals_side(OrcSim *s, int d, int side)
{
  if (s->p.Nx[d] > 1 && s->p.bc[2*d + side]) bc_fill(s, d, side, s->p.bc[2*d + side]);
}

/* bvals_ionrad.c:63 + outflow_flux_ix1 :308 (dir=-1), outflow_flux_ix2 :357 (dir=-2); bvals_ionrad_init :176-232
 * enrols the function of the lit face only */
void orc_bvals_ionrad(OrcSim *s)
{
  int i, j, k, n0 = s->p.Nx[0]+1, n1 = s->p.Nx[1]+1;
  if (!s->p.ion) return;
  if (s->rad_dir == -1)
    for (k = 0; k <= s->p.Nx[2]; k++) for (j = 0; j <= s->p.Nx[1]; j++)
      s->EdgeFlux[((size_t)k*n1 + j)*n0 + 0] = s->flux_i;
  else if (s->rad_dir == -2)
    for (k = 0; k <= s->p.Nx[2]; k++) for (i = 0; i <= s->p.Nx[0]; i++)
      s->EdgeFlux[((size_t)k*n1 + 0)*n0 + i] = s->flux_i;
}

/* ------------------------------------------------------------------------------------ */
/* CFL: new_dt.c:32-198 */

/* new_dt.c:72-140 for one Grid; max_v and max_dti are carried from Grid to Grid (:33, never reset) */
static void cfl_accumulate(OrcSim *s, Real max_v[3], Real *pmax_dti)
{
  const Real Gamma = s->Gamma, Gamma_1 = s->Gamma_1;
  Real max_dti = *pmax_dti;
  int i, j, k, d;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    const Cons *u = &s->U[IDX(s,k,j,i)];
    Real di = 1.0/u->d, v1 = u->M[0]*di, v2 = u->M[1]*di, v3 = u->M[2]*di;
    Real qsq = v1*v1 + v2*v2 + v3*v3;
    Real p = MAXR(Gamma_1*(u->E - 0.5*u->d*qsq), TINY_NUMBER);
    Real asq = Gamma*p*di, v[3];
    v[0] = v1; v[1] = v2; v[2] = v3;
    for (d = 0; d < 3; d++) if (s->p.Nx[d] > 1) max_v[d] = MAXR(max_v[d], fabs(v[d]) + sqrt(asq));
  }
  for (d = 0; d < 3; d++) if (s->p.Nx[d] > 1) max_dti = MAXR(max_dti, max_v[d]/s->dx[d]);
  *pmax_dti = max_dti;
}

double orc_new_dt_local(OrcSim *s)
{
  Real max_v[3] = {0.0, 0.0, 0.0}, max_dti = 0.0;
  cfl_accumulate(s, max_v, &max_dti);
  return s->p.cour_no/max_dti;
}

void orc_new_dt(OrcSim *s)
{
  Real dtc = orc_new_dt_local(s);
  if (s->nstep == 0) s->dt = dtc; else s->dt = MINR(2.0*s->dt, dtc);
  if ((s->time < s->p.tlim) && ((s->p.tlim - s->time) < s->dt)) s->dt = s->p.tlim - s->time;
}

/* ------------------------------------------------------------------------------------ */
/* ionizing radiation: ionradiation/ionrad_3d.c, ionradplane_3d.c, ionrad_chemistry.c */

typedef struct { Real n_H, n_Hplus, n_e, x, ke, e_th, e_sp, T; } IonCell;

static IonCell ion_cell(const OrcSim *s, const Cons *u)   /* ionrad_3d.c:82-101 */
{
  const OrcParams *p = &s->p; IonCell c;
  c.n_H = u->s / p->m_H;
  c.n_Hplus = (u->d - u->s) / p->m_H;
  c.n_e = c.n_Hplus + u->d * p->alpha_C / (14.0 * p->m_H);
  c.x = c.n_e / (c.n_H + c.n_Hplus);
  c.ke = 0.5 * (u->M[0]*u->M[0] + u->M[1]*u->M[1] + u->M[2]*u->M[2]) / u->d;
  c.e_th = u->E - c.ke;
  c.e_sp = c.e_th / u->d;
  c.T = s->Gamma_1 * c.e_sp * (c.x*0.5*p->m_H+(1.0-c.x)*p->mu)/ p->k_B;
  return c;
}

static void apply_temp_floor(OrcSim *s)            /* ionrad_3d.c:70-131 */
{
  const OrcParams *p = &s->p; int i, j, k;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    Cons *u = &s->U[IDX(s,k,j,i)];
    IonCell c = ion_cell(s, u); Real e_sp;
    if (c.T < p->tfloor) {
      e_sp = p->tfloor * p->k_B / ((c.x*0.5*p->m_H+(1.0-c.x)*p->mu) * s->Gamma_1);
      u->E = 0.5 * (u->M[0]*u->M[0] + u->M[1]*u->M[1] + u->M[2]*u->M[2]) / u->d + e_sp * u->d;
    }
    if ((c.T > p->tceil) && (p->tceil > 0)) {
      e_sp = p->tceil * p->k_B / ((c.x*0.5*p->m_H+(1.0-c.x)*p->mu) * s->Gamma_1);
      u->E = 0.5 * (u->M[0]*u->M[0] + u->M[1]*u->M[1] + u->M[2]*u->M[2]) / u->d + e_sp * u->d;
    }
  }
}

static Real neutral_lim(const OrcSim *s, Real d)   /* ionrad_3d.c:147-148 */
{ Real d_nlim = d*IONFRACFLOOR; return d_nlim < s->d_nlo ? d_nlim : s->d_nlo; }

static void apply_neutral_floor(OrcSim *s)         /* ionrad_3d.c:140-156 */
{
  int i, j, k;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    Cons *u = &s->U[IDX(s,k,j,i)];
    Real d_nlim = neutral_lim(s, u->d);
    if (u->s < d_nlim) u->s = d_nlim; else if (u->s > u->d) u->s = u->d;
  }
}

static void save_energy_and_x(OrcSim *s)           /* ionrad_3d.c:162-196 */
{
  int i, j, k;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    size_t m = IDX(s,k,j,i); const Cons *u = &s->U[m];
    IonCell c = ion_cell(s, u);
    Real e_thermal = u->E - 0.5 * (u->M[0]*u->M[0] + u->M[1]*u->M[1] + u->M[2]*u->M[2]) / u->d;
    s->e_init[m] = u->E; s->e_th_init[m] = e_thermal; s->x_init[m] = c.n_e / (c.n_H + c.n_Hplus);
    s->last_sign[m] = 0; s->sign_count[m] = 0;
  }
}

void orc_ion_begin(OrcSim *s) { apply_temp_floor(s); apply_neutral_floor(s); save_energy_and_x(s); }

/* rays along +x2 (dir=-2): ionradplane_3d.c:323-354.  Unlike case -1 the incident flux is the plane's flux_i without
 * the time ramp, the optical depth of a zone still uses dx1 (:337) while the rate divides by dx2 (:339, cell_len :134),
 * the cut-off compares flux/flux_i, and EdgeFlux behind the cut keeps what earlier sweeps left there */
static void get_ph_rate_plane_x2(OrcSim *s)
{
  const OrcParams *p = &s->p;
  const int n0 = p->Nx[0]+1, n1 = p->Nx[1]+1;
  int i, j, k;
  for (k = s->ks; k <= s->ke; k++) for (i = s->is; i <= s->ie; i++) {
    Real flux = s->flux_i, flux_frac;
    for (j = s->js; j <= s->je; j++) {
      size_t m = IDX(s,k,j,i);
      Real n_H, tau, etau, kph;
      s->EdgeFlux[((size_t)(k-s->ks)*n1 + (j-s->js))*n0 + (i-s->is)] = flux;
      n_H = s->U[m].s / p->m_H;
      tau = p->sigma_ph * n_H * s->dx[0];
      etau = exp(-tau);
      kph = flux * (1.0-etau) / (n_H*s->dx[1]);
      s->ph_rate[m] += kph;
      flux *= etau;
      flux_frac = flux / s->flux_i;
      if (flux_frac < MINFLUXFRAC) break;
    }
  }
}

/* ray sweep, dir=-1: ionradplane_3d.c:88-320 */
static void get_ph_rate_plane(OrcSim *s)
{
  const OrcParams *p = &s->p;
  const int n0 = p->Nx[0]+1, n1 = p->Nx[1]+1;
  const int st = s->is, e = s->ie;
  int i, j, k, ii;
  if (s->rad_dir == -2) { get_ph_rate_plane_x2(s); return; }
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) {
    Real *ef = &s->EdgeFlux[((size_t)(k-s->ks)*n1 + (j-s->js))*n0];
    /* :264-270: the root level carries the time ramp, finer levels start from the flux their
     * parent left at the shared face (ionrad_prolong_rcv) */
    Real flux = (s->level == 0) ? s->flux_i*(5.*(erf((s->time - 1.2e5)/8e4)+1)+0.1) : ef[0];
    Real flux_frac = 0.0;
    for (i = st; i <= e; i++) {
      size_t m = IDX(s,k,j,i);
      Real n_H, tau, etau, kph;
      ef[i-st] = flux;
      n_H = s->U[m].s / p->m_H;
      tau = p->sigma_ph * n_H * s->dx[0];
      etau = exp(-tau);
      kph = flux * (1.0-etau) / (n_H*s->dx[0]);
      s->ph_rate[m] += kph;
      flux *= etau;
      flux_frac = flux / (ef[0] + 1e-12);
      if (flux_frac < MINFLUXFRAC) {
        for (ii = i; ii <= e; ii++) ef[ii-st+1] = 0.0;
        break;
      }
    }
    ef[e-st+1] = flux_frac < MINFLUXFRAC ? 0.0 : flux;
  }
}

static Real recomb_rate_coef(Real T) { return 2.59e-13*pow(T/1.0e4, -0.7); }     /* chemistry :111 */
static Real recomb_cool_rate_coef(Real T)                                        /* :137 */
{ if (T < 100.0) return 0.0; return 6.11e-10*pow(T,-0.89)*KB_CHEM*T; }
static Real lya_cool_rate(Real nh, Real nhplus, Real T)                          /* :350 */
{ return -7.5e-19*nhplus*nh*exp(-118348/T); }

static Real compute_chem_rates(OrcSim *s)          /* ionrad_3d.c:288-408 */
{
  const OrcParams *p = &s->p; Real dt_chem_min = LARGE; int i, j, k, n;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    size_t m = IDX(s,k,j,i); const Cons *u = &s->U[m];
    IonCell c = ion_cell(s, u);
    Real e_sp = (u->E - 0.5 * (u->M[0]*u->M[0] + u->M[1]*u->M[1] + u->M[2]*u->M[2]) / u->d) / u->d;
    Real T = s->Gamma_1 * e_sp * (c.x*0.5*p->m_H+(1.0-c.x)*p->mu)/ p->k_B;
    Real d_nlim, dt1, dt2, dt_chem;
    if (T < p->tfloor) T = p->tfloor;
    s->nHdot[m] = recomb_rate_coef(T) * p->time_unit * c.n_e * c.n_Hplus - s->ph_rate[m] * c.n_H;
    if (s->nHdot[m] < 0.0) {
      if (s->last_sign[m] == 1) s->sign_count[m]++;
      else if (s->sign_count[m] > 0) s->sign_count[m]--;
      s->last_sign[m] = -1;
    } else if (s->nHdot[m] > 0.0) {
      if (s->last_sign[m] == -1) s->sign_count[m]++;
      else if (s->sign_count[m] > 0) s->sign_count[m]--;
      s->last_sign[m] = 1;
    } else {
      s->sign_count[m] = s->last_sign[m] = 0;
    }
    for (n = MAXSIGNCOUNT; n < s->sign_count[m]; n++) { s->edot[m] *= DAMPFACTOR; s->nHdot[m] *= DAMPFACTOR; }
    d_nlim = neutral_lim(s, u->d);
    if (s->nHdot[m] == 0.0) {
      dt1 = dt2 = LARGE;
    } else if (s->nHdot[m] > 0.0) {
      dt1 = p->max_dx_iter / (1+p->max_dx_iter) * c.n_e / s->nHdot[m];
      dt2 = p->max_dx_iter * c.n_H / s->nHdot[m];
    } else if (u->s > 1.0001*d_nlim) {
      dt1 = -p->max_dx_iter * c.n_e / s->nHdot[m];
      dt2 = -p->max_dx_iter / (1+p->max_dx_iter) * c.n_H / s->nHdot[m];
    } else {
      dt1 = dt2 = LARGE;
    }
    dt_chem = (dt1 < dt2) ? dt1 : dt2;
    if (dt_chem < dt_chem_min) dt_chem_min = dt_chem;
    if (dt_chem < 0) { fprintf(stderr, "[oracle compute_chem_rates]: cell %d %d %d: dt_chem = %e\n", i, j, k, dt_chem); abort(); }
  }
  return dt_chem_min;
}

static Real compute_therm_rates(OrcSim *s)         /* ionrad_3d.c:414-561 */
{
  const OrcParams *p = &s->p; Real dt_therm_min = LARGE; int i, j, k;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    size_t m = IDX(s,k,j,i); const Cons *u = &s->U[m];
    IonCell c = ion_cell(s, u);
    Real e_thermal = u->E - 0.5 * (u->M[0]*u->M[0] + u->M[1]*u->M[1] + u->M[2]*u->M[2]) / u->d;
    Real e_sp = e_thermal / u->d;
    Real T = s->Gamma_1 * e_sp * (c.x*0.5*p->m_H+(1.0-c.x)*p->mu)/ p->k_B;
    Real d_nlim, dt1, dt2, dt_therm, e_sp_min, e_th_min, e_min;
    if (T < p->tfloor) { s->edot[m] = 0.0; continue; }
    d_nlim = neutral_lim(s, u->d);
    if ((s->nHdot[m] < 0) && (u->s < 1.0001*d_nlim)) { s->edot[m] = 0.0; continue; }
    s->edot[m] = s->ph_rate[m] * p->e_gamma * c.n_H
      - recomb_cool_rate_coef(T) * p->time_unit * c.n_Hplus * c.n_e
      + lya_cool_rate(c.n_H, c.n_e, T) * p->time_unit;
    if (s->edot[m] == 0.0) {
      dt1 = dt2 = LARGE;
    } else if (s->edot[m] > 0.0) {
      dt1 = p->max_de_iter * u->E / s->edot[m];
      dt2 = p->max_de_therm_iter * e_thermal / s->edot[m];
    } else {
      e_sp_min = p->tfloor * p->k_B / ((c.x*0.5*p->m_H+(1.0-c.x)*p->mu) * s->Gamma_1);
      e_th_min = e_sp_min * u->d;
      e_min = 0.5 * (u->M[0]*u->M[0] + u->M[1]*u->M[1] + u->M[2]*u->M[2]) / u->d + e_th_min;
      if ((e_thermal/(1.0+p->max_de_therm_iter) < e_th_min) && (u->E/(1.0+p->max_de_iter) < e_min)) continue;
      dt1 = -p->max_de_iter / (1+p->max_de_iter) * u->E / s->edot[m];
      dt2 = -p->max_de_therm_iter / (1+p->max_de_therm_iter) * e_thermal / s->edot[m];
    }
    dt_therm = (dt1 < dt2) ? dt1 : dt2;
    if (dt_therm < dt_therm_min) dt_therm_min = dt_therm;
  }
  return dt_therm_min;
}

void orc_ion_rates(OrcSim *s, double *dt_chem, double *dt_therm)
{
  int i, j, k;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++)
    s->ph_rate[IDX(s,k,j,i)] = 0.0;                                     /* ph_rate_init :55 */
  if (s->nradplane > 0) get_ph_rate_plane(s);
  *dt_chem = compute_chem_rates(s);
  *dt_therm = compute_therm_rates(s);
}

void orc_ion_update(OrcSim *s, double dt)          /* ionization_update :565-588 + floors */
{
  const OrcParams *p = &s->p; int i, j, k;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    size_t m = IDX(s,k,j,i); Cons *u = &s->U[m];
    Real d_nlim = neutral_lim(s, u->d);
    if ((s->nHdot[m] > 0) || (u->s > 1.0001*d_nlim)) {
      u->E += s->edot[m] * dt;
      u->s += s->nHdot[m] * dt * p->m_H;
    }
  }
  apply_temp_floor(s); apply_neutral_floor(s);
}

long orc_ion_check_range_count(OrcSim *s)          /* check_range :206-264 */
{
  const OrcParams *p = &s->p; long cellcount = 0; int i, j, k;
  Real e_thermal = 0.0;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    size_t m = IDX(s,k,j,i); const Cons *u = &s->U[m];
    Real n_H = u->s / p->m_H, n_Hplus, n_e, x;
    if (s->ph_rate[m] / (s->min_area * n_H) > 2.0*CION) continue;
    if (p->max_de_therm_step > 0)
      e_thermal = u->E - 0.5 * (u->M[0]*u->M[0] + u->M[1]*u->M[1] + u->M[2]*u->M[2]) / u->d;
    if ((e_thermal / s->e_th_init[m] >= 1 + p->max_de_therm_step) ||
        (s->e_th_init[m] / e_thermal >= 1 + p->max_de_therm_step)) { cellcount++; continue; }
    if (p->max_de_step > 0) {
      if ((u->E / s->e_init[m] >= 1 + p->max_de_step) || (s->e_init[m] / u->E >= 1 + p->max_de_step)) {
        cellcount++; continue;
      }
    }
    if (p->max_dx_step > 0) {
      n_Hplus = (u->d - u->s) / p->m_H;
      n_e = n_Hplus + u->d * p->alpha_C / (14.0 * p->m_H);
      x = n_e / (n_H + n_Hplus);
      if ((x / s->x_init[m] >= 1 + p->max_dx_step) || (s->x_init[m] / x >= 1 + p->max_dx_step)) {
        cellcount++; continue;
      }
    }
  }
  return cellcount;
}

double orc_ion_dt_hydro(OrcSim *s)                 /* compute_dt_hydro :593-669 */
{
  const Real Gamma = s->Gamma, Gamma_1 = s->Gamma_1; Real max_dti = 0.0; int i, j, k;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    const Cons *u = &s->U[IDX(s,k,j,i)];
    Real di = 1.0/u->d, v1 = u->M[0]*di, v2 = u->M[1]*di, v3 = u->M[2]*di;
    Real qsq = v1*v1 + v2*v2 + v3*v3;
    Real pp = MAXR(Gamma_1*(u->E - 0.5*u->d*qsq), TINY_NUMBER);
    Real asq = Gamma*pp*di;
    if (s->p.Nx[0] > 1) max_dti = MAXR(max_dti, (fabs(v1)+sqrt(asq))/s->dx[0]);
    if (s->p.Nx[1] > 1) max_dti = MAXR(max_dti, (fabs(v2)+sqrt(asq))/s->dx[1]);
    if (s->p.Nx[2] > 1) max_dti = MAXR(max_dti, (fabs(v3)+sqrt(asq))/s->dx[2]);
  }
  return s->p.cour_no/max_dti;
}

int orc_ion_radtransfer(OrcSim *s)                 /* ion_radtransfer_3d :862-1047, root level */
{
  Real dt_chem, dt_therm, dt_hydro, dt, dt_done = 0.0;
  int niter = 0, hydro_done = 0, nchem = 0, ntherm = 0;
  orc_ion_begin(s);
  while (!hydro_done) {
    orc_ion_rates(s, &dt_chem, &dt_therm);
    if (dt_chem < dt_therm) nchem++; else ntherm++;
    dt = MINR(dt_therm, dt_chem);
    if (dt_done + dt > s->dt) { dt = s->dt - dt_done; hydro_done = 1; }
    orc_ion_update(s, dt);
    dt_done += dt;
    niter++;
    if (orc_ion_check_range_count(s) > MAXCELLCOUNT) { s->dt = dt_done; break; }
    if (hydro_done) break;
    dt_hydro = orc_ion_dt_hydro(s);
    if (dt_hydro < dt_done) { s->dt = dt_done; break; }
  }
  if (niter == s->p.maxiter) s->dt = dt_done;
  s->niter_last = niter; s->nchem_last = nchem; s->ntherm_last = ntherm;
  return niter;
}

/* ------------------------------------------------------------------------------------ */
/* ioniz_sphere Userwork_in_loop: prob/ioniz_sphere.c:255-306 */

void orc_userwork(OrcSim *s)
{
  const OrcParams *p = &s->p; int i, j, k;
  Real powindex = 1.0/s->Gamma_1;
  if (p->userwork != 1) return;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    Cons *u = &s->U[IDX(s,k,j,i)]; Real x[3], rad2, myrho;
    cc_pos(s, i, j, k, x);
    rad2 = x[0]*x[0] + x[1]*x[1] + x[2]*x[2];
    if (rad2 <= p->uw_rreset2) {
      myrho = pow(s->Gamma_1/s->Gamma*p->pot_GM/p->uw_K/MAXR(sqrt(rad2),TINY_NUMBER) + p->uw_Cp, powindex);
      myrho = MINR(myrho, p->uw_rho0);
      u->d = myrho; u->M[0] = 0.0; u->M[1] = 0.0; u->M[2] = 0.0;
      u->E = p->uw_K*pow(myrho,s->Gamma)/s->Gamma_1;
      u->s = u->d;
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* main loop: main.c:519-669 */

void orc_start(OrcSim *s) { orc_bvals(s); orc_bvals_ionrad(s); orc_new_dt(s); }

int orc_step(OrcSim *s)
{
  int niter = 0;
  if (s->p.ion && s->nradplane > 0) { niter = orc_ion_radtransfer(s); orc_bvals(s); }
  orc_integrate(s);
  orc_userwork(s);
  s->nstep++;
  s->time += s->dt;
  orc_new_dt(s);
  orc_bvals(s);
  return niter;
}

/* ------------------------------------------------------------------------------------ */
/* Static mesh refinement: smr.c, ionradiation/ionrad_smr.c, the SMR branches of main.c,
 * new_dt.c and ionrad_3d.c.  One Domain (= one Grid) per level, each nested in the level below.
 * Level l is an ordinary OrcSim with dx = root dx / 2^l; this section only adds what the
 * reference adds between the levels.  No send/receive buffers: values are taken from the
 * other level directly, in the reference's summation order.                                */

#define ORC_MAXLEV 16
/* Grids in the order of the reference's loops: level by level from the root, Domains of a level in deck order
 * (MeshS.Domain[nl][nd]).  Link L joins grid L+1 (the child) to grid par[L] (its parent): with one Domain per level
 * par[L] = L.  Domains of a level neither overlap nor touch (init_mesh.c:398-418), so a child has ONE parent. */
struct OrcMesh {
  int nl;                    /* number of Grids */
  OrcSim *lev[ORC_MAXLEV];
  int par[ORC_MAXLEV];       /* par[L]: parent grid of grid L+1 */
  int f2c[ORC_MAXLEV];       /* the links finest level first, deck order inside a level (RestrictCorrect, smr.c:1224) */
  /* overlap of level l+1 on level l in level-l indices (init_grid.c: CGrid.ijks/ijke) */
  int cs[ORC_MAXLEV][3], ce[ORC_MAXLEV][3];
  int prol[ORC_MAXLEV][6];   /* level l+1 has a fine/coarse boundary on this side (myFlx != NULL): its ghost
                                zones there are prolonged ...                                          */
  int corr[ORC_MAXLEV][6];   /* ... and the parent zone outside is flux-corrected here (0 when that zone
                                belongs to another slab of the parent: orc_flux_x3_export/_apply)       */
  int cdisp[ORC_MAXLEV][3];  /* origin of level l+1 minus twice the origin of level l, in zones of level
                                l+1 (= DomainS.Disp of the child when the parent is not displaced)       */
  Real *ionflx[ORC_MAXLEV];  /* CGrid.ionFlx[0] of level l: (n3+1) x (n2+1)                     */
  Cons *box[ORC_MAXLEV];     /* Prolongate: level-l zones around child l+1, taken at "send" time */
  Cons *rU[ORC_MAXLEV];      /* RestrictCorrect: restricted solution of level l+1 (send_bufRC)   */
  Cons *rF[ORC_MAXLEV][6];   /*                  restricted boundary fluxes of level l+1          */
  Real tcoarse;              /* ionrad_3d.c:44 */
  Real time, dt; int nstep;  /* MeshS */
};

static Cons *cz(size_t n) { return (Cons*)calloc(n ? n : 1, sizeof(Cons)); }

/* order of the links for the fine-to-coarse passes */
static void mesh_order(OrcMesh *m)
{
  int L, n = 0, lev, maxlev = 0;
  for (L = 0; L + 1 < m->nl; L++) if (m->lev[L+1]->level > maxlev) maxlev = m->lev[L+1]->level;
  for (lev = maxlev; lev >= 1; lev--) for (L = 0; L + 1 < m->nl; L++) if (m->lev[L+1]->level == lev) m->f2c[n++] = L;
}

/* links[21*l ..]: cs[3] (local parent index incl. ghosts), n[3], prol[6], corr[6], cdisp[3] of level l+1 on level l */
static OrcMesh *mesh_create_links(int nlevels, const OrcParams *p, const int *links, const int *level, const int *par);
OrcMesh *orc_mesh_create_local(int nlevels, const OrcParams *p, const int *links)
{ return mesh_create_links(nlevels, p, links, NULL, NULL); }
static OrcMesh *mesh_create_links(int nlevels, const OrcParams *p, const int *links, const int *level, const int *par)
{
  OrcMesh *m; int l, d;
  if (nlevels < 1 || nlevels > ORC_MAXLEV) return NULL;
  m = (OrcMesh*)calloc(1, sizeof(OrcMesh));
  m->nl = nlevels;
  for (l = 0; l < nlevels; l++) {
    OrcSim *s = orc_create(&p[l]);
    s->level = level ? level[l] : l;
    for (d = 0; d < 3; d++) s->dx[d] = s->rootdx[d]/(Real)(1 << s->level);               /* init_mesh.c:245 */
    m->lev[l] = s;
  }
  for (l = 0; l + 1 < nlevels; l++) m->par[l] = par ? par[l] : l;
  mesh_order(m);
  for (l = 0; l + 1 < nlevels; l++) {
    const int *L = links + 21*l; int n[3];
    for (d = 0; d < 3; d++) { m->cs[l][d] = L[d]; n[d] = L[3+d]; m->ce[l][d] = L[d] + n[d] - 1; m->cdisp[l][d] = L[18+d]; }
    for (d = 0; d < 6; d++) { m->prol[l][d] = L[6+d]; m->corr[l][d] = L[12+d]; }
    m->ionflx[l] = (Real*)calloc((size_t)(n[2]+1)*(n[1]+1), sizeof(Real));
    m->box[l] = cz((size_t)(n[0]+6)*(n[1]+6)*(n[2]+6));
    m->rU[l] = cz((size_t)n[0]*n[1]*n[2]);
    for (d = 0; d < 3; d++) {
      size_t nf = (size_t)n[(d+1)%3]*n[(d+2)%3];
      m->rF[l][2*d] = cz(nf); m->rF[l][2*d+1] = cz(nf);
    }
  }
  return m;
}

/* level[g]: DomainS.Level of grid g (grids level by level from the root, deck order inside a level); disp[3g..]: its
 * iDisp/jDisp/kDisp in zones of its level.  The parent of a grid is the Domain of the level below that contains it. */
OrcMesh *orc_mesh_create_tree(int ngrids, const OrcParams *p, const int *level, const int *disp)
{
  int links[21*ORC_MAXLEV], par[ORC_MAXLEV], l, d, c;
  if (ngrids < 1 || ngrids > ORC_MAXLEV || level[0] != 0) return NULL;
  for (c = 1; c < ngrids; c++) {
    int *L = links + 21*(c-1), irefine = 1 << level[c], P = -1, q;
    const int *dc = disp + 3*c, *dp;
    if (level[c] < level[c-1] || level[c] < 1) return NULL;
    for (q = 0; q < c && P < 0; q++) {
      int inside = (level[q] == level[c] - 1);
      for (d = 0; d < 3 && inside; d++) {
        const int dq = (level[q] ? disp[3*q + d] : 0);
        if (dc[d]/2 < dq || (dc[d] + p[c].Nx[d])/2 > dq + p[q].Nx[d]) inside = 0;
      }
      if (inside) P = q;
    }
    if (P < 0) { fprintf(stderr, "[orc_mesh_create]: grid %d (level %d) is not nested in a Domain of level %d\n", c, level[c], level[c]-1); return NULL; }
    par[c-1] = P; l = P;
    dp = disp + 3*P;
    for (d = 0; d < 3; d++) {
      /* init_grid.c: G3 = child extent/2 clipped to this Grid; the child must be nested */
      const int dpd = level[P] ? dp[d] : 0;
      int a = dc[d]/2 - dpd, b = (dc[d] + p[c].Nx[d])/2 - dpd;
      if ((dc[d] & 1) || (p[c].Nx[d] & 1) || a < 0 || b > p[l].Nx[d]) {
        fprintf(stderr, "[orc_mesh_create]: grid %d is not nested in grid %d along x%d\n", c, l, d+1);
        return NULL;
      }
      L[d] = a + NGHOST; L[3+d] = b - a; L[18+d] = dc[d];
      L[6+2*d]   = L[12+2*d]   = (dc[d] != 0);
      L[6+2*d+1] = L[12+2*d+1] = ((dc[d] + p[c].Nx[d])/irefine != p[c].rootNx[d]);
    }
    /* ionrad_smr.c:97-98 mixes an index local to the parent Grid with the child's root-relative Disp:
     * only meaningful while the parent is not displaced across the rays.  ORC_SMR_DEEP_RADIATION=fixed
     * selects what the formula evidently means (child origin minus twice the parent's); that mode has
     * no reference behaviour behind it and only serves to check the product's same-named mode. */
    if (p[0].ion && level[P] && (dp[1] || dp[2])) {
      const char *e = getenv("ORC_SMR_DEEP_RADIATION");
      if (!(e && strcmp(e, "fixed") == 0)) {
        fprintf(stderr, "[orc_mesh_create]: radiation across a displaced parent (grid %d) is undefined in the reference\n", l);
        return NULL;
      }
      for (d = 0; d < 3; d++) L[18+d] = dc[d] - 2*dp[d];
    }
  }
  return mesh_create_links(ngrids, p, links, level, par);
}

/* one Domain per level: level l = grid l */
OrcMesh *orc_mesh_create(int nlevels, const OrcParams *p, const int *disp)
{
  int level[ORC_MAXLEV], l;
  if (nlevels < 1 || nlevels > ORC_MAXLEV) return NULL;
  for (l = 0; l < nlevels; l++) level[l] = l;
  return orc_mesh_create_tree(nlevels, p, level, disp);
}

void orc_mesh_destroy(OrcMesh *m)
{
  int l, d;
  if (!m) return;
  for (l = 0; l < m->nl; l++) {
    orc_destroy(m->lev[l]); free(m->ionflx[l]); free(m->box[l]); free(m->rU[l]);
    for (d = 0; d < 6; d++) free(m->rF[l][d]);
  }
  free(m);
}

OrcSim *orc_mesh_level(OrcMesh *m, int l) { return (l >= 0 && l < m->nl) ? m->lev[l] : NULL; }
double orc_mesh_time(const OrcMesh *m) { return m->time; }
double orc_mesh_dt(const OrcMesh *m) { return m->dt; }
int    orc_mesh_nstep(const OrcMesh *m) { return m->nstep; }

/* smr.c:1391-1458 (Step 3a): conservative average of the 2x2x2 fine zones under one coarse zone */
static Cons restrict_zone(const OrcSim *F, int i, int j, int k)
{
  const size_t s1 = 1, s2 = F->N[0], s3 = (size_t)F->N[0]*F->N[1];
  const size_t m0 = IDX(F,k,j,i);
  Cons r; int n;
  for (n = 0; n < 6; n++) {
    const Real *a = ((const Real*)&F->U[m0]) + n;
#define A(off) a[6*(off)]
    Real v = A(0) + A(s1);
    v += A(s2) + A(s2+s1);
    v += A(s3) + A(s3+s1) + A(s3+s2) + A(s3+s2+s1);
    v *= 0.125;
#undef A
    ((Real*)&r)[n] = v;
  }
  return r;
}

/* smr.c:1464-1640 (Step 3c): average of the 2x2 fine fluxes on one coarse face.  dir = normal
 * direction; (a,b) = fine indices along the two transverse directions in the reference's loop
 * order: x1-faces [k][j], x2-faces [k][i], x3-faces [j][i] (fast index second). */
static Cons restrict_flux(const OrcSim *F, int dir, int nidx, int slow, int fast)
{
  const size_t str[3] = {1, (size_t)F->N[0], (size_t)F->N[0]*F->N[1]};
  const int dslow = (dir == 2) ? 1 : 2, dfast = (dir == 0) ? 1 : 0;
  const size_t m0 = (size_t)nidx*str[dir] + (size_t)slow*str[dslow] + (size_t)fast*str[dfast];
  Cons r; int n;
  for (n = 0; n < 6; n++) {
    const Real *a = ((const Real*)&F->F[dir][m0]) + n;
    Real v = a[0] + a[6*str[dfast]];
    v += a[6*str[dslow]] + a[6*(str[dslow] + str[dfast])];
    v *= 0.25;
    ((Real*)&r)[n] = v;
  }
  return r;
}

/* One level pair of smr.c:1207 RestrictCorrect: Step 3 of level l+1 (restrict its solution and its
 * boundary fluxes, :1391-1640) followed by Steps 1-2 of level l (inject, :1256-1269; flux-correct the
 * zones just outside, :1277-1340).  RestrictCorrect is this for l = nl-2 ... 0. */
static void restrict_correct_pair(OrcMesh *m, int l)
{
  OrcSim *G = m->lev[m->par[l]]; const OrcSim *F = m->lev[l+1];      /* link l: grid l+1 on its parent */
  const int *cs = m->cs[l], *ce = m->ce[l];
  const int nn[3] = {ce[0]-cs[0]+1, ce[1]-cs[1]+1, ce[2]-cs[2]+1};
  int i, j, k, n, dim;
  {
    Cons *r = m->rU[l];
    for (k = F->ks; k <= F->ke; k += 2) for (j = F->js; j <= F->je; j += 2) for (i = F->is; i <= F->ie; i += 2)
      *r++ = restrict_zone(F, i, j, k);
    for (dim = 0; dim < 6; dim++) {
      const int d = dim/2;
      Cons *rf = m->rF[l][dim];
      if (!m->prol[l][dim]) continue;
      if (d == 0) { const int ii = (dim & 1) ? F->ie+1 : F->is;
        for (k = F->ks; k <= F->ke; k += 2) for (j = F->js; j <= F->je; j += 2) *rf++ = restrict_flux(F, 0, ii, k, j); }
      if (d == 1) { const int jj = (dim & 1) ? F->je+1 : F->js;
        for (k = F->ks; k <= F->ke; k += 2) for (i = F->is; i <= F->ie; i += 2) *rf++ = restrict_flux(F, 1, jj, k, i); }
      if (d == 2) { const int kk = (dim & 1) ? F->ke+1 : F->ks;
        for (j = F->js; j <= F->je; j += 2) for (i = F->is; i <= F->ie; i += 2) *rf++ = restrict_flux(F, 2, kk, j, i); }
    }
  }
  {
    const Cons *r = m->rU[l];
    for (k = cs[2]; k <= ce[2]; k++) for (j = cs[1]; j <= ce[1]; j++) for (i = cs[0]; i <= ce[0]; i++)
      G->U[IDX(G,k,j,i)] = *r++;
    for (dim = 0; dim < 6; dim++) {
      const int d = dim/2, d1 = (d == 0) ? 1 : 0, d2 = (d == 2) ? 1 : 2;       /* fast, slow */
      Real q; int a, b, idx[3];
      if (!m->corr[l][dim]) continue;
      if (dim & 1) { idx[d] = ce[d]+1; q =  (G->dt/G->dx[d]); }
      else         { idx[d] = cs[d]-1; q = -(G->dt/G->dx[d]); }
      for (b = 0; b < nn[d2]; b++) for (a = 0; a < nn[d1]; a++) {
        size_t mc, mf; Real *u; const Real *mine, *fine;
        idx[d1] = cs[d1] + a; idx[d2] = cs[d2] + b;
        mc = IDX(G, idx[2], idx[1], idx[0]);
        /* this level's own flux through the shared face (Step 12e of the integrator, :3072) */
        idx[d] = (dim & 1) ? ce[d]+1 : cs[d];
        mf = IDX(G, idx[2], idx[1], idx[0]);
        idx[d] = (dim & 1) ? ce[d]+1 : cs[d]-1;
        u = (Real*)&G->U[mc]; mine = (const Real*)&G->F[d][mf];
        fine = (const Real*)&m->rF[l][dim][(size_t)b*nn[d1] + a];
        for (n = 0; n < 6; n++) u[n] -= q*(mine[n] - fine[n]);
      }
    }
  }
}

/* smr.c:1207.  (Before the first step, main.c:401, all fluxes are still zero and dt = 0: the
 * correction adds q*(0-0).) */
static void restrict_correct(OrcMesh *m, int first)
{
  int l;
  (void)first;
  for (l = 0; l + 1 < m->nl; l++) restrict_correct_pair(m, m->f2c[l]);      /* finest level first, deck order inside a level */
}

/* smr.c:85 ionradRestrictCorrect: E and s[0] only, after the radiation step.  Finest level first; a Grid takes what its
 * children restricted (in their order), then restricts itself for its parent. */
static void ion_restrict_correct(OrcMesh *m)
{
  int g, L, i, j, k, lev, maxlev = 0;
  for (g = 0; g < m->nl; g++) if (m->lev[g]->level > maxlev) maxlev = m->lev[g]->level;
  for (lev = maxlev; lev >= 0; lev--) for (g = 0; g < m->nl; g++) {
    OrcSim *G = m->lev[g];
    if (G->level != lev) continue;
    for (L = 0; L + 1 < m->nl; L++) if (m->par[L] == g) {
      const int *cs = m->cs[L], *ce = m->ce[L];
      const Cons *r = m->rU[L];
      for (k = cs[2]; k <= ce[2]; k++) for (j = cs[1]; j <= ce[1]; j++) for (i = cs[0]; i <= ce[0]; i++, r++) {
        Cons *u = &G->U[IDX(G,k,j,i)];
        u->E = r->E; u->s = r->s;
      }
    }
    if (g > 0) {
      Cons *r = m->rU[g-1];
      for (k = G->ks; k <= G->ke; k += 2) for (j = G->js; j <= G->je; j += 2) for (i = G->is; i <= G->ie; i += 2)
        *r++ = restrict_zone(G, i, j, k);
    }
  }
}

/* smr.c:3478 */
static Real mcd_slope(const Real vl, const Real vc, const Real vr)
{
  Real dvl = (vc - vl), dvr = (vr - vc), dv, dvm;
  if (dvl > 0.0 && dvr > 0.0) {
    dv = 2.0*(dvl < dvr ? dvl : dvr); dvm = 0.5*(dvl + dvr);
    return (dvm < dv ? dvm : dv);
  } else if (dvl < 0.0 && dvr < 0.0) {
    dv = 2.0*(dvl > dvr ? dvl : dvr); dvm = 0.5*(dvl + dvr);
    return (dvm > dv ? dvm : dv);
  }
  return 0.0;
}

static Real eint(const Cons *u) { return u->E - 0.5*(SQR(u->M[0]) + SQR(u->M[1]) + SQR(u->M[2]))/u->d; }

/* smr.c:3068 ProCon: 2x2x2 prolongation with monotonized slopes; the internal energy, not E, is
 * interpolated (:3146-3168) */
static void pro_con(const Cons *im1, const Cons *c, const Cons *ip1, const Cons *jm1, const Cons *jp1,
                    const Cons *km1, const Cons *kp1, Cons P[2][2][2])
{
  int i, j, k, n;
  static const int fld[5] = {0, 1, 2, 3, 5};     /* d, M1, M2, M3, s[0] as doubles inside Cons */
  Real dq1, dq2, dq3, Pi;
  for (n = 0; n < 5; n++) {
    const int f = fld[n];
#define V(u) (((const Real*)(u))[f])
    dq1 = mcd_slope(V(im1), V(c), V(ip1));
    dq2 = mcd_slope(V(jm1), V(c), V(jp1));
    dq3 = mcd_slope(V(km1), V(c), V(kp1));
    for (k = 0; k < 2; k++) for (j = 0; j < 2; j++) for (i = 0; i < 2; i++)
      ((Real*)&P[k][j][i])[f] = V(c) + (0.5*i - 0.25)*dq1 + (0.5*j - 0.25)*dq2 + (0.5*k - 0.25)*dq3;
#undef V
  }
  Pi = eint(c);
  dq1 = mcd_slope(eint(im1), Pi, eint(ip1));
  dq2 = mcd_slope(eint(jm1), Pi, eint(jp1));
  dq3 = mcd_slope(eint(km1), Pi, eint(kp1));
  for (k = 0; k < 2; k++) for (j = 0; j < 2; j++) for (i = 0; i < 2; i++) {
    Cons *q = &P[k][j][i];
    q->E = Pi + (0.5*i - 0.25)*dq1 + (0.5*j - 0.25)*dq2 + (0.5*k - 0.25)*dq3;
    q->E += 0.5*(SQR(q->M[0]) + SQR(q->M[1]) + SQR(q->M[2]))/q->d;
  }
}

/* smr.c:2359 Prolongate.  Level by level from the root: a level first hands its zones around the
 * child to the child ("send", Step 1: before its own ghost zones are refreshed in this call),
 * then fills its own ghost zones from what its parent handed over (Steps 2-3). */
static void prolongate(OrcMesh *m)
{
  int l, i, j, k, dim;
  for (l = 0; l < m->nl; l++) {
    OrcSim *G = m->lev[l]; int L;
    for (L = 0; L + 1 < m->nl; L++) if (m->par[L] == l) {          /* Step 1 :2397-2470, for every child */
      const int *cs = m->cs[L], *ce = m->ce[L];
      Cons *b = m->box[L];
      for (k = cs[2]-3; k <= ce[2]+3; k++) for (j = cs[1]-3; j <= ce[1]+3; j++) for (i = cs[0]-3; i <= ce[0]+3; i++)
        *b++ = G->U[IDX(G,k,j,i)];
    }
    if (l > 0) {                                                   /* Steps 2-3 :2520-2900 */
      const int *cs = m->cs[l-1], *ce = m->ce[l-1];
      const int bn0 = ce[0]-cs[0]+7, bn1 = ce[1]-cs[1]+7;
      const Cons *box = m->box[l-1];
      const int lo[3] = {G->is, G->js, G->ks}, hi[3] = {G->ie, G->je, G->ke};
#define BOX(kc,jc,ic) box[((size_t)(kc)*bn1 + (jc))*bn0 + (ic)]
      for (dim = 0; dim < 6; dim++) {
        int ps[3], pe[3], d;
        if (!m->prol[l-1][dim]) continue;
        for (d = 0; d < 3; d++) { ps[d] = lo[d] - NGHOST; pe[d] = hi[d] + NGHOST; }
        if (dim & 1) ps[dim/2] = hi[dim/2] + 1; else pe[dim/2] = lo[dim/2] - 1;
        for (k = ps[2]; k <= pe[2]; k += 2) for (j = ps[1]; j <= pe[1]; j += 2) for (i = ps[0]; i <= pe[0]; i += 2) {
          /* coarse zone under fine zones (i,i+1)x(j,j+1)x(k,k+1), as an index into the box whose
           * origin is the coarse zone cs-3: fine lo-4 lies in coarse cs-2 */
          const int ic = (i - (lo[0] - NGHOST))/2 + 1, jc = (j - (lo[1] - NGHOST))/2 + 1, kc = (k - (lo[2] - NGHOST))/2 + 1;
          Cons P[2][2][2]; int a, b2, c2;
          pro_con(&BOX(kc,jc,ic-1), &BOX(kc,jc,ic), &BOX(kc,jc,ic+1), &BOX(kc,jc-1,ic), &BOX(kc,jc+1,ic),
                  &BOX(kc-1,jc,ic), &BOX(kc+1,jc,ic), P);
          for (c2 = 0; c2 < 2; c2++) for (b2 = 0; b2 < 2; b2++) for (a = 0; a < 2; a++)
            G->U[IDX(G,k+c2,j+b2,i+a)] = P[c2][b2][a];
        }
      }
#undef BOX
    }
  }
}

/* ionrad_smr.c:345 ionrad_prolong_snd (dim 0: rays along +x1): the flux this level left at the
 * upstream face of its child, one value per coarse ray plus one extra row and column */
static void ionrad_prolong_snd(OrcMesh *m, int g)
{
  const OrcSim *G = m->lev[g]; const int *cs, *ce; int j, k, n0, n1, fixed, w, l;
  for (l = 0; l + 1 < m->nl; l++) {                               /* every child of grid g */
    if (m->par[l] != g || !m->prol[l][0]) continue;
    cs = m->cs[l]; ce = m->ce[l];
    n0 = G->p.Nx[0]+1; n1 = G->p.Nx[1]+1;
    fixed = cs[0] - NGHOST; w = ce[1] - cs[1] + 2;
    for (k = cs[2] - NGHOST; k <= ce[2]+1 - NGHOST; k++) for (j = cs[1] - NGHOST; j <= ce[1]+1 - NGHOST; j++)
      m->ionflx[l][(size_t)(k-(cs[2]-NGHOST))*w + j-(cs[1]-NGHOST)] = G->EdgeFlux[((size_t)k*n1 + j)*n0 + fixed];
  }
}

/* ionrad_smr.c:34 ionrad_prolong_rcv: piecewise-constant copy onto the 2x2 fine rays */
static void ionrad_prolong_rcv(OrcMesh *m, int l)
{
  OrcSim *G = m->lev[l]; const int *cs, *ce; int j, k, n0, n1, w;
  const int fixed = 0;      /* (PGrid.ijks[0] - nghost)*2 with the child nested: ijks[0] = is */
  if (l == 0 || !m->prol[l-1][0]) return;
  cs = m->cs[l-1]; ce = m->ce[l-1];
  n0 = G->p.Nx[0]+1; n1 = G->p.Nx[1]+1; w = ce[1] - cs[1] + 2;
#define EF(kk,jj) G->EdgeFlux[((size_t)(kk)*n1 + (jj))*n0 + fixed]
  for (k = cs[2] - NGHOST; k <= ce[2]+1 - NGHOST; k++) for (j = cs[1] - NGHOST; j <= ce[1]+1 - NGHOST; j++) {
    const Real v = m->ionflx[l-1][(size_t)(k-(cs[2]-NGHOST))*w + j-(cs[1]-NGHOST)];
    /* fine index of coarse ray (j,k): coarse active index is relative to the PARENT Grid, Disp of the
     * child is relative to the root, both in the reference (ionrad_smr.c:97-98) */
    const int ks = k*2 - m->cdisp[l-1][2], js = j*2 - m->cdisp[l-1][1];
    EF(ks,js) = v;
    if (j < ce[1]+1 - NGHOST) {
      if (k < ce[2]+1 - NGHOST) { EF(ks+1,js+1) = v; EF(ks,js+1) = v; EF(ks+1,js) = v; }
      else EF(ks,js+1) = v;
    } else if (k < ce[2]+1 - NGHOST) EF(ks+1,js) = v;
  }
#undef EF
}

/* ionrad_3d.c:862 ion_radtransfer_3d with STATIC_MESH_REFINEMENT: the root sub-cycles to its own
 * stopping criteria and publishes the time it covered (tcoarse); finer levels sub-cycle until
 * they have covered exactly that time (:919-1040) */
static int ion_radtransfer_level(OrcMesh *m, int l)
{
  OrcSim *s = m->lev[l];
  const int finegrid = (s->level != 0);
  Real dt_chem, dt_therm, dt_hydro, dt, dt_done = 0.0;
  int niter = 0, hydro_done = 0, coarsetime_done = 0, nchem = 0, ntherm = 0;
  if (finegrid) ionrad_prolong_rcv(m, l); else m->tcoarse = 0;
  orc_ion_begin(s);
  while (finegrid || !hydro_done) {
    orc_ion_rates(s, &dt_chem, &dt_therm);
    if (dt_chem < dt_therm) nchem++; else ntherm++;
    dt = MINR(dt_therm, dt_chem);
    if (!finegrid) {
      if (dt_done + dt > s->dt) { dt = s->dt - dt_done; hydro_done = 1; }
    } else {
      if (dt_done + dt > m->tcoarse) { dt = m->tcoarse - dt_done; coarsetime_done = 1; }
    }
    orc_ion_update(s, dt);
    dt_done += dt;
    niter++;
    if (!finegrid) {
      if (orc_ion_check_range_count(s) > MAXCELLCOUNT) { s->dt = dt_done; break; }
      if (hydro_done) break;
      dt_hydro = orc_ion_dt_hydro(s);
      if (dt_hydro < dt_done) { s->dt = dt_done; break; }
    } else if (coarsetime_done) { s->dt = dt_done; break; }
  }
  if (!finegrid) {
    if (niter == s->p.maxiter) s->dt = dt_done;
    m->tcoarse = dt_done;
  }
  m->dt = s->dt;                                                   /* :1030 pMesh->dt = pGrid->dt */
  ionrad_prolong_snd(m, l);
  s->niter_last = niter; s->nchem_last = nchem; s->ntherm_last = ntherm;
  return niter;
}

/* new_dt.c:32 with several levels: one CFL reduction over all Grids, the same dt everywhere */
static void mesh_new_dt(OrcMesh *m)
{
  Real max_v[3] = {0.0, 0.0, 0.0}, max_dti = 0.0, dtc; int l;
  const OrcParams *p = &m->lev[0]->p;
  for (l = 0; l < m->nl; l++) cfl_accumulate(m->lev[l], max_v, &max_dti);
  dtc = p->cour_no/max_dti;
  if (m->nstep == 0) m->dt = dtc; else m->dt = MINR(2.0*m->dt, dtc);
  if ((m->time < p->tlim) && ((p->tlim - m->time) < m->dt)) m->dt = p->tlim - m->time;
  for (l = 0; l < m->nl; l++) m->lev[l]->dt = m->dt;
}

/* ---- pieces for a driver that cuts every level into x3 slabs (one OrcMesh per slab stack) ---- */
void orc_mesh_restrict_correct(OrcMesh *m) { restrict_correct(m, 0); }
void orc_mesh_restrict_correct_pair(OrcMesh *m, int l) { restrict_correct_pair(m, l); }
void orc_mesh_ion_restrict_correct(OrcMesh *m) { ion_restrict_correct(m); }
void orc_mesh_prolongate(OrcMesh *m) { prolongate(m); }
void orc_mesh_ionflux_prolong(OrcMesh *m, int l) { ionrad_prolong_snd(m, m->par[l-1]); ionrad_prolong_rcv(m, l); }
void orc_cfl_max_v(OrcSim *s, double v[3])
{ Real dti = 0.0; v[0] = v[1] = v[2] = 0.0; cfl_accumulate(s, v, &dti); }

/* restricted x3-flux of a child slab at its lower (side 0) / upper (side 1) boundary, [j/2][i/2][6]:
 * what RestrictCorrect sends to the parent (smr.c:1592-1640) when the parent plane outside the
 * boundary lives on another slab */
void orc_flux_x3_export(OrcSim *c, int side, double *buf)
{
  int i, j; Cons *out = (Cons*)buf;
  const int kk = side ? c->ke+1 : c->ks;
  for (j = c->js; j <= c->je; j += 2) for (i = c->is; i <= c->ie; i += 2) *out++ = restrict_flux(c, 2, kk, j, i);
}

/* the matching correction on the parent slab (smr.c:1322-1340): side 0 = the child's LOWER boundary
 * coincides with this slab's upper edge (plane ke, face ke+1), side 1 = its upper boundary with this
 * slab's lower edge (plane ks, face ks); (i0,j0) = first parent zone under the child, n1 x n2 zones */
void orc_flux_x3_apply(OrcSim *p, int side, int i0, int j0, int n1, int n2, const double *buf)
{
  const Cons *fine = (const Cons*)buf; int a, b, n;
  const int kc = side ? p->ks : p->ke, kf = side ? p->ks : p->ke+1;
  const Real q = side ? (p->dt/p->dx[2]) : -(p->dt/p->dx[2]);
  for (b = 0; b < n2; b++) for (a = 0; a < n1; a++, fine++) {
    Real *u = (Real*)&p->U[IDX(p, kc, j0+b, i0+a)];
    const Real *mine = (const Real*)&p->F[2][IDX(p, kf, j0+b, i0+a)];
    for (n = 0; n < 6; n++) u[n] -= q*(mine[n] - ((const Real*)fine)[n]);
  }
}

/* main.c:395-447: after problem() ran on every level */
void orc_mesh_start(OrcMesh *m)
{
  int l;
  restrict_correct(m, 1);
  for (l = 0; l < m->nl; l++) { orc_bvals(m->lev[l]); orc_bvals_ionrad(m->lev[l]); }
  prolongate(m);
  mesh_new_dt(m);
}

/* main.c:519-669 with STATIC_MESH_REFINEMENT; niter[l] = radiation sub-cycles of level l */
void orc_mesh_step(OrcMesh *m, int *niter)
{
  int l;
  OrcSim *root = m->lev[0];
  if (root->p.ion && root->nradplane > 0) {                        /* :546-562 */
    for (l = 0; l < m->nl; l++) {
      int n = ion_radtransfer_level(m, l);
      if (niter) niter[l] = n;
      orc_bvals(m->lev[l]);
    }
    ion_restrict_correct(m);
  } else if (niter) for (l = 0; l < m->nl; l++) niter[l] = 0;
  for (l = 0; l < m->nl; l++) orc_integrate(m->lev[l]);            /* :572-585 */
  restrict_correct(m, 0);                                          /* :591 */
  for (l = 0; l < m->nl; l++) orc_userwork(m->lev[l]);             /* :597 */
  m->nstep++;
  m->time += m->dt;                                                /* :618-626 */
  for (l = 0; l < m->nl; l++) { m->lev[l]->time = m->time; m->lev[l]->nstep = m->nstep; }
  mesh_new_dt(m);                                                  /* :629 */
  for (l = 0; l < m->nl; l++) orc_bvals(m->lev[l]);                /* :635-644 */
  prolongate(m);                                                   /* :647 */
}

/* ------------------------------------------------------------------------------------ */
/* problem generators */

void orc_add_radplane(OrcSim *s, int dir, double flux) { s->rad_dir = dir; s->flux_i = flux; s->nradplane = 1; }

void orc_problem_ifront(OrcSim *s, double n_H, double cs, double flux)   /* prob/ifront.c:36-75 */
{
  int i, j, k;
  for (k = s->ks; k <= s->ke+1; k++) for (j = s->js; j <= s->je+1; j++) for (i = s->is; i <= s->ie+1; i++) {
    Cons *u = &s->U[IDX(s,k,j,i)];
    Real rho = n_H*s->p.m_H, pressure = rho*cs*cs;
    u->d = rho; u->M[0] = 0.0; u->M[1] = 0.0; u->M[2] = 0.0;
    u->E = pressure/s->Gamma_1; u->s = rho;
  }
  orc_add_radplane(s, -1, flux);
}

/* prob/ioniz_sphere.c:36-185; also fills the parameter slots PlanetPot/Userwork need */
void orc_problem_ioniz_sphere(OrcSim *s, double n_H, double cs, double flux,
                              double rp, double mp, double np)
{
  const Real Gamma = s->Gamma, Gamma_1 = s->Gamma_1;
  Real mu = s->p.mu, Ggrav = 6.67e-8, GM, rhop, Rsoft, rin, rreset2, powindex, K, rho0, Cp, rhoedge, rout, rhoout;
  int i, j, k;
  (void)n_H;
  GM = Ggrav * mp; rhop = np * mu; Rsoft = 0.01*rp;
  rin = 0.5*rp; rreset2 = 0.5625*rp*rp;
  powindex = 1.0/Gamma_1;
  K = pow(rhop,-Gamma_1)*cs*cs;
  rho0 = pow( pow(rhop,Gamma_1) - Gamma_1/Gamma*GM/K*(1.0/rp - 1.0/rin),powindex);
  Cp = pow(rho0,Gamma_1) - (Gamma_1/Gamma)*GM/K/rin;
  rhoedge = rhop/10;
  rout = 1./(Gamma/Gamma_1/GM*K*(pow(rhoedge, Gamma_1) - pow(rho0, Gamma_1)) + 1./rin);
  rhoout = rhoedge/10000.;
  s->p.pot = 1; s->p.pot_GM = GM; s->p.pot_Rsoft = Rsoft;
  s->p.userwork = 1; s->p.uw_K = K; s->p.uw_Cp = Cp; s->p.uw_rho0 = rho0; s->p.uw_rreset2 = rreset2;
  for (k = s->ks; k <= s->ke+1; k++) for (j = s->js; j <= s->je+1; j++) for (i = s->is; i <= s->ie+1; i++) {
    Cons *u = &s->U[IDX(s,k,j,i)]; Real x[3], rad;
    cc_pos(s, i, j, k, x);
    rad = sqrt(x[0]*x[0] + x[1]*x[1] + x[2]*x[2]);
    u->M[0] = 0.0; u->M[1] = 0.0; u->M[2] = 0.0;
    if (rad <= rin) {
      u->d = rho0; u->E = K*pow(u->d,Gamma)/Gamma_1; u->s = u->d;
    } else if (rad > rout) {
      u->d = rhoout; u->E = K*pow(rhoedge,Gamma)/Gamma_1; u->s = u->d * 1.0e-4;
    } else {
      u->d = pow(Gamma_1/Gamma*GM/K/MAXR(rad,TINY_NUMBER) + Cp,powindex);
      u->E = K*pow(u->d,Gamma)/Gamma_1; u->s = u->d;
    }
  }
  orc_add_radplane(s, -1, flux);
}

void orc_problem_blast(OrcSim *s, double radius, double pamb, double damb, double drat, double prat)
{                                                                         /* prob/blast.c:35-79 */
  int i, j, k;
  for (k = s->ks; k <= s->ke; k++) for (j = s->js; j <= s->je; j++) for (i = s->is; i <= s->ie; i++) {
    Cons *u = &s->U[IDX(s,k,j,i)]; Real x[3], rad; P1 W; C1 c;
    cc_pos(s, i, j, k, x);
    rad = sqrt(x[0]*x[0] + x[1]*x[1] + x[2]*x[2]);
    W.Vx = 0.0; W.Vy = 0.0; W.Vz = 0.0; W.r = 0.0;
    W.P = pamb; if (rad < radius) W.P = prat*pamb;
    W.d = damb; if (rad < radius) W.d = drat*damb;
    c = prim_to_cons(&W, s->Gamma_1, 0);
    u->d = c.d; u->M[0] = c.Mx; u->M[1] = c.My; u->M[2] = c.Mz; u->E = c.E; u->s = 0.0;
  }
}

void orc_problem_shkset1d(OrcSim *s, const double *wl, const double *wr, int shk_dir)
{                            /* prob/shkset1d.c:41-215; wl, wr = {d, P, v1, v2, v3} of the two states */
  int i, j, k, side;
  C1 c[2];
  for (side = 0; side < 2; side++) {
    const double *w = side ? wr : wl; P1 W;
    W.d = w[0]; W.P = w[1]; W.Vx = w[2]; W.Vy = w[3]; W.Vz = w[4]; W.r = 0.0;
    c[side] = prim_to_cons(&W, s->Gamma_1, 0);
  }
  for (k = 0; k < s->N[2]; k++) for (j = 0; j < s->N[1]; j++) for (i = 0; i < s->N[0]; i++) {   /* ghosts too (:89-108) */
    Cons *u = &s->U[IDX(s,k,j,i)]; Real x[3]; const C1 *q;
    cc_pos(s, i, j, k, x);
    q = (x[shk_dir-1] <= 0.0) ? &c[0] : &c[1];
    u->d = q->d; u->E = q->E; u->s = 0.0;
    if (shk_dir == 1)      { u->M[0] = q->Mx; u->M[1] = q->My; u->M[2] = q->Mz; }     /* :121-123 */
    else if (shk_dir == 2) { u->M[0] = q->Mz; u->M[1] = q->Mx; u->M[2] = q->My; }     /* :159-161 */
    else                   { u->M[0] = q->My; u->M[1] = q->Mz; u->M[2] = q->Mx; }     /* :197-199 */
  }
}

/* ------------------------------------------------------------------------------------ */
/* lifecycle */

OrcSim *orc_create(const OrcParams *p)
{
  OrcSim *s = (OrcSim*)calloc(1, sizeof(OrcSim));
  size_t nc; int d;
  s->p = *p;
  for (d = 0; d < 3; d++) {
    s->N[d] = p->Nx[d] + 2*NGHOST;
    s->rootdx[d] = (p->xmax[d] - p->xmin[d])/(Real)(p->rootNx[d]);      /* init_mesh.c:225 */
    s->dx[d] = s->rootdx[d];
  }
  s->is = s->js = s->ks = NGHOST;
  s->ie = s->is + p->Nx[0] - 1; s->je = s->js + p->Nx[1] - 1; s->ke = s->ks + p->Nx[2] - 1;
  s->Gamma = p->gamma; s->Gamma_1 = p->gamma - 1.0;
  nc = (size_t)s->N[0]*s->N[1]*s->N[2];
  s->U = (Cons*)calloc(nc, sizeof(Cons));
  for (d = 0; d < 3; d++) {
    s->Ul[d] = (Cons*)calloc(nc, sizeof(Cons)); s->Ur[d] = (Cons*)calloc(nc, sizeof(Cons));
    s->F[d] = (Cons*)calloc(nc, sizeof(Cons));  s->eta[d] = (Real*)calloc(nc, sizeof(Real));
  }
  s->dhalf = (Real*)calloc(nc, sizeof(Real));
  s->phalf = (Real*)calloc(nc, sizeof(Real));
  s->EdgeFlux = (Real*)calloc((size_t)(p->Nx[0]+1)*(p->Nx[1]+1)*(p->Nx[2]+1), sizeof(Real));
  if (p->ion) {
    Real a1, a2, a3, maxdx;
    s->ph_rate = (Real*)calloc(nc, sizeof(Real)); s->edot = (Real*)calloc(nc, sizeof(Real));
    s->nHdot = (Real*)calloc(nc, sizeof(Real));   s->e_init = (Real*)calloc(nc, sizeof(Real));
    s->e_th_init = (Real*)calloc(nc, sizeof(Real)); s->x_init = (Real*)calloc(nc, sizeof(Real));
    s->last_sign = (int*)calloc(nc, sizeof(int)); s->sign_count = (int*)calloc(nc, sizeof(int));
    /* ionrad.c:112-131 (root dx; note the reference's dx[1] fallback at :129) */
    a1 = s->rootdx[0]*s->rootdx[1]; a2 = s->rootdx[0]*s->rootdx[2]; a3 = s->rootdx[1]*s->rootdx[2];
    if (a1 < a2) { s->min_area = (a1 < a3) ? a1 : a3; } else { s->min_area = (a2 < a3) ? a2 : a3; }
    maxdx = s->rootdx[0] > s->rootdx[1] ? s->rootdx[0] : s->rootdx[1];
    maxdx = maxdx > s->rootdx[2] ? maxdx : s->rootdx[1];
    s->d_nlo = MINOPTDEPTH * p->m_H / (p->sigma_ph * maxdx);
  }
  return s;
}

void orc_destroy(OrcSim *s)
{
  int d;
  if (!s) return;
  free(s->U); free(s->EdgeFlux); free(s->dhalf); free(s->phalf);
  for (d = 0; d < 3; d++) { free(s->Ul[d]); free(s->Ur[d]); free(s->F[d]); free(s->eta[d]); }
  free(s->ph_rate); free(s->edot); free(s->nHdot); free(s->e_init); free(s->e_th_init); free(s->x_init);
  free(s->last_sign); free(s->sign_count);
  free(s);
}

double *orc_U(OrcSim *s) { return (double*)s->U; }
double *orc_edgeflux(OrcSim *s) { return s->EdgeFlux; }
void orc_dims(const OrcSim *s, int N[3]) { N[0] = s->N[0]; N[1] = s->N[1]; N[2] = s->N[2]; }
double orc_get_time(const OrcSim *s) { return s->time; }
double orc_get_dt(const OrcSim *s) { return s->dt; }
int orc_get_nstep(const OrcSim *s) { return s->nstep; }
void orc_set_time(OrcSim *s, double t) { s->time = t; }
void orc_set_dt(OrcSim *s, double dt) { s->dt = dt; }
void orc_set_nstep(OrcSim *s, int n) { s->nstep = n; }

/* ------------------------------------------------------------------------------------ */
/* function-level entry points */

void orc_cons_to_prim(int n, int nscal, double gamma, const double *U, double *W)
{
  int i, nv = NW + nscal;
  for (i = 0; i < n; i++) {
    C1 u; P1 w; memset(&u, 0, sizeof u); memcpy(&u, U + (size_t)i*nv, nv*sizeof(double));
    w = cons_to_prim(&u, gamma - 1.0, nscal); memcpy(W + (size_t)i*nv, &w, nv*sizeof(double));
  }
}

void orc_cfast(int n, int nscal, double gamma, const double *U, double *c)
{
  int i, nv = NW + nscal;
  for (i = 0; i < n; i++) {
    C1 u; memset(&u, 0, sizeof u); memcpy(&u, U + (size_t)i*nv, nv*sizeof(double));
    c[i] = cfast_c1(&u, gamma, gamma - 1.0);
  }
}

void orc_fluxes(int n, int nscal, double gamma, const double *Ul, const double *Ur,
                const double *eta, double *F)
{
  int i, nv = NW + nscal;
  for (i = 0; i < n; i++) {
    C1 ul, ur, f; P1 wl, wr;
    memset(&ul, 0, sizeof ul); memset(&ur, 0, sizeof ur);
    memcpy(&ul, Ul + (size_t)i*nv, nv*sizeof(double)); memcpy(&ur, Ur + (size_t)i*nv, nv*sizeof(double));
    wl = cons_to_prim(&ul, gamma - 1.0, nscal); wr = cons_to_prim(&ur, gamma - 1.0, nscal);
    flux_roe(&ul, &ur, &wl, &wr, eta[i], gamma, gamma - 1.0, nscal, &f);
    memcpy(F + (size_t)i*nv, &f, nv*sizeof(double));
  }
}

static void lr_states_any(int order, int n, int nscal, double gamma, const double *W, double dt, double dx,
                          int il, int iu, double *Wl, double *Wr)
{
  int i, nv = NW + nscal;
  P1 *w = (P1*)calloc(n, sizeof(P1)), *wl = (P1*)calloc(n, sizeof(P1)), *wr = (P1*)calloc(n, sizeof(P1));
  for (i = 0; i < n; i++) {
    memcpy(&w[i], W + (size_t)i*nv, nv*sizeof(double));
    memcpy(&wl[i], Wl + (size_t)i*nv, nv*sizeof(double));
    memcpy(&wr[i], Wr + (size_t)i*nv, nv*sizeof(double));
  }
  lr_states(w, dt, dx, il, iu, wl, wr, gamma, nscal, order);
  for (i = 0; i < n; i++) {
    memcpy(Wl + (size_t)i*nv, &wl[i], nv*sizeof(double));
    memcpy(Wr + (size_t)i*nv, &wr[i], nv*sizeof(double));
  }
  free(w); free(wl); free(wr);
}
void orc_lr_states(int n, int nscal, double gamma, const double *W, double dt, double dx,
                   int il, int iu, double *Wl, double *Wr)
{ lr_states_any(2, n, nscal, gamma, W, dt, dx, il, iu, Wl, Wr); }
void orc_lr_states_ppm(int n, int nscal, double gamma, const double *W, double dt, double dx,
                       int il, int iu, double *Wl, double *Wr)
{ lr_states_any(3, n, nscal, gamma, W, dt, dx, il, iu, Wl, Wr); }
