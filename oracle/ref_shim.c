/* oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Array-in/array-out wrappers around the *reference's own* kernels, compiled and linked
 * against the reference objects by oracle/Makefile.ref into oracle/_ref/libref_<cfg>.so.
 * Used (in the build container only) to produce function-level known-answer vectors for
 *   fluxes()            /root/reference/src/rsolvers/roe.c:59
 *   lr_states()         /root/reference/src/reconstruction/lr_states_plm.c:62
 *   Cons1D_to_Prim1D()  /root/reference/src/convert_var.c:389
 *   cfast()             /root/reference/src/convert_var.c:470
 * The shim only defines the globals main.c would define (via the reference's globals.h
 * with MAIN_C set) and marshals flat double arrays into the reference's structs.
 */
#include <stdlib.h>
#include <string.h>
#include "defs.h"
#include "athena.h"
#define MAIN_C
#include "globals.h"
#undef MAIN_C
#include "prototypes.h"

extern Real etah; /* roe.c:33 */

enum { NV = NWAVE + NSCALARS };

int ref_nvar(void) { return NV; }

void ref_set_gamma(double g) { Gamma = g; Gamma_1 = g - 1.0; Gamma_2 = g - 2.0; }

/* n interfaces; Ul,Ur: [n][NV] conserved (d,Mx,My,Mz,E[,s0]); eta[n]; F out [n][NV] */
void ref_fluxes(int n, const double *Ul, const double *Ur, const double *eta, double *F)
{
  int i; Real Bx = 0.0;
  for (i = 0; i < n; i++) {
    Cons1DS ul, ur, f; Prim1DS wl, wr;
    memset(&f, 0, sizeof f);
    memcpy(&ul, Ul + (size_t)i*NV, NV*sizeof(double));
    memcpy(&ur, Ur + (size_t)i*NV, NV*sizeof(double));
    wl = Cons1D_to_Prim1D(&ul, &Bx);
    wr = Cons1D_to_Prim1D(&ur, &Bx);
    etah = eta[i];
    fluxes(ul, ur, wl, wr, Bx, &f);
    memcpy(F + (size_t)i*NV, &f, NV*sizeof(double));
  }
  etah = 0.0;
}

void ref_cons_to_prim(int n, const double *U, double *W)
{
  int i; Real Bx = 0.0;
  for (i = 0; i < n; i++) {
    Cons1DS u; Prim1DS w;
    memcpy(&u, U + (size_t)i*NV, NV*sizeof(double));
    w = Cons1D_to_Prim1D(&u, &Bx);
    memcpy(W + (size_t)i*NV, &w, NV*sizeof(double));
  }
}

void ref_cfast(int n, const double *U, double *c)
{
  int i; Real Bx = 0.0;
  for (i = 0; i < n; i++) {
    Cons1DS u;
    memcpy(&u, U + (size_t)i*NV, NV*sizeof(double));
    c[i] = cfast(&u, &Bx);
  }
}

/* W: [n][NV] primitive pencil; computes Wl,Wr over interfaces [il..iu+1] exactly as the
 * integrators call it (W valid on [il-2..iu+2]). Wl,Wr: [n][NV], untouched elsewhere. */
void ref_lr_states(int n, const double *W, double dt, double dx, int il, int iu,
                   double *Wl, double *Wr)
{
  static int inited = 0;
  static MeshS M; static DomainS D; static DomainS *Dp; static GridS G; static int dpl;
  Prim1DS *w, *wl, *wr; Real *bxc; int i;
  if (!inited) {
    memset(&M, 0, sizeof M); memset(&D, 0, sizeof D); memset(&G, 0, sizeof G);
    G.Nx[0] = 1 << 16; G.Nx[1] = 1; G.Nx[2] = 1;
    D.Grid = &G; Dp = &D; dpl = 1;
    M.NLevels = 1; M.DomainsPerLevel = &dpl; M.Domain = &Dp;
    lr_states_init(&M);
    inited = 1;
  }
  w  = (Prim1DS*)malloc(n*sizeof(Prim1DS));
  wl = (Prim1DS*)calloc(n, sizeof(Prim1DS));
  wr = (Prim1DS*)calloc(n, sizeof(Prim1DS));
  bxc = (Real*)calloc(n, sizeof(Real));
  for (i = 0; i < n; i++) memcpy(&w[i], W + (size_t)i*NV, NV*sizeof(double));
  for (i = 0; i < n; i++) { memcpy(&wl[i], Wl + (size_t)i*NV, NV*sizeof(double));
                            memcpy(&wr[i], Wr + (size_t)i*NV, NV*sizeof(double)); }
  lr_states(&G, w, bxc, dt, dx, il, iu, wl, wr, 1);
  for (i = 0; i < n; i++) { memcpy(Wl + (size_t)i*NV, &wl[i], NV*sizeof(double));
                            memcpy(Wr + (size_t)i*NV, &wr[i], NV*sizeof(double)); }
  free(w); free(wl); free(wr); free(bxc);
}
