/* tests/fixtures/userwork_late.c -- a USER problem file written against the reference's public problem-file API
 * (prototypes.h:199-205), used to check the host/device coherence policy of the drop-in shim: its Userwork_in_loop is a
 * fixed imprint for the first steps (a block of zones reset to the same values, like prob/ioniz_sphere.c's core) and
 * WAKES UP later: from step `tstart` on it also stirs a second block with values that depend on the time -- the pattern
 * `if (time > t0)` of the incident-flux ramp.  AA_COHERENCE=step must reproduce the all-CPU reference; AA_COHERENCE=auto
 * must notice at its next re-validation and say so.  Not derived from any reference problem file.
 * Keys: <problem> pamb, prat, radius, nwake.
 */
#include <math.h>
#include <stdio.h>
#include "defs.h"
#include "athena.h"
#include "globals.h"
#include "prototypes.h"

static Real pamb;
static int nwake;

void problem(DomainS *pDomain)
{
  GridS *pG = pDomain->Grid;
  int i, j, k;
  Real x1, x2, x3, prat, radius, p;
  pamb = par_getd("problem", "pamb");
  prat = par_getd("problem", "prat");
  radius = par_getd("problem", "radius");
  nwake = par_geti_def("problem", "nwake", 5);
  for (k = pG->ks; k <= pG->ke; k++) for (j = pG->js; j <= pG->je; j++) for (i = pG->is; i <= pG->ie; i++) {
    cc_pos(pG, i, j, k, &x1, &x2, &x3);
    p = (x1*x1 + x2*x2 + x3*x3 < radius*radius) ? prat*pamb : pamb;
    pG->U[k][j][i].d = 1.0; pG->U[k][j][i].M1 = 0.0; pG->U[k][j][i].M2 = 0.0; pG->U[k][j][i].M3 = 0.0;
    pG->U[k][j][i].E = p/Gamma_1;
  }
}

void Userwork_in_loop(MeshS *pM)
{
  GridS *pG = pM->Domain[0][0].Grid;
  int i, j, k;
  /* the fixed imprint: a 3x3x3 block held at the ambient state */
  for (k = pG->ks; k < pG->ks + 3; k++) for (j = pG->js; j < pG->js + 3; j++) for (i = pG->is; i < pG->is + 3; i++) {
    pG->U[k][j][i].d = 1.0; pG->U[k][j][i].M1 = 0.0; pG->U[k][j][i].M2 = 0.0; pG->U[k][j][i].M3 = 0.0;
    pG->U[k][j][i].E = pamb/Gamma_1;
  }
  /* ... and, once awake, a second block that is pushed along +x1 with a strength that grows with time and depends on
   * the state the integrator left (so it needs the fresh host block) */
  if (pM->nstep >= nwake) {
    for (k = pG->ke - 2; k <= pG->ke; k++) for (j = pG->je - 2; j <= pG->je; j++) for (i = pG->ie - 2; i <= pG->ie; i++) {
      ConsS *u = &pG->U[k][j][i];
      Real dv = 0.05*(1.0 + pM->time);
      u->E += u->M1*dv + 0.5*u->d*dv*dv;
      u->M1 += u->d*dv;
    }
  }
}

void problem_write_restart(MeshS *pM, FILE *fp) { return; }
void problem_read_restart(MeshS *pM, FILE *fp) { return; }
ConsFun_t get_usr_expr(const char *expr) { return NULL; }
VOutFun_t get_usr_out_fun(const char *name) { return NULL; }
void Userwork_after_loop(MeshS *pM) { return; }
