/* tests/fixtures/userbc_blast.c -- a USER problem file written against the reference's public
 * problem-file API (prototypes.h:199-205, bvals_mhd_fun :77), used to check that the drop-in shim
 * honours boundary functions enrolled by problem().  Not derived from any reference problem file.
 *
 * Flow: uniform gas moving in +x1 with an over-pressured sphere; the inner-x1 boundary is a
 * user-defined fixed inflow, the outer-x2 boundary a user-defined zero-gradient copy; the other
 * four sides use the <domain1> bc flags.  Keys: <problem> gamma, pamb, prat, radius, v0.
 */
#include <math.h>
#include <stdio.h>
#include "defs.h"
#include "athena.h"
#include "globals.h"
#include "prototypes.h"

static Real v0, pamb;

static void inflow_ix1(GridS *pG)
{
  int i, j, k;
  for (k = pG->ks; k <= pG->ke; k++) for (j = pG->js; j <= pG->je; j++) for (i = 1; i <= nghost; i++) {
    ConsS *u = &pG->U[k][j][pG->is - i];
    u->d = 1.0; u->M1 = v0; u->M2 = 0.0; u->M3 = 0.0;
    u->E = pamb/Gamma_1 + 0.5*v0*v0;
  }
}

static void copy_ox2(GridS *pG)
{
  int i, j, k;
  for (k = pG->ks; k <= pG->ke; k++) for (j = 1; j <= nghost; j++)
    for (i = pG->is - nghost; i <= pG->ie + nghost; i++)
      pG->U[k][pG->je + j][i] = pG->U[k][pG->je][i];
}

void problem(DomainS *pDomain)
{
  GridS *pG = pDomain->Grid;
  int i, j, k;
  Real x1, x2, x3, prat, radius, p;
  pamb = par_getd("problem", "pamb");
  prat = par_getd("problem", "prat");
  radius = par_getd("problem", "radius");
  v0 = par_getd_def("problem", "v0", 0.3);
  for (k = pG->ks; k <= pG->ke; k++) for (j = pG->js; j <= pG->je; j++) for (i = pG->is; i <= pG->ie; i++) {
    cc_pos(pG, i, j, k, &x1, &x2, &x3);
    p = (x1*x1 + x2*x2 + x3*x3 < radius*radius) ? prat*pamb : pamb;
    pG->U[k][j][i].d = 1.0; pG->U[k][j][i].M1 = v0; pG->U[k][j][i].M2 = 0.0; pG->U[k][j][i].M3 = 0.0;
    pG->U[k][j][i].E = p/Gamma_1 + 0.5*v0*v0;
  }
  bvals_mhd_fun(pDomain, left_x1, inflow_ix1);
  bvals_mhd_fun(pDomain, right_x2, copy_ox2);
}

void problem_write_restart(MeshS *pM, FILE *fp) { return; }
void problem_read_restart(MeshS *pM, FILE *fp) { return; }
ConsFun_t get_usr_expr(const char *expr) { return NULL; }
VOutFun_t get_usr_out_fun(const char *name) { return NULL; }
void Userwork_in_loop(MeshS *pM) { return; }
void Userwork_after_loop(MeshS *pM) { return; }
