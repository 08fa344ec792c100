/* tests/fixtures/rayplane_dir.c -- a USER problem file written against the reference's public problem-file API
 * (prototypes.h:199-205; add_radplane_3d, ionradiation/prototypes.h:66), used to pin the reference's behaviour for
 * radiation planes whose rays travel along +x2 or +x3 (get_ph_rate_plane cases -2 / -3, ionradplane_3d.c:323-388).
 * Not derived from any reference problem file.
 *
 * Gas at rest, neutral, with a density that varies from zone to zone by a fixed integer pattern (so that every ray
 * sees its own column and rays end in different zones); no cc_pos, no random numbers: the initial state depends on
 * the zone indices only.  Keys: <problem> n_H, cs, flux, raydir (-1, -2 or -3).
 */
#include <math.h>
#include <stdio.h>
#include "defs.h"
#include "athena.h"
#include "globals.h"
#include "prototypes.h"

void problem(DomainS *pDomain)
{
  GridS *pG = pDomain->Grid;
  int i, j, k;
  Real n_H = par_getd("problem", "n_H"), cs = par_getd("problem", "cs"), flux = par_getd("problem", "flux");
  Real m_H = par_getd("ionradiation", "m_H");
  int raydir = par_geti("problem", "raydir");
  for (k = pG->ks; k <= pG->ke; k++) for (j = pG->js; j <= pG->je; j++) for (i = pG->is; i <= pG->ie; i++) {
    int pat = (7*(i - pG->is) + 3*(j - pG->js) + 5*(k - pG->ks)) % 11;
    Real rho = n_H*m_H*(0.02 + 0.09*(Real)pat);
    pG->U[k][j][i].d = rho;
    pG->U[k][j][i].M1 = 0.0; pG->U[k][j][i].M2 = 0.0; pG->U[k][j][i].M3 = 0.0;
    pG->U[k][j][i].E = rho*cs*cs/Gamma_1;
    pG->U[k][j][i].s[0] = rho;
  }
  add_radplane_3d(pG, raydir, flux);
}

void problem_write_restart(MeshS *pM, FILE *fp) { return; }
void problem_read_restart(MeshS *pM, FILE *fp) { return; }
ConsFun_t get_usr_expr(const char *expr) { return NULL; }
VOutFun_t get_usr_out_fun(const char *name) { return NULL; }
void Userwork_in_loop(MeshS *pM) { return; }
void Userwork_after_loop(MeshS *pM) { return; }
