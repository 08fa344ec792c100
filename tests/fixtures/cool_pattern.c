/* tests/fixtures/cool_pattern.c -- a USER problem file written against the reference's public problem-file API
 * (prototypes.h:199-205; the CoolingFunc hook of globals.h:25, microphysics/prototypes.h:23), used to pin the reference's
 * optically thin cooling terms in integrate_3d_ctu (Steps 1c-3c, 8b, 11c) with the one cooling function the reference
 * ships, KoyInut (microphysics/cool.c:48).  Not derived from any reference problem file.
 *
 * Diffuse gas in cgs units whose number density, temperature and velocity vary from zone to zone by fixed integer
 * patterns of the zone indices (no cc_pos, no random numbers), from 150 K -- below the 185 K switch of the cooling
 * function -- to 6000 K.  Keys: <problem> n0 [cm^-3], T0 [K], v0 [cm/s], gamma; cool = 0 leaves CoolingFunc NULL.
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
  const Real mbar = 1.37*1.6733e-24, kb = 1.380658e-16;
  Real n0 = par_getd("problem", "n0"), T0 = par_getd("problem", "T0"), v0 = par_getd("problem", "v0");
  for (k = pG->ks; k <= pG->ke; k++) for (j = pG->js; j <= pG->je; j++) for (i = pG->is; i <= pG->ie; i++) {
    int a = i - pG->is, b = j - pG->js, c = k - pG->ks;
    Real n = n0*(0.5 + 0.25*(Real)((7*a + 3*b + 5*c) % 11));
    Real T = T0*(0.06 + 0.4*(Real)((5*a + 7*b + 3*c) % 7));
    Real rho = n*mbar;
    Real v1 = v0*((Real)((3*a + 5*b + 7*c) % 5) - 2.0), v2 = v0*((Real)((a + 2*b + 3*c) % 7) - 3.0), v3 = v0*((Real)((2*a + b + 4*c) % 3) - 1.0);
    pG->U[k][j][i].d = rho;
    pG->U[k][j][i].M1 = rho*v1; pG->U[k][j][i].M2 = rho*v2; pG->U[k][j][i].M3 = rho*v3;
    pG->U[k][j][i].E = n*kb*T/Gamma_1 + 0.5*rho*(v1*v1 + v2*v2 + v3*v3);
  }
  if (par_geti("problem", "cool") != 0) CoolingFunc = KoyInut;
}

void problem_write_restart(MeshS *pM, FILE *fp) { return; }
void problem_read_restart(MeshS *pM, FILE *fp) { return; }
ConsFun_t get_usr_expr(const char *expr) { return NULL; }
VOutFun_t get_usr_out_fun(const char *name) { return NULL; }
void Userwork_in_loop(MeshS *pM) { return; }
void Userwork_after_loop(MeshS *pM) { return; }
