/* host/problems.c -- problem generators of the three decks this package ships, written
 * against the C-ABI (include/athena_amd.h).  They play the role of the reference's
 * src/prob/<problem>.c: fill the host ConsS block, register the radiation plane, install the
 * static potential and the per-step fix-up.  Host C like the reference's; no GPU code here.
 *
 *   ifront        prob/ifront.c:36-86        uniform medium + ionizing plane at ix1
 *   ioniz_sphere  prob/ioniz_sphere.c:36-185 hydrostatic polytropic atmosphere of a planet,
 *                                            ionized low-density ambient gas, PlanetPot
 *                                            (:316-330), core re-imposed every step (:255-306)
 *   blast         prob/blast.c:35-79         over-pressured sphere in a uniform medium
 */
#include <math.h>
#include <stddef.h>
#include "../../include/athena_amd.h"

#define NG AA_NGHOST
#define TINY_NUMBER 1.0e-20
#define MAXR(a,b) (((a) > (b)) ? (a) : (b))
#define MINR(a,b) (((a) < (b)) ? (a) : (b))
#define SQR(x) ((x)*(x))

static void centre(const aa_params *p, int i, int j, int k, double x[3])   /* cc_pos.c:36-43 */
{
  double dx[3]; int d;
  for (d = 0; d < 3; d++) dx[d] = (p->xmax[d] - p->xmin[d])/(double)p->rootNx[d]/(double)(1 << p->level);   /* init_mesh.c:225,245 */
  x[0] = p->MinX[0] + ((double)(i - NG) + 0.5)*dx[0];
  x[1] = p->MinX[1] + ((double)(j - NG) + 0.5)*dx[1];
  x[2] = p->MinX[2] + ((double)(k - NG) + 0.5)*dx[2];
}

static double *cell(const aa_params *p, double *U, int i, int j, int k)
{
  const int nvar = 5 + p->nscal, N1 = p->Nx[0] + 2*NG, N2 = p->Nx[1] + 2*NG;
  return U + (((size_t)k*N2 + j)*N1 + i)*nvar;
}

/* ---- ifront -------------------------------------------------------------------------- */
int aa_problem_ifront(const aa_params *p, double n_H, double cs, double *U)
{
  const double Gamma_1 = p->gamma - 1.0;
  int i, j, k;
  if (p->nscal != 1) return -1;
  for (k = NG; k <= NG + p->Nx[2]; k++) for (j = NG; j <= NG + p->Nx[1]; j++) for (i = NG; i <= NG + p->Nx[0]; i++) {
    double *u = cell(p, U, i, j, k);
    double rho = n_H*p->m_H, pressure = rho*cs*cs;
    u[0] = rho; u[1] = 0.0; u[2] = 0.0; u[3] = 0.0; u[4] = pressure/Gamma_1; u[5] = rho;
  }
  return 0;
}

/* ---- ioniz_sphere ------------------------------------------------------------------------ */
typedef struct { double GM, Rsoft, K, Cp, rho0, rreset2, rin, rout, rhoedge, rhoout; } sphere_t;
static sphere_t S;     /* file-scope like the reference's statics (ioniz_sphere.c:24) */

static void sphere_setup(const aa_params *p, double cs, double rp, double mp, double np)
{
  const double Gamma = p->gamma, Gamma_1 = Gamma - 1.0, Ggrav = 6.67e-8, powindex = 1.0/Gamma_1;
  double rhop = np * p->mu;
  S.GM = Ggrav * mp; S.Rsoft = 0.01*rp;
  S.rin = 0.5*rp; S.rreset2 = 0.5625*rp*rp;
  S.K = pow(rhop,-Gamma_1)*cs*cs;
  S.rho0 = pow( pow(rhop,Gamma_1) - Gamma_1/Gamma*S.GM/S.K*(1.0/rp - 1.0/S.rin),powindex);
  S.Cp = pow(S.rho0,Gamma_1) - (Gamma_1/Gamma)*S.GM/S.K/S.rin;
  S.rhoedge = rhop/10;
  S.rout = 1./(Gamma/Gamma_1/S.GM*S.K*(pow(S.rhoedge, Gamma_1) - pow(S.rho0, Gamma_1)) + 1./S.rin);
  S.rhoout = S.rhoedge/10000.;
}

int aa_problem_ioniz_sphere(const aa_params *p, double cs, double rp, double mp, double np, double *U)
{
  const double Gamma = p->gamma, Gamma_1 = Gamma - 1.0, powindex = 1.0/Gamma_1;
  int i, j, k;
  if (p->nscal != 1) return -1;
  sphere_setup(p, cs, rp, mp, np);
#pragma omp parallel for private(i, j)
  for (k = NG; k <= NG + p->Nx[2]; k++) for (j = NG; j <= NG + p->Nx[1]; j++) for (i = NG; i <= NG + p->Nx[0]; i++) {
    double *u = cell(p, U, i, j, k), x[3], rad;
    centre(p, i, j, k, x);
    rad = sqrt(x[0]*x[0] + x[1]*x[1] + x[2]*x[2]);
    u[1] = 0.0; u[2] = 0.0; u[3] = 0.0;
    if (rad <= S.rin) {
      u[0] = S.rho0; u[4] = S.K*pow(u[0],Gamma)/Gamma_1; u[5] = u[0];
    } else if (rad > S.rout) {
      u[0] = S.rhoout; u[4] = S.K*pow(S.rhoedge,Gamma)/Gamma_1; u[5] = u[0] * 1.0e-4;
    } else {
      u[0] = pow(Gamma_1/Gamma*S.GM/S.K/MAXR(rad,TINY_NUMBER) + S.Cp,powindex);
      u[4] = S.K*pow(u[0],Gamma)/Gamma_1; u[5] = u[0];
    }
  }
  return 0;
}

/* StaticGravPot = PlanetPot (ioniz_sphere.c:316-330, non-shearing-box branch) */
double aa_planet_pot(double x1, double x2, double x3)
{
  double rad = sqrt(SQR(x1)+SQR(x2)+SQR(x3));
  double adist = 7.48e11;
  double GMstar = 6.67e-8 * 1.99e33;
  double omega = sqrt(GMstar / (pow(adist,3)));
  double radstar = sqrt(SQR(x1+adist) + SQR(x2) + SQR(x3));
  double rcentrif = sqrt(SQR(x1+adist) + SQR(x2));
  return -S.GM/(rad+S.Rsoft)-GMstar/radstar -.5*SQR(omega*rcentrif);
}

/* Userwork_in_loop (ioniz_sphere.c:255-306) writes the same values into the same cells every
 * step; returns that cell list (count only when index == NULL).  index = linear [k][j][i]
 * position in the host block, values = nvar doubles per cell. */
long long aa_ioniz_sphere_pinned(const aa_params *p, long long *index, double *values)
{
  const double Gamma = p->gamma, Gamma_1 = Gamma - 1.0, powindex = 1.0/Gamma_1;
  const int N1 = p->Nx[0] + 2*NG, N2 = p->Nx[1] + 2*NG;
  long long n = 0; int i, j, k;
  for (k = NG; k < NG + p->Nx[2]; k++) for (j = NG; j < NG + p->Nx[1]; j++) for (i = NG; i < NG + p->Nx[0]; i++) {
    double x[3], rad2, myrho;
    centre(p, i, j, k, x);
    rad2 = x[0]*x[0] + x[1]*x[1] + x[2]*x[2];
    if (rad2 <= S.rreset2) {
      if (index) {
        myrho = pow(Gamma_1/Gamma*S.GM/S.K/MAXR(sqrt(rad2),TINY_NUMBER) + S.Cp,powindex);
        myrho = MINR(myrho, S.rho0);
        index[n] = ((long long)k*N2 + j)*N1 + i;
        values[6*n + 0] = myrho; values[6*n + 1] = 0.0; values[6*n + 2] = 0.0; values[6*n + 3] = 0.0;
        values[6*n + 4] = S.K*pow(myrho,Gamma)/Gamma_1; values[6*n + 5] = myrho;
      }
      n++;
    }
  }
  return n;
}

/* ---- blast ------------------------------------------------------------------------------- */
int aa_problem_blast(const aa_params *p, double radius, double pamb, double damb, double drat,
                     double prat, double *U)
{
  const double Gamma_1 = p->gamma - 1.0;
  int i, j, k;
  if (p->nscal != 0) return -1;
#pragma omp parallel for private(i, j)
  for (k = NG; k < NG + p->Nx[2]; k++) for (j = NG; j < NG + p->Nx[1]; j++) for (i = NG; i < NG + p->Nx[0]; i++) {
    double *u = cell(p, U, i, j, k), x[3], rad, P, d;
    centre(p, i, j, k, x);
    rad = sqrt(x[0]*x[0] + x[1]*x[1] + x[2]*x[2]);
    P = pamb; if (rad < radius) P = prat*pamb;
    d = damb; if (rad < radius) d = drat*damb;
    u[0] = d; u[1] = d*0.0; u[2] = d*0.0; u[3] = d*0.0;
    u[4] = P/Gamma_1 + 0.5*d*(SQR(0.0) + SQR(0.0) + SQR(0.0));
  }
  return 0;
}

/* ---- shock tube along x1, x2 or x3 (prob/shkset1d.c:41-215); wl, wr = {d, P, v1, v2, v3} ------- */
int aa_problem_shkset1d(const aa_params *p, const double *wl, const double *wr, int shk_dir, double *U)
{
  const double Gamma_1 = p->gamma - 1.0;
  const int N1 = p->Nx[0] + 2*NG, N2 = p->Nx[1] + 2*NG, N3 = p->Nx[2] + 2*NG;
  double c[2][5];
  int i, j, k, side;
  if (p->nscal != 0 || shk_dir < 1 || shk_dir > 3) return -1;
  for (side = 0; side < 2; side++) {                     /* Prim1D_to_Cons1D, convert_var.c:432 */
    const double *w = side ? wr : wl;
    c[side][0] = w[0]; c[side][1] = w[0]*w[2]; c[side][2] = w[0]*w[3]; c[side][3] = w[0]*w[4];
    c[side][4] = w[1]/Gamma_1 + 0.5*w[0]*(SQR(w[2]) + SQR(w[3]) + SQR(w[4]));
  }
  for (k = 0; k < N3; k++) for (j = 0; j < N2; j++) for (i = 0; i < N1; i++) {    /* ghost zones too (:89-108) */
    double *u = cell(p, U, i, j, k), x[3]; const double *q;
    centre(p, i, j, k, x);
    q = (x[shk_dir-1] <= 0.0) ? c[0] : c[1];
    u[0] = q[0]; u[4] = q[4];
    if (shk_dir == 1)      { u[1] = q[1]; u[2] = q[2]; u[3] = q[3]; }
    else if (shk_dir == 2) { u[1] = q[3]; u[2] = q[1]; u[3] = q[2]; }
    else                   { u[1] = q[2]; u[2] = q[3]; u[3] = q[1]; }
  }
  return 0;
}
