// ion_kernels.hip -- plane-parallel ionizing-radiation step (ionradiation/ionrad_3d.c,
// ionradplane_3d.c, ionrad_chemistry.c) as HIP kernels for gfx950.
//
// During the ion step only E and s0 change; d and the momenta (hence the kinetic energy and the
// speeds) are frozen.  ion_begin therefore stores ke and max_d|v_d| once, and one radiation
// sub-cycle is three kernels that touch 1, 5 and 9 fields per zone (the reference's loops touch
// 6 + scratch each):
//
//   ray_sweep   get_ph_rate_plane (ionradplane_3d.c:88).  Every (j,k) ray is an exclusive prefix
//               product of exp(-tau) along x1 with a data-dependent cut-off.  A block stages a
//               64-ray x RS_CH-cell tile: all 256 threads evaluate exp(-tau) (the expensive part)
//               with coalesced loads; one wavefront -- one lane per ray -- carries the product
//               serially through LDS in exactly the reference's multiplication order; then all
//               threads turn the staged incoming fluxes into ph_rate / EdgeFlux with coalesced
//               stores.  The flux is carried across tiles in LDS; a block whose rays are all
//               extinguished only streams zeros.
//   ion_rates   compute_chem_rates + compute_therm_rates (ionrad_3d.c:288, :414): the two
//               time-step limits (block MIN -> 2 atomics per block).  The rates themselves are NOT
//               stored.
//   ion_update  ionization_update + apply_temp_floor + apply_neutral_floor + check_range +
//               compute_dt_hydro (:565, :70, :140, :206, :593): re-evaluates the two rates from
//               (state, ph_rate) -- a log and three exp per cell are cheaper than 32 B/zone of
//               HBM round trip -- applies them with the globally reduced dt, and reduces the
//               out-of-range count and the hydro CFL limit.
// HBM-bound FP64 streaming; no MFMA.
#include "ion_dev.h"

namespace aa {

// ---- entry of ion_radtransfer_3d: floors + save_energy_and_x (:896-905, :162-196) -------------
// also freezes ke and max_d |v_d|
__global__ void __launch_bounds__(256)
k_ion_begin(DevGrid g, IonPar p)
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
  g.sign[m] = make_int2(0, 0);
  g.kin[m] = c.ke;
  // compute_dt_hydro takes max_d (|v_d| + a)/dx_d with one sound speed a: for dx1=dx2=dx3 that is
  // (max_d|v_d| + a)/dx exactly, so one frozen number per zone replaces the three momenta
  g.vmax[m] = rmax(rmax(fabs(M1*di), fabs(M2*di)), fabs(M3*di));
}

// ---- ray sweep --------------------------------------------------------------------------------
#ifndef RS_CH
#define RS_CH 32
#endif
#define RS_RAYS 64
#define RS_PASS (RS_RAYS*RS_CH/256)      /* cells per thread per tile */
// RATES: the chemistry / thermal rates of every zone (compute_chem_rates + compute_therm_rates) are
// evaluated right where its ph_rate is produced, and the two time-step limits reduced here: the separate
// k_ion_rates pass (which re-reads s0 and ph_rate and is bound by its log + 3 exp per zone) then overlaps
// with the loads / barriers / serial product this kernel is bound by.
template <bool RATES>
__global__ void __launch_bounds__(256, RATES ? 3 : 4)
k_ray_sweep(DevGrid g, IonPar p, Real flux0, int from_edgeflux, DevScalars *sc)
{
  __shared__ Real s_etau[RS_RAYS][RS_CH + 1];
  __shared__ Real s_fin[RS_RAYS][RS_CH + 1];
  __shared__ Real s_flux[RS_RAYS];
  __shared__ Real s_f0[RS_RAYS];                              // flux entering the ray (denominator of :299)
  __shared__ int  s_dead[RS_RAYS];
  __shared__ int  s_nalive;
  __shared__ Real red[RATES ? 256 : 1];
  const int tid = threadIdx.x;
  const int j0 = g.js + blockIdx.x*RS_RAYS;                 // rays: 64 consecutive j at one k
  const int k = g.ks + blockIdx.y;
  const int nrays = min(RS_RAYS, g.je - j0 + 1);
  const int col = tid % RS_CH, rsub = tid / RS_CH;           // 256/RS_CH ray-rows per pass
  constexpr int RSTEP = 256/RS_CH;
  const long efp = (long)(g.Nx1 + 1), efrow = (long)(g.Nx2 + 1)*efp;
  Real dt_chem_min = DBL_MAX, dt_therm_min = DBL_MAX;
  if (tid < RS_RAYS) {
    Real f0 = flux0;
    if (from_edgeflux && tid < nrays)                          // :271 refined level: the parent's flux
      f0 = g.edgeflux[(long)(k - g.ks)*efrow + (long)(j0 + tid - g.js)*efp];
    s_flux[tid] = f0; s_f0[tid] = f0; s_dead[tid] = (tid < nrays) ? 0 : 1;
  }
  if (tid == 0) s_nalive = nrays;
  __syncthreads();
  // the neutral densities of a tile are loaded one tile ahead (during the serial product of the tile
  // before), so the chain: loads -> exp -> barrier -> serial product -> barrier -> stores does not leave the
  // memory pipe idle
  Real sv[RS_PASS], sn[RS_PASS];
#pragma unroll
  for (int q = 0; q < RS_PASS; q++) {
    const int r = rsub + RSTEP*q;
    sv[q] = 0.0; sn[q] = 0.0;
    if (g.is + col <= g.ie && r < nrays) sv[q] = Uq(g,5)[(long)k*g.sK + (long)(j0 + r)*g.sJ + g.is + col];
  }
  for (int c0 = g.is; c0 <= g.ie; c0 += RS_CH) {
    const int i = c0 + col;
    const bool incol = (i <= g.ie);
    const int ncol = min(RS_CH, g.ie - c0 + 1);
    const bool alive = (s_nalive > 0);                        // block-uniform
    Real nH[RS_PASS];
    if (alive) {
#pragma unroll
      for (int q = 0; q < RS_PASS; q++) {
        const int r = rsub + RSTEP*q;
        nH[q] = 1.0;
        if (incol && r < nrays) {
          const Real n_H = sv[q] * p.inv_mH;                   // ionradplane_3d.c:281
          const Real tau = p.sigma_ph * n_H * g.dx[0];        // :294
          nH[q] = n_H;
          s_etau[r][col] = exp(-tau);
        }
      }
    }
    if ((alive || RATES) && c0 + RS_CH <= g.ie) {             // next tile (without RATES wasted once, when every ray dies in this one)
#pragma unroll
      for (int q = 0; q < RS_PASS; q++) {
        const int r = rsub + RSTEP*q;
        if (i + RS_CH <= g.ie && r < nrays) sn[q] = Uq(g,5)[(long)k*g.sK + (long)(j0 + r)*g.sJ + i + RS_CH];
      }
    }
    __syncthreads();
    if (alive && tid < nrays) {                               // one lane per ray: serial product
      const int r = tid;
      Real flux = s_flux[r]; int dead = s_dead[r];
      // flux/(f0 + 1e-12) < MINFLUXFRAC (:299-300) as flux < MINFLUXFRAC*(f0 + 1e-12): no division in the
      // serial chain (a ray whose ratio rounds exactly onto the threshold may be cut one zone apart)
      const Real cut = MINFLUXFRAC*(s_f0[r] + 1e-12);
      for (int cc = 0; cc < ncol; cc++) {
        if (dead) { s_fin[r][cc] = 0.0; continue; }
        s_fin[r][cc] = flux;                                  // EdgeFlux[..][i-s] = flux  (:279)
        flux *= s_etau[r][cc];                                // :298
        if (flux < cut) { dead = 1; flux = 0.0; atomicSub(&s_nalive, 1); }   // :300-306
      }
      s_flux[r] = flux; s_dead[r] = dead;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RS_PASS; q++) {
      const int r = rsub + RSTEP*q;
      if (incol && r < nrays) {
        const long m = (long)k*g.sK + (long)(j0 + r)*g.sJ + i;
        Real kph = 0.0, fin = 0.0;
        if (alive) {
          fin = s_fin[r][col];
          kph = fin * (1.0 - s_etau[r][col]) / (nH[q]*g.dx[0]);   // :296
        }
        g.ph_rate[m] = kph;                                   // ph_rate_init + "+=" (:55, :297)
        g.edgeflux[(long)(k - g.ks)*efrow + (long)(j0 + r - g.js)*efp + (i - g.is)] = fin;
        if (RATES) {
          Cell c; c.d = Uq(g,0)[m]; c.ke = g.kin[m]; c.E = Uq(g,4)[m]; c.s = sv[q];
          int2 sg = g.sign[m];
          const int2 sg0 = sg;
          Real dt_chem, dt_therm;
          rates_cell(c, kph, sg, p, g.Gamma_1, sc, dt_chem, dt_therm);
          if (sg.x != sg0.x || sg.y != sg0.y) g.sign[m] = sg;
          dt_chem_min = rmin(dt_chem_min, dt_chem);
          dt_therm_min = rmin(dt_therm_min, dt_therm);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RS_PASS; q++) sv[q] = sn[q];
    __syncthreads();
  }
  if (tid < nrays)                                            // :308
    g.edgeflux[(long)(k - g.ks)*efrow + (long)(j0 + tid - g.js)*efp + g.Nx1] = s_dead[tid] ? 0.0 : s_flux[tid];
  if (RATES) {
    block_min_to(&sc->dt_chem, dt_chem_min, red);
    block_min_to(&sc->dt_therm, dt_therm_min, red);
  }
}

// ---- rates: compute_chem_rates :334-394 + compute_therm_rates :460-557.  Streams d, ke, E, s0,
// ph_rate (+ the int2 sign bookkeeping), stores nothing but the sign changes, and reduces the two
// time-step limits (grid-stride; 2 atomics per block) --------------------------------------------
__global__ void __launch_bounds__(256)
k_ion_rates(DevGrid g, IonPar p, DevScalars *sc)
{
  __shared__ Real red[256];
  long m;
  Real dt_chem_min = DBL_MAX, dt_therm_min = DBL_MAX;
  // software-pipelined like k_ion_update: the next zone's operands are in flight during this zone's rates
  const long stride = (long)gridDim.x*blockDim.x;
  long lin = (long)blockIdx.x*blockDim.x + threadIdx.x, m_n = 0;
  bool have = active_cell(g, lin, m);
  Real n_d = 0, n_ke = 0, n_E = 0, n_s = 0, n_ph = 0; int2 n_sg = make_int2(0, 0);
  if (have) { n_d = Uq(g,0)[m]; n_ke = g.kin[m]; n_E = Uq(g,4)[m]; n_s = Uq(g,5)[m]; n_ph = g.ph_rate[m]; n_sg = g.sign[m]; }
  for (; have; lin += stride, m = m_n) {
    Cell c; c.d = n_d; c.ke = n_ke; c.E = n_E; c.s = n_s;
    const Real ph = n_ph;
    int2 sg = n_sg;
    have = active_cell(g, lin + stride, m_n);
    if (have) { n_d = Uq(g,0)[m_n]; n_ke = g.kin[m_n]; n_E = Uq(g,4)[m_n]; n_s = Uq(g,5)[m_n]; n_ph = g.ph_rate[m_n]; n_sg = g.sign[m_n]; }
    const int2 sg0 = sg;
    Real dt_chem, dt_therm;
    rates_cell(c, ph, sg, p, g.Gamma_1, sc, dt_chem, dt_therm);
    if (sg.x != sg0.x || sg.y != sg0.y) g.sign[m] = sg;
    dt_chem_min = rmin(dt_chem_min, dt_chem);
    dt_therm_min = rmin(dt_therm_min, dt_therm);
  }
  // dt values are > 0 (or +DBL_MAX): their bit patterns order like the values
  block_min_to(&sc->dt_chem, dt_chem_min, red);
  block_min_to(&sc->dt_therm, dt_therm_min, red);
}

// ---- rays along +x2 (dir = -2): get_ph_rate_plane case -2, ionradplane_3d.c:323-354.  Thread = one ray (i,k), lanes
// along i (every load and store of a wavefront is one contiguous row segment), marching along j with the flux in a
// register: the reference's multiplication order, no staging.  As in the reference: the incident flux is flux_i
// without the time ramp, tau uses dx1 (:337) while the rate divides by dx2 (cell_len :134, :339), the cut-off tests
// flux/flux_i (:342), and EdgeFlux behind the cut keeps what earlier sweeps left there (ph_rate is 0 there: ph_rate_init).
__global__ void __launch_bounds__(256)
k_ray_sweep_x2(DevGrid g, IonPar p, Real flux_i)
{
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (lin >= (long)g.Nx1*g.Nx3) return;
  const int i = (int)(lin % g.Nx1), k = (int)(lin / g.Nx1);
  const long efp = (long)(g.Nx1 + 1), efrow = (long)(g.Nx2 + 1)*efp;
  Real flux = flux_i;
  bool dead = false;
  for (int j = 0; j < g.Nx2; j++) {
    const long m = (long)(k + g.ks)*g.sK + (long)(j + g.js)*g.sJ + (i + g.is);
    Real kph = 0.0;
    if (!dead) {
      g.edgeflux[(long)k*efrow + (long)j*efp + i] = flux;
      const Real n_H = Uq(g,5)[m] / p.m_H;
      const Real tau = p.sigma_ph * n_H * g.dx[0];
      const Real etau = exp(-tau);
      kph = flux * (1.0 - etau) / (n_H*g.dx[1]);
      flux *= etau;
      if (flux / flux_i < MINFLUXFRAC) dead = true;
    }
    g.ph_rate[m] = kph;
  }
}
// bvals_ionrad.c:357 outflow_flux_ix2: EdgeFlux[k][0][i] = flux_i for k<=Nx3, i<=Nx1
__global__ void k_edgeflux_bc_x2(DevGrid g, Real flux_i)
{
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  const long n1 = g.Nx1 + 1;
  if (lin >= n1*(g.Nx3 + 1)) return;
  const long i = lin % n1, k = lin / n1;
  g.edgeflux[k*(long)(g.Nx2 + 1)*n1 + i] = flux_i;
}

// bvals_ionrad.c:63 / outflow_flux_ix1 :308: EdgeFlux[k][j][0] = flux_i for k<=Nx3, j<=Nx2
__global__ void k_edgeflux_bc(DevGrid g, Real flux_i)
{
  const long lin = (long)blockIdx.x*blockDim.x + threadIdx.x;
  const long n = (long)(g.Nx2 + 1)*(g.Nx3 + 1);
  if (lin >= n) return;
  g.edgeflux[lin*(long)(g.Nx1 + 1)] = flux_i;
}

// ---- update + floors + range check + hydro CFL --------------------------------------------------
__global__ void __launch_bounds__(256)
k_ion_update(DevGrid g, IonPar p, Real dt_arg, DevScalars *sc, int dt_from_sc)
{
  const Real dt = dt_from_sc ? sc->dt_sel : dt_arg;
  __shared__ Real red[256];
  __shared__ unsigned int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  long m;
  Real dti = 0.0;
  unsigned int mycnt = 0;
  const Real *ke_f = g.kin, *vmx_f = g.vmax;
  // grid-stride: a capped grid keeps the number of same-address atomics at 2 per block (one
  // word sustains only ~90 atomics/us on MI355X)
  // software-pipelined: the next zone's operands are in flight while this zone's rates are evaluated
  // (a log and three exp between the loads and the stores otherwise leave the memory pipe idle)
  const long stride = (long)gridDim.x*blockDim.x;
  long lin = (long)blockIdx.x*blockDim.x + threadIdx.x, m_n = 0;
  bool have = active_cell(g, lin, m);
  Real n_d = 0, n_ke = 0, n_E = 0, n_s = 0, n_ph = 0, n_e0 = 0; int n_sy = 0;
  if (have) { n_d = Uq(g,0)[m]; n_ke = ke_f[m]; n_E = Uq(g,4)[m]; n_s = Uq(g,5)[m]; n_ph = g.ph_rate[m]; n_sy = g.sign[m].y; n_e0 = g.e_init[m]; }
  for (; have; lin += stride, m = m_n) {
    Cell c; c.d = n_d; c.ke = n_ke; c.E = n_E; c.s = n_s;
    const Real E0 = c.E, s0 = c.s;
    const Real ph = n_ph, e_init0 = n_e0;
    const int sign_y = n_sy;
    have = active_cell(g, lin + stride, m_n);
    if (have) { n_d = Uq(g,0)[m_n]; n_ke = ke_f[m_n]; n_E = Uq(g,4)[m_n]; n_s = Uq(g,5)[m_n]; n_ph = g.ph_rate[m_n]; n_sy = g.sign[m_n].y; n_e0 = g.e_init[m_n]; }
    {   // the rates ion_rates derived its time-step limits from, re-evaluated (same code path)
      const IonQ q0 = ion_q(c, p, g.Gamma_1);
      Real lnT; bool cold;
      const Real nHdot = damp(chem_rate(q0, ph, p, lnT, cold), sign_y);
      const Real d_nlim = neutral_lim(c.d, p);
      const bool skip = cold || ((nHdot < 0) && (c.s < 1.0001*d_nlim));
      const Real edot = skip ? 0.0 : therm_rate(q0, ph, lnT, p);
      if ((nHdot > 0) || (c.s > 1.0001*d_nlim)) {          // ionization_update :577-585
        c.E += edot * dt;
        c.s += nHdot * dt * p.m_H;
      }
    }
    IonQ q; bool floored;
    floors(c, p, g.Gamma_1, q, floored);
    if (c.E != E0) Uq(g,4)[m] = c.E;
    if (c.s != s0) Uq(g,5)[m] = c.s;
    if (floored) q = ion_q(c, p, g.Gamma_1);               // otherwise what floors() derived still holds
    // check_range :223-264.  a/b >= L is tested as a >= L*b when both are positive (the common
    // case; ratios sit near 1, limits at 11), by division otherwise.
    {
      bool counted = false;
      const bool dtype = (q.n_H > 0.0) ? (ph > 2.0*CION*p.min_area*q.n_H) : (ph / (p.min_area * q.n_H) > 2.0*CION);
      if (!dtype) {
        // e_th_init (ionrad_3d.c:176) = e_init - ke with ke frozen over the ion step: not stored
        const Real e0 = e_init0, eth0 = e0 - c.ke;
        const Real L1 = 1 + p.max_de_therm_step, L2 = 1 + p.max_de_step, L3 = 1 + p.max_dx_step;
        if (ratio_ge(q.e_th, eth0, L1) || ratio_ge(eth0, q.e_th, L1)) counted = true;
        else if ((p.max_de_step > 0) && (ratio_ge(c.E, e0, L2) || ratio_ge(e0, c.E, L2))) counted = true;
        else if (p.max_dx_step > 0) {
          const Real x0 = g.x_init[m];
          if (ratio_ge(q.x, x0, L3) || ratio_ge(x0, q.x, L3)) counted = true;
        }
      }
      if (counted) mycnt++;
    }
    // compute_dt_hydro :609-660 (only compared against dt_done, never used as a time step)
    {
      const Real pp = rmax(g.Gamma_1*(c.E - c.ke), AA_TINY);
      const Real a = sqrt(g.Gamma*pp*q.di);
      Real t3;
      if (p.iso) t3 = (vmx_f[m] + a)*p.inv_dx[0];
      else {                                               // anisotropic zones: per-direction speeds
        const Real v1 = fabs(Uq(g,1)[m]*q.di), v2 = fabs(Uq(g,2)[m]*q.di), v3 = fabs(Uq(g,3)[m]*q.di);
        t3 = rmax(rmax((v1 + a)*p.inv_dx[0], (v2 + a)*p.inv_dx[1]), (v3 + a)*p.inv_dx[2]);
      }
      if (t3 == t3) dti = rmax(dti, t3);
    }
  }
  if (mycnt) atomicAdd(&cnt, mycnt);
  red[threadIdx.x] = dti;
  __syncthreads();
  for (int s = blockDim.x/2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = rmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicMax(&sc->max_dti, (unsigned long long)__double_as_longlong(red[0]));
    if (cnt) atomicAdd(&sc->cellcount, (unsigned long long)cnt);
  }
}

// =============================================================================================
static inline unsigned nblk(long n, int b) { return (unsigned)((n + b - 1)/b); }

void launch_ion_begin(const DevGrid &g, const IonPar &p, hipStream_t st)
{ const long n = (long)g.Nx1*g.Nx2*g.Nx3; hipLaunchKernelGGL(k_ion_begin, dim3(nblk(n, 256)), dim3(256), 0, st, g, p); }
void launch_ray_sweep(const DevGrid &g, const IonPar &p, Real flux0, bool from_edgeflux, hipStream_t st)
{ hipLaunchKernelGGL(k_ray_sweep<false>, dim3((g.Nx2 + RS_RAYS - 1)/RS_RAYS, g.Nx3), dim3(256), 0, st, g, p, flux0, from_edgeflux ? 1 : 0, (DevScalars*)nullptr); }
// ray sweep + the rates of every zone + the two time-step limits (what launch_ray_sweep + launch_ion_rates do)
void launch_ray_sweep_rates(const DevGrid &g, const IonPar &p, Real flux0, bool from_edgeflux, DevScalars *sc, hipStream_t st)
{ hipLaunchKernelGGL(k_ray_sweep<true>, dim3((g.Nx2 + RS_RAYS - 1)/RS_RAYS, g.Nx3), dim3(256), 0, st, g, p, flux0, from_edgeflux ? 1 : 0, sc); }
void launch_ray_sweep_x2(const DevGrid &g, const IonPar &p, Real flux_i, hipStream_t st)
{ const long n = (long)g.Nx1*g.Nx3; hipLaunchKernelGGL(k_ray_sweep_x2, dim3(nblk(n, 256)), dim3(256), 0, st, g, p, flux_i); }
void launch_edgeflux_bc_x2(const DevGrid &g, Real flux_i, hipStream_t st)
{ const long n = (long)(g.Nx1 + 1)*(g.Nx3 + 1); hipLaunchKernelGGL(k_edgeflux_bc_x2, dim3(nblk(n, 256)), dim3(256), 0, st, g, flux_i); }
void launch_ion_rates(const DevGrid &g, const IonPar &p, DevScalars *sc, hipStream_t st)
{ const long n = (long)g.Nx1*g.Nx2*g.Nx3; const unsigned nb = reduce_blocks(n);
  hipLaunchKernelGGL(k_ion_rates, dim3(nb), dim3(256), 0, st, g, p, sc); }
void launch_ion_update(const DevGrid &g, const IonPar &p, Real dt, DevScalars *sc, hipStream_t st)
{ const long n = (long)g.Nx1*g.Nx2*g.Nx3; const unsigned nb = reduce_blocks(n);
  hipLaunchKernelGGL(k_ion_update, dim3(nb), dim3(256), 0, st, g, p, dt, sc, 0); }
void launch_ion_update_sel(const DevGrid &g, const IonPar &p, DevScalars *sc, hipStream_t st)
{ const long n = (long)g.Nx1*g.Nx2*g.Nx3; const unsigned nb = reduce_blocks(n);
  hipLaunchKernelGGL(k_ion_update, dim3(nb), dim3(256), 0, st, g, p, 0.0, sc, 1); }

// ionrad_3d.c:941-963 for one sub-cycle, on the device: dt = MIN(dt_therm, dt_chem), cut back to what is
// left of the hydro step (root) or of the coarse time (refined level).  Also re-arms the reductions of the
// next round, so that a sub-cycle costs one read-back of the scalars instead of two plus two resets.
__global__ void k_ion_pick(DevScalars *sc, Real dt_done, Real dt_limit)
{
  const Real dt_chem = __longlong_as_double((long long)sc->dt_chem), dt_therm = __longlong_as_double((long long)sc->dt_therm);
  Real dt = (dt_therm < dt_chem) ? dt_therm : dt_chem;
  int hit = 0;
  if (dt_done + dt > dt_limit) { dt = dt_limit - dt_done; hit = 1; }
  sc->dt_sel = dt; sc->limit_hit = hit;
  sc->dt_chem_out = dt_chem; sc->dt_therm_out = dt_therm; sc->neg_out = sc->neg_dt_chem;
  sc->dt_chem = (unsigned long long)__double_as_longlong(DBL_MAX); sc->dt_therm = sc->dt_chem; sc->neg_dt_chem = 0;
  sc->max_dti = 0; sc->cellcount = 0;
}
void launch_ion_pick(DevScalars *sc, Real dt_done, Real dt_limit, hipStream_t st)
{ hipLaunchKernelGGL(k_ion_pick, dim3(1), dim3(1), 0, st, sc, dt_done, dt_limit); }
// grid.h Mailbox: the scalars to pinned host memory, then the stamp (system scope: the host polls it)
__global__ void __launch_bounds__(64)
k_publish(const DevScalars *sc, Mailbox *mb, unsigned long long seq)
{
  constexpr int n = (int)(sizeof(DevScalars)/sizeof(unsigned long long));
  static_assert(sizeof(DevScalars) % sizeof(unsigned long long) == 0, "DevScalars is copied word by word");
  const unsigned long long *src = (const unsigned long long*)sc;
  unsigned long long *dst = (unsigned long long*)&mb->s;
  for (int i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&mb->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_publish(const DevScalars *sc, Mailbox *mb_dev, unsigned long long seq, hipStream_t st)
{ hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, sc, mb_dev, seq); }
void launch_edgeflux_bc(const DevGrid &g, Real flux_i, hipStream_t st)
{ const long n = (long)(g.Nx2 + 1)*(g.Nx3 + 1); hipLaunchKernelGGL(k_edgeflux_bc, dim3(nblk(n, 256)), dim3(256), 0, st, g, flux_i); }

}  // namespace aa
