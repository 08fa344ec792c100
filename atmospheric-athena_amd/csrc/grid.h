// grid.h -- device-resident Grid descriptor and the launch interface between the C-ABI
// layer (api.hip) and the kernel translation units (hydro_kernels.hip, ion_kernels.hip).
//
// HBM layout (one Grid = one GPU):
//   every field is its own [N3][N2][N1] array of doubles (struct-of-arrays, i fastest, ghost
//   zones included) so that a wavefront's 64 lanes on 64 consecutive i issue one 512-byte
//   coalesced request per field.  Arrays of one family are spaced `nc` doubles apart:
//     U    : 6*nc   d, M1, M2, M3, E, s0                      (GridS.U, athena.h:290)
//     LR   : 36*nc  [dir][L|R][var] face states, GLOBAL momentum frame
//                                                            (Ul/Ur_x?Face, integrate_3d_ctu.c:61-63)
//     F    : 18*nc  [dir][var] fluxes, global frame           (x?Flux, :64)
//     eta  : 3*nc   H-correction wave-speed spread per face   (eta1..3, :74)
//     dhalf: nc                                               (:71)
//     phi  : 4*nc   static potential at cell centres and at the lower x1/x2/x3 faces
//   ion module (ionrad_3d.c:33-50): ph_rate, e_init, x_init, plus ke and max|v|/dx frozen (e_th_init = e_init - ke is re-derived)
//   at the start of the ion step (the reference's edot/nHdot arrays are recomputed, not stored),
//   last_sign/sign_count packed in one int2 array; EdgeFlux [Nx3+1][Nx2+1][Nx1+1].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aa {

typedef double Real;
#define AA_NGHOST_ 4

struct DevGrid {
  int N1, N2, N3;                 // zones incl. ghosts
  int is, ie, js, je, ks, ke;     // active range (nghost = 4)
  long sJ, sK, nc;                // strides of j and k (i stride is 1); doubles per field
  Real dx[3];
  Real Gamma, Gamma_1;
  Real rGamma_1;                  // 1/(gamma-1), correctly rounded (host division): prim_to_cons' quotient in three instructions (hydro_dev.h, AA_XDIV)
  Real *U, *LR, *F, *eta, *dhalf;
  Real *phalf;                    // P^{n+1/2}: only with a cooling function (aa_set_cooling), else null
  Real *phi;                      // null when StaticGravPot == NULL; [0]=centre, [1+d]=lower face d
  Real *slope;                    // --with-order=3 only: 18*nc, [dir][prim var] monotonised slopes (lr_states_ppm.c dWm)
  // ion
  Real *ph_rate, *kin, *vmax, *e_init, *x_init;   // kin (kinetic energy), vmax: frozen during the ion step
  Real *s_init;                                   // one-kernel sub-cycle: the scalar after the entry floors (see k_ion_pass, speculation)
  int2 *sign;                     // .x = last_sign, .y = sign_count
  Real *edgeflux;
  // one-kernel sub-cycle (ion_pass.hip): flux entering every zone, double-buffered (the sweep after a
  // data-dependent stop is speculative); flux leaving every ray [2][Nx3*Nx2]; last_sign/sign_count in 2 bytes
  Real *fin[2];
  Real *raylast;
  unsigned short *sg16;
  int Nx1, Nx2, Nx3;
};

struct IonPar {                   // ionrad.h:54-91 globals
  Real sigma_ph, m_H, mu, e_gamma, alpha_C, k_B, time_unit;
  Real max_de_iter, max_de_therm_iter, max_dx_iter;
  Real max_de_step, max_de_therm_step, max_dx_step;
  Real tfloor, tceil;
  Real min_area, d_nlo;
  Real cour_no;
  // host-computed reciprocals / constants (FP64 division is the expensive op in the ion kernels)
  Real inv_mH, inv_kB, aC14, rec_floor, cx1, ce1, ce2, ie1, ie2, inv_dx[3];
  int iso;                          // dx1 == dx2 == dx3
};

// device scalars written by reduction kernels (all reductions are MIN/MAX of non-negative
// doubles via their bit patterns, or integer sums: decomposition- and order-independent)
struct DevScalars {
  unsigned long long max_v[3];       // new_dt: max(|v_d| + a) per direction
  unsigned long long dt_chem, dt_therm;   // min over cells
  unsigned long long max_dti;        // compute_dt_hydro
  unsigned long long cellcount;      // check_range
  int neg_dt_chem;                   // error flag (ionrad_3d.c:389-391)
  int pad;
  // written by k_ion_pick (the step of one sub-cycle chosen on the device, so that the host reads the
  // scalars once per sub-cycle): the dt handed to k_ion_update, the values it was derived from
  Real dt_sel, dt_chem_out, dt_therm_out;
  int limit_hit, neg_out;
  // one-kernel sub-cycle (ion_pass.hip): k_ion_pick2 also keeps the time the applied updates have covered and hands
  // the host what belongs to the update the last pass applied, so that the host reads back ONCE per sub-cycle
  Real dt_done;                      // sum of the steps applied so far in this ion step
  Real dt_applied;                   // the step of the update the last pass applied
  int hit_applied, neg_applied;      // ... whether it was cut back to the limit; negative-dt_chem flag of the rates behind it
  // the first pass of an ion step may already have applied the update with the whole step (k_ion_pass<BEG>, spec_dt):
  // 1 = k_ion_pick2 found that step to be the one (the closing update pass has nothing left to do), 2 = it was not (the
  // next pass starts again from e_init / s_init), 0 otherwise
  int spec_state, pad2;
};

// what one block of k_ion_pass contributes to the reductions of a sub-cycle (all doubles: the record is
// also what ranks exchange)
struct IonPart { Real dt_chem, dt_therm, max_dti, cellcount, neg; };
#ifndef AA_ION_WORDS
#define AA_ION_WORDS 8               /* doubles per rank in the sub-cycle's reduction (5 used) */
#endif

// Host-side launch choices of ONE Grid: read from the environment by aa_create and kept with the Grid, so that no later change of
// the environment (a test's monkeypatch, a second Grid of another caller) can reach a Grid that already exists and no Grid depends
// on what an earlier one found there.  Every choice changes the launch geometry only; the results are the same bit for bit in the
// strict build (tests/test_gpu_layout_switches.py).
struct LaunchCfg {
  int strip = 64, xcd = 1;   // AA_STRIP / AA_XCD: zone order of the unfused stencil kernels (k_flux2, k_update)
  int x1_flat = 1;           // AA_X1_FLAT=0: the x1 first-pass sweep with a block per piece of a row
  int slopes_march = 1;      // AA_SLOPES_MARCH=0: the PPM slope arrays along x2 / x3 one zone per thread
  int ca_kc = 0, fu_kc = 0;  // AA_CA_KC / AA_FU_KC: planes per block of k_correct_all / k_flux2_update (0: by size)
  int ion_pass_cap = 4096;   // AA_ION_PASS_BLOCKS: most blocks of a k_ion_pass launch
  int pitch_align = 1;       // AA_PITCH_ALIGN=0: dense device rows
  int mailbox = 1;           // AA_MAILBOX=0: scalars come back by hipMemcpyAsync + hipStreamSynchronize instead of the polled mailbox
  int mailbox_spin_us = 300; // AA_MAILBOX_SPIN_US: how long the host polls before it falls back to hipStreamSynchronize
  int bc_one = 1;            // AA_BC_ONE=0: bvals_mhd as one launch per direction instead of one for the whole ghost shell
  int edge_overlap = 0;      // AA_EDGE_OVERLAP=1: k_x1_edge_flux beside the x2 sweep on a side stream instead of in front of k_correct_all on the Grid's stream
  int pin_one = 1;           // AA_PIN_ONE=0: the pinned zones' share of new_dt's maxima by a kernel of its own behind k_pinned
  int fuse_pick = 1;         // AA_ION_FUSE_PICK=0: k_ion_reduce and k_ion_pick2 as two launches also where one rank reduces alone
};
// the descriptor as the host keeps it: what the kernels get (DevGrid, passed by value: the launch slices it off) + the launch choices
struct HostGrid : DevGrid { LaunchCfg cfg; };

// the scalars' way back to the host (api.hip aa_fetch_scalars): pinned, device-visible host memory that a one-wave kernel fills
// and stamps with a sequence number the host polls for -- a read-back costs a ~5 us launch instead of a copy + a stream wait
struct Mailbox { DevScalars s; unsigned long long seq; };
void launch_publish(const DevScalars *sc, Mailbox *mb_dev, unsigned long long seq, hipStream_t st);

// face planes (index along the normal, incl. ghost offset) whose second-pass fluxes the fused kernel also
// stores: the level boundaries static mesh refinement reads back (smr.hip)
struct KeepPlanes { int n; int p[3][8]; };       // own two boundary planes + the outlines of up to three children (unused slots repeat the first)
#include "hydro_launch.h"
}  // namespace aa
namespace aa_cool {       // hydro_kernels.hip compiled with -DAA_COOLING=1 (see hydro_launch.h)
using namespace aa;
#include "hydro_launch.h"
}
namespace aa {

// ---- launch wrappers (ion_kernels.hip) ---------------------------------------------
void launch_ion_begin(const DevGrid &g, const IonPar &p, hipStream_t st);
void launch_ray_sweep(const DevGrid &g, const IonPar &p, Real flux0, bool from_edgeflux, hipStream_t st);
void launch_ray_sweep_rates(const DevGrid &g, const IonPar &p, Real flux0, bool from_edgeflux, DevScalars *sc, hipStream_t st);
void launch_ion_rates(const DevGrid &g, const IonPar &p, DevScalars *sc, hipStream_t st);
void launch_ion_update(const DevGrid &g, const IonPar &p, Real dt, DevScalars *sc, hipStream_t st);
void launch_ion_pick(DevScalars *sc, Real dt_done, Real dt_limit, hipStream_t st);      // ionrad_3d.c:941-963 on the device
void launch_ion_update_sel(const DevGrid &g, const IonPar &p, DevScalars *sc, hipStream_t st);   // dt = sc->dt_sel
void launch_edgeflux_bc(const DevGrid &g, Real flux_i, hipStream_t st);
void launch_ray_sweep_x2(const DevGrid &g, const IonPar &p, Real flux_i, hipStream_t st);    // rays along +x2 (dir = -2)
void launch_edgeflux_bc_x2(const DevGrid &g, Real flux_i, hipStream_t st);

// ---- launch wrappers (ion_pass.hip): the one-kernel radiation sub-cycle ------------------
int  ion_pass_blocks(const HostGrid &g);                       // launch size = number of IonPart records
void launch_ion_begin16(const DevGrid &g, const IonPar &p, hipStream_t st);
// [entry of the ion step: floors, save_energy_and_x, if `begin`;] update(n-1) with sc->dt_sel, then sweep(n) + rates(n)
// into buffer cur^1; folds the records into `words`
void launch_ion_pass(const HostGrid &g, const IonPar &p, bool update, bool sweep, bool begin, Real flux0, bool from_edgeflux,
                     const DevScalars *sc, int cur, IonPart *part, Real *words, hipStream_t st, Real spec_dt = -1.0, bool reduce = true);
void launch_ion_pick2(const Real *words, int nranks, DevScalars *sc, int first, Real dt_limit, hipStream_t st, int spec_armed = 0);
// one rank: the fold of the pass's records (k_ion_reduce) and the pick in ONE launch; launch_ion_pass(..., reduce = false) goes before it
void launch_ion_reduce_pick(const HostGrid &g, const IonPart *part, Real *words, DevScalars *sc, int first, Real dt_limit, hipStream_t st, int spec_armed,
                            Mailbox *mb_dev = nullptr, unsigned long long seq = 0);      // mb_dev: also publish the scalars (stamp seq)
void launch_ion_finish(const DevGrid &g, int cur, hipStream_t st);
void launch_test_explog(int n, const Real *x, Real *ye, Real *yl, hipStream_t st);   // n a multiple of 4

}  // namespace aa
