// api_internal.h -- state shared by the translation units that implement the C-ABI (api.hip: one
// Grid; smr.hip: the nested levels of a static-mesh-refinement Mesh).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/athena_amd.h"
#include "grid.h"

int aa_fail(int code, const char *fmt, ...);   // records the message aa_last_error() returns
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
  return aa_fail(-2, "[athena_amd] HIP error %s at %s:%d: %s", #x, __FILE__, __LINE__, hipGetErrorString(e_)); } while (0)

#define MAXCELLCOUNT 20   /* ionrad.h:38 */

struct ProfEntry { std::string name; std::vector<hipEvent_t> ev; double total_ms = 0; long long launches = 0; };

struct aa_grid {
  aa_params p;
  aa::HostGrid d;
  aa::IonPar ion;
  hipStream_t st = nullptr; bool own_stream = false;
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;      // aa_integrate_3d_ctu: the tile-edge x1 fluxes beside the x2 sweep
  aa::Real *pool = nullptr; size_t pool_doubles = 0;
  aa::DevScalars *sc = nullptr;        // device
  aa::DevScalars *sc_host = nullptr;   // pinned (= &mb->s)
  aa::Mailbox *mb = nullptr, *mb_dev = nullptr;   // the pinned mailbox and its device address (grid.h Mailbox); mb_seq: stamps handed out
  unsigned long long mb_seq = 0;
  unsigned long long mb_prepublished = 0;   // stamp of the scalars the last fused pick already put into the mailbox (0: none): the fetch right behind it only polls
  bool ion_fuse_pick = false;          // the pass just queued left its records unfolded: aa_ion_pick folds and picks in one launch (one rank)
  long long *pin_idx = nullptr; aa::Real *pin_val = nullptr; long long npin = 0;
  double *cfl_part = nullptr; long cfl_part_n = 0;   // the blocks' CFL maxima of k_update<CFL> (van Leer integrator), 3 x update_blocks
  unsigned char *pin_mask = nullptr;   // 1 where a zone is pinned (k_flux2_update<CFL> leaves those to k_pinned_cfl)
  bool cfl_in_update = false;          // aa_cfl_in_update: the integrator also leaves new_dt's maxima behind
  bool ion_spec_on = true;             // AA_ION_SPECULATE=0: the first pass of an ion step never applies an update
  double ion_spec_dt = -1.0, ion_spec_limit = 0.0;   // aa_ion_speculate: armed for the next first pass / what it was told
  bool ion_spec_armed = false;         //   ... the first pass has speculated: aa_ion_pick(first) settles it
  bool cfl_force = false;              // AA_CFL_FUSED=2 (until round 4: the way to have it in the strict build too; now the same as 1)
  bool cfl_step = true;                // aa_step does so by itself (AA_CFL_FUSED=0: k_cfl)
  bool cfl_ready = false;              //   ... and has done so for the state as it is now
  bool active_dirty = true;            // a call since the last full upload / download of U wrote ACTIVE zones on the device (aa_download_ghost_zones then moves the whole block)
  bool grav = false;
  int cool = 0;                        // aa_set_cooling: 1 = KoyInut; the integrator then launches the kernels of namespace aa_cool
  int rad_dir = 0, nradplane = 0; aa::Real flux_i = 0;
  int level = 0;                       // DomainS.Level: > 0 only as a level of an aa_mesh
  bool fused_update = false;           // second-pass fluxes + update in one kernel (AA_FUSED_UPDATE=0 at aa_create: the unfused chain)
  int x3_fused_mode = -1;              // k_correct_all also does the x3 first pass: -1 by configuration (api.hip), AA_X3_FUSED=0/1 forces
  bool inner_swept = false;            // aa_integrate_begin has done the first-pass x1 / x2 sweeps of the planes ks .. ke
  double inner_dt = 0.0;               //   ... with this dt
  bool correct_all = false;            // the three correct passes in one kernel (k_correct_all): Grids of 4e5 zones or more (2^21 until round 4), or AA_CORRECT_ALL
  bool fused_rates = false;            // rates evaluated inside the ray sweep (k_ray_sweep<true>): 2^17 rays or more, or AA_FUSED_RATES
  bool ion_fused = false;              // one-kernel radiation sub-cycle with the scan sweep (ion_pass.hip): rays of 64 zones or more, or AA_ION_FUSED
  bool ion_begin_fused = true;         // the entry of the ion step rides on its first pass (AA_ION_BEGIN_FUSED=0: k_ion_begin16 on its own)
  bool ion_begin_due = false;          // aa_ion_begin was called: the next first pass does the entry
  bool ef_stale = false;               // GridS.EdgeFlux has not been filled from the last sweep yet (done when somebody reads it)
  int ion_cur = 0; bool ion_pending = false;   // buffer of the last sweep that counts; a sweep launched but not yet relied upon
  aa::IonPart *ion_part = nullptr; aa::Real *ion_words = nullptr;   // per-block records of a pass; this Grid's folded words
  int host_syncs = 0;                  // stream synchronisations that return scalars to the host (bench: per step)
  bool vl_predict = false;             // van Leer predictor as one kernel (k_vl_predict): 2^18 zones or more, or AA_VL_PREDICT
  bool keep_flux = false;              // a level of an aa_mesh: RestrictCorrect reads the second-pass fluxes ...
  aa::KeepPlanes keep = {0, {{0}}};        // ... on these face planes (own boundaries + the child's outline)
  double time = 0, dt = 0; int nstep = 0;
  long long bytes = 0;
  // composite (slabs.hip): this handle stands for ONE Grid of the caller cut into x3 slabs, one aa_grid per
  // device; every entry point of the C-ABI forwards to the slabs and does the neighbour exchange / reductions
  std::vector<aa_grid*> slab;
  struct SlabLink *link = nullptr;
  bool prof = false;
  std::vector<ProfEntry> pe;
  std::vector<hipEvent_t> ev_pool;     // events of drained scopes, reused (a small Grid's step is ~25 scopes: creating 50 events per step showed)
};

// ---- profiling: an event pair around every kernel-chain stage, on the launch stream ----------
struct Scope {
  aa_grid *g; int id; hipEvent_t a = nullptr, b = nullptr;
  Scope(aa_grid *g_, const char *name) : g(g_), id(-1) {
    if (!g->prof) return;
    for (size_t i = 0; i < g->pe.size(); i++) if (g->pe[i].name == name) { id = (int)i; break; }
    if (id < 0) { g->pe.push_back(ProfEntry()); id = (int)g->pe.size() - 1; g->pe[id].name = name; }
    if (g->ev_pool.size() >= 2) { a = g->ev_pool.back(); g->ev_pool.pop_back(); b = g->ev_pool.back(); g->ev_pool.pop_back(); }
    else { hipEventCreate(&a); hipEventCreate(&b); }
    hipEventRecord(a, g->st);
  }
  ~Scope() {
    if (id < 0) return;
    hipEventRecord(b, g->st);
    g->pe[id].ev.push_back(a); g->pe[id].ev.push_back(b); g->pe[id].launches++;
  }
};

// ---- composite Grids (slabs.hip) -------------------------------------------------------------
int slabs_create(const aa_params *p, int nslab, aa_grid **out);
void slabs_destroy(aa_grid *g);
int slabs_sync(aa_grid *g);
long long slabs_device_bytes(const aa_grid *g);
int slabs_upload_cons(aa_grid *g, const double *U);
int slabs_download_cons(aa_grid *g, double *U);
int slabs_upload_edgeflux(aa_grid *g, const double *ef);
int slabs_download_edgeflux(aa_grid *g, double *ef);
int slabs_set_grav_tables(aa_grid *g, const double *pc, const double *p1, const double *p2, const double *p3);
int slabs_set_cooling(aa_grid *g, int kind);
int slabs_set_pinned_cells(aa_grid *g, long long n, const long long *index, const double *values);
int slabs_apply_pinned_cells(aa_grid *g);
int slabs_add_radplane(aa_grid *g, int dir, double flux);
int slabs_bvals_mhd(aa_grid *g);
int slabs_bvals_mhd_side(aa_grid *g, int dir, int side);
int slabs_bvals_ionrad(aa_grid *g);
int slabs_new_dt_local(aa_grid *g, double *dt_cfl);
int slabs_cfl_max_v(aa_grid *g, double *v);
int slabs_integrate(aa_grid *g, int vl);
int slabs_ion_begin(aa_grid *g);
int slabs_ion_rates(aa_grid *g, double *dt_chem, double *dt_therm);
int slabs_ion_update(aa_grid *g, double dt, long long *cellcount, double *dt_hydro);
int slabs_ion_pass(aa_grid *g, int update, int sweep);
int slabs_ion_pick(aa_grid *g, int first, double limit);
int slabs_fetch_scalars(aa_grid *g);
int slabs_ion_finish(aa_grid *g);
int slabs_ion_run_phased(aa_grid *g, double limit, int *niter_out, double *dt_done_out);
int slabs_history(aa_grid *g, double *sums);
void slabs_push_state(aa_grid *g);
// evaluation of a StaticGravPot callback at zone centres and lower faces of a Grid (api.hip)
void aa_eval_grav_tables(const aa_params &p, const double dx[3], int N1, int N2, int N3, aa_gravpot_fn fn, std::vector<double> t[4]);
int aa_edgeflux_ready(aa_grid *g);    // fill GridS.EdgeFlux from the buffers of the one-kernel sub-cycle if that is still due
extern "C" int aa_download_cons_planes(aa_grid *g, int k_first, int nplanes, double *dst);
extern "C" int aa_download_edgeflux_planes(aa_grid *g, int nplanes, double *dst);

static inline double bits_to_double(unsigned long long b) { double x; memcpy(&x, &b, 8); return x; }
static inline unsigned long long double_to_bits(double x) { unsigned long long b; memcpy(&b, &x, 8); return b; }
extern "C" int aa_fetch_scalars(aa_grid *g);     // DevScalars device -> pinned host, synchronous
// one radiation sub-cycle with the step chosen on the device (api.hip); used by the single-Grid and the
// Mesh drivers (the multi-rank drivers need the reductions on the host and use aa_ion_rates/_update)
// ionrad_3d.c:862-1047 for the Grid of one level (api.hip): root (finegrid 0, limit = its dt) or refined level
// (limit = the time the root covered); *dt_done_out = the time the sub-cycles covered
extern "C" int aa_ion_run(aa_grid *g, int finegrid, double limit, int *niter_out, double *dt_done_out);
extern "C" int aa_ion_arm(aa_grid *g);
extern "C" int aa_ion_subcycle(aa_grid *g, double dt_done, double limit, double *dt, int *limit_hit, double *dt_chem,
                               double *dt_therm, long long *cellcount, double *dt_hydro);
