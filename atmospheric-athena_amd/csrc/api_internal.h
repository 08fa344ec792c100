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
  aa::DevGrid d;
  aa::IonPar ion;
  hipStream_t st = nullptr; bool own_stream = false;
  aa::Real *pool = nullptr; size_t pool_doubles = 0;
  aa::DevScalars *sc = nullptr;        // device
  aa::DevScalars *sc_host = nullptr;   // pinned
  long long *pin_idx = nullptr; aa::Real *pin_val = nullptr; long long npin = 0;
  bool grav = false;
  int rad_dir = 0, nradplane = 0; aa::Real flux_i = 0;
  int level = 0;                       // DomainS.Level: > 0 only as a level of an aa_mesh
  bool fused_update = false;           // second-pass fluxes + update in one kernel (AA_FUSED_UPDATE=0 at aa_create: the unfused chain)
  bool correct_all = false;            // the three correct passes in one kernel (k_correct_all): Grids of 2^21 zones or more, or AA_CORRECT_ALL
  bool fused_rates = false;            // rates evaluated inside the ray sweep (k_ray_sweep<true>): 2^17 rays or more, or AA_FUSED_RATES
  bool vl_predict = false;             // van Leer predictor as one kernel (k_vl_predict): 2^18 zones or more, or AA_VL_PREDICT
  bool keep_flux = false;              // a level of an aa_mesh: RestrictCorrect reads the second-pass fluxes ...
  aa::KeepPlanes keep = {0, {{0}}};        // ... on these face planes (own boundaries + the child's outline)
  double time = 0, dt = 0; int nstep = 0;
  long long bytes = 0;
  bool prof = false;
  std::vector<ProfEntry> pe;
};

// ---- profiling: an event pair around every kernel-chain stage, on the launch stream ----------
struct Scope {
  aa_grid *g; int id; hipEvent_t a = nullptr, b = nullptr;
  Scope(aa_grid *g_, const char *name) : g(g_), id(-1) {
    if (!g->prof) return;
    for (size_t i = 0; i < g->pe.size(); i++) if (g->pe[i].name == name) { id = (int)i; break; }
    if (id < 0) { g->pe.push_back(ProfEntry()); id = (int)g->pe.size() - 1; g->pe[id].name = name; }
    hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, g->st);
  }
  ~Scope() {
    if (id < 0) return;
    hipEventRecord(b, g->st);
    g->pe[id].ev.push_back(a); g->pe[id].ev.push_back(b); g->pe[id].launches++;
  }
};

static inline double bits_to_double(unsigned long long b) { double x; memcpy(&x, &b, 8); return x; }
static inline unsigned long long double_to_bits(double x) { unsigned long long b; memcpy(&b, &x, 8); return b; }
extern "C" int aa_fetch_scalars(aa_grid *g);     // DevScalars device -> pinned host, synchronous
// one radiation sub-cycle with the step chosen on the device (api.hip); used by the single-Grid and the
// Mesh drivers (the multi-rank drivers need the reductions on the host and use aa_ion_rates/_update)
extern "C" int aa_ion_arm(aa_grid *g);
extern "C" int aa_ion_subcycle(aa_grid *g, double dt_done, double limit, double *dt, int *limit_hit, double *dt_chem,
                               double *dt_therm, long long *cellcount, double *dt_hydro);
