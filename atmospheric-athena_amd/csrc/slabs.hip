// slabs.hip -- ONE Grid of the caller on several GPUs, behind the unchanged C-ABI.
//
// The reference reaches N ranks through MPI: init_mesh.c:583-620 cuts a Domain into Grids, bvals_mhd.c:423-493
// exchanges the x3 ghost zones, new_dt.c:177 and ionrad_3d.c:275,399,554,672 reduce scalars.  A driver that owns
// ONE host Grid per Domain (the reference's main() linked on host/athena_shim.c, one process) gets the same
// decomposition here, inside the library: aa_create with aa_params.nslab > 1 (or AA_NGPU in the environment)
// returns a composite handle whose slabs are ordinary aa_grid objects, one per device, cut along x3 only (rays
// travel along x1; x3 planes are contiguous).  Every entry point of include/athena_amd.h forwards to the slabs:
//   * host transfers scatter / gather k-plane ranges of the caller's block (a slab's block incl. ghost planes is a
//     contiguous sub-block of it);
//   * bvals_mhd: x1, x2 and physical x3 boundaries per slab, then both directions of the 4-plane x3 halo as
//     device-to-device copies between neighbouring slabs (hipMemcpyPeerAsync over xGMI), ordered by events on the
//     slabs' own streams -- no host synchronisation;
//   * new_dt: every slab leaves its maxima in pinned host memory, the host folds them (one wait per step).  The radiation
//     sub-cycle's reductions stay on the devices: slab 0 pulls every slab's words (ion_pass.hip) with small peer copies,
//     every slab pulls the gathered set back and picks the step itself (k_ion_pick2 with nranks = nslab, as the
//     multi-process driver does behind its all-gather), all ordered by events -- the host reads back ONCE per
//     sub-cycle (slab 0's scalars), and the next pass is already queued behind it (MIN / MAX / integer SUM: bitwise
//     independent of the cut);
//   * peer access between neighbouring devices (halo) and between slab 0's device and every other (sub-cycle words) is asked for
//     and CHECKED; where it is refused (or with AA_SLAB_NO_PEER=1 / AA_SLAB_WORDS_STAGED=1) the halo / the words travel through
//     pinned host buffers instead, and the library says so on stderr;
//   * StaticGravPot is evaluated at the positions of the caller's undivided Grid, and the problem generator /
//     Userwork hooks work on the caller's block, so an N-slab run reproduces the 1-slab run bit for bit.
// Block decomposition as init_mesh.c:583-620: Nx3/N planes each, the remainder to the first slabs.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "api_internal.h"

using namespace aa;

struct SlabLink {
  int n = 0;
  std::vector<int> k0, nk, dev;            // first active plane (0-based, global), planes, HIP device of every slab
  std::vector<int> lo, hi;                 // neighbour slab below / above (-1: physical boundary)
  std::vector<Real*> send[2], recv[2];     // halo buffers on every slab's device
  std::vector<hipEvent_t> packed, copied;  // slab's send buffers are full / slab has pulled its neighbours' buffers
  std::vector<bool> copied_valid;
  // The peer copies run on a stream of their own beside the slab's kernels, and the unpack is left to whoever reads
  // the ghost planes next (halo_finish): the integrator first sweeps the planes that hold no neighbour data
  // (aa_integrate_begin), the ion step and new_dt never read them.
  std::vector<hipStream_t> xfer;
  std::vector<hipEvent_t> unpacked;        // slab's recv buffers have been unpacked (free for the next copy)
  std::vector<bool> unpacked_valid;
  bool halo_due = false;                   // messages posted, ghost planes not yet written
  DevScalars *hsc = nullptr;               // pinned: n
  std::vector<Real*> dwords_all;           // device: n x AA_ION_WORDS on every slab
  std::vector<hipEvent_t> wdone;           // slab's words of the pass just queued are complete
  hipEvent_t gathered = nullptr;           // slab 0 holds every slab's words
  // no peer access between two neighbouring devices (or AA_SLAB_NO_PEER=1): the halo goes device -> pinned host -> device
  bool staged = false;
  std::vector<Real*> hstage[2];            // pinned: the slab's two send buffers on the host
  std::vector<hipStream_t> xout;           // the slab's device -> host copies
  std::vector<hipEvent_t> staged_ev;       // ... are complete
  // the radiation sub-cycle's words: slab 0 <-> every slab.  Where peer access between device 0 and ANY slab's device is missing (checked at
  // create time for every pair, not only neighbours), or the halo is staged anyway, the words go through pinned host memory instead: every
  // slab copies its words into hwords[round & 1][s] behind its pass and, once all have, reads the whole set back -- no peer copy at all.
  // Two rounds of buffers: a slab can be one pick ahead of another, never two (its next pass waits for everybody's words of this one).
  bool words_staged = false;
  Real *hwords[2] = {nullptr, nullptr};    // pinned: n x AA_ION_WORDS each
  unsigned wround = 0;
  size_t halo = 0;
  int home = 0;                            // the caller's current device when the handle was made
};

// every composite entry point leaves the caller's current HIP device as it found it
struct DevGuard { int d = -1; DevGuard() { if (hipGetDevice(&d) != hipSuccess) d = -1; } ~DevGuard() { if (d >= 0) (void)hipSetDevice(d); } };
#define SLAB_DEV(L, s) HIPCHK(hipSetDevice((L)->dev[s]))
static int halo_finish(aa_grid *g);          // the ghost planes of a posted exchange are written before anything reads them
#define HALO_FLUSH(g) do { int rc_ = halo_finish(g); if (rc_) return rc_; } while (0)

void slabs_push_state(aa_grid *g)
{ for (aa_grid *c : g->slab) { c->time = g->time; c->dt = g->dt; c->nstep = g->nstep; } }

int slabs_create(const aa_params *p, int nslab, aa_grid **out)
{
  if (nslab > 64) return aa_fail(-1, "[aa_create]: %d slabs", nslab);
  if (p->level != 0) return aa_fail(-1, "[aa_create]: only the root level can be cut into slabs");
  if (p->Nx[2]/nslab < AA_NGHOST) return aa_fail(-1, "[aa_create]: %d x3 planes cut into %d slabs leave fewer than nghost = %d planes each",
                                                  p->Nx[2], nslab, AA_NGHOST);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return aa_fail(-3, "[aa_create]: no HIP device visible -- this library has no CPU path");
  DevGuard keep;
  aa_grid *g = new aa_grid();
  g->p = *p; g->p.nslab = nslab;
  memset(&g->d, 0, sizeof g->d);
  g->level = 0;
  SlabLink *L = new SlabLink(); g->link = L; L->n = nslab; L->home = keep.d;
#define CREATE_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { slabs_destroy(g); \
  return aa_fail(-2, "[aa_create] HIP error %s at %s:%d: %s", #x, __FILE__, __LINE__, hipGetErrorString(e_)); } } while (0)
  // devices: AA_SLAB_DEVICES="0,1,2,.." or round robin over the visible ones (one device: a rehearsal of the
  // multi-GPU path, every slab on it)
  std::vector<int> devs;
  if (const char *e = getenv("AA_SLAB_DEVICES")) { for (const char *q = e; *q; ) { devs.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q) q++; } }
  const bool periodic = (p->bc[4] == AA_BC_PERIODIC && p->bc[5] == AA_BC_PERIODIC);
  const Real dx3 = (p->xmax[2] - p->xmin[2])/(Real)(p->rootNx[2]);
  int k0 = 0; Real minx3 = p->MinX[2];
  for (int s = 0; s < nslab; s++) {
    const int nk = p->Nx[2]/nslab + (s < p->Nx[2] % nslab ? 1 : 0);      // init_mesh.c:583-620
    aa_params ps = *p;
    ps.nslab = 1;
    ps.Nx[2] = nk; ps.MinX[2] = minx3;                                    // init_grid.c:104-111
    ps.device = devs.empty() ? (p->device + s) % ndev : devs[s % devs.size()];
    if (ps.device >= ndev) { slabs_destroy(g); return aa_fail(-1, "[aa_create]: slab %d wants HIP device %d of %d", s, ps.device, ndev); }
    const int lo = (s > 0) ? s - 1 : (periodic ? nslab - 1 : -1), hi = (s < nslab - 1) ? s + 1 : (periodic ? 0 : -1);
    if (lo >= 0) ps.bc[4] = AA_BC_NONE;
    if (hi >= 0) ps.bc[5] = AA_BC_NONE;
    aa_grid *c = nullptr;
    int rc = aa_create(&ps, &c);
    if (rc) { slabs_destroy(g); return rc; }
    g->slab.push_back(c);
    L->k0.push_back(k0); L->nk.push_back(nk); L->dev.push_back(ps.device); L->lo.push_back(lo); L->hi.push_back(hi);
    k0 += nk; minx3 += (Real)nk*dx3;
  }
  g->ion_fused = g->slab[0]->ion_fused;
  for (aa_grid *c : g->slab) if (c->ion_fused != g->ion_fused) { slabs_destroy(g); return aa_fail(-1, "[aa_create]: slabs disagree on the sub-cycle path"); }
  L->halo = (size_t)aa_halo_doubles(g->slab[0]);
  L->packed.assign(nslab, nullptr); L->copied.assign(nslab, nullptr); L->copied_valid.assign(nslab, false);
  L->xfer.assign(nslab, nullptr); L->unpacked.assign(nslab, nullptr); L->unpacked_valid.assign(nslab, false);
  L->dwords_all.assign(nslab, nullptr); L->wdone.assign(nslab, nullptr);
  L->xout.assign(nslab, nullptr); L->staged_ev.assign(nslab, nullptr);
  for (int w = 0; w < 2; w++) { L->send[w].assign(nslab, nullptr); L->recv[w].assign(nslab, nullptr); L->hstage[w].assign(nslab, nullptr); }
  // peer access between neighbouring devices: asked for, and CHECKED
  { const char *e = getenv("AA_SLAB_NO_PEER"); L->staged = e && atoi(e) != 0; }
  if (L->staged) fprintf(stderr, "[athena_amd] AA_SLAB_NO_PEER: the x3 halo of the slabs travels through pinned host memory\n");
  for (int s = 0; s < nslab && !L->staged; s++) {
    CREATE_CHK(hipSetDevice(L->dev[s]));
    for (int o : {L->lo[s], L->hi[s]})
      if (o >= 0 && L->dev[o] != L->dev[s]) {
        int can = 0;
        hipError_t e = hipDeviceCanAccessPeer(&can, L->dev[s], L->dev[o]);
        if (e == hipSuccess && can) {
          e = hipDeviceEnablePeerAccess(L->dev[o], 0);
          if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
        }
        if (e != hipSuccess || !can) {
          (void)hipGetLastError();
          fprintf(stderr, "[athena_amd] WARNING: no peer access from HIP device %d to %d (%s): the x3 halo of the slabs travels through "
                          "pinned host memory instead of xGMI\n", L->dev[s], L->dev[o], e != hipSuccess ? hipGetErrorString(e) : "refused");
          L->staged = true;
        }
      }
  }
  // ... and between slab 0's device and EVERY slab's device for the sub-cycle's words (slab 0 pulls from all, all pull from slab 0)
  { const char *e = getenv("AA_SLAB_WORDS_STAGED"); L->words_staged = L->staged || (e && atoi(e) != 0); }
  for (int s = 1; s < nslab && !L->words_staged; s++) {
    if (L->dev[s] == L->dev[0]) continue;
    for (int dir = 0; dir < 2 && !L->words_staged; dir++) {
      const int from = dir ? L->dev[s] : L->dev[0], to = dir ? L->dev[0] : L->dev[s];
      CREATE_CHK(hipSetDevice(from));
      int can = 0;
      hipError_t e = hipDeviceCanAccessPeer(&can, from, to);
      if (e == hipSuccess && can) {
        e = hipDeviceEnablePeerAccess(to, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
      }
      if (e != hipSuccess || !can) {
        (void)hipGetLastError();
        fprintf(stderr, "[athena_amd] WARNING: no peer access from HIP device %d to %d (%s): the radiation sub-cycle's reduction words of the slabs "
                        "travel through pinned host memory\n", from, to, e != hipSuccess ? hipGetErrorString(e) : "refused");
        L->words_staged = true;
      }
    }
  }
  if (L->words_staged) for (int w = 0; w < 2; w++) {
    CREATE_CHK(hipHostMalloc(&L->hwords[w], (size_t)nslab*AA_ION_WORDS*sizeof(Real)));
    memset(L->hwords[w], 0, (size_t)nslab*AA_ION_WORDS*sizeof(Real));
  }
  for (int s = 0; s < nslab; s++) {
    CREATE_CHK(hipSetDevice(L->dev[s]));
    for (int w = 0; w < 2; w++) {
      CREATE_CHK(hipMalloc(&L->send[w][s], L->halo*sizeof(Real)));
      CREATE_CHK(hipMalloc(&L->recv[w][s], L->halo*sizeof(Real)));
      if (L->staged) CREATE_CHK(hipHostMalloc(&L->hstage[w][s], L->halo*sizeof(Real)));
    }
    CREATE_CHK(hipMalloc(&L->dwords_all[s], (size_t)nslab*AA_ION_WORDS*sizeof(Real)));
    CREATE_CHK(hipEventCreateWithFlags(&L->packed[s], hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&L->copied[s], hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&L->unpacked[s], hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&L->wdone[s], hipEventDisableTiming));
    CREATE_CHK(hipStreamCreateWithFlags(&L->xfer[s], hipStreamNonBlocking));
    if (L->staged) {
      CREATE_CHK(hipStreamCreateWithFlags(&L->xout[s], hipStreamNonBlocking));
      CREATE_CHK(hipEventCreateWithFlags(&L->staged_ev[s], hipEventDisableTiming));
    }
  }
  CREATE_CHK(hipSetDevice(L->dev[0]));
  CREATE_CHK(hipEventCreateWithFlags(&L->gathered, hipEventDisableTiming));
  CREATE_CHK(hipHostMalloc(&L->hsc, (size_t)nslab*sizeof(DevScalars)));
  CREATE_CHK(hipHostMalloc(&g->sc_host, sizeof(DevScalars)));
#undef CREATE_CHK
  for (aa_grid *c : g->slab) g->bytes += c->bytes;
  *out = g;
  return 0;
}

void slabs_destroy(aa_grid *g)      // (also the error path of slabs_create: whatever exists so far)
{
  DevGuard keep;
  SlabLink *L = g->link;
  for (size_t s = 0; s < g->slab.size(); s++) {
    hipSetDevice(L->dev[s]);
    hipStreamSynchronize(g->slab[s]->st);
    if (s < L->xfer.size() && L->xfer[s]) { hipStreamSynchronize(L->xfer[s]); hipStreamDestroy(L->xfer[s]); }
    if (s < L->xout.size() && L->xout[s]) { hipStreamSynchronize(L->xout[s]); hipStreamDestroy(L->xout[s]); }
    if (s < L->unpacked.size() && L->unpacked[s]) hipEventDestroy(L->unpacked[s]);
    if (s < L->staged_ev.size() && L->staged_ev[s]) hipEventDestroy(L->staged_ev[s]);
    if (s < L->wdone.size() && L->wdone[s]) hipEventDestroy(L->wdone[s]);
    if (s < L->send[0].size()) for (int w = 0; w < 2; w++) {
      if (L->send[w][s]) hipFree(L->send[w][s]);
      if (L->recv[w][s]) hipFree(L->recv[w][s]);
      if (L->hstage[w][s]) hipHostFree(L->hstage[w][s]);
    }
    if (s < L->dwords_all.size() && L->dwords_all[s]) hipFree(L->dwords_all[s]);
    if (s < L->packed.size() && L->packed[s]) hipEventDestroy(L->packed[s]);
    if (s < L->copied.size() && L->copied[s]) hipEventDestroy(L->copied[s]);
    aa_destroy(g->slab[s]);
  }
  if (L->gathered) hipEventDestroy(L->gathered);
  if (L->hsc) hipHostFree(L->hsc);
  for (int w = 0; w < 2; w++) if (L->hwords[w]) hipHostFree(L->hwords[w]);
  if (g->sc_host) hipHostFree(g->sc_host);
  delete L;
  g->link = nullptr;
  g->slab.clear();
  delete g;
}

int slabs_sync(aa_grid *g)
{
  DevGuard keep;
  HALO_FLUSH(g);
  SlabLink *L = g->link;
  for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); HIPCHK(hipStreamSynchronize(g->slab[s]->st)); }
  return 0;
}
long long slabs_device_bytes(const aa_grid *g) { return g->bytes; }

// ---- host transfers: k-plane ranges of the caller's block --------------------------------------------
int slabs_upload_cons(aa_grid *g, const double *U)
{
  DevGuard keep;
  HALO_FLUSH(g);
  SlabLink *L = g->link;
  const size_t pl = (size_t)(g->p.Nx[0] + 2*AA_NGHOST)*(g->p.Nx[1] + 2*AA_NGHOST)*(5 + g->p.nscal);
  for (int s = 0; s < L->n; s++) {
    SLAB_DEV(L, s);
    int rc = aa_upload_cons(g->slab[s], U + (size_t)L->k0[s]*pl);       // planes k0 .. k0 + nk + 8 of the block
    if (rc) return rc;
  }
  return 0;
}
int slabs_download_cons(aa_grid *g, double *U)
{
  DevGuard keep;
  HALO_FLUSH(g);
  SlabLink *L = g->link;
  const size_t pl = (size_t)(g->p.Nx[0] + 2*AA_NGHOST)*(g->p.Nx[1] + 2*AA_NGHOST)*(5 + g->p.nscal);
  for (int s = 0; s < L->n; s++) {
    SLAB_DEV(L, s);
    // the slab's active planes, plus the ghost planes only where it ends at the Grid's own boundary
    int kf = AA_NGHOST, np = L->nk[s];
    if (s == 0) { kf = 0; np += AA_NGHOST; }
    if (s == L->n - 1) np += AA_NGHOST;
    int rc = aa_download_cons_planes(g->slab[s], kf, np, U + (size_t)(L->k0[s] + kf)*pl);
    if (rc) return rc;
  }
  return 0;
}
int slabs_upload_edgeflux(aa_grid *g, const double *ef)
{
  DevGuard keep;
  SlabLink *L = g->link;
  const size_t pl = (size_t)(g->p.Nx[0] + 1)*(g->p.Nx[1] + 1);
  for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); int rc = aa_upload_edgeflux(g->slab[s], ef + (size_t)L->k0[s]*pl); if (rc) return rc; }
  return 0;
}
int slabs_download_edgeflux(aa_grid *g, double *ef)
{
  DevGuard keep;
  SlabLink *L = g->link;
  const size_t pl = (size_t)(g->p.Nx[0] + 1)*(g->p.Nx[1] + 1);
  for (int s = 0; s < L->n; s++) {
    SLAB_DEV(L, s);
    int rc = aa_download_edgeflux_planes(g->slab[s], L->nk[s] + (s == L->n - 1 ? 1 : 0), ef + (size_t)L->k0[s]*pl);
    if (rc) return rc;
  }
  return 0;
}

// ---- hooks ---------------------------------------------------------------------------------------------
int slabs_set_cooling(aa_grid *g, int kind)
{
  DevGuard keep;
  SlabLink *L = g->link;
  for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); int rc = aa_set_cooling(g->slab[s], kind); if (rc) return rc; }
  g->cool = kind;
  return 0;
}
int slabs_set_grav_tables(aa_grid *g, const double *pc, const double *p1, const double *p2, const double *p3)
{
  DevGuard keep;
  SlabLink *L = g->link;
  const size_t pl = (size_t)(g->p.Nx[0] + 2*AA_NGHOST)*(g->p.Nx[1] + 2*AA_NGHOST);
  for (int s = 0; s < L->n; s++) {
    SLAB_DEV(L, s);
    const size_t o = (size_t)L->k0[s]*pl;
    int rc = pc ? aa_set_static_grav_tables(g->slab[s], pc + o, p1 + o, p2 + o, p3 + o)
                : aa_set_static_grav_tables(g->slab[s], nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
  }
  g->grav = (pc != nullptr);
  return 0;
}

int slabs_set_pinned_cells(aa_grid *g, long long n, const long long *index, const double *values)
{
  DevGuard keep;
  HALO_FLUSH(g);
  SlabLink *L = g->link;
  const int nvar = 5 + g->p.nscal;
  const long long pl = (long long)(g->p.Nx[0] + 2*AA_NGHOST)*(g->p.Nx[1] + 2*AA_NGHOST);
  g->npin = 0;
  for (int s = 0; s < L->n; s++) {
    // a zone belongs to the slab in which it is active (or a ghost zone of the Grid's own boundary)
    const long long klo = (s == 0) ? 0 : L->k0[s] + AA_NGHOST;
    const long long khi = (s == L->n - 1) ? g->p.Nx[2] + 2*AA_NGHOST : L->k0[s] + AA_NGHOST + L->nk[s];
    std::vector<long long> idx; std::vector<double> val;
    for (long long q = 0; q < n; q++) {
      const long long k = index[q]/pl;
      if (k < klo || k >= khi) continue;
      idx.push_back(index[q] - (long long)L->k0[s]*pl);
      for (int v = 0; v < nvar; v++) val.push_back(values[q*nvar + v]);
    }
    SLAB_DEV(L, s);
    int rc = aa_set_pinned_cells(g->slab[s], (long long)idx.size(), idx.data(), val.data());
    if (rc) return rc;
    g->npin += (long long)idx.size();
  }
  return 0;
}
int slabs_apply_pinned_cells(aa_grid *g)
{
  DevGuard keep;
  HALO_FLUSH(g);
  SlabLink *L = g->link;
  for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); int rc = aa_apply_pinned_cells(g->slab[s]); if (rc) return rc; }
  return 0;
}
int slabs_add_radplane(aa_grid *g, int dir, double flux)
{
  DevGuard keep;
  SlabLink *L = g->link;
  if (dir == -2) g->ion_fused = false;
  for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); int rc = aa_add_radplane_3d(g->slab[s], dir, flux); if (rc) return rc; }
  return 0;
}
int slabs_bvals_ionrad(aa_grid *g)
{
  DevGuard keep;
  SlabLink *L = g->link;
  for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); int rc = aa_bvals_ionrad(g->slab[s]); if (rc) return rc; }
  return 0;
}

// ---- ghost zones: bvals_mhd.c:423-493 between slabs ---------------------------------------------------
// MPI_Waitall + unpack_ix3 / unpack_ox3: the slab's kernel stream waits for its copies and writes its ghost planes
static int halo_finish(aa_grid *g)
{
  DevGuard keep;
  SlabLink *L = g->link;
  if (!L->halo_due) return 0;
  L->halo_due = false;
  const int nvar = 5 + g->p.nscal;
  for (int r = 0; r < L->n; r++) {
    if (L->lo[r] < 0 && L->hi[r] < 0) continue;
    SLAB_DEV(L, r);
    aa_grid *c = g->slab[r];
    HIPCHK(hipStreamWaitEvent(c->st, L->copied[r], 0));
    if (L->lo[r] >= 0) launch_unpack_x3(c->d, nvar, c->d.ks - AA_NGHOST, L->recv[0][r], c->st);
    if (L->hi[r] >= 0) launch_unpack_x3(c->d, nvar, c->d.ke + 1, L->recv[1][r], c->st);
    HIPCHK(hipEventRecord(L->unpacked[r], c->st));
    L->unpacked_valid[r] = true;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// pack_ix3 / pack_ox3 + the messages (MPI_Isend / MPI_Irecv): packed on the kernel streams, pulled by the receivers on
// their copy streams
static int halo_post(aa_grid *g)
{
  DevGuard keep;
  SlabLink *L = g->link;
  HALO_FLUSH(g);
  const int nvar = 5 + g->p.nscal;
  const size_t bytes = L->halo*sizeof(Real);
  bool any = false;
  for (int s = 0; s < L->n; s++) {
    if (L->lo[s] < 0 && L->hi[s] < 0) continue;
    any = true;
    SLAB_DEV(L, s);
    aa_grid *c = g->slab[s];
    // my send buffers are free once both neighbours have pulled the previous exchange out of them
    for (int o : {L->lo[s], L->hi[s]}) if (o >= 0 && L->copied_valid[o]) HIPCHK(hipStreamWaitEvent(c->st, L->copied[o], 0));
    if (L->lo[s] >= 0) launch_pack_x3(c->d, nvar, c->d.ks, L->send[0][s], c->st);                       // pack_ix3
    if (L->hi[s] >= 0) launch_pack_x3(c->d, nvar, c->d.ke - AA_NGHOST + 1, L->send[1][s], c->st);       // pack_ox3
    HIPCHK(hipEventRecord(L->packed[s], c->st));
    if (L->staged) {             // no peer access: the send buffers go to pinned host memory on a copy stream of the sender
      HIPCHK(hipStreamWaitEvent(L->xout[s], L->packed[s], 0));
      if (L->lo[s] >= 0) HIPCHK(hipMemcpyAsync(L->hstage[0][s], L->send[0][s], bytes, hipMemcpyDeviceToHost, L->xout[s]));
      if (L->hi[s] >= 0) HIPCHK(hipMemcpyAsync(L->hstage[1][s], L->send[1][s], bytes, hipMemcpyDeviceToHost, L->xout[s]));
      HIPCHK(hipEventRecord(L->staged_ev[s], L->xout[s]));
    }
  }
  for (int r = 0; r < L->n; r++) {
    if (L->lo[r] < 0 && L->hi[r] < 0) continue;
    SLAB_DEV(L, r);
    hipStream_t x = L->xfer[r];
    if (L->unpacked_valid[r]) HIPCHK(hipStreamWaitEvent(x, L->unpacked[r], 0));     // my recv buffers are free
    if (L->lo[r] >= 0) {         // the lower neighbour's upper planes fill my lower ghost planes
      const int o = L->lo[r];
      HIPCHK(hipStreamWaitEvent(x, L->staged ? L->staged_ev[o] : L->packed[o], 0));
      if (L->staged) HIPCHK(hipMemcpyAsync(L->recv[0][r], L->hstage[1][o], bytes, hipMemcpyHostToDevice, x));
      else HIPCHK(hipMemcpyPeerAsync(L->recv[0][r], L->dev[r], L->send[1][o], L->dev[o], bytes, x));
    }
    if (L->hi[r] >= 0) {
      const int o = L->hi[r];
      HIPCHK(hipStreamWaitEvent(x, L->staged ? L->staged_ev[o] : L->packed[o], 0));
      if (L->staged) HIPCHK(hipMemcpyAsync(L->recv[1][r], L->hstage[0][o], bytes, hipMemcpyHostToDevice, x));
      else HIPCHK(hipMemcpyPeerAsync(L->recv[1][r], L->dev[r], L->send[0][o], L->dev[o], bytes, x));
    }
    HIPCHK(hipEventRecord(L->copied[r], x));
    L->copied_valid[r] = true;
  }
  L->halo_due = any;
  HIPCHK(hipGetLastError());
  return 0;
}

int slabs_bvals_mhd(aa_grid *g)
{
  DevGuard keep;
  SlabLink *L = g->link;
  HALO_FLUSH(g);
  for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); int rc = aa_bvals_mhd(g->slab[s]); if (rc) return rc; }   // x1, x2, physical x3
  return halo_post(g);          // after x1 and x2, all i and j incl. ghosts: the corners travel (bvals_mhd.c:170)
}

// one (*BCFun)(pGrid) call of bvals_mhd.c:196-420 for drivers that interleave user boundary functions: the slabs'
// exchange belongs to the x3 step and is done with its inner side
int slabs_bvals_mhd_side(aa_grid *g, int dir, int side)
{
  DevGuard keep;
  SlabLink *L = g->link;
  HALO_FLUSH(g);
  if (dir < 2) {
    for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); int rc = aa_bvals_mhd_side(g->slab[s], dir, side); if (rc) return rc; }
    return 0;
  }
  const int s = side ? L->n - 1 : 0;
  SLAB_DEV(L, s);
  int rc = aa_bvals_mhd_side(g->slab[s], dir, side); if (rc) return rc;
  return side == 0 ? halo_post(g) : 0;
}

// ---- time step: new_dt.c:72-177 ----------------------------------------------------------------------
static int cfl_all(aa_grid *g)
{
  DevGuard keep;
  SlabLink *L = g->link;
  for (int s = 0; s < L->n; s++) {
    SLAB_DEV(L, s);
    aa_grid *c = g->slab[s];
    { Scope sc(c, "new_dt");
      HIPCHK(hipMemsetAsync(c->sc->max_v, 0, 3*sizeof(unsigned long long), c->st));
      launch_cfl(c->d, c->sc, c->st); }
    HIPCHK(hipMemcpyAsync(&L->hsc[s], c->sc, sizeof(DevScalars), hipMemcpyDeviceToHost, c->st));
  }
  for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); HIPCHK(hipStreamSynchronize(g->slab[s]->st)); }
  g->host_syncs++;
  return 0;
}
int slabs_cfl_max_v(aa_grid *g, double *v)
{
  int rc = cfl_all(g); if (rc) return rc;
  for (int d = 0; d < 3; d++) {
    v[d] = 0.0;
    for (int s = 0; s < g->link->n; s++) { const double q = bits_to_double(g->link->hsc[s].max_v[d]); v[d] = (v[d] > q) ? v[d] : q; }
  }
  return 0;
}
int slabs_new_dt_local(aa_grid *g, double *dt_cfl)
{
  // every slab's own dt as new_dt.c:159-170 computes it, then the MIN over slabs (:177)
  int rc = cfl_all(g); if (rc) return rc;
  double dt = DBL_MAX;
  for (int s = 0; s < g->link->n; s++) {
    double max_dti = 0.0;
    for (int d = 0; d < 3; d++) { const double v = bits_to_double(g->link->hsc[s].max_v[d])/g->slab[s]->d.dx[d]; max_dti = (max_dti > v) ? max_dti : v; }
    const double q = g->p.cour_no/max_dti;
    dt = (q < dt) ? q : dt;
  }
  *dt_cfl = dt;
  return 0;
}

int slabs_integrate(aa_grid *g, int vl)
{
  DevGuard keep;
  SlabLink *L = g->link;
  slabs_push_state(g);
  if (L->halo_due) {     // the ghost planes are still on their way: first the sweeps that do not read them
    if (!vl) for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); int rc = aa_integrate_begin(g->slab[s]); if (rc) return rc; }
    HALO_FLUSH(g);
  }
  for (int s = 0; s < L->n; s++) {
    SLAB_DEV(L, s);
    int rc = vl ? aa_integrate_3d_vl(g->slab[s]) : aa_integrate_3d_ctu(g->slab[s]);
    if (rc) return rc;
  }
  return 0;
}

// ---- radiation -----------------------------------------------------------------------------------------
int slabs_ion_begin(aa_grid *g)
{
  DevGuard keep;
  SlabLink *L = g->link;
  slabs_push_state(g);
  for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); int rc = aa_ion_begin(g->slab[s]); if (rc) return rc; }
  return 0;
}

// the one-kernel sub-cycle: every slab runs its pass and leaves its words in its own device memory
int slabs_ion_pass(aa_grid *g, int update, int sweep)
{
  DevGuard keep;
  SlabLink *L = g->link;
  for (int s = 0; s < L->n; s++) {
    SLAB_DEV(L, s);
    aa_grid *c = g->slab[s];
    int rc = aa_ion_pass(c, update, sweep, nullptr); if (rc) return rc;
    if (L->words_staged)      // the slab's words to the host buffer of this round (slabs_ion_pick closes the round)
      HIPCHK(hipMemcpyAsync(L->hwords[L->wround & 1] + (size_t)s*AA_ION_WORDS, c->ion_words, AA_ION_WORDS*sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipEventRecord(L->wdone[s], c->st));
  }
  return 0;
}

// ionrad_3d.c:275,399,554,672 (the MPI_Allreduce calls) without the host: slab 0 pulls every slab's words (after ALL of
// them are complete: by then every slab has also finished reading the previous set out of slab 0), every slab pulls the
// gathered set back and picks the step itself.  The next pass of slab s is queued behind its own pick, i.e. behind the
// gather: its words are not overwritten before slab 0 has them.
int slabs_ion_pick(aa_grid *g, int first, double limit)
{
  DevGuard keep;
  SlabLink *L = g->link;
  const size_t wb = AA_ION_WORDS*sizeof(double);
  if (L->words_staged) {
    // through pinned host memory: every slab waits for everybody's words of this round and reads the whole set back itself
    const Real *hw = L->hwords[L->wround & 1];
    L->wround++;
    for (int s = 0; s < L->n; s++) {
      SLAB_DEV(L, s);
      aa_grid *c = g->slab[s];
      if (first && c->ion_spec_armed && limit != c->ion_spec_limit)
        return aa_fail(-1, "[aa_ion_pick]: limit %.17g, but aa_ion_speculate was told %.17g", limit, c->ion_spec_limit);
      for (int t = 0; t < L->n; t++) if (t != s) HIPCHK(hipStreamWaitEvent(c->st, L->wdone[t], 0));
      HIPCHK(hipMemcpyAsync(L->dwords_all[s], hw, (size_t)L->n*wb, hipMemcpyHostToDevice, c->st));
      launch_ion_pick2(L->dwords_all[s], L->n, c->sc, first, limit, c->st, first ? (c->ion_spec_armed ? 1 : 0) : 0);
      if (!first) c->ion_spec_armed = false;
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  {
    SLAB_DEV(L, 0);
    aa_grid *c0 = g->slab[0];
    for (int s = 1; s < L->n; s++) HIPCHK(hipStreamWaitEvent(c0->st, L->wdone[s], 0));
    for (int s = 0; s < L->n; s++) {
      if (L->dev[s] == L->dev[0]) HIPCHK(hipMemcpyAsync(L->dwords_all[0] + (size_t)s*AA_ION_WORDS, g->slab[s]->ion_words, wb, hipMemcpyDeviceToDevice, c0->st));
      else HIPCHK(hipMemcpyPeerAsync(L->dwords_all[0] + (size_t)s*AA_ION_WORDS, L->dev[0], g->slab[s]->ion_words, L->dev[s], wb, c0->st));
    }
    HIPCHK(hipEventRecord(L->gathered, c0->st));
  }
  for (int s = 0; s < L->n; s++) {
    SLAB_DEV(L, s);
    aa_grid *c = g->slab[s];
    if (first && c->ion_spec_armed && limit != c->ion_spec_limit)
      return aa_fail(-1, "[aa_ion_pick]: limit %.17g, but aa_ion_speculate was told %.17g", limit, c->ion_spec_limit);
    if (s > 0) {
      HIPCHK(hipStreamWaitEvent(c->st, L->gathered, 0));
      if (L->dev[s] == L->dev[0]) HIPCHK(hipMemcpyAsync(L->dwords_all[s], L->dwords_all[0], (size_t)L->n*wb, hipMemcpyDeviceToDevice, c->st));
      else HIPCHK(hipMemcpyPeerAsync(L->dwords_all[s], L->dev[s], L->dwords_all[0], L->dev[0], (size_t)L->n*wb, c->st));
    }
    launch_ion_pick2(L->dwords_all[s], L->n, c->sc, first, limit, c->st, first ? (c->ion_spec_armed ? 1 : 0) : 0);
    if (!first) c->ion_spec_armed = false;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// the ONE read-back of a sub-cycle: slab 0's scalars (every slab holds the same ones)
int slabs_fetch_scalars(aa_grid *g)
{
  DevGuard keep;
  SlabLink *L = g->link;
  SLAB_DEV(L, 0);
  aa_grid *c0 = g->slab[0];
  HIPCHK(hipMemcpyAsync(g->sc_host, c0->sc, sizeof(DevScalars), hipMemcpyDeviceToHost, c0->st));
  HIPCHK(hipStreamSynchronize(c0->st));
  g->host_syncs++;
  return 0;
}

int slabs_ion_finish(aa_grid *g)
{
  DevGuard keep;
  SlabLink *L = g->link;
  for (int s = 0; s < L->n; s++) { SLAB_DEV(L, s); int rc = aa_ion_finish(g->slab[s]); if (rc) return rc; }
  return 0;
}

// the two-kernel sub-cycle (short rays): the reductions go through the host twice per sub-cycle, as under MPI
int slabs_ion_rates(aa_grid *g, double *dt_chem, double *dt_therm)
{
  DevGuard keep;
  SlabLink *L = g->link;
  *dt_chem = DBL_MAX; *dt_therm = DBL_MAX;
  for (int s = 0; s < L->n; s++) {
    SLAB_DEV(L, s);
    double a, b; int rc = aa_ion_rates(g->slab[s], &a, &b); if (rc) return rc;
    *dt_chem = (a < *dt_chem) ? a : *dt_chem; *dt_therm = (b < *dt_therm) ? b : *dt_therm;
  }
  return 0;
}
int slabs_ion_update(aa_grid *g, double dt, long long *cellcount, double *dt_hydro)
{
  DevGuard keep;
  SlabLink *L = g->link;
  long long n = 0; double h = DBL_MAX;
  for (int s = 0; s < L->n; s++) {
    SLAB_DEV(L, s);
    long long a; double b; int rc = aa_ion_update(g->slab[s], dt, &a, &b); if (rc) return rc;
    n += a; h = (b < h) ? b : h;
  }
  if (cellcount) *cellcount = n;
  if (dt_hydro) *dt_hydro = h;
  return 0;
}
int slabs_ion_run_phased(aa_grid *g, double limit, int *niter_out, double *dt_done_out)
{
  // ionrad_3d.c:862-1012, root level
  double dt_chem, dt_therm, dt_hydro = 0, dt, dt_done = 0.0;
  long long cellcount = 0;
  int niter = 0, hydro_done = 0, rc;
  if ((rc = slabs_ion_begin(g))) return rc;
  while (!hydro_done) {
    if ((rc = slabs_ion_rates(g, &dt_chem, &dt_therm))) return rc;
    dt = (dt_therm < dt_chem) ? dt_therm : dt_chem;
    if (dt_done + dt > limit) { dt = limit - dt_done; hydro_done = 1; }
    if ((rc = slabs_ion_update(g, dt, &cellcount, &dt_hydro))) return rc;
    dt_done += dt;
    niter++;
    if (cellcount > MAXCELLCOUNT) { g->dt = dt_done; break; }
    if (hydro_done) break;
    if (dt_hydro < dt_done) { g->dt = dt_done; break; }
  }
  *niter_out = niter; *dt_done_out = dt_done;
  return 0;
}

int slabs_history(aa_grid *g, double *sums)
{
  DevGuard keep;
  SlabLink *L = g->link;
  for (int q = 0; q < 9; q++) sums[q] = 0.0;
  for (int s = 0; s < L->n; s++) {            // dump_history.c:257 MPI_Reduce(SUM) over the Grids of a Domain
    SLAB_DEV(L, s);
    double part[9]; int rc = aa_history(g->slab[s], part); if (rc) return rc;
    for (int q = 0; q < 9; q++) sums[q] += part[q];
  }
  return 0;
}
