// hydro_launch.h -- launch interface of hydro_kernels.hip.  Included inside a namespace: `aa` for the kernels of every run,
// and once more inside `aa_cool` for the second compilation of the same file with -DAA_COOLING=1 (optically thin cooling in the
// CTU integrator: integrate_3d_ctu.c Steps 1c-3c, 8b, 11c), whose kernels api.hip launches when a cooling function is enrolled.
// ---- launch wrappers (hydro_kernels.hip) ------------------------------------------
void launch_slopes(const HostGrid &g, int nscal, int dir, hipStream_t st, const Real *src = nullptr);   // order 3: before the sweeps of a step (src: the conserved state they reconstruct; null = U)
// first-pass sweep of one direction; for dir 0 / 1 optionally only the k-planes ks-2+koff .. +kcnt-1 (kcnt < 0: to the end)
void launch_sweep(const HostGrid &g, int nscal, int dir, Real dt, bool grav, hipStream_t st, int koff = 0, int kcnt = -1);
void launch_correct(const HostGrid &g, int nscal, int dir, Real dt, bool grav, hipStream_t st);
void launch_sweep_correct_x1(const HostGrid &g, int nscal, Real dt, bool grav, hipStream_t st);
// the three correct passes in one kernel; x3f: and the x3 first pass (then launch_sweep(.., 2, ..) is not needed)
void launch_correct_all(const HostGrid &g, int nscal, Real dt, bool grav, bool x3f, hipStream_t st, bool edges_done = false);
// the tile-edge x1 fluxes launch_correct_all(x3f) otherwise starts with (edges_done): they need U only, so a caller may run them on
// another stream beside the x2 sweep
void launch_x1_edges(const HostGrid &g, int nscal, Real dt, bool grav, hipStream_t st);
bool ca_x1_on_board();     // with x3f, launch_correct_all also does the x1 first pass (then launch_sweep(.., 0, ..) is not needed)
void launch_flux2(const HostGrid &g, int nscal, int dir, hipStream_t st);
void launch_update(const HostGrid &g, int nscal, const Real *dhalf, Real dt, bool grav, hipStream_t st, DevScalars *sc = nullptr,
                   Real *cfl_part = nullptr, const unsigned char *pinmask = nullptr);
long update_blocks(const DevGrid &g);
void launch_flux2_update(const HostGrid &g, int nscal, Real dt, bool grav, const KeepPlanes *keep, hipStream_t st,
                         DevScalars *sc = nullptr, const unsigned char *pinmask = nullptr);   // sc: also new_dt's maxima (k_flux2_update<CFL>)
void launch_pinned_cfl(const DevGrid &g, long long n, const long long *idx, DevScalars *sc, hipStream_t st);
void launch_pin_mask(const DevGrid &g, long long n, const long long *idx, unsigned char *mask, hipStream_t st);   // flux2 x3 + update fused
void launch_vl_flux1(const DevGrid &g, int nscal, int dir, hipStream_t st);
void launch_vl_uhalf(const DevGrid &g, int nscal, Real dt, bool grav, hipStream_t st);
void launch_vl_predict(const DevGrid &g, int nscal, Real dt, bool grav, hipStream_t st);   // vl_flux1 x3 + vl_uhalf fused
void launch_vl_flux2(const HostGrid &g, int nscal, int dir, Real dt, hipStream_t st);
void launch_bc(const DevGrid &g, int nscal, int dir, int side, int flag, hipStream_t st);
void launch_bc_dir(const DevGrid &g, int nscal, int dir, int flag_in, int flag_out, hipStream_t st);   // both sides, one launch
void launch_bc_shell(const DevGrid &g, int nscal, const int flags[6], hipStream_t st);                 // all six sides, one launch (k_bc_shell)
unsigned reduce_blocks(long nzones);      // launch size of the grid-stride reduction kernels
void launch_cfl(const DevGrid &g, DevScalars *sc, hipStream_t st);
int  launch_history(const DevGrid &g, int nscal, Real *partial, hipStream_t st);   // returns the number of partial rows
void launch_aos_to_soa(const DevGrid &g, int nvar, const Real *aos, hipStream_t st);
void launch_soa_to_aos(const DevGrid &g, int nvar, Real *aos, hipStream_t st);
void launch_pinned(const DevGrid &g, int nvar, long long n, const long long *idx, const Real *vals,
                   hipStream_t st, DevScalars *sc = nullptr);      // sc: and the zones' share of new_dt's maxima, from the values written
void launch_pack_x3(const DevGrid &g, int nvar, int k0, Real *buf, hipStream_t st);
void launch_unpack_x3(const DevGrid &g, int nvar, int k0, const Real *buf, hipStream_t st);
void launch_pack_x2(const DevGrid &g, int nvar, int j0, Real *buf, hipStream_t st);      // pencils: the x2 halo (bvals_mhd.c:2462)
void launch_unpack_x2(const DevGrid &g, int nvar, int j0, const Real *buf, hipStream_t st);
void launch_test_fluxes(int nscal, Real gamma, int n, const Real *Ul, const Real *Ur, const Real *eta,
                        Real *F, hipStream_t st);
void launch_test_lr(int nscal, Real gamma, int n, const Real *W, Real dt, Real dx, int il, int iu,
                    Real *Wl, Real *Wr, hipStream_t st);
void launch_test_xdiv(int n, const Real *a, const Real *b, Real *out, hipStream_t st);     // out[5][n]: x_div, a/b, x_sqrt, sqrt, x_div_r
void launch_test_lr_ppm(int nscal, Real gamma, int n, const Real *W, Real dt, Real dx, int il, int iu,
                        Real *Wl, Real *Wr, hipStream_t st);

