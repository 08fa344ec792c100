/* oracle/athena_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, one thread) of the hot path of tripathi/Atmospheric-Athena
 * for the configuration HYDRO / ADIABATIC / CARTESIAN / SECOND_ORDER_CHAR / ROE_FLUX /
 * CTU_INTEGRATOR / H_CORRECTION / ION_RADIATION + ION_RADPLANE / NSCALARS in {0,1}.
 *
 * It is the CHECKER for the HIP product, never the product: only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it.  It is pinned against the
 * real reference (oracle/_ref, built by oracle/Makefile.ref) through the golden fixtures
 * in tests/golden/ (whole-run restart dumps + function-level vectors).
 */
#ifndef ATHENA_ORACLE_H
#define ATHENA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcParams {
  int    Nx[3];        /* active zones of THIS grid                                   */
  int    rootNx[3];    /* active zones of the root Domain (dx = (xmax-xmin)/rootNx)   */
  double xmin[3], xmax[3]; /* root Domain extent  (init_mesh.c:225)                   */
  double MinX[3];      /* lower edge of THIS grid (init_grid.c:104-111)               */
  int    bc[6];        /* ix1,ox1,ix2,ox2,ix3,ox3: 1 reflect 2 outflow 4 periodic, 0 = none
                          (filled by a neighbour exchange)  (bvals_mhd.c:560-586)     */
  int    nscal;        /* NSCALARS: 0 or 1                                            */
  int    ion;          /* ION_RADIATION on/off                                        */
  double gamma, cour_no, tlim;
  /* <ionradiation> block (ionrad_3d.c:742-757) */
  double sigma_ph, m_H, mu, e_gamma, alpha_C, k_B, time_unit;
  double max_de_iter, max_de_therm_iter, max_dx_iter;
  double max_de_step, max_de_therm_step, max_dx_step;
  double tfloor, tceil;
  int    maxiter;
  int    pot;          /* 0: StaticGravPot == NULL, 1: ioniz_sphere PlanetPot         */
  double pot_GM, pot_Rsoft;
  int    userwork;     /* 0: none, 1: ioniz_sphere Userwork_in_loop                   */
  double uw_K, uw_Cp, uw_rho0, uw_rreset2;
  int    integrator;   /* 0: CTU + H-correction (README.rst:25); 1: VL, no H-correction (the
                          only VL combination the reference compiles); 2: CTU without H-correction
                          (the reference's configure default)                           */
  int    order;        /* 2 (0 = default): PLM, configure --with-order=2; 3: PPM, --with-order=3
                          (reconstruction/lr_states_ppm.c), CTU integrator only          */
} OrcParams;

typedef struct OrcSim OrcSim;

OrcSim *orc_create(const OrcParams *p);
void    orc_destroy(OrcSim *s);
double *orc_U(OrcSim *s);        /* [N3][N2][N1][6]: d,M1,M2,M3,E,s0 (s0 unused if nscal=0) */
double *orc_edgeflux(OrcSim *s); /* [Nx3+1][Nx2+1][Nx1+1]                                    */
void    orc_dims(const OrcSim *s, int N[3]);
double  orc_get_time(const OrcSim *s);
double  orc_get_dt(const OrcSim *s);
int     orc_get_nstep(const OrcSim *s);
void    orc_set_time(OrcSim *s, double t);
void    orc_set_dt(OrcSim *s, double dt);
void    orc_set_nstep(OrcSim *s, int n);
void    orc_set_cooling(OrcSim *s, int kind);   /* CoolingFunc: 0 = NULL, 1 = KoyInut (microphysics/cool.c:48); CTU integrator only */

/* problem generators (prob/ifront.c, prob/ioniz_sphere.c, prob/blast.c, prob/shkset1d.c) */
void orc_problem_ifront(OrcSim *s, double n_H, double cs, double flux);
void orc_problem_ioniz_sphere(OrcSim *s, double n_H, double cs, double flux,
                              double rp, double mp, double np);
void orc_problem_blast(OrcSim *s, double radius, double pamb, double damb,
                       double drat, double prat);
void orc_problem_shkset1d(OrcSim *s, const double *wl, const double *wr, int shk_dir); /* {d,P,v1,v2,v3} x 2 */
void orc_add_radplane(OrcSim *s, int dir, double flux);

/* main.c steps */
void   orc_start(OrcSim *s);            /* bvals + bvals_ionrad + first new_dt (main.c:412-451) */
void   orc_bvals(OrcSim *s);            /* bvals_mhd.c:174                                      */
void   orc_bvals_side(OrcSim *s, int d, int side);   /* one side of it (bvals_mhd.c:196-420)          */
void   orc_bvals_ionrad(OrcSim *s);     /* bvals_ionrad.c:63                                    */
double orc_new_dt_local(OrcSim *s);     /* new_dt.c:72-170: returns CourNo/max_dti of this grid */
void   orc_new_dt(OrcSim *s);           /* new_dt.c:32                                          */
void   orc_integrate(OrcSim *s);        /* integrate_3d_ctu.c:110 or integrate_3d_vl.c:96       */
int    orc_ion_radtransfer(OrcSim *s);  /* ionrad_3d.c:862; returns niter                       */
void   orc_userwork(OrcSim *s);         /* ioniz_sphere.c:255                                   */
int    orc_step(OrcSim *s);             /* one pass of the main loop (main.c:519-669); niter    */

/* sub-phases of the ion step, for slab-decomposed drivers */
void   orc_ion_begin(OrcSim *s);                                   /* ionrad_3d.c:896-905 */
void   orc_ion_rates(OrcSim *s, double *dt_chem, double *dt_therm);/* :922-938            */
void   orc_ion_update(OrcSim *s, double dt);                       /* :965-971            */
long   orc_ion_check_range_count(OrcSim *s);                       /* :206-264            */
double orc_ion_dt_hydro(OrcSim *s);                                /* :593-669            */

/* Static mesh refinement (smr.c, ionrad_smr.c, the SMR branches of main.c / new_dt.c / ionrad_3d.c):
 * nlevels nested levels, one Domain each.  p[l] describes level l as a Grid of its own (Nx, MinX,
 * bc = 0 on fine/coarse sides; rootNx/xmin/xmax are those of the ROOT); disp[3*l+d] = <domainN>
 * iDisp/jDisp/kDisp in zones of level l.  Run the problem generator on every orc_mesh_level()
 * before orc_mesh_start(). */
typedef struct OrcMesh OrcMesh;
OrcMesh *orc_mesh_create(int nlevels, const OrcParams *p, const int *disp);
/* several Domains on a level (MeshS.Domain[nl][nd]): grids level by level from the root, deck order inside a level;
 * level[g] = DomainS.Level, disp[3g..] = iDisp/jDisp/kDisp.  Domains of a level neither overlap nor touch (init_mesh.c:398-418). */
OrcMesh *orc_mesh_create_tree(int ngrids, const OrcParams *p, const int *level, const int *disp);
void     orc_mesh_destroy(OrcMesh *m);
OrcSim  *orc_mesh_level(OrcMesh *m, int l);
void     orc_mesh_start(OrcMesh *m);               /* main.c:395-447 */
void     orc_mesh_step(OrcMesh *m, int *niter);    /* main.c:519-669; niter[nlevels] */
/* pieces for drivers that cut every level into x3 slabs (tests of the multi-GPU SMR driver): a local
 * stack of slabs with explicit links (21 ints per link: cs[3] local parent index incl. ghosts, n[3],
 * prol[6], corr[6], cdisp[3]), the inter-level operations one by one, and the flux correction of a
 * parent plane that lies on another slab */
OrcMesh *orc_mesh_create_local(int nlevels, const OrcParams *p, const int *links);
void     orc_mesh_restrict_correct(OrcMesh *m);
void     orc_mesh_restrict_correct_pair(OrcMesh *m, int l);  /* level l+1 -> level l only */
void     orc_mesh_ion_restrict_correct(OrcMesh *m);
void     orc_mesh_prolongate(OrcMesh *m);
void     orc_mesh_ionflux_prolong(OrcMesh *m, int l);       /* ionrad_prolong_snd(l-1) + _rcv(l) */
void     orc_cfl_max_v(OrcSim *s, double v[3]);             /* new_dt.c:72-140, this Grid only  */
void     orc_flux_x3_export(OrcSim *child, int side, double *buf);
void     orc_flux_x3_apply(OrcSim *parent, int side, int i0, int j0, int n1, int n2, const double *buf);
double   orc_mesh_time(const OrcMesh *m);
double   orc_mesh_dt(const OrcMesh *m);
int      orc_mesh_nstep(const OrcMesh *m);

/* function-level kernels, array-in/array-out, NV = 5+nscal doubles per state */
void orc_cons_to_prim(int n, int nscal, double gamma, const double *U, double *W);
void orc_cfast(int n, int nscal, double gamma, const double *U, double *c);
void orc_fluxes(int n, int nscal, double gamma, const double *Ul, const double *Ur,
                const double *eta, double *F);
void orc_lr_states(int n, int nscal, double gamma, const double *W, double dt, double dx,
                   int il, int iu, double *Wl, double *Wr);
void orc_lr_states_ppm(int n, int nscal, double gamma, const double *W, double dt, double dx,
                       int il, int iu, double *Wl, double *Wr);

#ifdef __cplusplus
}
#endif
#endif
