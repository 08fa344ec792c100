/* include/athena_compat.h -- the reference's host-side data structures, restated for the one
 * configuration this package accelerates:
 *   HYDRO, ADIABATIC, CARTESIAN, NSCALARS = AA_NSCALARS (default 1), ION_RADIATION + ION_RADPLANE,
 *   NO_MPI_PARALLEL (or MPI_PARALLEL with -DAA_MPI), NO_MESH_REFINEMENT (or STATIC_MESH_REFINEMENT with -DAA_SMR), also both
 *   together (README.rst:25: --enable-mpi ... --enable-smr),
 *   no particles / self-gravity / shearing box.
 * A driver or problem file compiled against the reference's athena.h with those macros and one
 * compiled against this header agree on every offset, so the reference's own main.o, init_mesh.o,
 * problem.o ... can be linked against host/athena_shim.c unchanged.  Field order
 * follows /root/reference/src/athena.h at the cited lines; the reference states that the order of
 * ConsS / Cons1DS / Prim1DS "CANNOT be changed" (athena.h:79,105,146,171).
 */
#ifndef ATHENA_COMPAT_H
#define ATHENA_COMPAT_H

#ifndef AA_NSCALARS
#define AA_NSCALARS 1
#endif
#ifndef AA_ION_RADPLANE            /* --enable-ion-radiation --enable-ion-plane (configure.ac:219-246) */
#define AA_ION_RADPLANE (AA_NSCALARS > 0)
#endif

#ifdef AA_MPI                              /* the reference configured with --enable-mpi (MPI_PARALLEL) */
#include <mpi.h>
#endif
typedef double Real;                       /* athena.h:33-34 (DOUBLE_PREC) */
struct Mesh_s;

typedef struct GridsData_s {               /* athena.h:66-72 */
  int Nx[3], Disp[3];
  int ID_Comm_world, ID_Comm_Domain;
#ifdef AA_SMR
  int ID_Comm_Children, ID_Comm_Parent;    /* athena.h:70-73 */
#endif
} GridsDataS;

typedef struct Cons_s {                    /* athena.h:81-100 */
  Real d, M1, M2, M3, E;
#if AA_NSCALARS > 0
  Real s[AA_NSCALARS];
#endif
} ConsS;

typedef struct Radplane_s {                /* athena.h:134-142 */
  int *dir;
  int nradplane;
  Real flux_i;
} Radplane;

#ifdef AA_SMR
typedef struct GridOvrlp_s {               /* athena.h:257-276 */
  int ijks[3], ijke[3];
  int ID, DomN;
  int nWordsRC, nWordsP;
  ConsS **myFlx[6];
#if AA_ION_RADPLANE
  Real *ionFlx[6];
#ifdef AA_MPI
  Real ion_mpitag;                         /* athena.h:266-268 */
#endif
#endif
} GridOvrlpS;
#endif

typedef struct Grid_s {                    /* athena.h:289-321 */
  ConsS ***U;
  Real MinX[3], MaxX[3];
  Real dx1, dx2, dx3;
  Real time, dt;
  int is, ie, js, je, ks, ke;
  int Nx[3], Disp[3];
  int rx1_id, lx1_id, rx2_id, lx2_id, rx3_id, lx3_id;
#if AA_ION_RADPLANE
  Real ***EdgeFlux;                        /* athena.h:316-319 */
  struct Mesh_s *Mesh;
#endif
#ifdef AA_SMR
  int NCGrid, NPGrid, NmyCGrid, NmyPGrid;  /* athena.h:332-342 */
  GridOvrlpS *CGrid, *PGrid;
#endif
} GridS;

typedef void (*VGFun_t)(GridS *pG);        /* athena.h:325 */

typedef struct Domain_s {                  /* athena.h:340-386 */
  Real RootMinX[3], RootMaxX[3], MinX[3], MaxX[3], dx[3];
  int Nx[3], NGrid[3], Disp[3];
  int Level, DomNumber, InputBlock;
  GridS *Grid;
  GridsDataS ***GData;
  VGFun_t ix1_BCFun, ox1_BCFun, ix2_BCFun, ox2_BCFun, ix3_BCFun, ox3_BCFun;
#if AA_ION_RADPLANE
  struct Mesh_s *Mesh;                     /* athena.h:383-385 */
#endif
#ifdef AA_MPI
  MPI_Comm Comm_Domain;                    /* athena.h:387-389 */
  MPI_Group Group_Domain;
#ifdef AA_SMR
  MPI_Comm Comm_Parent, Comm_Children;     /* athena.h:390-394 */
  MPI_Group Group_Children;
#endif
#endif
} DomainS;

typedef void (*VDFun_t)(DomainS *pD);      /* athena.h:390 */

typedef struct Mesh_s {                    /* athena.h:397-425 */
  Real RootMinX[3], RootMaxX[3], dx[3];
  Real time, dt;
  int Nx[3];
  int nstep;
  int BCFlag_ix1, BCFlag_ox1, BCFlag_ix2, BCFlag_ox2, BCFlag_ix3, BCFlag_ox3;
  int NLevels;
  int *DomainsPerLevel;
  DomainS **Domain;
  char *outfilename;
#if AA_ION_RADPLANE
  Radplane *radplanelist;                  /* athena.h:421-423 */
#endif
} MeshS;

typedef Real (*GravPotFun_t)(const Real x1, const Real x2, const Real x3);   /* athena.h:540 */
typedef Real (*CoolingFun_t)(const Real d, const Real p, const Real dt);     /* athena.h:545 */

/* ---- the 15 entry points the reference's driver and problem files link against
 *      (SURVEY.md 8b); implemented by host/athena_shim.c on top of include/athena_amd.h ---- */
void    lr_states_init(MeshS *pM);                 /* reconstruction/prototypes.h:43 */
void    lr_states_destruct(void);                  /* :42 */
VDFun_t integrate_init(MeshS *pM);                 /* integrators/prototypes.h:16 */
void    integrate_destruct(void);                  /* :17 */
void    ion_radtransfer_init_domain(MeshS *pM);    /* ionradiation/prototypes.h:29 */
VDFun_t ion_radtransfer_init(MeshS *pM, int ires); /* :30 */
void    bvals_ionrad_init(MeshS *pM);              /* :66 */
void    bvals_ionrad(DomainS *pD);                 /* :67 */
void    set_coarse_time(void);                     /* :33 */
void    clear_coarse_time(void);                   /* :34 */
void    add_radplane_3d(GridS *pGrid, int dir, Real flux);   /* :58 */
void    bvals_mhd_init(MeshS *pM);                 /* prototypes.h:75 */
void    bvals_mhd(DomainS *pD);                    /* :76 */
enum BCDirection {left_x1, right_x1, left_x2, right_x2, left_x3, right_x3};   /* athena.h:543 */
void    bvals_mhd_fun(DomainS *pD, enum BCDirection dir, VGFun_t prob_bc);    /* :77 */
void    new_dt(MeshS *pM);                         /* :147 */
#ifdef AA_SMR                                      /* prototypes.h:163-168 (smr.c) */
void    SMR_init(MeshS *pM);
void    RestrictCorrect(MeshS *pM);
void    Prolongate(MeshS *pM);
void    ionradRestrictCorrect(MeshS *pM);
#endif

#endif
