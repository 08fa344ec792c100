"""Run configuration: what init_mesh.c / init_grid.c / main.c derive from an athinput deck
for the single-level hot path, plus the x3-slab decomposition used across GPUs.

Reference rules reproduced here (file:line under /root/reference/src):
  * root ``dx_i = (x_imax - x_imin)/Nx_i``                                 init_mesh.c:225
  * BC flags: 1 reflect, 2 outflow, 4 periodic                              bvals_mhd.c:560-586
  * block decomposition: ``Nx/NGrid`` cells per Grid, the remainder goes to the first
    Grids of that direction                                                 init_mesh.c:583-620
  * a Grid's lower edge is accumulated ``MinX += (Real)Nx_prev*dx``         init_grid.c:104-111
  * periodic neighbours wrap (rank +- 1 mod N)                              bvals_mhd.c:570-577
  * ``--enable-ion-radiation`` adds one passive scalar (NSCALARS=1)         configure.ac:241-243
The rays travel along +x1 (dir=-1), so Grids are never cut along x1: only
x3 slabs are produced.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional, Tuple

from .athinput import ParTable, ParError

NGHOST = 4  # defs.h.in:129-140

ION_KEYS = ("sigma_ph", "m_H", "mu", "e_gamma", "alpha_C", "k_B", "time_unit",
            "max_de_iter", "max_de_therm_iter", "max_dx_iter",
            "max_de_step", "max_de_therm_step", "max_dx_step", "tfloor", "tceil")

PROBLEMS = {
    # name: (ion radiation on?, NSCALARS)
    "ifront": (True, 1),
    "ioniz_sphere": (True, 1),
    "blast": (False, 0),
}


@dataclass
class RunConfig:
    problem: str
    rootNx: Tuple[int, int, int]
    xmin: Tuple[float, float, float]
    xmax: Tuple[float, float, float]
    bc: Tuple[int, int, int, int, int, int]          # ix1 ox1 ix2 ox2 ix3 ox3 (root Domain)
    nscal: int
    ion: bool
    gamma: float
    cour_no: float
    tlim: float
    nlim: int
    ionp: Dict[str, float] = field(default_factory=dict)   # <ionradiation> block
    maxiter: int = 0
    prob: Dict[str, float] = field(default_factory=dict)   # <problem> block, numeric keys
    # compile-time choice of the reference (configure --with-integrator): "ctu" (+ H-correction, the
    # README.rst:25 configuration) or "vl" (no H-correction: integrate_2d_vl.c:533 does not compile
    # with it, so that is the only VL build the reference has)
    integrator: str = "ctu"

    @property
    def dx(self) -> Tuple[float, float, float]:
        return tuple((self.xmax[d] - self.xmin[d]) / float(self.rootNx[d]) for d in range(3))


@dataclass
class GridConfig:
    """One Grid (= one GPU's slab) of the root Domain."""
    run: RunConfig
    rank: int
    nranks: int
    Nx: Tuple[int, int, int]
    disp: Tuple[int, int, int]                        # cell displacement inside the Domain
    MinX: Tuple[float, float, float]
    bc: Tuple[int, int, int, int, int, int]           # 0 where a neighbour Grid fills the ghosts
    lx3: int                                          # neighbour rank below (-1: physical BC)
    rx3: int                                          # neighbour rank above


def from_par(par: ParTable, problem: Optional[str] = None) -> RunConfig:
    problem = problem or par.gets("job", "problem_id").lower()
    if problem not in PROBLEMS:
        raise ParError(f"[config]: unsupported problem \"{problem}\" (have {sorted(PROBLEMS)})")
    ion, nscal = PROBLEMS[problem]
    blk = "domain1"
    Nx = tuple(par.geti(blk, f"Nx{d}") for d in (1, 2, 3))
    if min(Nx) <= 1:
        raise ParError("[config]: the MI355X path is 3-D only (Nx1,Nx2,Nx3 > 1)")
    xmin = tuple(par.getd(blk, f"x{d}min") for d in (1, 2, 3))
    xmax = tuple(par.getd(blk, f"x{d}max") for d in (1, 2, 3))
    bc = tuple(par.geti_def(blk, k, 0) for k in
               ("bc_ix1", "bc_ox1", "bc_ix2", "bc_ox2", "bc_ix3", "bc_ox3"))
    for b in bc:
        if b not in (1, 2, 4):
            raise ParError(f"[bvals_init]: bc flag = {b} unknown")       # bvals_mhd.c:586
    cour_no = par.getd("time", "cour_no")
    if cour_no > 0.5:                                                    # integrate.c:66-68
        raise ParError("<time>cour_no was set to %g: must be <= 0.5 with 3D integrator" % cour_no)
    cfg = RunConfig(problem=problem, rootNx=Nx, xmin=xmin, xmax=xmax, bc=bc, nscal=nscal, ion=ion,
                    gamma=par.getd("problem", "gamma"), cour_no=cour_no,
                    tlim=par.getd("time", "tlim"), nlim=par.geti_def("time", "nlim", -1))
    if ion:
        cfg.ionp = {k: par.getd("ionradiation", k) for k in ION_KEYS}
        cfg.maxiter = int(par.getd("ionradiation", "maxiter"))           # ionrad_3d.c:757
        if par.geti("problem", "nradplanes") != 1:
            raise ParError("Invalid number of radplanes specified in input file")
    for k, v in par.blocks.get("problem", {}).items():
        try:
            cfg.prob[k] = float(v)
        except ValueError:
            pass
    return cfg


def load(path: str, overrides=None, problem: Optional[str] = None, integrator: str = "ctu") -> RunConfig:
    run = from_par(ParTable.from_file(path).cmdline(overrides), problem)
    if integrator not in ("ctu", "vl"):
        raise ParError(f"[integrate_init]: unknown integrator {integrator}")
    run.integrator = integrator
    return run


def split_cells(n: int, parts: int) -> List[int]:
    """init_mesh.c:583-620: n/parts each, remainder to the first Grids."""
    base, rem = divmod(n, parts)
    return [base + (1 if r < rem else 0) for r in range(parts)]


def slab(run: RunConfig, rank: int = 0, nranks: int = 1) -> GridConfig:
    if nranks < 1 or not (0 <= rank < nranks):
        raise ParError(f"[config]: bad rank {rank} of {nranks}")
    nx3 = split_cells(run.rootNx[2], nranks)
    if min(nx3) < NGHOST:
        raise ParError(f"[config]: x3 slabs thinner than nghost={NGHOST} ({run.rootNx[2]}/{nranks})")
    dx = run.dx
    disp3 = sum(nx3[:rank])
    minx3 = run.xmin[2]
    for r in range(rank):                     # accumulate exactly as init_grid.c:109-110
        minx3 += float(nx3[r]) * dx[2]
    periodic3 = (run.bc[4] == 4 and run.bc[5] == 4)
    lx3 = rank - 1 if rank > 0 else (nranks - 1 if (periodic3 and nranks > 1) else -1)
    rx3 = rank + 1 if rank < nranks - 1 else (0 if (periodic3 and nranks > 1) else -1)
    bc = list(run.bc)
    if lx3 >= 0:
        bc[4] = 0
    if rx3 >= 0:
        bc[5] = 0
    return GridConfig(run=run, rank=rank, nranks=nranks,
                      Nx=(run.rootNx[0], run.rootNx[1], nx3[rank]),
                      disp=(0, 0, disp3), MinX=(run.xmin[0], run.xmin[1], minx3),
                      bc=tuple(bc), lx3=lx3, rx3=rx3)
