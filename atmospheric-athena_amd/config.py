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

import os
from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional, Sequence, Tuple

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
    "shkset1d": (False, 0),      # tst/1D-hydro/athinput.sod run as a slab of a 3-D box
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
    # configure --with-order: 2 = piecewise linear (lr_states_plm.c, the default), 3 = piecewise parabolic
    # (lr_states_ppm.c); characteristic variables in both cases
    order: int = 2

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
    level: int = 0                                    # DomainS.Level (static mesh refinement)
    lx2: int = -1                                     # x2 x x3 pencils: neighbour ranks in x2 (-1: physical BC / not cut)
    rx2: int = -1
    p2: int = 1                                       # Grids along x2 (NGrid_x2); nranks = p2 * (Grids along x3)


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
    if integrator not in ("ctu", "vl", "ctu-noh"):      # ctu-noh: CTU without --enable-h-correction (the reference's configure default)
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


def pencil(run: RunConfig, rank: int, p2: int, p3: int) -> GridConfig:
    """One Grid of an NGrid_x2 x NGrid_x3 = p2 x p3 decomposition of the root Domain (init_mesh.c:526-620; x1 is never cut:
    the rays travel along it and it is the contiguous axis).  Ranks are dealt x2-fastest, then x3 (init_mesh.c:589-596 with
    NGrid_x1 = 1); cells per Grid and the remainder rule as :583-620, MinX accumulated as init_grid.c:104-111."""
    if p2 < 1 or p3 < 1 or not (0 <= rank < p2 * p3):
        raise ParError(f"[config]: bad rank {rank} of {p2}x{p3}")
    if p2 == 1:
        return slab(run, rank, p3)
    r2, r3 = rank % p2, rank // p2
    nx2, nx3 = split_cells(run.rootNx[1], p2), split_cells(run.rootNx[2], p3)
    if min(nx2) < NGHOST or min(nx3) < NGHOST:
        raise ParError(f"[config]: pencils thinner than nghost={NGHOST} ({run.rootNx[1]}/{p2} x {run.rootNx[2]}/{p3})")
    dx = run.dx
    minx2, minx3 = run.xmin[1], run.xmin[2]
    for r in range(r2):
        minx2 += float(nx2[r]) * dx[1]
    for r in range(r3):
        minx3 += float(nx3[r]) * dx[2]
    per2 = (run.bc[2] == 4 and run.bc[3] == 4)
    per3 = (run.bc[4] == 4 and run.bc[5] == 4)

    def nb(r, p, per):
        lo = r - 1 if r > 0 else (p - 1 if (per and p > 1) else -1)
        hi = r + 1 if r < p - 1 else (0 if (per and p > 1) else -1)
        return lo, hi

    l2, h2 = nb(r2, p2, per2)
    l3, h3 = nb(r3, p3, per3)
    bc = list(run.bc)
    for side, n in ((2, l2), (3, h2), (4, l3), (5, h3)):
        if n >= 0:
            bc[side] = 0
    return GridConfig(run=run, rank=rank, nranks=p2 * p3,
                      Nx=(run.rootNx[0], nx2[r2], nx3[r3]),
                      disp=(0, sum(nx2[:r2]), sum(nx3[:r3])), MinX=(run.xmin[0], minx2, minx3), bc=tuple(bc),
                      lx3=(l3 * p2 + r2) if l3 >= 0 else -1, rx3=(h3 * p2 + r2) if h3 >= 0 else -1,
                      lx2=(r3 * p2 + l2) if l2 >= 0 else -1, rx2=(r3 * p2 + h2) if h2 >= 0 else -1, p2=p2)


def levels(par: ParTable, run: RunConfig) -> List[GridConfig]:
    """The Domains of a static-mesh-refinement deck in the order of the reference's loops -- level by level from the root,
    Domains of a level in deck order (MeshS.Domain[nl][nd], init_mesh.c:170-295).  Level l has dx = root dx / 2^l; ``disp``
    is <domainN> iDisp/jDisp/kDisp in zones of that level; its lower edge is ``root_xmin + Disp*dx_l`` (:281-286); sides that
    are not on the root boundary get bc = 0 (bvals_mhd.c:193-361, ProlongateLater).  Domains of one level neither overlap nor
    touch (:398-418), so every Domain lies in ONE Domain of the level below."""
    nd = par.geti_def("job", "num_domains", 1)
    doms = sorted((par.geti(f"domain{n}", "level"), n) for n in range(1, nd + 1))
    levs = sorted({lev for lev, _ in doms})
    if levs != list(range(len(levs))) or sum(1 for lev, _ in doms if lev == 0) != 1:
        raise ParError("[init_mesh]: levels must be contiguous from 0, with one root Domain")
    out = [slab(run, 0, 1)]
    ext = [((0, 0, 0), tuple(run.rootNx))]                # (origin, size) of every Domain in zones of its level
    for lev, n in doms[1:]:
        blk = f"domain{n}"
        irefine = 2 ** lev
        Nx = tuple(par.geti(blk, f"Nx{d}") for d in (1, 2, 3))
        disp = tuple(par.geti(blk, k) for k in ("iDisp", "jDisp", "kDisp"))
        for d in range(3):
            if Nx[d] % irefine:
                raise ParError(f"[init_mesh]: {blk}/Nx{d + 1} = {Nx[d]} must be divisible by {irefine}")
            if disp[d] % irefine:
                raise ParError(f"[init_mesh]: {blk}/Disp{d + 1} = {disp[d]} must be divisible by {irefine}")
        # init_mesh.c:398-418: Domains on the same level may neither overlap nor touch
        for g, (o, sz) in zip(out, ext):
            if g.level == lev and all(disp[d] <= o[d] + sz[d] and o[d] <= disp[d] + Nx[d] for d in range(3)):
                raise ParError(f"[init_mesh]: Domains at level {lev} overlap or touch ({blk})")
        # init_mesh.c:320-360, :448-470: inside ONE Domain of the level below; it may touch its edge only where that is the root boundary
        parent = None
        for g, (o, sz) in zip(out, ext):
            if g.level == lev - 1 and all(disp[d] // 2 >= o[d] and (disp[d] + Nx[d]) // 2 <= o[d] + sz[d] for d in range(3)):
                parent = (o, sz)
        if parent is None:
            raise ParError(f"[init_mesh]: {blk} is not inside the Domain of level {lev - 1}")
        pdisp, pNx = parent
        for d in range(3):
            lo, hi = disp[d] // 2, (disp[d] + Nx[d]) // 2
            if (lo == pdisp[d] and disp[d] != 0) or \
               (hi == pdisp[d] + pNx[d] and (disp[d] + Nx[d]) // irefine != run.rootNx[d]):
                raise ParError(f"[init_mesh]: child Domain {blk} touches its parent in x{d + 1}")
            # init_mesh.c:484-499: ... or lies closer than nghost/2 parent zones to its edge (the prolongation stencil)
            if 0 < 2 * (lo - pdisp[d]) < NGHOST or 0 < 2 * (pdisp[d] + pNx[d] - hi) < NGHOST:
                raise ParError(f"[init_mesh]: child Domain {blk} closer than nghost/2 to its parent in x{d + 1}")
        dxl = tuple(run.dx[d] / float(irefine) for d in range(3))
        MinX = tuple(run.xmin[d] if disp[d] == 0 else run.xmin[d] + float(disp[d]) * dxl[d] for d in range(3))
        bc = list(run.bc)
        for d in range(3):
            if disp[d] != 0:
                bc[2 * d] = 0
            if (disp[d] + Nx[d]) // irefine != run.rootNx[d]:
                bc[2 * d + 1] = 0
        out.append(GridConfig(run=run, rank=0, nranks=1, Nx=Nx, disp=disp, MinX=MinX, bc=tuple(bc),
                              lx3=-1, rx3=-1, level=lev))
        ext.append((disp, Nx))
    # ionrad_smr.c:97-98: the coarse->fine radiation hand-off is only defined while the parent is not
    # displaced across the rays (2 levels); AA_SMR_DEEP_RADIATION=fixed opts into the corrected index
    if run.ion and os.environ.get("AA_SMR_DEEP_RADIATION") != "fixed":
        for g, (o, sz) in zip(out, ext):
            has_child = any(c.level == g.level + 1 and all(c.disp[d] // 2 >= o[d] and (c.disp[d] + c.Nx[d]) // 2 <= o[d] + sz[d]
                                                           for d in range(3)) for c in out)
            if g.level and has_child and (g.disp[1] or g.disp[2]):
                raise ParError(f"[config]: radiation across a displaced parent (level {g.level}) is undefined in the "
                               "reference; set AA_SMR_DEEP_RADIATION=fixed for the corrected hand-off")
    return out


# Every level is cut at the SAME planes as the root (x3 slabs, SURVEY.md 8e "co-partitioned"): rank r
# owns root planes [K_r, K_r+1) and, of level l, the zones lying over them.  Restriction, the
# radiation hand-off and prolongation then stay inside a rank (prolongation reads the parent's
# ghost planes where a level starts exactly at a cut); what crosses ranks is the x3 halo of every
# level and the flux correction of the one parent plane that lies across a cut from the child.

@dataclass
class LinkConfig:
    """Level l+1 (this rank's slab of it) seen from this rank's slab of level l."""
    cs: Tuple[int, int, int]          # first parent zone under the child slab, local parent index incl. ghosts
    n: Tuple[int, int, int]           # overlap, parent zones
    prol: Tuple[int, ...]             # [6] the child's ghost zones on this side are prolonged from the parent
    corr: Tuple[int, ...]             # [6] the parent zone outside this side is on this rank: flux-correct it here
    corr_to: Tuple[int, int]          # x3 sides: rank owning the parent plane outside (-1 if local or no boundary)
    cdisp: Tuple[int, int, int]       # child slab origin minus 2 x parent slab origin (zones of the child level)


@dataclass
class MeshSlabConfig:
    rank: int
    nranks: int
    cuts: Tuple[int, ...]             # root planes K_0=0 < ... < K_N
    table: List[List[Optional[Tuple[int, int]]]]   # table[r][l] = (k0, k1) of level l on rank r (level units) or None
    levels: List[GridConfig]          # the levels present on this rank, root first
    links: List[LinkConfig]           # links[l]: levels[l+1] on levels[l]
    # flux corrections arriving from a neighbour's child slab: (level of the parent, side of the child
    # boundary 0 lower / 1 upper, source rank, i0, j0, n1, n2) with i0, j0 local parent indices incl. ghosts
    corr_in: List[Tuple[int, int, int, int, int, int, int]]


def balanced_cuts(levs: List[GridConfig], nranks: int) -> Tuple[int, ...]:
    """Root planes K_r such that every rank gets about the same number of zones summed over levels
    (a root plane under level l carries 2^l planes of that level)."""
    root = levs[0]
    n3 = root.Nx[2]
    w = [float(root.Nx[0] * root.Nx[1])] * n3
    for g in levs[1:]:
        f = 2 ** g.level
        for k in range(g.disp[2] // f, (g.disp[2] + g.Nx[2]) // f):
            w[k] += float(g.Nx[0] * g.Nx[1]) * f
    tot = sum(w)
    cuts, acc, k = [0], 0.0, 0
    for r in range(1, nranks):
        while k < n3 and acc + 0.5 * w[k] < tot * r / nranks:
            acc += w[k]; k += 1
        k = max(k, cuts[-1] + NGHOST)
        cuts.append(k)
    cuts.append(n3)
    return tuple(cuts)


def mesh_slabs(par: ParTable, run: RunConfig, rank: int, nranks: int,
               cuts: Optional[Sequence[int]] = None) -> MeshSlabConfig:
    levs = levels(par, run)
    if len({g.level for g in levs}) != len(levs):
        raise ParError("[config]: several Domains on a level are not cut across GPUs (one process: Mesh / the drop-in shim)")
    n3 = run.rootNx[2]
    cuts = tuple(cuts) if cuts is not None else balanced_cuts(levs, nranks)
    if len(cuts) != nranks + 1 or cuts[0] != 0 or cuts[-1] != n3 or any(b - a < NGHOST for a, b in zip(cuts, cuts[1:])):
        raise ParError(f"[config]: bad x3 cuts {cuts} for {nranks} ranks (slabs need >= {NGHOST} root planes)")
    table: List[List[Optional[Tuple[int, int]]]] = []
    for r in range(nranks):
        row: List[Optional[Tuple[int, int]]] = []
        for g in levs:
            f = 2 ** g.level
            d3 = g.disp[2] if g.level else 0
            k0, k1 = max(d3, f * cuts[r]), min(d3 + g.Nx[2], f * cuts[r + 1])
            if k1 <= k0 or (row and row[-1] is None):
                row.append(None)
            else:
                if k1 - k0 < NGHOST:
                    raise ParError(f"[config]: level {g.level} slab of rank {r} is thinner than nghost; choose other cuts")
                row.append((k0, k1))
        table.append(row)

    def has(r, l):
        return 0 <= r < nranks and table[r][l] is not None

    periodic3 = (run.bc[4] == 4 and run.bc[5] == 4)
    mine: List[GridConfig] = []
    for g in levs:
        span = table[rank][g.level]
        if span is None:
            break
        k0, k1 = span
        f = 2 ** g.level
        d3 = g.disp[2] if g.level else 0
        dx3 = run.dx[2] / float(f)
        bc = list(g.bc)
        lo_nb = hi_nb = -1
        if k0 > d3:                                  # the cut is interior to this level: neighbour slab below
            lo_nb = rank - 1; bc[4] = 0
        elif g.level == 0 and periodic3 and nranks > 1:
            lo_nb = nranks - 1; bc[4] = 0
        if k1 < d3 + g.Nx[2]:
            hi_nb = rank + 1; bc[5] = 0
        elif g.level == 0 and periodic3 and nranks > 1:
            hi_nb = 0; bc[5] = 0
        minx3 = run.xmin[2] if k0 == 0 else run.xmin[2] + float(k0) * dx3        # init_mesh.c:281-286 form
        mine.append(GridConfig(run=run, rank=rank, nranks=nranks, Nx=(g.Nx[0], g.Nx[1], k1 - k0),
                               disp=(g.disp[0] if g.level else 0, g.disp[1] if g.level else 0, k0),
                               MinX=(g.MinX[0], g.MinX[1], minx3), bc=tuple(bc), lx3=lo_nb, rx3=hi_nb, level=g.level))
    links: List[LinkConfig] = []
    for l in range(len(mine) - 1):
        P, C, Cg = mine[l], mine[l + 1], levs[l + 1]
        irefine = 2 ** (l + 1)
        cs, n, prol, corr = [], [], [0] * 6, [0] * 6
        corr_to = [-1, -1]
        for d in range(3):
            a = C.disp[d] // 2 - P.disp[d]
            b = (C.disp[d] + C.Nx[d]) // 2 - P.disp[d]
            cs.append(a + NGHOST); n.append(b - a)
            glo, ghi = Cg.disp[d], Cg.disp[d] + Cg.Nx[d]           # the level's global extent
            at_lo = (C.disp[d] == glo) and glo != 0                # a fine/coarse boundary of the LEVEL
            at_hi = (C.disp[d] + C.Nx[d] == ghi) and (ghi // irefine != run.rootNx[d])
            prol[2 * d], prol[2 * d + 1] = int(at_lo), int(at_hi)
            if at_lo:
                if a > 0 or d < 2:
                    corr[2 * d] = 1
                else:
                    corr_to[0] = rank - 1
            if at_hi:
                if b < P.Nx[d] or d < 2:
                    corr[2 * d + 1] = 1
                else:
                    corr_to[1] = rank + 1
        links.append(LinkConfig(cs=tuple(cs), n=tuple(n), prol=tuple(prol), corr=tuple(corr), corr_to=tuple(corr_to),
                                cdisp=tuple(C.disp[d] - 2 * P.disp[d] for d in range(3))))
    corr_in = []
    for l in range(len(mine)):
        if l + 1 >= len(levs):
            break
        P, Cg = mine[l], levs[l + 1]
        irefine = 2 ** (l + 1)
        k0, k1 = table[rank][l]
        i0 = Cg.disp[0] // 2 - P.disp[0] + NGHOST; j0 = Cg.disp[1] // 2 - P.disp[1] + NGHOST
        n1, n2 = Cg.Nx[0] // 2, Cg.Nx[1] // 2
        # the child level starts exactly at my upper cut: my top plane lies outside its lower boundary
        if Cg.disp[2] == 2 * k1 and Cg.disp[2] != 0 and has(rank + 1, l + 1):
            corr_in.append((l, 0, rank + 1, i0, j0, n1, n2))
        ghi = Cg.disp[2] + Cg.Nx[2]
        if ghi == 2 * k0 and ghi // irefine != run.rootNx[2] and has(rank - 1, l + 1):
            corr_in.append((l, 1, rank - 1, i0, j0, n1, n2))
    return MeshSlabConfig(rank=rank, nranks=nranks, cuts=cuts, table=table, levels=mine, links=links, corr_in=corr_in)
