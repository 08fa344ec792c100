"""The CPU oracle against the golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  Pins the oracle: every comparison is bit-for-bit."""
import glob
import os

import numpy as np
import pytest

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("nscal", [0, 1])
def test_kernels_bitwise(nscal):
    g = np.load(os.path.join(GOLD, f"kernels_nscal{nscal}.npz"))
    gam = float(g["gamma"])
    assert _same(orc.cons_to_prim(g["Ul"], gam, nscal), g["W"])
    assert _same(orc.cfast(g["Ul"], gam, nscal), g["cfast"])
    assert _same(orc.fluxes(g["Ul"], g["Ur"], g["eta"], gam, nscal), g["F"])
    Wl, Wr = orc.lr_states(g["Wp"], float(g["dt"]), float(g["dx"]), int(g["il"]), int(g["iu"]), gam, nscal)
    assert _same(Wl, g["Wl"]) and _same(Wr, g["Wr"])


@pytest.mark.parametrize("nscal", [0, 1])
def test_ppm_reconstruction_bitwise(nscal):
    """lr_states_ppm.c (--with-order=3) on the 2048-cell pencils, incl. the scalar column whose work
    arrays overlap in the reference (see the oracle's comment)."""
    g = np.load(os.path.join(GOLD, f"kernels_ppm_nscal{nscal}.npz"))
    Wl, Wr = orc.lr_states(g["Wp"], float(g["dt"]), float(g["dx"]), int(g["il"]), int(g["iu"]), float(g["gamma"]), nscal, order=3)
    assert _same(Wl, g["Wl"]) and _same(Wr, g["Wr"])


def test_kernel_vectors_cover_all_roe_branches():
    """The Riemann vectors must exercise the HLLE fallback and the supersonic returns."""
    import ctypes as C
    L = orc.lib()
    h = C.c_long.in_dll(L, "orc_dbg_hlle"); s = C.c_long.in_dll(L, "orc_dbg_supersonic")
    g = np.load(os.path.join(GOLD, "kernels_nscal1.npz"))
    h.value = 0; s.value = 0
    orc.fluxes(g["Ul"], g["Ur"], g["eta"], float(g["gamma"]), 1)
    assert h.value > 100 and s.value > 100


def _rayplane(name, g):
    """Rays along +x1 / +x2 from our own problem file (tests/fixtures/rayplane_dir.c) run by the reference:
    get_ph_rate_plane cases -1 and -2 (ionradplane_3d.c:254-354), bvals_ionrad's lit face."""
    s = orc.make_rayplane_sim(g["nx"], -int(name.split("_dir")[1][0]))
    assert _same(s.active, g["U0"]), "initial condition"
    s.start()
    assert s.dt == float(g["dt0"])
    niter = [s.step() for _ in range(int(g["nstep"]))]
    assert niter == [int(x) for x in g["niter"]], "radiation sub-cycle counts"
    assert s.time == float(g["time"]) and s.dt == float(g["dt"])
    assert _same(s.active, g["U"])
    assert _same(s.edgeflux, g["edgeflux"])


def _coolpat(name, g):
    """Optically thin cooling (integrate_3d_ctu.c Steps 1c-3c, 8b, 11c; CoolingFunc = KoyInut, microphysics/cool.c:48) from our own
    problem file (tests/fixtures/cool_pattern.c) run by the reference, with and without the cooling function enrolled."""
    s = orc.make_coolpat_sim(g)
    assert _same(s.active[..., :5], g["U0"][..., :5]), "initial condition"
    s.start()
    assert s.dt == float(g["dt0"])
    for _ in range(int(g["nstep"])): s.step()
    assert s.time == float(g["time"]) and s.dt == float(g["dt"])
    assert _same(s.active[..., :5], g["U"][..., :5])


def test_cooling_fixtures_differ_from_the_run_without():
    a = np.load(os.path.join(GOLD, "coolpat_c1_16x12x10_n4.npz")); b = np.load(os.path.join(GOLD, "coolpat_c0_16x12x10_n4.npz"))
    assert _same(a["U0"], b["U0"]) and float(a["time"]) != float(b["time"])
    assert np.abs(a["U"][..., 4] / b["U"][..., 4] - 1).max() > 1e-2          # the cooling terms move the energy by per cent


RUNS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*_n[0-9]*.npz")))


@pytest.mark.parametrize("name", RUNS)
def test_whole_run_bitwise(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    if name.startswith("rayplane"):
        return _rayplane(name, g)
    if name.startswith("coolpat"):
        return _coolpat(name, g)
    prob = name.rsplit("_", 2)[0]
    integrator = "ctu"
    order = 2
    if prob.startswith("vl_"):
        prob, integrator = prob[3:], "vl"
    if prob.startswith("noh_"):
        prob, integrator = prob[4:], "ctu-noh"             # CTU without --enable-h-correction
    if prob.startswith("ppm_"):
        prob, order = prob[4:], 3
    if prob.startswith("shkset1d"):
        prob = "shkset1d"                       # shkset1d_d<dir>_...
    nx = g["nx"]
    s = orc.make_sim(prob, [f"domain1/Nx{d + 1}={int(nx[d])}" for d in range(3)] + [str(o) for o in g["overrides"]],
                     integrator=integrator, order=order)
    nv = 5 + s.grid.run.nscal
    assert _same(s.active[..., :nv], g["U0"][..., :nv]), "initial condition"
    s.start()
    assert s.dt == float(g["dt0"])
    niter = [s.step() for _ in range(int(g["nstep"]))]
    if s.grid.run.ion:
        assert niter == [int(x) for x in g["niter"]], "radiation sub-cycle counts"
    assert s.time == float(g["time"]) and s.dt == float(g["dt"])
    assert _same(s.active[..., :nv], g["U"][..., :nv])
    if "edgeflux" in g.files:
        assert _same(s.edgeflux, g["edgeflux"])


DEV = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "dev_*.npz")))


@pytest.mark.parametrize("name", DEV)
def test_developed_state_pairs_bitwise(name):
    """Start from a reference state deep into the run (what a restart carries) and reach the
    reference's later state exactly."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    prob = name[4:].rsplit("_", 3)[0]
    nx = g["nx"]
    over = [str(o) for o in g["overrides"]] if "overrides" in g.files else []      # e.g. the zoomed box of the one-sub-cycle pair
    s = orc.make_sim(prob, [f"domain1/Nx{d + 1}={int(nx[d])}" for d in range(3)] + over)
    nv = 5 + s.grid.run.nscal
    s.active[..., :nv] = g["UA"][..., :nv]
    s.time = float(g["timeA"]); s.dt = float(g["dtA"]); s.nstep = int(g["nstepA"])
    s.bvals(); s.bvals_ionrad()
    niter = [s.step() for _ in range(int(g["nstepB"]) - int(g["nstepA"]))]
    if s.grid.run.ion:
        assert niter == [int(x) for x in g["niter"]]
    assert s.time == float(g["timeB"]) and s.dt == float(g["dtB"])
    assert _same(s.active[..., :nv], g["UB"][..., :nv])
    if "edgefluxB" in g.files:
        assert _same(s.edgeflux, g["edgefluxB"])


def test_the_headline_regime_is_pinned():
    """The benchmark's timed region takes ONE radiation sub-cycle per step (ionrad_3d.c:919-1012 with the loop body
    executed once): one of the developed reference pairs must be in exactly that regime, NaN-free, over several steps."""
    g = np.load(os.path.join(GOLD, "dev_ioniz_sphere_36x36x36_s27_s33.npz"))
    assert [int(x) for x in g["niter"]] == [1] * 6
    assert np.isfinite(g["UA"]).all() and np.isfinite(g["UB"]).all()
    assert float(g["dtB"]) < 1.5 * float(g["dtA"])            # dt is no longer doubling: it sits at the CFL limit


SMR = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "smr_*.npz")))


@pytest.mark.parametrize("name", SMR)
def test_smr_runs_bitwise(name):
    """Static mesh refinement (smr.c RestrictCorrect / Prolongate, ionrad_smr.c, the SMR branches of
    main.c, new_dt.c and ionrad_3d.c): nested levels against runs of the reference built with
    STATIC_MESH_REFINEMENT, bit for bit on every level."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    prob = "ioniz_sphere" if "ioniz_sphere" in name else "blast"
    m = orc.make_mesh(prob, orc.deck_for(prob, g), [str(o) for o in g["overrides"]], integrator="vl" if name.startswith("smr_vl_") else "ctu",
                      order=3 if name.startswith("smr_ppm_") else 2)
    assert len(m.lev) == int(g["nlevels"])
    nv = 5 + m.lev[0].grid.run.nscal
    m.start()
    assert m.dt == float(g["dt0"])
    niter = []
    for _ in range(int(g["nstep"])):
        niter += m.step()
    if m.lev[0].grid.run.ion:
        assert niter == [int(x) for x in g["niter"]], "radiation sub-cycle counts, level by level"
    assert m.time == float(g["time"]) and m.dt == float(g["dt"]) and m.nstep == int(g["nstep"])
    for l, s in enumerate(m.lev):
        assert _same(s.active[..., :nv], g[f"U{l}"][..., :nv]), f"level {l}"
        if f"edgeflux{l}" in g.files:
            assert _same(s.edgeflux, g[f"edgeflux{l}"]), f"EdgeFlux level {l}"
