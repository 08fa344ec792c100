"""Static mesh refinement on the GPU (csrc/smr.hip through the C-ABI aa_mesh_*) against the golden
fixtures of the reference's own SMR build and against the CPU oracle.

Hydro-only meshes (blast, 3 levels): the strict build must agree BIT FOR BIT on every level --
restriction, flux correction and prolongation are pure sums and products.  With radiation the
device exp/log differ from glibc's in the last bits, so the 2-level sphere is held to a tolerance
(written at the assert; north_star's bar is 1e-6) with identical sub-cycle counts on both levels.
"""
import importlib
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def lib():
    return importlib.import_module("atmospheric-athena_amd.lib")


@pytest.fixture(scope="module")
def aa():
    return importlib.import_module("atmospheric-athena_amd")


def make_gpu_mesh(aa, lib, problem, overrides, strict, integrator="ctu", order=2, deck=None):
    par = aa.athinput.ParTable.from_file(deck or os.path.join(orc.DECKS, "athinput." + problem)).cmdline(overrides)
    run = aa.config.from_par(par, problem)
    run.integrator, run.order = integrator, order
    return lib.Mesh(aa.config.levels(par, run), 0, strict)


def relerr(a, b):
    out = []
    for c in range(a.shape[-1]):
        scale = np.nanmax(np.abs(b[..., c]))
        out.append(0.0 if scale == 0 else float(np.nanmax(np.abs(a[..., c] - b[..., c])) / scale))
    return out


@pytest.mark.parametrize("name", ["smr_blast_3lev_s6", "smr_blast_3lev_edge_s8", "smr_vl_blast_2lev_s5", "smr_ppm_blast_2lev_s5",
                                  "smr_blast_2dom_s6", "smr_blast_tree_s5"])
@pytest.mark.parametrize("strict", [True, False])
def test_blast_three_levels_vs_reference(aa, lib, name, strict):
    """(also 2 levels with the van Leer integrator and with third-order reconstruction: the reference's
    --enable-smr --with-integrator=vl and --with-order=3 builds; and TWO Domains on level 1 a root zone apart, whose
    flux corrections meet in the root zones between them, and a tree -- a level-2 Domain under the second of two level-1
    Domains: MeshS.Domain[nl][nd], athena.h:355-361)"""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    m = make_gpu_mesh(aa, lib, "blast", [str(o) for o in g["overrides"]], strict,
                      "vl" if name.startswith("smr_vl_") else "ctu", 3 if name.startswith("smr_ppm_") else 2, orc.deck_for("blast", g))
    try:
        m.start()
        assert m.dt == float(g["dt0"]) if strict else abs(m.dt / float(g["dt0"]) - 1) < 1e-13
        for _ in range(int(g["nstep"])):
            m.step()
        assert m.nstep == int(g["nstep"])
        for l, lev in enumerate(m.lev):
            U = lev.download()[4:-4, 4:-4, 4:-4, :5]
            ref = g[f"U{l}"][..., :5]
            if strict:
                assert m.time == float(g["time"]) and m.dt == float(g["dt"])
                assert np.array_equal(U, ref), f"level {l}: {relerr(U, ref)}"
            else:
                # fused multiply-adds only: rounding accumulation over 5-8 steps (the tree's blast of radius 0.25 fills its
                # 20^3 root: 2.9e-11 there)
                assert max(relerr(U, ref)) < (1e-10 if "tree" in name else 1e-11), f"level {l}: {relerr(U, ref)}"
    finally:
        m.close()


@pytest.mark.parametrize("name", ["smr_blast_3lev_s6", "smr_blast_3lev_edge_s8", "smr_blast_2dom_s6", "smr_blast_tree_s5"])
def test_levels_on_the_big_grid_kernels_vs_reference(aa, lib, name, monkeypatch):
    """The levels of the fixtures are small enough for the tile kernels; Grids of 4e5 zones or more (the 80^3 root of the reference's own
    deck since round 4) take k_correct_all with the x1 / x3 first passes on board, whose second-pass fluxes on the level boundaries
    (KeepPlanes) feed the flux correction.  Forced here (AA_CORRECT_ALL=1): strict build, every level bit for bit the reference's."""
    monkeypatch.setenv("AA_CORRECT_ALL", "1")
    g = np.load(os.path.join(GOLD, name + ".npz"))
    m = make_gpu_mesh(aa, lib, "blast", [str(o) for o in g["overrides"]], True, "ctu", 2, orc.deck_for("blast", g))
    try:
        m.start()
        for _ in range(int(g["nstep"])):
            m.step()
        assert m.time == float(g["time"]) and m.dt == float(g["dt"])
        for l, lev in enumerate(m.lev):
            assert np.array_equal(lev.download()[4:-4, 4:-4, 4:-4, :5], g[f"U{l}"][..., :5]), f"level {l}"
    finally:
        m.close()


@pytest.mark.parametrize("name", ["smr_blast_3lev_edge_s8", "smr_blast_tree_s5"])
@pytest.mark.parametrize("strict", [True, False])
def test_one_launch_per_coupling_step_and_overlapped_levels_change_no_bit(aa, lib, name, strict, monkeypatch):
    """Round 4: the six sides of a flux correction and of a prolongation are one launch each (the sides correct disjoint parent
    zones; where two sides' ghost regions overlap both write the same values), and aa_mesh_step integrates the levels side by side on
    streams of their own.  AA_SMR_ONE_LAUNCH=0 + AA_MESH_OVERLAP=0 is the round-3 schedule: every level's whole block, ghost zones
    and corners included, must come out bit for bit, in both builds."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    out = []
    for old in (False, True):
        for k in ("AA_SMR_ONE_LAUNCH", "AA_MESH_OVERLAP"):
            if old: monkeypatch.setenv(k, "0")
            else: monkeypatch.delenv(k, raising=False)
        m = make_gpu_mesh(aa, lib, "blast", [str(o) for o in g["overrides"]], strict, "ctu", 2, orc.deck_for("blast", g))
        try:
            m.start()
            for _ in range(int(g["nstep"])):
                m.step()
            out.append(([lev.download() for lev in m.lev], m.time, m.dt))
        finally:
            m.close()
    assert out[0][1:] == out[1][1:]
    for a, b in zip(out[0][0], out[1][0]):
        assert np.array_equal(a, b, equal_nan=True)


def test_ghost_zones_after_prolongation_bitwise(aa, lib):
    """Prolongate alone: the ghost zones of the refined levels (incl. edges and corners, written by
    several sides) must equal the oracle's after start(), bit for bit."""
    g = np.load(os.path.join(GOLD, "smr_blast_3lev_edge_s8.npz"))
    ov = [str(o) for o in g["overrides"]]
    o = orc.make_mesh("blast", None, ov).start()
    m = make_gpu_mesh(aa, lib, "blast", ov, True)
    try:
        m.start()
        for l, lev in enumerate(m.lev):
            assert np.array_equal(lev.download()[..., :5], o.lev[l].U[..., :5]), f"level {l}"
    finally:
        m.close()


@pytest.mark.parametrize("strict", [True, False])
def test_sphere_two_domains_on_a_level_vs_reference(aa, lib, strict):
    """Two level-1 Domains that both take the radiation from the root (ionrad_prolong_snd to every child), gravity, the
    pinned core inside one of them: sub-cycle counts of all three Grids and the state against the reference."""
    g = np.load(os.path.join(GOLD, "smr_ioniz_sphere_2dom_s3.npz"))
    m = make_gpu_mesh(aa, lib, "ioniz_sphere", [str(o) for o in g["overrides"]], strict)
    try:
        assert len(m.lev) == 3 and [lev.cfg.level for lev in m.lev] == [0, 1, 1]
        m.start()
        assert abs(m.dt / float(g["dt0"]) - 1) < 1e-13
        niter = []
        for _ in range(int(g["nstep"])):
            niter += m.step()
        assert niter == [int(x) for x in g["niter"]]
        assert abs(m.time / float(g["time"]) - 1) < 1e-9 and abs(m.dt / float(g["dt"]) - 1) < 1e-9
        for l, lev in enumerate(m.lev):
            U = lev.download()[4:-4, 4:-4, 4:-4, :]
            err = relerr(U, g[f"U{l}"])
            assert max(err) < 1e-8, f"grid {l}: {err}"            # north_star: 1e-6
            ef = lev.download_edgeflux(); ref = g[f"edgeflux{l}"]
            assert np.nanmax(np.abs(ef - ref)) <= 1e-9 * np.nanmax(np.abs(ref)), f"EdgeFlux grid {l}"
    finally:
        m.close()


@pytest.mark.parametrize("chain", ["by size", "correct_all"])
@pytest.mark.parametrize("strict", [True, False])
def test_sphere_two_levels_vs_reference(aa, lib, strict, chain, monkeypatch):
    """(chain "correct_all": AA_CORRECT_ALL=1, the kernels the 80^3 root of the reference's own deck takes since round 4 -- gravity,
    scalar, radiation and the kept level-boundary fluxes together)"""
    if chain == "correct_all":
        monkeypatch.setenv("AA_CORRECT_ALL", "1")
    g = np.load(os.path.join(GOLD, "smr_ioniz_sphere_2lev_s4.npz"))
    m = make_gpu_mesh(aa, lib, "ioniz_sphere", [str(o) for o in g["overrides"]], strict)
    try:
        m.start()
        assert abs(m.dt / float(g["dt0"]) - 1) < 1e-13
        niter = []
        for _ in range(int(g["nstep"])):
            niter += m.step()
        if niter != [int(x) for x in g["niter"]]:
            # seen ONCE in eleven full-suite runs of round 3 and never in isolation: say what a second Mesh in the same
            # process does, so that the next occurrence tells a persistent state from a transient one
            m2 = make_gpu_mesh(aa, lib, "ioniz_sphere", [str(o) for o in g["overrides"]], strict)
            m2.start()
            again = []
            for _ in range(int(g["nstep"])):
                again += m2.step()
            same0 = [bool(np.array_equal(a.download(), b.download(), equal_nan=True)) for a, b in zip(m.lev, m2.lev)]
            m2.close()
            raise AssertionError(f"radiation sub-cycle counts on both levels: {niter} (a second Mesh in this process: {again}; "
                                 f"final states equal: {same0}) against the reference's {[int(x) for x in g['niter']]}")
        assert abs(m.time / float(g["time"]) - 1) < 1e-9 and abs(m.dt / float(g["dt"]) - 1) < 1e-9
        for l, lev in enumerate(m.lev):
            U = lev.download()[4:-4, 4:-4, 4:-4, :]
            err = relerr(U, g[f"U{l}"])
            assert max(err) < 1e-8, f"level {l}: {err}"            # north_star: 1e-6
            ef = lev.download_edgeflux(); ref = g[f"edgeflux{l}"]
            assert np.nanmax(np.abs(ef - ref)) <= 1e-9 * np.nanmax(np.abs(ref)), f"EdgeFlux level {l}"
    finally:
        m.close()


def test_mesh_phases_match_oracle(aa, lib):
    """The individual SMR call sites (ionradRestrictCorrect, RestrictCorrect, Prolongate, new_dt) in
    the order of main.c, each checked against the oracle, on the 2-level sphere."""
    g = np.load(os.path.join(GOLD, "smr_ioniz_sphere_2lev_s4.npz"))
    ov = [str(o) for o in g["overrides"]]
    o = orc.make_mesh("ioniz_sphere", None, ov).start()
    m = make_gpu_mesh(aa, lib, "ioniz_sphere", ov, True)
    try:
        m.start()
        for l in range(2):
            assert np.array_equal(m.lev[l].download(), o.lev[l].U), f"after start, level {l}"
        n_o = o.step()
        n_g = [0, 0]
        for l in range(2):
            m.lev[l].set_mesh_state(m.time, m.lev[l].dt, m.nstep)
            n_g[l] = m.ion_radtransfer_3d(l)
            m.lev[l].bvals_mhd()
        assert n_g == n_o
        m.ionradRestrictCorrect()
        for lev in m.lev:
            lev.integrate()
        m.RestrictCorrect()
        for lev in m.lev:
            lev.apply_pinned_cells()
        # (time advance happens inside aa_mesh_step; compare the state before new_dt / Prolongate
        #  through the active zones, which those two do not touch)
        for l in range(2):
            U = m.lev[l].download()[4:-4, 4:-4, 4:-4, :]
            err = relerr(U, o.lev[l].active)
            assert max(err) < 1e-9, f"level {l}: {err}"
    finally:
        m.close()


def test_mesh_create_rejects_bad_nesting(aa, lib):
    g = np.load(os.path.join(GOLD, "smr_blast_3lev_s6.npz"))
    ov = [str(o) for o in g["overrides"]]
    bad = [o for o in ov if not o.startswith("domain2/iDisp")] + ["domain2/iDisp=0"]   # touches the periodic root edge: allowed
    par = aa.athinput.ParTable.from_file(os.path.join(orc.DECKS, "athinput.blast")).cmdline(bad)
    run = aa.config.from_par(par, "blast")
    with pytest.raises(aa.athinput.ParError):
        # level 3 (iDisp=20 -> level-1 zone 10) now lies outside level 2 (zones 0..6 of level 1)
        aa.config.levels(par, run)
    # the C-ABI refuses levels that were not created with matching aa_params.level
    import ctypes as C
    par = aa.athinput.ParTable.from_file(os.path.join(orc.DECKS, "athinput.blast")).cmdline(ov)
    run = aa.config.from_par(par, "blast")
    levels = aa.config.levels(par, run)
    g0 = lib.setup_problem(levels[0], 0, True); g1 = lib.setup_problem(levels[0], 0, True)
    try:
        hs = (C.c_void_p * 2)(g0._h, g1._h); disp = (C.c_int * 6)(0, 0, 0, 8, 20, 6); h = C.c_void_p()
        assert g0.L.aa_mesh_create(2, hs, disp, C.byref(h)) != 0
        assert b"level" in g0.L.aa_last_error()
    finally:
        g0.close(); g1.close()


def test_three_levels_with_radiation_fixed_handoff(aa, lib, monkeypatch):
    """AA_SMR_DEEP_RADIATION=fixed: refused by default; in the corrected mode the
    GPU mesh agrees with the oracle's same mode."""
    ov = ["job/num_domains=3", "domain1/Nx1=32", "domain1/Nx2=32", "domain1/Nx3=32", "problem/rp=2.1e10",
          "domain2/Nx1=32", "domain2/Nx2=28", "domain2/Nx3=24", "domain2/iDisp=16", "domain2/jDisp=18", "domain2/kDisp=20",
          "domain3/Nx1=16", "domain3/Nx2=16", "domain3/Nx3=16", "domain3/iDisp=48", "domain3/jDisp=52", "domain3/kDisp=56"]
    monkeypatch.delenv("AA_SMR_DEEP_RADIATION", raising=False)
    with pytest.raises(aa.athinput.ParError):
        make_gpu_mesh(aa, lib, "ioniz_sphere", ov, False)
    monkeypatch.setenv("AA_SMR_DEEP_RADIATION", "fixed"); monkeypatch.setenv("ORC_SMR_DEEP_RADIATION", "fixed")
    o = orc.make_mesh("ioniz_sphere", None, ov).start()
    m = make_gpu_mesh(aa, lib, "ioniz_sphere", ov, True)
    try:
        m.start()
        for _ in range(3):
            assert m.step() == o.step()
        for l, lev in enumerate(m.lev):
            err = relerr(lev.download()[4:-4, 4:-4, 4:-4, :], o.lev[l].active)
            assert max(err) < 1e-8, (l, err)
            assert np.allclose(lev.download_edgeflux(), o.lev[l].edgeflux, rtol=1e-8, atol=0)
        assert (o.lev[2].edgeflux[:-1, :-1, 0] > 0).mean() > 0.5
    finally:
        m.close()
