"""Function-level checks of the one-kernel radiation sub-cycle (csrc/ion_pass.hip): its batched exp and its ln against
numpy (glibc) in double precision, and its ray sweep (wavefront prefix product with the MINFLUXFRAC cut-off) against
the oracle's serial sweep on rays built to end exactly at, just above and just below the cut-off threshold."""
import importlib
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return importlib.import_module("atmospheric-athena_amd.lib")


def _dp(a):
    import ctypes as C
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_exp_and_log_of_the_pass_kernel(lib):
    L = lib.load(False)
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-745.0, 10.0, 400000), rng.uniform(-2.0, 2.0, 200000), -np.logspace(-12, 2.8, 100000),
                        np.array([0.0, -0.0, 1e-300, -1e-300, -744.9, -746.0, -1e4, -1e9, 1.0, np.log(2.0) / 2, -np.log(2.0) / 2, 6.5])])
    x = np.ascontiguousarray(x[: len(x) // 4 * 4])
    ye = np.zeros_like(x); yl = np.zeros_like(x)
    assert L.aa_test_explog(len(x), _dp(x), _dp(ye), _dp(yl)) == 0
    ref = np.exp(x)
    big = ref > 1e-300                                             # normal results: relative error
    rel = np.abs(ye[big] - ref[big]) / ref[big]
    assert rel.max() < 2.3e-16, rel.max()                          # ~1 ulp (2.2e-16 = 2^-52)
    assert np.all(ye[~big] <= 1.01e-300) and np.all(ye[~big] >= 0)
    # ln on the range temperatures live in, and far beyond
    t = np.ascontiguousarray(np.concatenate([np.logspace(-3, 12, 300000), rng.uniform(0.5, 2.0, 100000), np.logspace(-250, 250, 100000)]))
    te = np.zeros_like(t); tl = np.zeros_like(t)
    assert L.aa_test_explog(len(t), _dp(t), _dp(te), _dp(tl)) == 0
    lref = np.log(t)
    big = np.abs(lref) >= 0.5                                     # every temperature above the floor
    rel = np.abs(tl - lref)[big] / np.abs(lref[big])
    assert rel.max() < 4.5e-16, rel.max()                          # <= 2 ulp
    # |ln x| < 0.5: x = m 2^e with e = -1 and m just below 2 makes e ln2 and ln m cancel -- absolute accuracy there
    assert np.abs(tl - lref)[~big].max() < 3e-16, np.abs(tl - lref)[~big].max()
    # NaN in, NaN out
    z = np.array([np.nan, 1.0, 2.0, 3.0]); ze = np.zeros(4); zl = np.zeros(4)
    assert L.aa_test_explog(4, _dp(z), _dp(ze), _dp(zl)) == 0
    assert np.isnan(ze[0]) and np.isnan(zl[0]) and ze[1] == np.exp(1.0)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("strict", [True, False])
def test_ray_cut_off_at_the_threshold(lib, fused, strict):
    """Rays built so that flux/(f0 + 1e-12) passes MINFLUXFRAC = 1e-3 within a few units in the last place, in the
    first zone of a ray and after three attenuating zones, on both sides of the threshold (ionradplane_3d.c:298-306).
    The kernels test `flux < MINFLUXFRAC*(f0 + 1e-12)` (no division; the one-kernel sub-cycle also multiplies the
    attenuation factors in a tree): away from the threshold the cut must fall in the oracle's zone; within 8 units in
    the last place of it the cut may fall one zone later or earlier, and everything behind a cut must be zero."""
    aa = importlib.import_module("atmospheric-athena_amd")
    nx = (64, 48, 8) if fused else (16, 48, 8)
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
    run = aa.config.load(os.path.join(orc.DECKS, "athinput.ifront"), ov, "ifront")
    g = lib.Grid(aa.config.slab(run), 0, strict, ion_path=1 if fused else 2)
    o = orc.make_sim("ifront", ov)
    m_H, sig = run.ionp["m_H"], run.ionp["sigma_ph"]
    dx1 = (run.xmax[0] - run.xmin[0]) / nx[0]
    # column per ray: ray (j, k) wants exp(-tau_total) = 1e-3 * (1 + delta), delta on a grid of a few 1e-16 around 0;
    # rays with k even reach it in their first zone, the others after three zones of tau = 0.5
    U = g.new_host_block()
    act = U[4:-4, 4:-4, 4:-4]
    rho = 100.0 * run.prob["n_H"] * m_H
    act[..., 0] = 50.0 * rho; act[..., 4] = 50.0 * rho * run.prob["cs"] ** 2 / (run.gamma - 1.0)
    act[..., 5] = rho                                   # behind the designed zones: optically thick
    for k in range(nx[2]):
        for j in range(nx[1]):
            delta = (j - nx[1] // 2) * 1.1e-16 * (1 + k // 2)
            tau_tot = -np.log(1e-3 * (1.0 + delta))
            if k % 2 == 0:
                taus = [tau_tot]
            else:
                taus = [0.5, 0.5, 0.5, tau_tot - 1.5]
            for i, tau in enumerate(taus):
                act[k, j, i, 5] = tau * m_H / (sig * dx1)
    assert act[..., 5].max() < act[..., 0].min()
    g.upload(U); o.active[...] = act
    g.add_radplane_3d(-1, run.prob["flux"]); o.add_radplane(-1, run.prob["flux"])
    g.bvals_mhd(); g.bvals_ionrad(); o.bvals(); o.bvals_ionrad()
    # first sweep of an ion step on both sides
    o.ion_begin(); o.ion_rates()
    g.ion_begin()
    if fused:
        assert g.ion_is_fused()
        g.ion_pass(False, True); g.ion_pick(0, 1, True, 1e300); g.ion_pass(True, True); g.ion_pick(0, 1, False, 1e300)
        g.ion_fetch(); g.ion_finish()
    else:
        g.ion_rates()
    ef = g.download_edgeflux(); eo = o.edgeflux
    f0 = eo[:nx[2], :nx[1], 0]
    assert np.all(f0 > 0) and np.array_equal(ef[:nx[2], :nx[1], 0], f0)
    ncut_same = ncut_near = nlit = 0
    for k in range(nx[2]):
        for j in range(nx[1]):
            a, b = ef[k, j, :], eo[k, j, :]
            ca = int(np.argmax(a == 0)) if (a == 0).any() else len(a)        # first edge behind the cut
            cb = int(np.argmax(b == 0)) if (b == 0).any() else len(b)
            assert np.all(a[ca:] == 0) and np.all(b[cb:] == 0)                # zeros behind a cut, nothing else
            n = min(ca, cb)
            assert np.allclose(a[:n], b[:n], rtol=1e-14, atol=0)
            cz = 1 if k % 2 == 0 else 4                                       # the edge behind the designed zones
            frac = None
            if cb > cz:                                                      # the oracle's ray survived the designed zones
                frac = b[cz] / (b[0] + 1e-12)
                nlit += 1
            if ca == cb:
                ncut_same += 1
            else:
                # only a ray whose flux fraction behind the designed zones is within 8 ulp of the threshold may differ, by one zone
                fa = (a[cz] if ca > cz else b[cz] if cb > cz else 0.0) / (b[0] + 1e-12)
                assert abs(fa / 1e-3 - 1.0) < 8 * 2.3e-16 and {ca, cb} == {cz, cz + 1}, (k, j, ca, cb, fa)
                ncut_near += 1
    print(f"cut-off threshold: {ncut_same} rays cut in the oracle's zone, {ncut_near} within 8 ulp of the threshold cut one side apart, "
          f"{nlit} rays pass the designed zones (fused={fused}, strict={strict})")
    assert nlit > 20 and nlit < nx[1] * nx[2] - 20          # both sides of the threshold are populated
    assert ncut_near <= nx[1] * nx[2] // 8
    g.close()


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("problem,ov,nstep", [
    ("ioniz_sphere", ["domain1/Nx1=64", "domain1/Nx2=16", "domain1/Nx3=16"], 12),     # 80, 13, 12, 11, 10 sub-cycles, then 1 per step
    ("ifront", ["domain1/Nx1=64", "domain1/Nx2=8", "domain1/Nx3=8", "problem/flux=1e3"], 5),
    ("ifront", ["domain1/Nx1=128", "domain1/Nx2=6", "domain1/Nx3=5"], 4)])
def test_speculative_first_update_changes_no_bit(problem, ov, nstep, strict, monkeypatch):
    """aa_ion_speculate: the first pass of an ion step also applies the first sub-cycle's update with the whole hydro step;
    k_ion_pick2 then finds out whether that was the step (one sub-cycle per hydro step: the closing update pass is skipped) or
    not (the next pass restarts from e_init / s_init).  Against AA_ION_SPECULATE=0: the same sub-cycle counts, dt, state and
    EdgeFlux, bit for bit, in both builds -- the update is the same code on the same operands -- over steps of both kinds."""
    import importlib
    import os
    aa = importlib.import_module("atmospheric-athena_amd")
    lib = importlib.import_module("atmospheric-athena_amd.lib")
    decks = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "atmospheric-athena_amd", "decks")
    out = []
    for on in ("0", "1"):
        monkeypatch.setenv("AA_ION_SPECULATE", on)
        run = aa.config.load(os.path.join(decks, "athinput." + problem), ov, problem)
        g = lib.setup_problem(aa.config.slab(run), 0, strict)
        assert g.ion_is_fused()
        g.start()
        its, dts = [], []
        for _ in range(nstep):
            its.append(g.step()); dts.append(g.dt)
        out.append((its, dts, g.download(), g.download_edgeflux()))
        g.close()
    a, b = out
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2], equal_nan=True) and np.array_equal(a[3], b[3], equal_nan=True)
    if problem == "ioniz_sphere":
        assert 1 in a[0] and max(a[0]) > 1          # both kinds of step were exercised


def test_speculation_needs_the_limit_it_was_given(monkeypatch):
    """aa_ion_speculate(limit) followed by aa_ion_pick(first, another limit) is a caller's bug: refused, not guessed at."""
    import importlib
    import os
    aa = importlib.import_module("atmospheric-athena_amd")
    lib = importlib.import_module("atmospheric-athena_amd.lib")
    decks = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "atmospheric-athena_amd", "decks")
    run = aa.config.load(os.path.join(decks, "athinput.ifront"), ["domain1/Nx1=64", "domain1/Nx2=8", "domain1/Nx3=8"], "ifront")
    g = lib.setup_problem(aa.config.slab(run), 0, True)
    g.start()
    g.ion_begin()
    g.ion_speculate(g.dt)
    g.ion_pass(False, True, 0)
    with pytest.raises(Exception):
        g.ion_pick(0, 1, True, 0.5 * g.dt)
    g.close()
