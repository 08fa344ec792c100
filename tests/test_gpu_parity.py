"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the golden vectors.

Two builds of the same sources are exercised:
  * strict (-ffp-contract=off): hydro must agree with the oracle BIT FOR BIT -- any difference
    is a logic error, not rounding;
  * default (fused multiply-adds allowed): agreement to rounding accumulation.
The ion step calls exp/pow, whose device implementations differ from glibc's in the last
bits, so it is held to a tolerance in both builds.  north_star's bar is 1e-6 relative on
density and ion fraction; the tolerances asserted here are far tighter and written at each
assert.
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


def relerr(a, b):
    """max |a-b| / max|b| per variable (robust where a component passes through zero)"""
    out = []
    for c in range(a.shape[-1]):
        scale = np.nanmax(np.abs(b[..., c]))
        out.append(0.0 if scale == 0 else float(np.nanmax(np.abs(a[..., c] - b[..., c])) / scale))
    return out


def _dp(a):
    import ctypes as C
    return a.ctypes.data_as(C.POINTER(C.c_double))


# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nscal", [0, 1])
def test_kernels_vs_reference_vectors(lib, strict, nscal):
    L = lib.load(strict)
    g = np.load(os.path.join(GOLD, f"kernels_nscal{nscal}.npz"))
    gam = float(g["gamma"])
    Ul = np.ascontiguousarray(g["Ul"]); Ur = np.ascontiguousarray(g["Ur"]); eta = np.ascontiguousarray(g["eta"])
    F = np.zeros_like(Ul)
    assert L.aa_test_fluxes(nscal, gam, Ul.shape[0], _dp(Ul), _dp(Ur), _dp(eta), _dp(F)) == 0
    Wp = np.ascontiguousarray(g["Wp"]); Wl = np.zeros_like(Wp); Wr = np.zeros_like(Wp)
    assert L.aa_test_lr_states(nscal, gam, Wp.shape[0], _dp(Wp), float(g["dt"]), float(g["dx"]),
                               int(g["il"]), int(g["iu"]), _dp(Wl), _dp(Wr)) == 0
    if strict:
        assert np.array_equal(F, g["F"], equal_nan=True)
        assert np.array_equal(Wl, g["Wl"]) and np.array_equal(Wr, g["Wr"])
    else:
        # fused multiply-adds: rounding-level differences only; the Roe->HLLE switch can flip on
        # a knife edge, allow a handful of those
        scale = np.abs(g["F"]).max(axis=1, keepdims=True) + 1e-300
        bad = (np.abs(F - g["F"]) / scale > 1e-10).any(axis=1)
        assert bad.sum() <= 4, f"{bad.sum()} of {len(bad)} Riemann problems differ beyond 1e-10"
        assert np.allclose(Wl, g["Wl"], rtol=1e-11, atol=1e-13) and np.allclose(Wr, g["Wr"], rtol=1e-11, atol=1e-13)


# ---------------------------------------------------------------------------------------
def run_pair(aa, lib, problem, nx, nsteps, strict, integrator="ctu"):
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
    o = orc.make_sim(problem, ov, integrator=integrator)
    run = aa.config.load(os.path.join(orc.DECKS, "athinput." + problem), ov, problem, integrator)
    g = lib.setup_problem(aa.config.slab(run), 0, strict)
    nv = 5 + run.nscal
    assert np.array_equal(g.host_initial[4:-4, 4:-4, 4:-4, :nv], o.active[..., :nv]), "problem generators disagree"
    o.start(); g.start()
    trace = []
    for _ in range(nsteps):
        no = o.step(); ng = g.step()
        trace.append((no, ng, o.dt, g.dt, o.time, g.time))
    return o, g, nv, trace


@pytest.mark.parametrize("strict", [True, False])
def test_scaling_free_quotient_and_square_root_are_the_ieee_ones(lib, strict):
    """hydro_dev.h x_div / x_sqrt / x_div_r (the compiler's division and square-root sequences without their range
    scaling) against hipcc's own a/b and sqrt(a) on the device AND against numpy's correctly rounded ones, bit for bit:
    operands over 400 decades (far beyond what a run holds), zero numerators, the deck's gamma - 1 as a
    loop-invariant denominator with the host's reciprocal."""
    L = lib.load(strict)
    rng = np.random.default_rng(5)
    n = 1 << 20
    a = np.concatenate([10.0 ** rng.uniform(-200, 200, n) * rng.uniform(1.0, 10.0, n), rng.uniform(0.5, 2.0, n),
                        np.zeros(64), 10.0 ** rng.uniform(-30, 30, n)])
    b = np.concatenate([10.0 ** rng.uniform(-200, 200, n) * rng.uniform(1.0, 10.0, n), rng.uniform(0.5, 2.0, n),
                        10.0 ** rng.uniform(-30, 30, 64), rng.choice(np.array([5.0 / 3.0 - 1.0, 1.4 - 1.0, 1.1 - 1.0, 1.0001 - 1.0]), n)])
    with np.errstate(divide="ignore"):
        keep = (np.abs(np.log10(np.maximum(a, 1e-300) / b)) < 290) | (a == 0)    # quotients away from the ends of the range
    a = np.ascontiguousarray(a[keep]); b = np.ascontiguousarray(b[keep])
    out = np.zeros((5, len(a)))
    assert L.aa_test_xdiv(len(a), _dp(a), _dp(b), _dp(out)) == 0
    bits = lambda x: x.view(np.int64)
    assert np.array_equal(bits(out[0]), bits(out[1])), "x_div differs from the compiler's a/b"
    assert np.array_equal(bits(out[0]), bits(a / b)), "x_div differs from the IEEE quotient"
    assert np.array_equal(bits(out[4]), bits(a / b)), "x_div_r with the correctly rounded reciprocal differs from the IEEE quotient"
    pos = a > 0
    assert np.array_equal(bits(out[2][pos]), bits(out[3][pos])), "x_sqrt differs from the compiler's sqrt"
    assert np.array_equal(bits(out[2][pos]), bits(np.sqrt(a[pos]))), "x_sqrt differs from the IEEE square root"
    # NaN in, NaN out
    z = np.array([np.nan, 1.0, -1.0, 4.0]); w = np.array([2.0, np.nan, 3.0, 2.0]); o = np.zeros((5, 4))
    assert L.aa_test_xdiv(4, _dp(z), _dp(w), _dp(o)) == 0
    assert np.isnan(o[0][0]) and np.isnan(o[0][1]) and np.isnan(o[2][0]) and np.isnan(o[2][2]) and o[0][3] == 2.0 and o[2][3] == 2.0


@pytest.mark.parametrize("nx,nsteps", [((16, 16, 16), 5), ((12, 20, 16), 4), ((40, 24, 32), 3)])
def test_blast_hydro_bitwise_strict(aa, lib, nx, nsteps):
    """Hydro only (CTU + PLM + Roe/HLLE + H-correction, periodic BCs, CFL): the strict build
    reproduces the oracle bit for bit, including the dt sequence."""
    o, g, nv, trace = run_pair(aa, lib, "blast", nx, nsteps, True)
    for (_, _, dto, dtg, to, tg) in trace:
        assert dto == dtg and to == tg
    U = g.download()
    assert np.array_equal(U[..., :nv], o.U[..., :nv]), relerr(U[..., :nv], o.U[..., :nv])
    g.close()


@pytest.mark.parametrize("nx,nsteps", [((16, 16, 16), 5), ((40, 24, 32), 3)])
def test_blast_hydro_default_build(aa, lib, nx, nsteps):
    o, g, nv, trace = run_pair(aa, lib, "blast", nx, nsteps, False)
    U = g.download()
    err = relerr(U[4:-4, 4:-4, 4:-4, :nv], o.active[..., :nv])
    assert max(err) < 1e-11, err              # tolerance: 1e-11 of each field's max (bar: 1e-6)
    assert abs(trace[-1][3] / trace[-1][2] - 1) < 1e-12
    g.close()


def test_blast_golden_fixture(aa, lib):
    gz = np.load(os.path.join(GOLD, "blast_12x20x16_n4.npz"))
    o, g, nv, trace = run_pair(aa, lib, "blast", tuple(int(x) for x in gz["nx"]), int(gz["nstep"]), True)
    U = g.download()[4:-4, 4:-4, 4:-4, :nv]
    assert np.array_equal(U, gz["U"][..., :nv])
    assert g.time == float(gz["time"]) and g.dt == float(gz["dt"])
    g.close()


@pytest.mark.parametrize("nscal", [0, 1])
def test_ppm_pencils_vs_reference_vectors(lib, nscal):
    """--with-order=3: lr_states_ppm.c on 2048-cell pencils (vectors from the reference's own routine),
    strict build bit for bit, incl. the scalar column whose work arrays overlap in the reference."""
    g = np.load(os.path.join(GOLD, f"kernels_ppm_nscal{nscal}.npz"))
    Wp = np.ascontiguousarray(g["Wp"])
    for strict in (True, False):
        L = lib.load(strict)
        Wl = np.zeros_like(Wp); Wr = np.zeros_like(Wp)
        assert L.aa_test_lr_states_ppm(nscal, float(g["gamma"]), Wp.shape[0], _dp(Wp), float(g["dt"]), float(g["dx"]),
                                       int(g["il"]), int(g["iu"]), _dp(Wl), _dp(Wr)) == 0
        if strict:
            assert np.array_equal(Wl, g["Wl"]) and np.array_equal(Wr, g["Wr"])
        else:
            assert np.allclose(Wl, g["Wl"], rtol=1e-11, atol=1e-12) and np.allclose(Wr, g["Wr"], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("name", ["ppm_blast_16x12x20_n5", "ppm_ifront_16x8x8_n4", "ppm_ioniz_sphere_32x32x32_n2"])
def test_ppm_golden_fixtures(aa, lib, name, strict):
    """Whole runs of the reference configured with --with-order=3 (CTU + PPM + Roe + H-correction)."""
    gz = np.load(os.path.join(GOLD, name + ".npz"))
    prob = name[4:].rsplit("_", 2)[0]
    nx = tuple(int(x) for x in gz["nx"])
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)] + [str(o) for o in gz["overrides"]]
    run = aa.config.load(os.path.join(orc.DECKS, "athinput." + prob), ov, prob)
    run.order = 3
    g = lib.setup_problem(aa.config.slab(run), 0, strict)
    nv = 5 + run.nscal
    g.start()
    niter = [g.step() for _ in range(int(gz["nstep"]))]
    U = g.download()[4:-4, 4:-4, 4:-4, :nv]
    if not run.ion:
        if strict:
            assert g.time == float(gz["time"]) and g.dt == float(gz["dt"])
            assert np.array_equal(U, gz["U"][..., :nv]), relerr(U, gz["U"][..., :nv])
        else:
            assert max(relerr(U, gz["U"][..., :nv])) < 1e-11
    else:
        assert niter == [int(x) for x in gz["niter"]]
        assert abs(g.time / float(gz["time"]) - 1) < 1e-9
        assert max(relerr(U, gz["U"][..., :nv])) < 1e-8, relerr(U, gz["U"][..., :nv])     # north_star: 1e-6
    g.close()


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("name", ["shkset1d_d1_48x8x6_n12", "shkset1d_d2_6x48x8_n12", "shkset1d_d3_8x6x48_n12"])
def test_sod_shock_tube_golden_fixtures(aa, lib, name, strict):
    """BASELINE configs[0] on the GPU: Sod's shock tube (gamma 1.4, outflow boundaries) along x1, x2
    and x3 of a 3-D box, against restart dumps of the reference's 3-D CTU + H-correction build."""
    gz = np.load(os.path.join(GOLD, name + ".npz"))
    nx = tuple(int(x) for x in gz["nx"])
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)] + [str(o) for o in gz["overrides"]]
    run = aa.config.load(os.path.join(orc.DECKS, "athinput.shkset1d"), ov, "shkset1d")
    g = lib.setup_problem(aa.config.slab(run), 0, strict)
    assert np.array_equal(g.host_initial[4:-4, 4:-4, 4:-4, :5], gz["U0"][..., :5])
    g.start()
    for _ in range(int(gz["nstep"])):
        g.step()
    U = g.download()[4:-4, 4:-4, 4:-4, :5]
    if strict:
        assert g.time == float(gz["time"]) and g.dt == float(gz["dt"])
        assert np.array_equal(U, gz["U"][..., :5]), relerr(U, gz["U"][..., :5])
    else:
        assert max(relerr(U, gz["U"][..., :5])) < 1e-11      # FMA contraction only
    # the tube stays uniform across the two transverse directions, bit for bit
    d = int(name.split("_d")[1][0]) - 1                     # shock direction 0,1,2 -> array axis 2,1,0
    line = np.moveaxis(U, 2 - d, 0)
    assert np.array_equal(line, np.broadcast_to(line[:, :1, :1, :], line.shape))
    g.close()


@pytest.fixture(params=["tile", "scan"])
def ion_path(request, monkeypatch):
    """AA_ION_FUSED: the radiation sub-cycle as tile sweep + separate update (ion_kernels.hip, short rays' default) /
    as ONE kernel with the wavefront-scan sweep (ion_pass.hip, the default from 64 zones along the rays)"""
    monkeypatch.setenv("AA_ION_FUSED", "1" if request.param == "scan" else "0")
    return request.param


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nx,nsteps", [((16, 8, 8), 3), ((8, 12, 16), 4), ((32, 16, 16), 3), ((100, 8, 6), 3), ((192, 6, 5), 2)])
def test_ifront_vs_oracle(aa, lib, nx, nsteps, strict, ion_path):
    """Hydro + ion radiation.  Same sub-cycle counts; fields within 1e-9 of each field's max
    (device exp/pow differ from glibc in the last bits)."""
    o, g, nv, trace = run_pair(aa, lib, "ifront", nx, nsteps, strict)
    assert [t[0] for t in trace] == [t[1] for t in trace], trace
    for (_, _, dto, dtg, to, tg) in trace:
        assert abs(dtg / dto - 1) < 1e-10 and abs(tg / to - 1) < 1e-10
    U = g.download()
    err = relerr(U[4:-4, 4:-4, 4:-4, :nv], o.active[..., :nv])
    assert max(err) < 1e-9, err
    ef = g.download_edgeflux()
    assert np.allclose(ef, o.edgeflux, rtol=1e-9, atol=1e-9 * np.abs(o.edgeflux).max())
    g.close()


@pytest.fixture(params=["0", "1"])
def vl_predict(request, monkeypatch):
    """AA_VL_PREDICT: the van Leer predictor as four kernels (small Grids' default) / as one (big Grids')"""
    monkeypatch.setenv("AA_VL_PREDICT", request.param)
    return request.param


@pytest.fixture(params=["0", "1"])
def fused_rates(request, monkeypatch):
    """AA_FUSED_RATES: the rates in their own pass (small Grids' default) / inside the ray sweep (big Grids')"""
    monkeypatch.setenv("AA_FUSED_RATES", request.param)
    return request.param


def test_ifront_golden_fixture(aa, lib, fused_rates, ion_path):
    gz = np.load(os.path.join(GOLD, "ifront_16x8x8_n6.npz"))
    o, g, nv, trace = run_pair(aa, lib, "ifront", (16, 8, 8), 6, False)
    assert [t[1] for t in trace] == [int(x) for x in gz["niter"]]
    U = g.download()[4:-4, 4:-4, 4:-4, :nv]
    err = relerr(U, gz["U"][..., :nv])
    assert max(err) < 1e-8, err
    x_gpu = 1.0 - U[..., 5] / U[..., 0]; x_ref = 1.0 - gz["U"][..., 5] / gz["U"][..., 0]
    assert np.max(np.abs(x_gpu - x_ref)) < 1e-8      # ion fraction (north_star bar: 1e-6)
    g.close()


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nx,nsteps", [((20, 20, 20), 1), ((24, 16, 12), 2), ((40, 40, 40), 2)])
def test_ioniz_sphere_vs_oracle(aa, lib, nx, nsteps, strict, fused_rates, ion_path):
    """Hydro + static gravity (potential tables) + ion radiation + per-step core reset."""
    o, g, nv, trace = run_pair(aa, lib, "ioniz_sphere", nx, nsteps, strict)
    assert [t[0] for t in trace] == [t[1] for t in trace], trace
    for (_, _, dto, dtg, to, tg) in trace:
        assert abs(dtg / dto - 1) < 1e-9
    U = g.download()
    a = U[4:-4, 4:-4, 4:-4, :nv]; b = o.active[..., :nv]
    assert np.array_equal(np.isnan(a), np.isnan(b))
    err = relerr(a, b)
    # 24x16x12 puts the planet (radius 2 cells) next to a 1e5 density jump: face pressures go
    # negative there (NaN etas, Roe->HLLE switches), so any last-bit difference (device exp/log,
    # fused multiply-adds) moves a few cells by ~1e-6; every other case holds 1e-8.
    # tests/test_conditioning.py: the ORACLE run again from a start state moved by one unit in the last place spreads
    # by 5e-6 .. 7e-6 on this case -- 2e-5 is what arithmetic that differs in the last bit can be held to here.
    if nx == (24, 16, 12):
        # held to what the oracle's own 1-ulp twins do in this session (tests/twins.py: ~7e-6 on ~48 zones), not to a constant
        import twins
        twin_worst, twin_nflip = twins.sphere_24x16x12()
        scale = np.nanmax(np.abs(b), axis=(0, 1, 2))
        nflip = int((np.abs(a - b) / scale > 1e-8).any(axis=-1).sum())
        print(f"N_flip (zones beyond 1e-8) = {nflip} of {a[..., 0].size} (twins: {twin_nflip}), max {max(err):.2e} (twins: {twin_worst:.2e}), strict={strict}")
        assert max(err) <= 2.0 * twin_worst, (err, twin_worst)
        assert nflip <= 2 * twin_nflip, (nflip, twin_nflip)      # every other zone holds 1e-8
    else:
        assert max(err) < 1e-8, err
    g.close()


@pytest.mark.parametrize("nx,nsteps", [((16, 12, 20), 4), ((40, 24, 32), 3)])
def test_vl_blast_bitwise_strict(aa, lib, nx, nsteps, vl_predict):
    """van Leer integrator (integrate_3d_vl.c, NO_H_CORRECTION), hydro only: bit for bit."""
    o, g, nv, trace = run_pair(aa, lib, "blast", nx, nsteps, True, "vl")
    for (_, _, dto, dtg, to, tg) in trace:
        assert dto == dtg and to == tg
    U = g.download()
    assert np.array_equal(U[..., :nv], o.U[..., :nv]), relerr(U[..., :nv], o.U[..., :nv])
    g.close()


def test_vl_golden_fixtures(aa, lib, vl_predict):
    gz = np.load(os.path.join(GOLD, "vl_blast_16x12x20_n4.npz"))
    o, g, nv, trace = run_pair(aa, lib, "blast", (16, 12, 20), 4, True, "vl")
    assert np.array_equal(g.download()[4:-4, 4:-4, 4:-4, :nv], gz["U"][..., :nv]) and g.dt == float(gz["dt"])
    g.close()
    gz = np.load(os.path.join(GOLD, "vl_ifront_16x8x8_n3.npz"))
    o, g, nv, trace = run_pair(aa, lib, "ifront", (16, 8, 8), 3, False, "vl")
    assert [t[1] for t in trace] == [int(x) for x in gz["niter"]]
    assert max(relerr(g.download()[4:-4, 4:-4, 4:-4, :nv], gz["U"][..., :nv])) < 1e-8
    g.close()


@pytest.mark.parametrize("chain", ["tiles", "correct_all", "unfused"])
@pytest.mark.parametrize("strict", [True, False])
def test_ctu_without_h_correction_golden_fixtures(aa, lib, strict, chain, monkeypatch):
    """A reference built WITHOUT --enable-h-correction (its configure default): aa_params.integrator = 2, the CTU chain with the
    etas zeroed in front of the second-pass fluxes (roe.c:282-290 without etah = with etah 0), against whole runs of the
    reference built that way, through the three forms of the kernel chain."""
    if chain == "correct_all": monkeypatch.setenv("AA_CORRECT_ALL", "1")
    if chain == "unfused": monkeypatch.setenv("AA_FUSED_UPDATE", "0")
    for name, prob in (("noh_blast_16x12x20_n4", "blast"), ("noh_ioniz_sphere_20x16x12_n2", "ioniz_sphere")):
        gz = np.load(os.path.join(GOLD, name + ".npz"))
        nx = tuple(int(x) for x in gz["nx"])
        ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
        run = aa.config.load(os.path.join(orc.DECKS, "athinput." + prob), ov, prob, "ctu-noh")
        g = lib.setup_problem(aa.config.slab(run), 0, strict)
        nv = 5 + run.nscal
        g.start()
        niter = [g.step() for _ in range(int(gz["nstep"]))]
        out = g.download()[4:-4, 4:-4, 4:-4, :nv]
        if prob == "blast":
            if strict:
                assert g.time == float(gz["time"]) and g.dt == float(gz["dt"]) and np.array_equal(out, gz["U"][..., :nv])
            else:
                assert max(relerr(out, gz["U"][..., :nv])) < 1e-11
        else:
            assert niter == [int(x) for x in gz["niter"]]
            assert abs(g.time / float(gz["time"]) - 1) < 1e-9
            # 20x16x12 puts the planet (radius 2 zones) beside a 1e5 density jump; without the H-correction's dissipation the
            # default build's last-bit differences (fused multiply-adds, reciprocal forms) grow to 1.3e-7 in the x1 momentum in
            # two steps (every other field 4e-9; the strict build holds 1e-8 everywhere).  north_star: 1e-6
            assert max(relerr(out, gz["U"][..., :nv])) < (1e-8 if strict else 1e-6), relerr(out, gz["U"][..., :nv])
        g.close()


@pytest.mark.parametrize("strict", [True, False])
def test_vl_with_third_order_reconstruction_golden_fixtures(aa, lib, vl_predict, strict):
    """--with-integrator=vl --with-order=3: the van Leer corrector on piecewise parabolic states of U^{n+1/2} without tracing
    (lr_states_ppm.c:502-507), against whole runs of the reference built that way."""
    for name, prob in (("vl_ppm_blast_16x12x20_n4", "blast"), ("vl_ppm_ioniz_sphere_20x16x12_n2", "ioniz_sphere")):
        gz = np.load(os.path.join(GOLD, name + ".npz"))
        nx = tuple(int(x) for x in gz["nx"])
        ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
        run = aa.config.load(os.path.join(orc.DECKS, "athinput." + prob), ov, prob, "vl")
        run.order = 3
        g = lib.setup_problem(aa.config.slab(run), 0, strict)
        nv = 5 + run.nscal
        g.start()
        niter = [g.step() for _ in range(int(gz["nstep"]))]
        out = g.download()[4:-4, 4:-4, 4:-4, :nv]
        if prob == "blast":
            if strict:
                assert g.time == float(gz["time"]) and g.dt == float(gz["dt"]) and np.array_equal(out, gz["U"][..., :nv])
            else:
                assert max(relerr(out, gz["U"][..., :nv])) < 1e-11
        else:
            assert niter == [int(x) for x in gz["niter"]]
            assert abs(g.time / float(gz["time"]) - 1) < 1e-9
            assert max(relerr(out, gz["U"][..., :nv])) < 1e-8, relerr(out, gz["U"][..., :nv])     # north_star: 1e-6
        g.close()


@pytest.mark.parametrize("problem,nx,nsteps", [("ifront", (16, 8, 8), 3), ("ioniz_sphere", (32, 32, 32), 2)])
def test_vl_ion_problems_vs_oracle(aa, lib, problem, nx, nsteps, vl_predict):
    o, g, nv, trace = run_pair(aa, lib, problem, nx, nsteps, False, "vl")
    assert [t[0] for t in trace] == [t[1] for t in trace], trace
    err = relerr(g.download()[4:-4, 4:-4, 4:-4, :nv], o.active[..., :nv])
    assert max(err) < 1e-8, err
    g.close()


@pytest.mark.parametrize("problem,nx,integrator", [("blast", (4, 4, 4), "ctu"), ("blast", (5, 6, 7), "ctu"),
                                                   ("blast", (7, 4, 5), "vl"), ("ifront", (4, 5, 6), "ctu"),
                                                   ("blast", (65, 4, 4), "ctu"), ("blast", (4, 4, 67), "ctu"),
                                                   ("ifront", (70, 4, 66), "ctu")])
def test_tiny_and_awkward_grids(aa, lib, problem, nx, integrator, ion_path):
    """Grids as thin as nghost, sizes straddling the 64-lane / 256-thread / 32-cell tile edges."""
    strict = problem == "blast"
    o, g, nv, trace = run_pair(aa, lib, problem, nx, 2, strict, integrator)
    U = g.download()[4:-4, 4:-4, 4:-4, :nv]
    if strict:
        assert np.array_equal(U, o.active[..., :nv])
    else:
        assert [t[0] for t in trace] == [t[1] for t in trace]
        assert max(relerr(U, o.active[..., :nv])) < 1e-9
    g.close()


@pytest.mark.parametrize("name,strict,tol", [("dev_blast_24x24x24_s30_s34", True, 0.0),
                                             ("dev_ioniz_sphere_32x32x32_s12_s15", True, 1e-9),
                                             ("dev_ioniz_sphere_32x32x32_s12_s15", False, 1e-6),
                                             ("dev_ifront_24x8x8_s40_s44", False, 1e-8),
                                             # the regime the benchmark is timed in: ONE sub-cycle per step, six steps
                                             ("dev_ioniz_sphere_36x36x36_s27_s33", True, 1e-9),
                                             ("dev_ioniz_sphere_36x36x36_s27_s33", False, 1e-8)])
def test_from_developed_reference_state(aa, lib, name, strict, tol, ion_path):
    """Load a reference state deep into the run (shocks, an evolved ionization front with up to 65
    sub-cycles per step; the stationary regime with one sub-cycle per step) the way a restart would, advance,
    compare with the reference's later state."""
    gz = np.load(os.path.join(GOLD, name + ".npz"))
    prob = name[4:].rsplit("_", 3)[0]
    nx = tuple(int(x) for x in gz["nx"])
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)] + ([str(o) for o in gz["overrides"]] if "overrides" in gz.files else [])
    run = aa.config.load(os.path.join(orc.DECKS, "athinput." + prob), ov, prob)
    g = lib.setup_problem(aa.config.slab(run), 0, strict)
    nv = 5 + run.nscal
    U = g.new_host_block()
    U[4:-4, 4:-4, 4:-4, :] = gz["UA"][..., :nv]
    g.upload(U)
    g.set_mesh_state(float(gz["timeA"]), float(gz["dtA"]), int(gz["nstepA"]))
    g.bvals_mhd(); g.bvals_ionrad()
    niter = [g.step() for _ in range(int(gz["nstepB"]) - int(gz["nstepA"]))]
    out = g.download()[4:-4, 4:-4, 4:-4, :nv]
    if strict and tol == 0.0:
        assert np.array_equal(out, gz["UB"][..., :nv]) and g.time == float(gz["timeB"]) and g.dt == float(gz["dtB"])
    else:
        assert niter == [int(x) for x in gz["niter"]]
        assert abs(g.time / float(gz["timeB"]) - 1) < 1e-10
        ref = gz["UB"][..., :nv]
        scale = np.abs(ref).max(axis=(0, 1, 2))
        assert np.all(out[..., scale == 0] == 0)
        err = np.abs(out - ref)[..., scale > 0] / scale[scale > 0]
        assert err.max() < tol, err.max(axis=(0, 1, 2))
        if tol > 1e-8:
            # the 32^3 sphere has a planet 3 zones in radius beside a 1e5 density jump: with fused
            # multiply-adds a limiter / Roe->HLLE decision flips in a few dozen zones by the third step
            # (the strict build of the same sources matches to 1e-9, in fact bit for bit); everywhere
            # else the default build stays at rounding level.  Measured 1.8e-8 on 43 zones; asserted: north_star's 1e-6
            # (tol), and no worse than the ORACLE's own 1-ulp twins of this pair do in this session (tests/twins.py:
            # ~6e-5 on ~27 zones), in size and in the number of zones that move at all.
            import twins
            twin_worst, twin_nflip = twins.developed_sphere_32()
            nflip = int((err > 1e-9).any(axis=-1).sum())
            print(f"N_flip (zones beyond 1e-9) = {nflip} of {err[..., 0].size} (twins: {twin_nflip}), max {err.max():.2e} (twins: {twin_worst:.2e})")
            assert err.max() <= twin_worst and nflip <= 3 * twin_nflip, (err.max(), twin_worst, nflip, twin_nflip)
    g.close()


@pytest.mark.parametrize("spec", ["1", "0"])
@pytest.mark.parametrize("strict", [True, False])
def test_headline_regime_against_the_reference(aa, lib, strict, spec, ion_path, monkeypatch):
    """The regime bench.py times: dt at the CFL limit, ONE radiation sub-cycle per step, i.e. ionrad_3d.c:919-1012 with the
    loop body executed once.  On the device that is the speculated first update confirmed by k_ion_pick2 (spec_state 1), the
    closing pass skipped and new_dt's maxima taken in the update kernel (aa_step).  Reference pair: steps 27 -> 33 of the
    planet in a +-1.5e10 cm box at 36^3 (tests/golden/make_golden.py `dev`), sub-cycle counts 1 x 6, NaN-free.  Both builds,
    both forms of the sub-cycle, speculation on and off: same sub-cycle counts, dt sequence to 1e-12, fields <= 1e-9 (strict)
    / 1e-8 (default) of each field's maximum, ion fraction <= 1e-8, EdgeFlux <= 1e-9 (north_star's bar: 1e-6)."""
    monkeypatch.setenv("AA_ION_SPECULATE", spec)
    gz = np.load(os.path.join(GOLD, "dev_ioniz_sphere_36x36x36_s27_s33.npz"))
    nx = tuple(int(x) for x in gz["nx"])
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)] + [str(o) for o in gz["overrides"]]
    run = aa.config.load(os.path.join(orc.DECKS, "athinput.ioniz_sphere"), ov, "ioniz_sphere")
    g = lib.setup_problem(aa.config.slab(run), 0, strict)
    assert g.ion_is_fused() == (ion_path == "scan")
    nv = 5 + run.nscal
    U = g.new_host_block()
    U[4:-4, 4:-4, 4:-4, :] = gz["UA"][..., :nv]
    g.upload(U)
    g.set_mesh_state(float(gz["timeA"]), float(gz["dtA"]), int(gz["nstepA"]))
    g.bvals_mhd(); g.bvals_ionrad()
    niter = [g.step() for _ in range(int(gz["nstepB"]) - int(gz["nstepA"]))]
    assert niter == [1] * 6 == [int(x) for x in gz["niter"]]
    assert abs(g.time / float(gz["timeB"]) - 1) < 1e-12 and abs(g.dt / float(gz["dtB"]) - 1) < 1e-12
    out = g.download()[4:-4, 4:-4, 4:-4, :nv]; ref = gz["UB"][..., :nv]
    assert np.isfinite(out).all()
    err = relerr(out, ref)
    assert max(err) < (1e-9 if strict else 1e-8), err
    xg = 1.0 - out[..., 5] / out[..., 0]; xr = 1.0 - ref[..., 5] / ref[..., 0]
    assert np.max(np.abs(xg - xr)) < 1e-8
    ef = g.download_edgeflux()
    assert np.allclose(ef, gz["edgefluxB"], rtol=1e-9, atol=1e-9 * np.abs(gz["edgefluxB"]).max())
    g.close()


def test_ioniz_sphere_default_kernels_vs_oracle(aa, lib):
    """128^3, 2 steps (11 + 5 sub-cycles): the size from which every size-switched kernel is the default (k_correct_all,
    the one-kernel sub-cycle with the scan sweep, k_vl_predict's threshold too), compared with the oracle, default
    (fused multiply-add) build: north_star's bar is 1e-6 on density and ion fraction; asserted 1e-8."""
    o, g, nv, trace = run_pair(aa, lib, "ioniz_sphere", (128, 128, 128), 2, False)
    assert [t[0] for t in trace] == [t[1] for t in trace], trace
    a = g.download()[4:-4, 4:-4, 4:-4, :nv]; b = o.active[..., :nv]
    assert np.isfinite(a).all() and np.isfinite(b).all()
    err = relerr(a, b)
    assert max(err) < 1e-8, err
    xg = 1.0 - a[..., 5] / a[..., 0]; xo = 1.0 - b[..., 5] / b[..., 0]
    assert np.max(np.abs(xg - xo)) < 1e-8
    ef = g.download_edgeflux()
    assert np.allclose(ef, o.edgeflux, rtol=1e-9, atol=1e-9 * np.abs(o.edgeflux).max())
    g.close()


def test_128_cubed_12_steps_against_the_reference_mpi_run(aa, lib):
    """The run bench.py's cpu_baseline times anyway -- the reference's own MPI build (oracle/_ref/athena_ioniz_sphere_mpi)
    on the host cores, ioniz_sphere 128^3, 12 steps (11, 5, then 4 sub-cycles per step) -- with full-precision restart dumps,
    reassembled (tests/refmpi.py), against the DEFAULT build with every size-switched kernel in its default form
    (k_correct_all, k_flux2_update with new_dt's maxima, the one-kernel sub-cycle with the scan sweep and speculation).
    Twelve steps, default kernels, the real reference: north_star's bar is 1e-6 on density and ion fraction; asserted 1e-8,
    with identical sub-cycle counts and the dt sequence's end to 1e-12."""
    import refmpi
    if not refmpi.available():
        pytest.skip("oracle/_ref/athena_ioniz_sphere_mpi or mpiexec not on this box")
    nx = (128, 128, 128); nsteps = 12
    ref = refmpi.run(nx, nsteps)
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
    run = aa.config.load(os.path.join(orc.DECKS, "athinput.ioniz_sphere"), ov, "ioniz_sphere")
    g = lib.setup_problem(aa.config.slab(run), 0, False)
    assert g.ion_is_fused()
    nv = 5 + run.nscal
    g.start()
    niter = [g.step() for _ in range(nsteps)]
    assert niter == ref["niter"], (niter, ref["niter"])
    assert abs(g.time / ref["time"] - 1) < 1e-12 and abs(g.dt / ref["dt"] - 1) < 1e-12
    a = g.download()[4:-4, 4:-4, 4:-4, :nv]; b = ref["U"][..., :nv]
    assert np.isfinite(a).all() and np.isfinite(b).all()
    err = relerr(a, b)
    xg = 1.0 - a[..., 5] / a[..., 0]; xr = 1.0 - b[..., 5] / b[..., 0]
    ex = float(np.max(np.abs(xg - xr)))
    print(f"128^3 x 12 steps vs the reference's MPI run on {ref['ranks']} ranks: max rel err per field {err}, ion fraction {ex:.2e}")
    assert max(err) < 1e-8, err
    assert ex < 1e-8
    g.close()


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("name", ["rayplane_dir1_12x10x8_n3", "rayplane_dir2_12x10x8_n3", "rayplane_dir2_6x70x5_n2"])
def test_ray_directions_golden_fixtures(aa, lib, name, strict, ion_path):
    """Radiation planes with rays along +x1 (dir=-1) and +x2 (dir=-2; get_ph_rate_plane case -2, ionradplane_3d.c:323-354)
    against runs of the reference on our own problem file tests/fixtures/rayplane_dir.c.  dir=-3 and dir>0 have no
    defined behaviour in the reference and are refused."""
    gz = np.load(os.path.join(GOLD, name + ".npz"))
    nx = tuple(int(x) for x in gz["nx"]); raydir = -int(name.split("_dir")[1][0])
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
    run = aa.config.load(os.path.join(orc.DECKS, "athinput.ifront"), ov, "ifront")
    g = lib.Grid(aa.config.slab(run), 0, strict)
    U = g.new_host_block()
    U[4:-4, 4:-4, 4:-4, :] = orc.rayplane_pattern(nx, run.prob["n_H"], run.ionp["m_H"], run.prob["cs"], run.gamma)
    assert np.array_equal(U[4:-4, 4:-4, 4:-4], gz["U0"])
    g.upload(U)
    g.add_radplane_3d(raydir, run.prob["flux"])
    for bad in (1, 2, 3, -3):
        with pytest.raises(lib.AthenaError, match="defined behaviour"):
            g.add_radplane_3d(bad, 1.0)
    g.start()
    niter = [g.step() for _ in range(int(gz["nstep"]))]
    assert niter == [int(x) for x in gz["niter"]]
    assert abs(g.time / float(gz["time"]) - 1) < 1e-10 and abs(g.dt / float(gz["dt"]) - 1) < 1e-10
    out = g.download()[4:-4, 4:-4, 4:-4]
    assert max(relerr(out, gz["U"])) < 1e-9, relerr(out, gz["U"])            # north_star: 1e-6
    ef = g.download_edgeflux()
    assert np.allclose(ef, gz["edgeflux"], rtol=1e-9, atol=1e-9 * np.abs(gz["edgeflux"]).max())
    g.close()


@pytest.mark.parametrize("chain", ["tiles", "correct_all", "unfused"])
@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("name", ["coolpat_c1_16x12x10_n4", "coolpat_c1_12x8x20_n3", "coolpat_c0_16x12x10_n4"])
def test_optically_thin_cooling_against_the_reference(aa, lib, name, strict, chain, monkeypatch):
    """CoolingFunc = KoyInut (microphysics/cool.c:48) in the CTU integrator -- integrate_3d_ctu.c Steps 1c-3c (:359-368, :662-671,
    :846-855), 8b (:2133-2266), 11c (:2943-2953) -- against runs of the reference on our own problem file
    tests/fixtures/cool_pattern.c (diffuse gas in cgs units, 150 K ... 6000 K, density jumps of 6, velocities of either sign),
    through all three forms of the kernel chain (tile kernels / k_correct_all, fused / unfused update).  The device's log10 /
    exp / pow differ from glibc's in the last place, so the comparison has a tolerance in both builds; the fixture without the
    cooling function (c0) runs through the same entry point with AA_COOL_NONE."""
    if chain == "correct_all": monkeypatch.setenv("AA_CORRECT_ALL", "1")
    if chain == "unfused": monkeypatch.setenv("AA_FUSED_UPDATE", "0")
    gz = np.load(os.path.join(GOLD, name + ".npz"))
    ov, n0, T0, v0, cool = orc.coolpat_setup(gz)
    run = aa.config.load(os.path.join(orc.DECKS, "athinput.blast"), ov, "blast")
    g = lib.Grid(aa.config.slab(run), 0, strict)
    U = g.new_host_block()
    U[4:-4, 4:-4, 4:-4, :] = orc.cool_pattern(gz["nx"], n0, T0, v0, run.gamma)[..., :U.shape[-1]]
    assert np.array_equal(U[4:-4, 4:-4, 4:-4, :5], gz["U0"][..., :5])
    g.upload(U)
    g.set_cooling(cool)
    g.start()
    assert g.dt == float(gz["dt0"])
    for _ in range(int(gz["nstep"])): g.step()
    out = g.download()[4:-4, 4:-4, 4:-4, :5]
    err = max(relerr(out, gz["U"][..., :5]))
    if cool == 0 and strict:
        assert g.time == float(gz["time"]) and g.dt == float(gz["dt"]) and np.array_equal(out, gz["U"][..., :5])
    else:
        assert abs(g.time / float(gz["time"]) - 1) < 1e-11 and abs(g.dt / float(gz["dt"]) - 1) < 1e-11
        assert err < 1e-10, err                                      # north_star: 1e-6
    # and the terms matter: the run without them is per cent away
    if cool:
        ref0 = np.load(os.path.join(GOLD, "coolpat_c0_16x12x10_n4.npz"))
        if tuple(ref0["nx"]) == tuple(gz["nx"]):
            assert np.abs(out[..., 4] / ref0["U"][..., 4] - 1).max() > 1e-2
    g.close()


@pytest.mark.parametrize("chain", ["tiles", "correct_all"])
@pytest.mark.parametrize("order", [2, 3])
@pytest.mark.parametrize("strict", [True, False])
def test_cooling_beside_gravity_scalar_and_radiation_vs_oracle(aa, lib, strict, order, chain, monkeypatch):
    """The cooling terms in the instantiations the reference fixtures do not reach: with the static potential (its kick on the
    L/R states before the cooling, its term in P^{n+1/2}, :2166-2189), the passive scalar, third order reconstruction and the
    ion step in front -- ioniz_sphere 40^3 with KoyInut enrolled on both sides, HIP against the oracle (whose cooling terms the
    reference fixtures pin bit for bit)."""
    if chain == "correct_all": monkeypatch.setenv("AA_CORRECT_ALL", "1")
    nx = (40, 40, 40)
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
    o = orc.make_sim("ioniz_sphere", ov, order=order)
    run = aa.config.load(os.path.join(orc.DECKS, "athinput.ioniz_sphere"), ov, "ioniz_sphere")
    run.order = order
    g = lib.setup_problem(aa.config.slab(run), 0, strict)
    o.set_cooling(1); g.set_cooling(1)
    o.start(); g.start()
    for _ in range(2):
        assert o.step() == g.step()
        assert abs(g.dt / o.dt - 1) < 1e-9
    a = g.download()[4:-4, 4:-4, 4:-4]; b = o.active
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert max(relerr(a, b)) < 1e-8, relerr(a, b)
    # the same two steps without the cooling function differ from these
    o2 = orc.make_sim("ioniz_sphere", ov, order=order); o2.start()
    for _ in range(2): o2.step()
    assert max(relerr(o2.active, b)) > 1e-12
    g.close()


def test_cooling_on_a_grid_cut_into_slabs(aa, lib):
    """aa_set_cooling on a composite handle (aa_params.nslab = 2, the drop-in's AA_NGPU): every slab runs the cooling kernels and
    keeps its own P^{n+1/2}; 12 x 8 x 20 in two slabs of 10 planes against the reference run of the undivided Grid."""
    gz = np.load(os.path.join(GOLD, "coolpat_c1_12x8x20_n3.npz"))
    ov, n0, T0, v0, cool = orc.coolpat_setup(gz)
    run = aa.config.load(os.path.join(orc.DECKS, "athinput.blast"), ov, "blast")
    g = lib.Grid(aa.config.slab(run), 0, True, 0, 2)
    U = g.new_host_block()
    U[4:-4, 4:-4, 4:-4, :] = orc.cool_pattern(gz["nx"], n0, T0, v0, run.gamma)[..., :U.shape[-1]]
    g.upload(U)
    g.set_cooling(cool)
    g.start()
    for _ in range(int(gz["nstep"])): g.step()
    out = g.download()[4:-4, 4:-4, 4:-4, :5]
    assert abs(g.time / float(gz["time"]) - 1) < 1e-11 and abs(g.dt / float(gz["dt"]) - 1) < 1e-11
    assert max(relerr(out, gz["U"][..., :5])) < 1e-10
    g.close()


def test_cooling_is_refused_where_the_reference_has_none(aa, lib):
    run = aa.config.load(os.path.join(orc.DECKS, "athinput.blast"), ["domain1/Nx1=8", "domain1/Nx2=8", "domain1/Nx3=8"], "blast")
    run.integrator = "vl"
    g = lib.Grid(aa.config.slab(run), 0, False)
    with pytest.raises(lib.AthenaError, match="van Leer"):
        g.set_cooling(1)
    with pytest.raises(lib.AthenaError, match="AA_COOL"):
        g.set_cooling(7)
    g.close()


def test_round_trip_and_bc(aa, lib):
    """upload -> download is the identity; ghost zones after bvals_mhd equal the oracle's for
    reflect/outflow (ifront deck) and periodic (blast deck)."""
    for prob, nx in (("ifront", (8, 12, 16)), ("blast", (12, 8, 10))):
        ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
        o = orc.make_sim(prob, ov)
        run = aa.config.load(os.path.join(orc.DECKS, "athinput." + prob), ov, prob)
        g = lib.setup_problem(aa.config.slab(run), 0, True)
        nv = 5 + run.nscal
        rng = np.random.default_rng(7)
        U = g.new_host_block(); U[...] = rng.uniform(0.5, 2.0, U.shape)
        g.upload(U)
        assert np.array_equal(g.download(), U)
        o.U[..., :nv] = U
        o.bvals(); g.bvals_mhd()
        assert np.array_equal(g.download(), o.U[..., :nv])
        g.close()


def test_ghost_zone_refresh(aa, lib):
    """aa_download_ghost_zones rewrites exactly the ghost shell of the caller's block: after bvals_mhd it equals a full
    download, and a block whose active zones were scribbled on keeps the scribble (the shim's end-of-step refresh)."""
    for prob, nx in (("ifront", (8, 12, 16)), ("blast", (12, 8, 10)), ("blast", (70, 9, 5))):
        ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
        run = aa.config.load(os.path.join(orc.DECKS, "athinput." + prob), ov, prob)
        g = lib.setup_problem(aa.config.slab(run), 0, True)
        rng = np.random.default_rng(11)
        U = g.new_host_block(); U[...] = rng.uniform(0.5, 2.0, U.shape)
        g.upload(U)
        g.bvals_mhd()
        full = g.download()
        H = U.copy()                      # host copy: active zones current, ghost zones stale
        g.download_ghost_zones(H)
        assert np.array_equal(H, full)
        H = np.full_like(U, -7.0)
        g.download_ghost_zones(H)
        assert np.all(H[4:-4, 4:-4, 4:-4] == -7.0)
        shell = np.ones(U.shape[:3], bool); shell[4:-4, 4:-4, 4:-4] = False
        assert np.array_equal(H[shell], full[shell])
        g.close()


def test_missing_library_is_loud(lib, monkeypatch):
    monkeypatch.setattr(lib, "HERE", "/nonexistent")
    lib._libs.clear()
    with pytest.raises(lib.AthenaError):
        lib.load(False)
    monkeypatch.undo()
    lib._libs.clear()
    lib.load(False)
