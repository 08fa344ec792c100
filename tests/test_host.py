"""CPU-side checks: the athinput reader, the C-ABI libraries export every symbol the header
declares (loaded, never called: no GPU here), the host C problem files reproduce the oracle's
initial conditions bit for bit."""
import ctypes as C
import importlib
import os
import re
import sys

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKGDIR = os.path.join(ROOT, "atmospheric-athena_amd")


def test_athinput_reader(aa):
    P = aa.athinput.ParTable
    t = P.from_text("<job>\nproblem_id = x # c\nmaxout=3\n<time>\ncour_no = 0.4\nnlim = 100000000000\n"
                    "<ionradiation>\nmaxiter = 100000000.\n")
    assert t.gets("job", "problem_id") == "x" and t.geti("job", "maxout") == 3
    assert t.getd("time", "cour_no") == 0.4
    assert t.geti("ionradiation", "maxiter") == 100000000          # atoi() semantics
    t.cmdline(["time/cour_no=0.3"])
    assert t.getd("time", "cour_no") == 0.3
    with pytest.raises(aa.athinput.ParError):                       # par.c:194: only existing keys
        t.cmdline(["time/newkey=1"])
    with pytest.raises(aa.athinput.ParError):
        t.cmdline(["nosuchblock/x=1"])
    with pytest.raises(aa.athinput.ParError):
        t.getd("time", "missing")
    assert t.getd_def("time", "missing", 2.5) == 2.5 and t.exist("time", "missing")


def test_config_rejects_what_the_reference_rejects(aa):
    deck = os.path.join(PKGDIR, "decks", "athinput.blast")
    with pytest.raises(aa.athinput.ParError):                       # integrate.c:66-68
        aa.config.load(deck, ["time/cour_no=0.8"], "blast")
    with pytest.raises(aa.athinput.ParError):                       # bvals_mhd.c:586
        aa.config.load(deck, ["domain1/bc_ix1=7"], "blast")
    with pytest.raises(aa.athinput.ParError):
        aa.config.load(deck, ["domain1/Nx3=1"], "blast")


def test_mesh_of_several_domains_per_level_follows_init_mesh(aa):
    """config.levels: Domains level by level in deck order (MeshS.Domain[nl][nd]); what init_mesh.c refuses is refused: Domains
    of a level that overlap or touch (:398-418), a child that touches its parent's edge away from the root boundary or lies
    closer than nghost/2 parent zones to it (:448-499), a Domain outside every Domain of the level below."""
    deck = os.path.join(PKGDIR, "decks", "athinput.blast")
    base = ["job/num_domains=3", "domain1/Nx1=16", "domain1/Nx2=24", "domain1/Nx3=16",
            "domain2/Nx1=12", "domain2/Nx2=16", "domain2/Nx3=20", "domain2/iDisp=8", "domain2/jDisp=20", "domain2/kDisp=6", "domain3/level=1"]

    def lev(third):
        par = aa.athinput.ParTable.from_file(deck).cmdline(base + third)
        return aa.config.levels(par, aa.config.from_par(par, "blast"))

    def dom3(nx, disp):
        return [f"domain3/Nx{d + 1}={nx[d]}" for d in range(3)] + [f"domain3/{k}Disp={disp[d]}" for d, k in enumerate("ijk")]

    g = lev(dom3((6, 8, 8), (22, 24, 10)))
    assert [x.level for x in g] == [0, 1, 1] and g[1].Nx == (12, 16, 20) and g[2].disp == (22, 24, 10)
    assert g[2].bc == (0, 0, 0, 0, 0, 0)                                   # fine / coarse sides all round
    with pytest.raises(aa.athinput.ParError, match="overlap or touch"):
        lev(dom3((6, 8, 8), (20, 24, 10)))                                 # starts where domain2 ends in x1
    with pytest.raises(aa.athinput.ParError, match="overlap or touch"):
        lev(dom3((8, 8, 8), (14, 24, 10)))                                 # inside domain2
    with pytest.raises(aa.athinput.ParError, match="nghost/2"):
        lev(dom3((8, 8, 8), (22, 24, 10)))                                 # one root zone from the root Domain's upper x1 edge
    with pytest.raises(aa.athinput.ParError, match="not inside"):
        lev(dom3((8, 8, 8), (28, 24, 10)))                                 # sticks out of the root
    # a level-2 Domain under the SECOND level-1 Domain finds that one as its parent
    par = aa.athinput.ParTable.from_file(deck).cmdline(
        ["job/num_domains=3", "domain1/Nx1=32", "domain1/Nx2=32", "domain1/Nx3=32",
         "domain2/Nx1=16", "domain2/Nx2=16", "domain2/Nx3=16", "domain2/iDisp=8", "domain2/jDisp=8", "domain2/kDisp=8",
         "domain3/Nx1=16", "domain3/Nx2=16", "domain3/Nx3=16", "domain3/iDisp=24", "domain3/jDisp=24", "domain3/kDisp=24"])
    g = aa.config.levels(par, aa.config.from_par(par, "blast"))
    assert [x.level for x in g] == [0, 1, 2]


@pytest.mark.parametrize("so", ["libathena_amd.so", "libathena_amd_strict.so"])
def test_cabi_exports_every_declared_symbol(so):
    hdr = open(os.path.join(ROOT, "include", "athena_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(aa_[a-z0-9_]+)\s*\(", hdr)))
    names = [n for n in names if n != "aa_gravpot_fn"]
    assert len(names) >= 35
    L = C.CDLL(os.path.join(PKGDIR, so))
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_binding_signature_table_matches_header():
    lib = importlib.import_module("atmospheric-athena_amd.lib")
    hdr = open(os.path.join(ROOT, "include", "athena_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(aa_[a-z0-9_]+)\s*\(", hdr)) - {"aa_gravpot_fn"}
    L = lib.load(False)
    assert set(L._sig) == names


@pytest.mark.parametrize("problem,nx", [("ifront", (8, 6, 10)), ("ioniz_sphere", (20, 16, 12)), ("blast", (12, 8, 10))])
def test_host_problem_files_match_oracle(aa, problem, nx):
    lib = importlib.import_module("atmospheric-athena_amd.lib")
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
    o = orc.make_sim(problem, ov)
    run = aa.config.load(os.path.join(PKGDIR, "decks", "athinput." + problem), ov, problem)
    p = lib.params_from_grid(aa.config.slab(run))
    H = lib.host()
    nv = 5 + run.nscal
    U = np.zeros((nx[2] + 8, nx[1] + 8, nx[0] + 8, nv))
    dp = U.ctypes.data_as(C.POINTER(C.c_double)); pr = run.prob
    if problem == "ifront":
        assert H.aa_problem_ifront(C.byref(p), pr["n_H"], pr["cs"], dp) == 0
    elif problem == "ioniz_sphere":
        assert H.aa_problem_ioniz_sphere(C.byref(p), pr["cs"], pr["rp"], pr["mp"], pr["np"], dp) == 0
    else:
        assert H.aa_problem_blast(C.byref(p), pr["radius"], pr["pamb"], 1.0, 1.0, pr["prat"], dp) == 0
    assert np.array_equal(U[4:-4, 4:-4, 4:-4], o.active[..., :nv])
    if problem == "ioniz_sphere":
        # the pinned-cell list is Userwork_in_loop: applying it equals the oracle's userwork
        n = H.aa_ioniz_sphere_pinned(C.byref(p), None, None)
        idx = np.zeros(n, dtype=np.int64); val = np.zeros((n, 6))
        H.aa_ioniz_sphere_pinned(C.byref(p), idx.ctypes.data_as(C.POINTER(C.c_longlong)), val.ctypes.data_as(C.POINTER(C.c_double)))
        o.U[4:-4, 4:-4, 4:-4, :] *= 1.37                            # disturb, then reset
        V = o.U.copy().reshape(-1, 6)
        o.userwork()
        V[idx] = val
        assert n > 0 and np.array_equal(V.reshape(o.U.shape), o.U)
        # potential: same function as the oracle's (first step with gravity is bitwise in the gpu tests)
        assert np.isfinite(H.aa_planet_pot(1e9, 2e9, -3e9))


def test_restart_round_trip(aa, tmp_path):
    """restart.py writes what it reads (layout of src/restart.c; byte-identical rewrite of a real
    reference dump is checked in the build container, see restart.py)."""
    R = aa.restart
    rng = np.random.default_rng(3)
    nx = (6, 5, 4)
    U = rng.uniform(0.1, 2.0, (nx[2], nx[1], nx[0], 6)); ef = rng.uniform(0, 1, (nx[2] + 1, nx[1] + 1, nx[0] + 1))
    par = aa.athinput.ParTable.from_file(os.path.join(PKGDIR, "decks", "athinput.ifront"))
    p = str(tmp_path / "x.rst")
    R.write_rst(p, R.par_dump(par), 17, 1.25e8, 3.5e6, U, ef)
    r = R.read_rst(p, nx, 1, True)
    assert r["nstep"] == 17 and r["time"] == 1.25e8 and r["dt"] == 3.5e6
    assert np.array_equal(r["U"], U) and np.array_equal(r["edgeflux"], ef)
    assert r["par"].getd("ionradiation", "sigma_ph") == 6.3e-18
    R.write_rst(p, R.par_dump(par), 3, 0.5, 0.25, U[..., :5], None)          # NSCALARS=0, no ion radiation
    r = R.read_rst(p, nx, 0, False)
    assert np.array_equal(r["U"], U[..., :5]) and r["edgeflux"] is None
    with pytest.raises(ValueError):
        R.read_rst(p, nx, 1, True)


def test_restart_levels_round_trip(aa, tmp_path):
    """Multi-level dumps (static mesh refinement): the Domains' blocks follow each other under one header."""
    R = aa.restart
    rng = np.random.default_rng(5)
    nxs = [(6, 4, 8), (4, 4, 2)]
    levels = [(rng.normal(size=(nx[2], nx[1], nx[0], 6)), rng.normal(size=(nx[2] + 1, nx[1] + 1, nx[0] + 1))) for nx in nxs]
    par = aa.athinput.ParTable.from_text("<job>\nproblem_id = x\nnum_domains = 2\n")
    p = str(tmp_path / "l.rst")
    R.write_rst_levels(p, R.par_dump(par), 9, 2.5, 0.125, levels)
    r = R.read_rst_levels(p, nxs, 1, True)
    assert (r["nstep"], r["time"], r["dt"]) == (9, 2.5, 0.125) and r["par"].geti("job", "num_domains") == 2
    for (U, ef), (U2, ef2) in zip(levels, r["levels"]):
        assert np.array_equal(U, U2) and np.array_equal(ef, ef2)
    # a single-level file is the one-level case of the same layout
    R.write_rst(p, R.par_dump(par), 9, 2.5, 0.125, levels[0][0], levels[0][1])
    r1 = R.read_rst_levels(p, nxs[:1], 1, True)
    assert np.array_equal(r1["levels"][0][0], levels[0][0])


def test_bench_helpers():
    """bench.py's host-side bookkeeping: every kernel stage the library can report has a byte figure and a phase, the
    dominant kernel is the argmax of total time, the MPI baseline splits ranks over x2 and x3 only."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.kernel_class("correct_all") == "hydro" and bench.kernel_class("sweep_x3") == "hydro" and bench.kernel_class("vl_predict") == "hydro"
    assert bench.kernel_class("ion_pass") == "subcycle" and bench.kernel_class("ion_pass_begin") == "subcycle" and bench.kernel_class("ray_sweep_rates") == "subcycle"
    assert bench.kernel_class("ion_begin") == "ion_step" and bench.kernel_class("new_dt") == "other" and bench.kernel_class("bvals_mhd") == "other"
    # the stage names api.hip / slabs.hip / smr.hip hand to the profiler
    import re
    names = set()
    for f in ("api.hip", "slabs.hip", "smr.hip"):
        names |= set(re.findall(r'Scope \w+\(\w+, "(\w+)"\)', open(os.path.join(ROOT, "atmospheric-athena_amd", "csrc", f)).read()))
    names |= {"ion_pass", "ion_pass_last", "ion_pass_begin", "ion_pass_first"}          # chosen at run time in aa_ion_pass
    unknown = {n for n in names if n not in bench.KERNEL_BYTES and not n.startswith("halo_") and not n.startswith("smr_")
               and n not in ("restrict", "flux_correct", "prolongate", "ion_restrict", "ionflux_prolong")}
    assert not unknown, unknown
    prof = {"correct_all": (500.0, 20), "ion_pass": (900.0, 300), "bvals_mhd": (5000.0, 40), "ion_update": (100.0, 5)}
    assert bench.dominant_kernel(prof) == "ion_pass"                      # bvals_mhd has no per-zone byte figure
    assert bench.KERNEL_BYTES["ion_update"] > 0 and bench.KERNEL_BYTES["ion_rates"] > 0     # (round 1 lost these two to a trailing comment)
    assert bench._rank_grid(16, 128) == (4, 4) and bench._rank_grid(8, 128) == (2, 4) and bench._rank_grid(1, 128) == (1, 1)
    p2, p3 = bench._rank_grid(12, 96)
    assert p2 * p3 <= 12 and 96 % p2 == 0 and 96 % p3 == 0


def test_bench_line_from_a_window_without_a_gpu(aa):
    """bench.analyse(): the JSON object of one measurement window, fed with made-up timings (no GPU): the contract's keys, the
    roofline on SURVEY 8(d)'s unit (hydro chain on 96 B in a quiet window, radiation sub-cycle on 64 B in a burst window), the
    dominant kernel's own figure beside it."""
    import sys
    import types
    sys.path.insert(0, ROOT)
    import bench
    run = aa.config.load(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput.ioniz_sphere"),
                         ["domain1/Nx1=512", "domain1/Nx2=512", "domain1/Nx3=512"], "ioniz_sphere")
    a = types.SimpleNamespace(problem="ioniz_sphere", integrator="ctu", order=2, ionized_slab=False, strict=False)
    c = types.SimpleNamespace(a=a, world=1, multi=False)
    steps, z = 20, 512 ** 3

    def window(prof, niter, ms, spinup):
        return {"run": run, "nx": 512, "nx2": 512, "nx3": 512, "p2": 1, "strong": False, "elapsed": ms * 1e-3 * steps, "steps": steps, "warmup": 5,
                "prof": prof, "nsync": 2 * steps, "hist0": [1.0, 2.0], "hist1": [1.0, 2.0], "spin_log": [4] * 10, "t_spin": 1.0, "t_setup": 1.0,
                "niter": niter, "dt": 5.4, "time": 900.0, "hbm": 9e10, "spinup": spinup, "zones": z, "zones_gpu": z, "nslab": 1}

    quiet = {"correct_all": (20.0 * steps, steps), "flux2_update": (14.0 * steps, steps), "sweep_x1": (5.0 * steps, steps),
             "sweep_x2": (4.5 * steps, steps), "ion_pass_begin": (3.0 * steps, steps), "bvals_mhd": (0.5 * steps, 6 * steps)}
    d = bench.analyse(c, window(quiet, [1] * steps, 48.0, "auto"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "phases", "step_roofline", "state_check"):
        assert k in d, k
    assert abs(d["value"] - z / 48.0e-3) < 1e-3 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["kernel"] == "chain(correct_all+flux2_update+sweep_x1+sweep_x2)" and r["bytes_per_launch"] == 96 * z
    assert abs(r["avg_launch_ms"] - 43.5) < 1e-9 and abs(r["frac"] - 96 * z / 43.5e-3 / 1e9 / 8000.0) < 1e-12
    assert r["dominant_kernel"] == "correct_all" and r["kernel_own_bytes_per_zone"] == bench.CORRECT_ALL_X3_BYTES      # no sweep_x3 in the profile
    assert d["step_roofline"]["bytes_per_cell_update"] == 160.0 and d["state_check"]["ok"] is True
    burst = dict(quiet); burst["ion_pass"] = (2.65 * 47 * steps, 47 * steps)
    b = bench.analyse(c, window(burst, [47] * steps, 175.0, "burst"))
    rb = b["roofline"]
    assert rb["kernel"] == "chain(ion_pass+ion_pass_begin)" and rb["bytes_per_launch"] == 64 * z and "64 B" in rb["bytes_basis"]
    assert abs(b["phases"]["subcycle"]["full_pass_ms"] - 2.65) < 1e-9 and 0.39 < b["phases"]["subcycle"]["full_pass_frac_hbm"] < 0.42


def test_boundary_docs_match_the_shim():
    """The shim's defaults and the C header are the contract a maintainer of the reference reads: INTEGRATION.md's table of
    environment knobs must state the default the code has, and list every value the code accepts."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = open(os.path.join(root, "atmospheric-athena_amd", "host", "athena_shim.c")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    hdr = open(os.path.join(root, "include", "athena_amd.h")).read()
    # the default: no AA_COHERENCE in the environment => neither auto nor learn => step
    assert re.search(r'automode = \(env && strcmp\(env, "auto"\) == 0\)', shim)
    assert re.search(r'learn = automode \|\| \(env && strcmp\(env, "learn"\) == 0\)', shim)
    row = next(ln for ln in doc.splitlines() if ln.startswith("| `AA_COHERENCE`"))
    assert row.split("|")[2].strip() == "`step`"
    for mode in re.findall(r'strcmp\(env, "(\w+)"\) != 0', shim):          # every accepted value is documented
        assert f"`{mode}`" in row, mode
    m = re.search(r'reval_every = r \? atoi\(r\) : (\d+)', shim)
    row = next(ln for ln in doc.splitlines() if ln.startswith("| `AA_REVALIDATE_EVERY`"))
    assert row.split("|")[2].strip() == m.group(1)
    assert "dir = -1 (rays along +x1) or -2" in hdr and "dir must be -1" not in hdr
    assert "nslab > 1" in hdr and "AA_NGPU" in doc


def test_committed_traffic_profile_belongs_to_the_kernel_sources():
    """bench.py quotes roofline.traffic from profiles/r04_traffic.json only while the kernel sources have the fingerprint recorded in
    it; a kernel change without a new `profiles/prof_r04.sh` run must not pass unnoticed."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    for f in ("r04_traffic.json", "r04_burst_traffic.json"):
        tj = json.load(open(os.path.join(ROOT, "profiles", f)))
        assert tj["source_fingerprint"] == bench.source_fingerprint(), f"profiles/{f} was taken on other kernel sources: re-run profiles/prof_r04.sh"
        assert tj["workload"].startswith("ioniz_sphere 512x512x512")


def test_bench_helpers_without_a_gpu():
    """bench.py's CPU-side pieces: the rank grid of the CPU leg never cuts x1 and uses what divides; the kernel sources' fingerprint is
    a function of the files' bytes."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench._rank_grid(16, 128) == (4, 4) and bench._rank_grid(1, 128) == (1, 1)
    p2, p3 = bench._rank_grid(128, 256)
    assert p2 * p3 <= 128 and 256 % p2 == 0 and 256 % p3 == 0 and p2 * p3 >= 64
    fp = bench.source_fingerprint()
    assert len(fp) == 16 and fp == bench.source_fingerprint()
    assert bench.kernel_class("correct_all") == "hydro" and bench.kernel_class("ion_pass") == "subcycle" and bench.kernel_class("bvals_mhd") == "other"
