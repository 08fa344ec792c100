"""Drop-in proof on the GPU box: the REFERENCE'S OWN driver (main.o, init_mesh.o, par.o, outputs,
restart writer and the unmodified problem file, linked by oracle/Makefile.ref `dropin`) runs
with the product's C shim + HIP library in place of its integrators / reconstruction / Riemann
solvers / ion-radiation module / bvals_mhd / new_dt, and its restart dump is compared with the
one written by the all-CPU reference executable on the same deck."""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
REFBIN = os.path.join(ROOT, "oracle", "_ref")


def run(exe, problem, nx, nlim, env_extra=None):
    problem = problem.replace("_vl", "").replace("_ppm", "").replace("_noh", "")
    from make_golden import read_rst
    tmp = tempfile.mkdtemp(prefix="dropin_")
    deck = os.path.join(tmp, "athinput")
    text = open(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput." + problem)).read()
    text = text.replace("maxout      = 0", "maxout      = 1") + "\n<output1>\nout_fmt = rst\ndt = 1e300\n"
    open(deck, "w").write(text)
    env = dict(os.environ); env.update(env_extra or {})
    pr = subprocess.run([os.path.join(REFBIN, exe), "-i", deck, "-d", os.path.join(tmp, "run"),
                         f"domain1/Nx1={nx[0]}", f"domain1/Nx2={nx[1]}", f"domain1/Nx3={nx[2]}", f"time/nlim={nlim}"],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, env=env, timeout=600)
    assert pr.returncode == 0, pr.stdout[-1500:] + pr.stderr[-1500:]
    rsts = sorted(f for f in os.listdir(os.path.join(tmp, "run")) if f.endswith(".rst"))
    ion = not problem.startswith("blast")
    out = read_rst(os.path.join(tmp, "run", rsts[-1]), nx, 1 if ion else 0, ion)
    shutil.rmtree(tmp)
    return out, pr.stderr


@pytest.mark.parametrize("problem,nx,nlim,env", [
    ("blast", (24, 16, 20), 4, {}),
    ("ifront", (16, 8, 8), 4, {}),
    ("ioniz_sphere", (32, 32, 32), 3, {}),
    ("ioniz_sphere", (32, 32, 32), 3, {"AA_COHERENCE": "learn"}),
    ("blast_vl", (24, 16, 20), 4, {}),                  # --with-integrator=vl builds of the reference
    ("ioniz_sphere_vl", (32, 32, 32), 3, {}),
    ("blast_ppm", (24, 16, 20), 4, {}),                 # --with-order=3 builds of the reference
    ("ioniz_sphere_ppm", (32, 32, 32), 3, {}),
    ("blast_noh", (24, 16, 20), 4, {}),                 # builds WITHOUT --enable-h-correction (the configure default)
    ("ioniz_sphere_noh", (32, 32, 32), 3, {}),
    ("blast_vl_ppm", (24, 16, 20), 4, {}),              # --with-integrator=vl --with-order=3
    ("ioniz_sphere_vl_ppm", (32, 32, 32), 3, {}),
])
def test_reference_driver_on_gpu_library(problem, nx, nlim, env):
    if not os.path.exists(os.path.join(REFBIN, f"athena_{problem}_amd")):
        pytest.skip("oracle/_ref drop-in executables not built (make -C oracle ref)")
    ref, _ = run(f"athena_{problem}", problem, nx, nlim)
    gpu, err = run(f"athena_{problem}_amd", problem, nx, nlim, env)
    assert "[athena_amd] Grid" in err
    assert gpu["nstep"] == ref["nstep"] == nlim
    assert abs(gpu["time"] / ref["time"] - 1) < 1e-9 and abs(gpu["dt"] / ref["dt"] - 1) < 1e-9
    nv = 5 if problem.startswith("blast") else 6
    a, b = gpu["U"][..., :nv], ref["U"][..., :nv]
    scale = np.abs(b).max(axis=(0, 1, 2))
    diff = np.abs(a - b).max(axis=(0, 1, 2))
    assert np.all(diff[scale == 0] == 0)
    err = (diff[scale > 0] / scale[scale > 0]).max()
    assert err < 1e-8, err                       # north_star bar: 1e-6 on density / ion fraction
    if not problem.startswith("blast"):
        assert np.allclose(gpu["edgeflux"], ref["edgeflux"], rtol=1e-8, atol=1e-8 * np.abs(ref["edgeflux"]).max())


def test_user_boundary_functions_enrolled_by_problem():
    """bvals_mhd_fun (bvals_mhd.c:917): a user problem file (tests/fixtures/userbc_blast.c, written
    against the reference's problem-file API) enrols its own inner-x1 and outer-x2 boundary
    functions.  The same problem object is linked once into the pure reference and once into the
    drop-in; the shim must call the user functions in the reference's order between its device
    boundary kernels."""
    if not os.path.exists(os.path.join(REFBIN, "athena_userbc_amd")):
        pytest.skip("oracle/_ref userbc executables not built (make -C oracle -f Makefile.ref userbc)")
    from make_golden import read_rst
    nx, nlim = (20, 16, 12), 5
    outs = {}
    for exe in ("athena_userbc", "athena_userbc_amd"):
        tmp = tempfile.mkdtemp(prefix="userbc_")
        deck = os.path.join(tmp, "athinput")
        text = open(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput.blast")).read()
        text = text.replace("maxout      = 0", "maxout      = 1") + "\n<output1>\nout_fmt = rst\ndt = 1e300\n"
        open(deck, "w").write(text)
        pr = subprocess.run([os.path.join(REFBIN, exe), "-i", deck, "-d", os.path.join(tmp, "run"),
                             f"domain1/Nx1={nx[0]}", f"domain1/Nx2={nx[1]}", f"domain1/Nx3={nx[2]}", f"time/nlim={nlim}",
                             "domain1/bc_ix1=2", "domain1/bc_ox1=2", "domain1/bc_ix2=1", "domain1/bc_ox2=2"],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, timeout=600)
        assert pr.returncode == 0, pr.stdout[-1500:] + pr.stderr[-1500:]
        rsts = sorted(f for f in os.listdir(os.path.join(tmp, "run")) if f.endswith(".rst"))
        outs[exe] = read_rst(os.path.join(tmp, "run", rsts[-1]), nx, 0, False)
        shutil.rmtree(tmp)
    ref, gpu = outs["athena_userbc"], outs["athena_userbc_amd"]
    assert gpu["nstep"] == ref["nstep"] == nlim
    assert abs(gpu["time"] / ref["time"] - 1) < 1e-12 and abs(gpu["dt"] / ref["dt"] - 1) < 1e-12
    a, b = gpu["U"][..., :5], ref["U"][..., :5]
    scale = np.abs(b).max(axis=(0, 1, 2)); scale[scale == 0] = 1
    assert (np.abs(a - b).max(axis=(0, 1, 2)) / scale).max() < 1e-11      # default (FMA) build, 5 steps
    # the inflow must have mattered: x1-momentum near the inner-x1 face differs from an outflow run
    assert np.abs(b[:, :, 0, 1]).max() > 0


def test_cooling_function_enrolled_by_problem():
    """globals.h:25 CoolingFunc: a user problem file (tests/fixtures/cool_pattern.c) enrols the cooling function the reference
    ships (KoyInut, microphysics/cool.c:48); the shim recognises it and the device integrator carries its terms
    (aa_set_cooling).  The reference's own main() on the GPU library against the all-CPU reference run of the same problem
    object (tests/golden/coolpat_c1_*)."""
    if not os.path.exists(os.path.join(REFBIN, "athena_coolpat_amd")):
        pytest.skip("oracle/_ref coolpat executables not built (make -C oracle -f Makefile.ref coolpat)")
    from make_golden import read_rst
    gz = np.load(os.path.join(ROOT, "tests", "golden", "coolpat_c1_16x12x10_n4.npz"))
    nx = tuple(int(x) for x in gz["nx"]); nlim = int(gz["nstep"])
    kv = dict(str(o).split("=") for o in gz["overrides"])
    tmp = tempfile.mkdtemp(prefix="coolpat_")
    deck = os.path.join(tmp, "athinput")
    text = open(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput.blast")).read()
    text = text.replace("maxout      = 0", "maxout      = 1") + "\n<output1>\nout_fmt = rst\ndt = 1e300\n"
    text = text.replace("<problem>", "<problem>\n" + "".join(f"{k} = {kv[k]}\n" for k in ("n0", "T0", "v0")) + "cool = 1")
    open(deck, "w").write(text)
    over = [f"{k}={v}" for k, v in kv.items() if "/" in k]
    pr = subprocess.run([os.path.join(REFBIN, "athena_coolpat_amd"), "-i", deck, "-d", os.path.join(tmp, "run"),
                         f"domain1/Nx1={nx[0]}", f"domain1/Nx2={nx[1]}", f"domain1/Nx3={nx[2]}", f"time/nlim={nlim}"] + over,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, timeout=600)
    assert pr.returncode == 0, pr.stdout[-1500:] + pr.stderr[-1500:]
    rsts = sorted(f for f in os.listdir(os.path.join(tmp, "run")) if f.endswith(".rst"))
    gpu = read_rst(os.path.join(tmp, "run", rsts[-1]), nx, 0, False)
    shutil.rmtree(tmp)
    assert gpu["nstep"] == nlim
    assert abs(gpu["time"] / float(gz["time"]) - 1) < 1e-11 and abs(gpu["dt"] / float(gz["dt"]) - 1) < 1e-11
    a, b = gpu["U"][..., :5], gz["U"][..., :5]
    scale = np.abs(b).max(axis=(0, 1, 2)); scale[scale == 0] = 1
    assert (np.abs(a - b).max(axis=(0, 1, 2)) / scale).max() < 1e-10


@pytest.mark.parametrize("fixture,cfg,env", [
    ("smr_blast_3lev_s6", "blast_smr", {}),
    ("smr_blast_3lev_edge_s8", "blast_smr", {}),
    ("smr_ioniz_sphere_2lev_s4", "ioniz_sphere_smr", {}),
    ("smr_ioniz_sphere_2lev_s4", "ioniz_sphere_smr", {"AA_COHERENCE": "learn"}),
    ("smr_blast_2dom_s6", "blast_smr", {}),                  # two Domains on level 1 (MeshS.Domain[nl][nd])
    ("smr_ioniz_sphere_2dom_s3", "ioniz_sphere_smr", {}),
])
def test_reference_smr_driver_on_gpu_library(fixture, cfg, env):
    """The reference's --enable-smr driver (main.o, init_mesh.o, init_grid.o with its overlap tables,
    restart writer, unmodified problem file) linked against the shim compiled with -DAA_SMR: SMR_init,
    RestrictCorrect, Prolongate and ionradRestrictCorrect come from the HIP library.  Every level of
    the restart dump is compared with the all-CPU reference executable's."""
    if not os.path.exists(os.path.join(REFBIN, f"athena_{cfg}_amd")):
        pytest.skip("oracle/_ref SMR drop-in executables not built (make -C oracle ref)")
    from make_golden import read_rst_levels
    g = np.load(os.path.join(HERE, "golden", fixture + ".npz"))
    over = [str(o) for o in g["overrides"]]
    nlim = int(g["nstep"])
    problem = "blast" if "blast" in cfg else "ioniz_sphere"
    ion = problem != "blast"
    nlev = int(g["nlevels"])
    nxs = [tuple(int(next(o for o in over if o.startswith(f"domain{n}/Nx{d}=")).split("=")[1]) for d in (1, 2, 3))
           for n in range(1, nlev + 1)]
    outs = {}
    for exe, e in ((f"athena_{cfg}", {}), (f"athena_{cfg}_amd", env)):
        tmp = tempfile.mkdtemp(prefix="dropin_smr_")
        deck = os.path.join(tmp, "athinput")
        text = open(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput." + problem)).read()
        text = text.replace("maxout      = 0", "maxout      = 1") + "\n<output1>\nout_fmt = rst\ndt = 1e300\n"
        open(deck, "w").write(text)
        envp = dict(os.environ); envp.update(e)
        pr = subprocess.run([os.path.join(REFBIN, exe), "-i", deck, "-d", os.path.join(tmp, "run"), f"time/nlim={nlim}"] + over,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, env=envp, timeout=600)
        assert pr.returncode == 0, pr.stdout[-1500:] + pr.stderr[-1500:]
        rsts = sorted(f for f in os.listdir(os.path.join(tmp, "run")) if f.endswith(".rst"))
        outs[exe] = (read_rst_levels(os.path.join(tmp, "run", rsts[0]), nxs, 1 if ion else 0, ion),
                     read_rst_levels(os.path.join(tmp, "run", rsts[-1]), nxs, 1 if ion else 0, ion), pr.stderr)
        shutil.rmtree(tmp)
    (ref0, ref, _), (gpu0, gpu, err) = outs[f"athena_{cfg}"], outs[f"athena_{cfg}_amd"]
    assert "(level 1) on HIP device" in err
    assert gpu["nstep"] == ref["nstep"] == nlim
    assert abs(gpu["time"] / ref["time"] - 1) < 1e-9 and abs(gpu["dt"] / ref["dt"] - 1) < 1e-9
    nv = 6 if ion else 5
    for l in range(nlev):
        # the dump written before the first cycle already holds the restricted parent zones
        assert np.array_equal(gpu0["levels"][l][0][..., :nv], ref0["levels"][l][0][..., :nv]), f"initial dump, level {l}"
        a, b = gpu["levels"][l][0][..., :nv], ref["levels"][l][0][..., :nv]
        scale = np.abs(b).max(axis=(0, 1, 2)); scale[scale == 0] = 1
        e = (np.abs(a - b).max(axis=(0, 1, 2)) / scale).max()
        assert e < 1e-8, (l, e)                  # north_star bar: 1e-6


def test_restart_of_the_reference_driver_on_gpu_library():
    """`athena -r file.rst` through the reference's own restart_grids() (restart.c:52-456) and main():
    the host block it fills reaches the device on the shim's first call, time / dt / nstep travel in
    MeshS.  A run restarted from its own step-2 dump must arrive at step 4 exactly where the straight
    run does (the same library computes both: bit for bit)."""
    if not os.path.exists(os.path.join(REFBIN, "athena_blast_amd")):
        pytest.skip("oracle/_ref drop-in executables not built (make -C oracle ref)")
    from make_golden import read_rst
    nx = (20, 16, 12)
    tmp = tempfile.mkdtemp(prefix="dropin_rst_")
    deck = os.path.join(tmp, "athinput")
    text = open(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput.blast")).read()
    text = text.replace("maxout      = 0", "maxout      = 1") + "\n<output1>\nout_fmt = rst\ndt = 1e300\n"
    open(deck, "w").write(text)
    size = [f"domain1/Nx1={nx[0]}", f"domain1/Nx2={nx[1]}", f"domain1/Nx3={nx[2]}"]
    exe = os.path.join(REFBIN, "athena_blast_amd")

    def go(args, rundir):
        pr = subprocess.run([exe] + args + ["-d", os.path.join(tmp, rundir)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            text=True, cwd=tmp, timeout=600)
        assert pr.returncode == 0, pr.stdout[-1500:] + pr.stderr[-1500:]
        d = os.path.join(tmp, rundir)
        return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".rst"))

    straight = go(["-i", deck, "time/nlim=4"] + size, "a")
    half = go(["-i", deck, "time/nlim=2"] + size, "b")
    resumed = go(["-r", half[-1], "time/nlim=4"], "c")
    a = read_rst(straight[-1], nx, 0, False); c = read_rst(resumed[-1], nx, 0, False)
    shutil.rmtree(tmp)
    assert a["nstep"] == c["nstep"] == 4 and a["time"] == c["time"] and a["dt"] == c["dt"]
    assert np.array_equal(a["U"][..., :5], c["U"][..., :5])


@pytest.mark.parametrize("problem,nx,nlim,outdt", [("blast", (24, 16, 20), 9, 0.004), ("ioniz_sphere", (32, 32, 32), 7, 3.0e-5)])
def test_auto_coherence_refreshes_the_host_exactly_when_main_reads_it(problem, nx, nlim, outdt):
    """AA_COHERENCE=auto (opt-in; the default is `step`): after two verified steps the host block is refreshed only when an <outputN> block
    is due, the loop ends, or SIGTERM arrives.  A restart output every `outdt` of simulated time puts dumps in the
    middle of the run: every dump must equal the one of AA_COHERENCE=step (same library, host kept in the loop every
    step) bit for bit, and there must be several of them."""
    if not os.path.exists(os.path.join(REFBIN, f"athena_{problem}_amd")):
        pytest.skip("oracle/_ref drop-in executables not built (make -C oracle ref)")
    from make_golden import read_rst
    ion = problem != "blast"
    dumps = {}
    for mode in ("step", "auto"):
        tmp = tempfile.mkdtemp(prefix="dropin_auto_")
        deck = os.path.join(tmp, "athinput")
        text = open(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput." + problem)).read()
        text = text.replace("maxout      = 0", "maxout      = 1") + f"\n<output1>\nout_fmt = rst\ndt = {outdt!r}\n"
        open(deck, "w").write(text)
        env = dict(os.environ, AA_COHERENCE=mode)
        pr = subprocess.run([os.path.join(REFBIN, f"athena_{problem}_amd"), "-i", deck, "-d", os.path.join(tmp, "run"),
                             f"domain1/Nx1={nx[0]}", f"domain1/Nx2={nx[1]}", f"domain1/Nx3={nx[2]}", f"time/nlim={nlim}"],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, env=env, timeout=600)
        assert pr.returncode == 0, pr.stdout[-1500:] + pr.stderr[-1500:]
        assert f"coherence={mode}" in pr.stderr
        if mode == "auto":
            assert "re-imposed on the device from now on" in pr.stderr          # the imprint was verified and adopted
        rsts = sorted(f for f in os.listdir(os.path.join(tmp, "run")) if f.endswith(".rst"))
        dumps[mode] = [read_rst(os.path.join(tmp, "run", f), nx, 1 if ion else 0, ion) for f in rsts]
        shutil.rmtree(tmp)
    a, b = dumps["step"], dumps["auto"]
    assert len(a) == len(b) and len(a) >= 4, (len(a), len(b))                   # initial, >= 2 in the loop, final
    mid = [d["nstep"] for d in a[1:-1]]
    assert any(n > 2 for n in mid), mid                                         # at least one after auto took over
    for x, y in zip(a, b):
        assert x["nstep"] == y["nstep"] and x["time"] == y["time"] and x["dt"] == y["dt"]
        assert np.array_equal(x["U"], y["U"], equal_nan=True)
        if ion:
            assert np.array_equal(x["edgeflux"], y["edgeflux"])


def test_userwork_that_wakes_up_later():
    """A Userwork_in_loop that is a fixed imprint for the first steps and starts writing other, time-dependent values at
    step 5 (tests/fixtures/userwork_late.c: the `if (time > t0)` pattern).  The DEFAULT coherence (`step`: the host block
    takes part in every step) must follow the all-CPU reference through it.  AA_COHERENCE=auto adopts the imprint after two
    steps, and must find out at its next re-validation (every AA_REVALIDATE_EVERY steps), say so, and carry on as `step`;
    with re-validation on every step it is `step` in effect and must match the reference as well."""
    if not os.path.exists(os.path.join(REFBIN, "athena_userwork_amd")):
        pytest.skip("oracle/_ref/athena_userwork_amd not built (make -C oracle ref)")
    nx, nlim = (20, 16, 12), 12
    ref, _ = run("athena_userwork", "blast", nx, nlim)
    gpu, err = run("athena_userwork_amd", "blast", nx, nlim)
    assert "coherence=step" in err                                       # the default
    assert gpu["nstep"] == ref["nstep"] == nlim and abs(gpu["dt"] / ref["dt"] - 1) < 1e-12
    scale = np.abs(ref["U"][..., :5]).max(axis=(0, 1, 2)); scale[scale == 0] = 1
    assert (np.abs(gpu["U"][..., :5] - ref["U"][..., :5]).max(axis=(0, 1, 2)) / scale).max() < 1e-11
    # the stirred block really moved (the test would be empty otherwise)
    assert np.abs(ref["U"][-3:, -3:, -3:, 1]).min() > 0.01
    # auto, validated every step: equal to step
    every, err1 = run("athena_userwork_amd", "blast", nx, nlim, {"AA_COHERENCE": "auto", "AA_REVALIDATE_EVERY": "1"})
    assert "re-imposed on the device from now on" in err1 and "falls back to `step`" in err1
    assert np.array_equal(every["U"], gpu["U"]) and every["dt"] == gpu["dt"]
    # auto with the default interval: the wake-up at step 5 is caught by the validation of step 2 + 8 = 10 at the latest
    late, err8 = run("athena_userwork_amd", "blast", nx, nlim, {"AA_COHERENCE": "auto"})
    assert "coherence=auto" in err8 and "re-imposed on the device from now on" in err8
    assert "WARNING: Userwork_in_loop no longer writes the imprint" in err8 and "falls back to `step`" in err8
