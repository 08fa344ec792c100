"""One-off measurement (GPU box, needs oracle/_ref built here and shipped with the snapshot): the reference's own driver
linked to the GPU library vs the all-CPU reference over tens of steps.  Results are quoted in DESIGN.md section 6.
usage: python tests/tools/longrun.py"""
import os, sys, subprocess, tempfile, shutil, re
import numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import read_rst
REFBIN = os.path.join(ROOT, "oracle", "_ref")
def run(exe, problem, nx, nlim, extra=()):
    tmp = tempfile.mkdtemp(prefix="long_")
    deck = os.path.join(tmp, "athinput")
    text = open(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput." + problem)).read()
    text = text.replace("maxout      = 0", "maxout      = 1") + "\n<output1>\nout_fmt = rst\ndt = 1e300\n"
    open(deck, "w").write(text)
    pr = subprocess.run([os.path.join(REFBIN, exe), "-i", deck, "-d", os.path.join(tmp, "run"),
                         f"domain1/Nx1={nx[0]}", f"domain1/Nx2={nx[1]}", f"domain1/Nx3={nx[2]}", f"time/nlim={nlim}"] + list(extra),
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp)
    assert pr.returncode == 0, pr.stderr[-2000:]
    rsts = sorted(f for f in os.listdir(os.path.join(tmp, "run")) if f.endswith(".rst"))
    ion = problem != "blast"
    out = read_rst(os.path.join(tmp, "run", rsts[-1]), nx, 1 if ion else 0, ion)
    its = [int(x) for x in re.findall(r"Radiation done in (\d+) iterations", pr.stderr)]
    shutil.rmtree(tmp)
    return out, its
for problem, nx, nlims, extra in (("ioniz_sphere", (64, 64, 64), (10, 30, 60), ()), ("ifront", (64, 32, 32), (10, 30, 60), ()), ("blast", (48, 48, 48), (20, 60), ())):
    for nlim in nlims:
        ref, itr = run(f"athena_{problem}", problem, nx, nlim, extra)
        gpu, itg = run(f"athena_{problem}_amd", problem, nx, nlim, extra)
        nv = 5 if problem == "blast" else 6
        a, b = gpu["U"][..., :nv], ref["U"][..., :nv]
        scale = np.nanmax(np.abs(b), axis=(0, 1, 2)); scale[scale == 0] = 1
        err = np.nanmax(np.abs(a - b), axis=(0, 1, 2)) / scale
        rel_d = np.nanmax(np.abs(a[..., 0] - b[..., 0]) / np.abs(b[..., 0]))
        relx = np.nanmax(np.abs(a[..., 5] / a[..., 0] - b[..., 5] / b[..., 0])) if nv == 6 else 0.0
        print(problem, nx, "steps", nlim, "iters equal", itr == itg, "t", ref["time"], gpu["time"] / ref["time"] - 1,
              "max err/field-max", ["%.1e" % e for e in err], "max pointwise rel d %.1e" % rel_d, "max |d(neutral fraction)| %.1e" % relx,
              "nan", int(np.isnan(b).sum()), flush=True)
