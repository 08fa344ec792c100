"""TEST INFRASTRUCTURE: run the reference's own MPI build (oracle/_ref/athena_ioniz_sphere_mpi, the unmodified sources
compiled by oracle/Makefile.ref; it travels to the GPU box as a prebuilt binary) on the host cores and reassemble the
full-precision restart dumps its ranks write (restart.c:531-770; one file per rank under id<r>/, main.c:227-232).
Used by the -m gpu parity tests as the checker, never by the product."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import read_rst      # noqa: E402  (the .rst parser only; nothing of /root/reference is touched)

MPIEXEC = "/opt/conda/bin/mpiexec"
EXE = os.path.join(ROOT, "oracle", "_ref", "athena_ioniz_sphere_mpi")
DECK = os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput.ioniz_sphere")


def available():
    return os.path.exists(EXE) and os.path.exists(MPIEXEC)


def rank_grid(cores, nx):
    """x2 x x3 split (never x1: the rays travel along x1), as many ranks as the cores allow"""
    best = (1, 1)
    for p2 in range(1, cores + 1):
        for p3 in range(p2, cores + 1):
            if p2 * p3 <= cores and nx[1] % p2 == 0 and nx[2] % p3 == 0 and (p2 * p3, p2) > (best[0] * best[1], best[0]):
                best = (p2, p3)
    return best


def run(nx, nlim, overrides=(), cores=None, exe=None, problem="ioniz_sphere", grid=None, env_extra=None):
    """-> dict(U [Nx3][Nx2][Nx1][6] of the whole Domain after `nlim` steps, niter per step, time, dt, ranks, stderr).
    exe: another MPI executable of oracle/_ref (e.g. the drop-in athena_<cfg>_mpi_amd); grid = (NGrid_x2, NGrid_x3)."""
    exe = exe or EXE
    ion = problem != "blast"
    deck0 = os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput." + problem)
    cores = cores or min(len(os.sched_getaffinity(0)), 16)
    p2, p3 = grid if grid else rank_grid(cores, nx)
    tmp = tempfile.mkdtemp(prefix="refmpi_")
    try:
        txt = open(deck0).read().replace("<domain1>", f"<domain1>\nNGrid_x1 = 1\nNGrid_x2 = {p2}\nNGrid_x3 = {p3}", 1)
        txt = re.sub(r"(?m)^maxout\s*=.*$", "maxout = 1", txt, count=1) + "\n<output1>\nout_fmt = rst\ndt = 1e300\n"
        deck = os.path.join(tmp, "athinput")
        open(deck, "w").write(txt)
        rundir = os.path.join(tmp, "run")
        env = dict(os.environ); env.update(env_extra or {})
        if exe == EXE or not exe.endswith("_amd"):       # (the drop-in carries its own search path: conda's libstdc++ must not come first)
            env["LD_LIBRARY_PATH"] = "/opt/conda/lib:" + env.get("LD_LIBRARY_PATH", "")
        args = [MPIEXEC, "-n", str(p2 * p3), exe, "-i", deck, "-d", rundir] + [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)] \
            + [f"time/nlim={nlim}"] + list(overrides)
        pr = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, timeout=1800, env=env)
        if pr.returncode != 0:
            raise RuntimeError(f"reference MPI run failed rc={pr.returncode}: {pr.stderr[-500:]}")
        niter = [int(x) for x in re.findall(r"Radiation done in (\d+) iterations", pr.stderr)][::p2 * p3]
        l2, l3 = nx[1] // p2, nx[2] // p3
        U = np.zeros((nx[2], nx[1], nx[0], 6))
        time = dt = None
        for r in range(p2 * p3):
            d = os.path.join(rundir, f"id{r}")
            f = sorted(x for x in os.listdir(d) if x.endswith(".rst"))[-1]
            g = read_rst(os.path.join(d, f), (nx[0], l2, l3), 1 if ion else 0, ion)
            assert g["nstep"] == nlim
            j, k = r % p2, r // p2                  # ranks are dealt x1-fastest, then x2, then x3 (init_mesh.c:589-596)
            U[k * l3:(k + 1) * l3, j * l2:(j + 1) * l2] = g["U"]
            time, dt = g["time"], g["dt"]
        return dict(U=U, niter=niter, time=time, dt=dt, ranks=p2 * p3, stderr=pr.stderr)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_smr(problem, overrides, nlim, exe, ngrid3, nxs):
    """MPI + SMR (the reference's README.rst:25 configuration): every Domain of the deck cut into `ngrid3` Grids along x3, one per
    rank.  nxs: zones (Nx1, Nx2, Nx3) of every Domain, root first.  -> per level the reassembled U (and EdgeFlux rows of rank 0), the
    sub-cycle counts printed by all ranks (a multiset: the ranks' lines interleave), time, dt."""
    from collections import Counter
    from make_golden import read_rst_levels
    ion = problem != "blast"
    deck0 = os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput." + problem)
    tmp = tempfile.mkdtemp(prefix="refmpismr_")
    try:
        txt = open(deck0).read()
        for n in range(1, len(nxs) + 1):
            txt = txt.replace(f"<domain{n}>", f"<domain{n}>\nNGrid_x1 = 1\nNGrid_x2 = 1\nNGrid_x3 = {ngrid3}", 1)
        txt = re.sub(r"(?m)^maxout\s*=.*$", "maxout = 1", txt, count=1) + "\n<output1>\nout_fmt = rst\ndt = 1e300\n"
        deck = os.path.join(tmp, "athinput")
        open(deck, "w").write(txt)
        rundir = os.path.join(tmp, "run")
        env = dict(os.environ)
        if not exe.endswith("_amd"):
            env["LD_LIBRARY_PATH"] = "/opt/conda/lib:" + env.get("LD_LIBRARY_PATH", "")
        args = [MPIEXEC, "-n", str(ngrid3), exe, "-i", deck, "-d", rundir, f"job/num_domains={len(nxs)}", f"time/nlim={nlim}"] + list(overrides)
        pr = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, timeout=1800, env=env)
        if pr.returncode != 0:
            raise RuntimeError(f"MPI + SMR run failed rc={pr.returncode}: {pr.stderr[-1500:]}")
        niter = Counter(int(x) for x in re.findall(r"Radiation done in (\d+) iterations", pr.stderr))
        U = [np.zeros((nx[2], nx[1], nx[0], 6)) for nx in nxs]
        time = dt = None
        for r in range(ngrid3):
            d = os.path.join(rundir, f"id{r}")
            f = sorted(x for x in os.listdir(d) if x.endswith(".rst"))[-1]
            loc = [(nx[0], nx[1], nx[2] // ngrid3) for nx in nxs]
            g = read_rst_levels(os.path.join(d, f), loc, 1 if ion else 0, ion)
            assert g["nstep"] == nlim
            for l, (Ul, _ef) in enumerate(g["levels"]):
                n3 = loc[l][2]
                U[l][r * n3:(r + 1) * n3] = Ul
            time, dt = g["time"], g["dt"]
        return dict(U=U, niter=niter, time=time, dt=dt, stderr=pr.stderr)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
