#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the hot path on N MI355X of one node.

One "step" = one pass of the reference's main loop (main.c:519-669) on synthetic input already
resident in HBM: ion-radiation step with its sub-cycles (ionrad_3d.c:862) + ghost zones +
3-D CTU hydro step (integrate_3d_ctu.c:110) + per-step core reset + new_dt + ghost zones.
Workload at N=1: BASELINE.json configs[3] on one GPU, tst/massloss/athinput.ioniz_sphere_hires
keys at 512^3 single level (hydro + static gravity + ion radiation).

Regime.  The deck starts with dt = 4.7e-6 s, which new_dt lets double once per step; around step 19
(512^3) dt reaches the chemical time scale and the ion step takes 30-80 sub-cycles for a few dozen
steps; from about step 60 on dt sits at the CFL limit (5.4-6.5 s) and the ion step needs ONE
sub-cycle per step (a few more bursts until about step 150), for as long as the run can be followed
(profiles/r02_regime_512.txt).  The timed
region therefore starts after an untimed SPIN-UP that runs until that stationary state is reached
(--spinup auto: dt no longer doubling and the sub-cycle count unchanged over 48 consecutive steps -- bursts of
sub-cycles recur sporadically until about step 150 --, at most 320 steps; --spinup N: exactly N steps, e.g. 19 to land
in the bursts).
`value` is what the timed region gives; `phases` breaks it into the hydro chain, one radiation
sub-cycle and the rest, each with its roofline on SURVEY 8(d)'s algorithmic bytes (96 B per
cell-update, 64 B per cell and sub-cycle), so the number can be re-derived for any sub-cycle count.

N>1: one rank per GPU, x3 slabs, halo exchange and scalar reductions over RCCL.  Default = weak
scaling (every GPU holds nx^3 zones, the box grows along x3 at the same dx); --strong cuts ONE
nx^3 box into N slabs (BASELINE configs[3]: 512^3 on 8 GPUs = 64-plane slabs).

Prints ONE JSON line (rank 0).  `value` = active zones advanced per second by the whole job.
"""
import argparse
import importlib
import json
import math
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "atmospheric-athena_amd"

# Rehearsal of the N>1 code path on a ONE-GPU box (not a measurement): AA_BENCH_REHEARSAL=1 puts every rank on
# cuda:0 and uses gloo (RCCL refuses two ranks on one device); messages are staged through the host.
REHEARSAL = bool(os.environ.get("AA_BENCH_REHEARSAL"))


def init_pg(dist, torch, rank, world, local):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
    # RCCL prints its version banner on stdout when the first communicator is created: keep stdout for the
    # ONE JSON line by pointing fd 1 at stderr until the communicator exists
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    try:
        if REHEARSAL:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        t = torch.zeros(1, device="cpu" if REHEARSAL else torch.device("cuda", local))
        dist.all_reduce(t)
        if not REHEARSAL:
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(keep, 1)
        os.close(keep)


HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6  # vector FP64, public spec; hydro is bound by this, not by HBM
HYDRO_FLOP_PER_CELL = 5.0e3   # SURVEY 8(d) hand count for one CTU cell-update (+2e3 with an analytic potential: tabulated here)

# Compulsory bytes per cell of each kernel = distinct doubles it must read + write per zone
# (NVAR=6): the figures are derived in DESIGN.md section 3.
KERNEL_BYTES = {
    "sweep_x1": 8 * (6 + 6), "sweep_x2": 8 * (6 + 6), "sweep_x3": 8 * (6 + 6),
    "sweep_correct_x1": 8 * (6 + 12 + 6 + 12 + 1), "correct_x1": 8 * (6 + 12 + 12 + 1), "correct_x2": 8 * (6 + 12 + 12 + 1),
    "correct_x3": 8 * (6 + 12 + 12 + 1),
    "vl_predict": 8 * (6 + 6 + 4),                   # U in, U^{n+1/2} out, phi x4
    "vl_flux1": 8 * (6 + 18), "vl_uhalf": 8 * (6 + 18 + 6), "vl_flux2_x1": 8 * 12, "vl_flux2_x2": 8 * 12,
    "vl_flux2_x3": 8 * 12, "flux2_x1": 8 * (12 + 3 + 6), "flux2_x2": 8 * (12 + 3 + 6),
    "flux2_x3": 8 * (12 + 3 + 6), "update": 8 * (6 + 18 + 6),
    "correct_all": 8 * (6 + 18 + 4 + 36 + 3 + 1),   # U, first-pass fluxes, phi in; L/R states x3, eta x3, d^{n+1/2} out
    "flux2_update": 8 * (36 + 3 + 6 + 5 + 6),     # L/R states of 3 directions, eta x3, U in, phi x4 + d^{n+1/2}, U out
    "ray_sweep": 8 * (1 + 2),
    "ray_sweep_rates": 8 * (1 + 2) + 8 * (3 + 1),   # + d, ke, E and the int2 sign word
    "ion_rates": 8 * (5 + 1),
    "ion_update": 8 * (5 + 1 + 2 + 0.5 + 2),
    # the one-kernel sub-cycle: d, ke, E, s0, incoming flux of the previous sweep, e_init, vmax in;
    # E, s0, incoming flux out; 2-byte sign word in + out
    "ion_pass": 8 * (7 + 3) + 4,
    "ion_pass_first": 8 * (4 + 1) + 4, "ion_pass_last": 8 * (7 + 2) + 2,
    "ion_pass_begin": 8 * (6 + 4 + 1) + 2,      # the entry of the ion step on the first pass: U in; ke, max|v|, e_init, x_init, incoming flux, sign word out
    "ion_begin": 8 * (6 + 6), "ion_finish": 8 * 2,
    "bvals_mhd": 0, "new_dt": 8 * 5, "pinned_cells": 0, "ppm_slopes": 3 * 8 * (6 + 6), "no_h_correction": 8 * 3,
}
CORRECT_ALL_X3_BYTES = 8 * (6 + 12 + 4 + 36 + 3 + 1)   # k_correct_all with the x3 first pass inside (no x3 first-pass fluxes in HBM)
CORRECT_ALL_X1X3_BYTES = 8 * (6 + 6 + 4 + 36 + 3 + 1)   # ... and the x1 first pass (round 4): only the x2 first-pass fluxes are read
HYDRO_KERNELS = ("sweep_", "sweep_correct_x1", "correct_", "flux2_", "update", "vl_", "ppm_slopes", "no_h_correction")
SUBCYCLE_KERNELS = ("ray_sweep", "ray_sweep_rates", "ion_rates", "ion_update", "ion_pass", "ion_pass_first", "ion_pass_last", "ion_pass_begin",
                    "ion_pick")


def kernel_class(name):
    if name in SUBCYCLE_KERNELS:
        return "subcycle"
    if name.startswith("ion_"):
        return "ion_step"
    for h in HYDRO_KERNELS:
        if name == h or (h.endswith("_") and name.startswith(h)):
            return "hydro"
    return "other"


def dominant_kernel(prof):
    """argmax of total time over the kernels that have a per-zone byte figure (boundary kernels have none)."""
    cand = {k: v[0] for k, v in prof.items() if KERNEL_BYTES.get(k, 0) > 0 and v[1] > 0}
    return max(cand, key=cand.get) if cand else None


# ---- CPU baseline: the REAL reference, MPI build, on the host cores of this box ------------------------------
MPIEXEC = "/opt/conda/bin/mpiexec"   # MPICH 3.3.2 of the image (off PATH)


def _rank_grid(cores, nx):
    """x2 x x3 split (never x1: the rays travel along x1), as many ranks as cores allow."""
    best = (1, 1)
    for p2 in range(1, cores + 1):
        for p3 in range(p2, cores + 1):
            if p2 * p3 <= cores and nx % p2 == 0 and nx % p3 == 0 and (p2 * p3, p2) > (best[0] * best[1], best[0]):
                best = (p2, p3)
    return best


def _run_ref(exe, deck, tmp, tag, nx, nlim, ranks):
    args = [exe, "-i", deck, "-d", os.path.join(tmp, tag), f"domain1/Nx1={nx}", f"domain1/Nx2={nx}", f"domain1/Nx3={nx}",
            f"time/nlim={nlim}"]
    if ranks > 1:
        args = [MPIEXEC, "-n", str(ranks)] + args
    t0 = time.time()
    pr = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, timeout=900)
    wall = time.time() - t0
    if pr.returncode != 0:
        raise RuntimeError(f"{os.path.basename(exe)} rc={pr.returncode}: {pr.stderr[-300:]}")
    m = re.findall(r"zone-cycles/wall-second = ([0-9.eE+-]+)", pr.stdout)
    its = [int(x) for x in re.findall(r"Radiation done in (\d+) iterations", pr.stderr)][::max(1, ranks)]
    return float(m[-1]), its, wall     # the last line is the total over ranks (main.c:735)


def cpu_baseline(nx=128, nlim=12, gpus=1):
    """The reference's own MPI CPU path (oracle/_ref/athena_*_mpi: the unmodified sources built with
    MPI_PARALLEL by oracle/Makefile.ref) on the host cores of this box, on a bounded sample of the same
    deck: the coupled run, and the same deck without ion radiation, so that the cost of a hydro step and
    of a radiation sub-cycle can be stated separately (ns per zone), like the GPU's `phases`."""
    ref = os.path.join(ROOT, "oracle", "_ref")
    deck0 = os.path.join(ROOT, PKG, "decks", "athinput.ioniz_sphere")
    # The host cores that belong to the GPUs of this job: 16 per GPU (a one-GPU box's CPU share; its affinity mask shows the whole
    # host, whose other cores belong to other boxes), i.e. 16 N of an N-GPU node -- north_star's "the GPU box's host cores" --
    # on a sample scaled with them.  AA_CPU_BASELINE_CORES overrides.
    avail = len(os.sched_getaffinity(0))
    cores = min(avail, 16 * max(1, 1 if REHEARSAL else gpus))
    if os.environ.get("AA_CPU_BASELINE_CORES"):
        cores = max(1, min(avail, int(os.environ["AA_CPU_BASELINE_CORES"])))
    if cores > 32:
        nx = 256
    tmp = tempfile.mkdtemp(prefix="cpu_baseline_")
    try:
        if os.path.exists(os.path.join(ref, "athena_ioniz_sphere_mpi")) and os.path.exists(MPIEXEC):
            try:
                p2, p3 = _rank_grid(cores, nx)
                deck = os.path.join(tmp, "athinput")
                txt = open(deck0).read().replace("<domain1>", f"<domain1>\nNGrid_x1 = 1\nNGrid_x2 = {p2}\nNGrid_x3 = {p3}", 1)
                open(deck, "w").write(txt)
                zc, its, wall = _run_ref(os.path.join(ref, "athena_ioniz_sphere_mpi"), deck, tmp, "ion", nx, nlim, p2 * p3)
                out = {"value": zc, "unit": "cell-updates/s", "cores": p2 * p3, "cores_available": avail, "kind": "reference-mpi",
                       "sample": f"ioniz_sphere {nx}^3, {nlim} steps, NGrid 1x{p2}x{p3} (mpiexec -n {p2 * p3}), sub-cycles/step {its}, {wall:.1f} s wall"}
                nsub = sum(its) / max(1, len(its))
                out["nsub_mean"] = nsub
                if os.path.exists(os.path.join(ref, "athena_sphere_hydro_mpi")):
                    zh, _, wh = _run_ref(os.path.join(ref, "athena_sphere_hydro_mpi"), deck, tmp, "hyd", nx, nlim, p2 * p3)
                    out["per_hydro_step_ns_per_zone"] = 1e9 / zh
                    # (no per-sub-cycle figure: it would be the difference of two 3-s runs, < 1 % of a CPU step on this deck --
                    #  the rays end in the first optically thick zones -- i.e. below the run-to-run spread)
                    out["sample"] += f"; same deck without ion radiation {zh:.3e} zone-cycles/s ({wh:.1f} s wall)"
                return out
            except Exception as e:      # e.g. hydra cannot start on this host: fall back to one core
                sys.stderr.write(f"[bench] MPI reference baseline failed ({e}); falling back to one core\n")
        exe = os.path.join(ref, "athena_ioniz_sphere")
        if os.path.exists(exe):
            try:
                zc, its, wall = _run_ref(exe, deck0, tmp, "ser", 64, 40, 1)
                return {"value": zc, "unit": "cell-updates/s", "cores": 1, "kind": "reference",
                        "sample": f"ioniz_sphere 64^3, 40 steps, sub-cycles/step {its}, {wall:.1f} s wall",
                        "nsub_mean": sum(its) / max(1, len(its))}
            except Exception as e:
                sys.stderr.write(f"[bench] serial reference baseline failed ({e}); falling back to the CPU restatement\n")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    s = orc.make_sim("ioniz_sphere", [f"domain1/Nx{d}=64" for d in (1, 2, 3)]).start()
    t0 = time.time(); its = [s.step() for _ in range(40)]; wall = time.time() - t0
    return {"value": 64 ** 3 * 40 / wall, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": f"ioniz_sphere 64^3, 40 steps, sub-cycles/step {its}, {wall:.1f} s wall", "nsub_mean": sum(its) / 40.0}


def bench_smr(a, aa, torch, rank, world, local):
    """2-level nested mesh: every step runs the radiation step on both levels (the fine one sub-cycles
    to the time the root covered), ionradRestrictCorrect, both integrators, RestrictCorrect, new_dt over
    both levels and Prolongate (main.c:519-669 with STATIC_MESH_REFINEMENT).  One GPU: the whole Mesh in
    one aa_mesh.  N GPUs (weak scaling): root nx x nx x nx*N with level 1 over its central half, every
    level cut at the same root planes (driver.MeshDriver)."""
    nx = a.nx if a.nx != 512 else 320
    if nx % 4:
        sys.exit("--smr: nx must be a multiple of 4")
    deck = os.path.join(ROOT, PKG, "decks", "athinput." + a.problem)
    par = aa.athinput.ParTable.from_file(deck)
    x3min, x3max = par.getd("domain1", "x3min"), par.getd("domain1", "x3max")
    if a.smr_deck:
        if world != 1:
            sys.exit("--smr-deck is a one-GPU measurement")
        par.cmdline([f"job/num_domains={a.smr_levels}"])
    else:
        par.cmdline(["job/num_domains=2", f"domain1/Nx1={nx}", f"domain1/Nx2={nx}", f"domain1/Nx3={nx * world}",
                     f"domain1/x3max={x3min + (x3max - x3min) * world!r}",
                     f"domain2/Nx1={nx}", f"domain2/Nx2={nx}", f"domain2/Nx3={nx * world}",
                     f"domain2/iDisp={nx // 2}", f"domain2/jDisp={nx // 2}", f"domain2/kDisp={nx * world // 2}"])
    run = aa.config.from_par(par, a.problem)
    run.integrator = a.integrator
    t_setup = time.time()
    multi = world > 1 or bool(os.environ.get("AA_FORCE_DISTRIBUTED"))
    if multi:
        import torch.distributed as dist
        init_pg(dist, torch, rank, world, local)
        m = importlib.import_module(PKG + ".driver").MeshDriver(par, run, None, rank, world, local).start()
        grids = m.eng.lev
    else:
        m = importlib.import_module(PKG + ".lib").Mesh(aa.config.levels(par, run), local).start()
        grids = m.lev
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    trace = []
    for _ in range(a.warmup):
        m.step()
    for g in grids:
        g.profile_reset(); g.profile_enable(not a.no_kernel_times)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trace.append(m.step())
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    zones = 2 * nx ** 3 * world
    lv = aa.config.levels(par, run)
    if a.smr_deck:
        zones = sum(g.Nx[0] * g.Nx[1] * g.Nx[2] for g in lv)
    if rank == 0:
        out = {"metric": "cell-updates/sec (hydro+ion-rad step)", "value": zones * a.steps / elapsed, "unit": "cell-updates/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic (deck values on a nested mesh, generated in place)",
               "config": {"workload": f"{a.problem} {len(lv)}-level SMR: " + " + ".join(f"level {g.level} {g.Nx[0]}x{g.Nx[1]}x{g.Nx[2]}" for g in lv)
                                      + " (zones of all levels counted, as the reference's zone-cycles do)",
                          "zones": zones, "partition": (f"x3 cuts {list(m.cfg.cuts)} shared by both levels" if multi else "one aa_mesh"),
                          "subcycle_trace_per_level": trace, "final_dt": m.dt,
                          "hbm_resident_GB_rank0": sum(g.device_bytes() for g in grids) / 1e9, "setup_s": t_setup}}
        if not a.no_kernel_times:
            prof, dom = {}, None
            for l, g in enumerate(grids):
                for k, (ms, n) in g.profile().items():
                    prof[f"L{l}.{k}"] = ms / a.steps
                    if n and KERNEL_BYTES.get(k, 0) > 0 and (dom is None or ms > dom[2]):   # boundary / inter-level kernels have no per-zone figure
                        dom = (l, k, ms, n)
            out["kernel_ms_per_step_rank0"] = dict(sorted(prof.items(), key=lambda kv: -kv[1]))
            if dom:
                # the dominant kernel (of whichever level): compulsory bytes of one launch on that level's slab /
                # its mean duration
                l, k, ms, n = dom
                cfg = grids[l].cfg if hasattr(grids[l], "cfg") else lv[l]
                ncell = cfg.Nx[0] * cfg.Nx[1] * cfg.Nx[2]
                kb = KERNEL_BYTES.get(k, 0)
                if k == "correct_all" and f"L{l}.sweep_x3" not in prof:
                    kb = CORRECT_ALL_X3_BYTES if f"L{l}.sweep_x1" in prof else CORRECT_ALL_X1X3_BYTES
                bpl = kb * ncell
                ach = bpl / (ms / n * 1e-3) / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": f"L{l}.{k}", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": ach / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": bpl, "avg_launch_ms": ms / n}
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    else:
        m.close()


def spin_up(drv, mode, log):
    """Untimed steps that carry the deck from its initial state into the regime to be timed (module docstring):
    'auto' the stationary one, 'burst' the first burst of sub-cycles after the doubling phase, N exactly N steps."""
    if mode == "burst":
        while len(log) < 96:
            n = drv.step()
            log.append(n)
            if len(log) >= 4 and n >= 8:          # (the doubling phase takes 4 per step; 512^3: 30-81 from step 19 on)
                break
        return
    if mode != "auto":
        for _ in range(int(mode)):
            log.append(drv.step())
        return
    quiet = 0
    while len(log) < 320 and quiet < 48:
        dt0 = drv.dt
        n = drv.step()
        log.append(n)
        steady = drv.dt < 1.5 * dt0 and len(log) >= 2 and n == log[-2]
        quiet = quiet + 1 if steady else 0


def self_launch(a, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: the program starts its own ranks, as the
    reference's binary does under MPI_Init (main.c:139-140, :210-211).  The ranks are CHILD processes of this one, which
    has not touched the GPU (`import torch` alone does not); their one JSON line is relayed and their exit code kept."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in pr.stdout.splitlines() if ln.strip().startswith("{")]
    if lines:
        print(lines[-1])
    elif pr.stdout:
        sys.stderr.write(pr.stdout[-4000:])
    sys.exit(pr.returncode if pr.returncode or lines else 1)


class Ctx:
    """What every measurement window of one invocation shares."""
    def __init__(self, a, aa, driver, torch, dist, rank, world, local, multi):
        self.a, self.aa, self.driver, self.torch, self.dist = a, aa, driver, torch, dist
        self.rank, self.world, self.local, self.multi = rank, world, local, multi

    def barrier(self):
        if self.multi:
            self.dist.barrier()
        self.torch.cuda.synchronize()


def run_window(c, strong, spinup, steps, warmup, nslab=1):
    """One Driver from the deck's initial state: spin-up, warm-up, then EXACTLY `steps` timed steps between barrier +
    synchronize pairs, MAX over ranks.  Returns the raw measurements (rank 0 builds the JSON from them)."""
    a, aa, torch = c.a, c.aa, c.torch
    nx, world = a.nx, c.world
    p2 = a.p2 if nslab == 1 else 1                       # Grids along x2 (pencils); the rest of the ranks along x3
    if world % p2:
        sys.exit(f"--p2 {p2} does not divide {world} ranks")
    nx2 = nx if strong else nx * p2
    nx3 = nx if strong else nx * (world // p2)
    deck = os.path.join(ROOT, PKG, "decks", "athinput." + a.problem)
    par = aa.athinput.ParTable.from_file(deck)
    x2min, x2max = par.getd("domain1", "x2min"), par.getd("domain1", "x2max")
    x3min, x3max = par.getd("domain1", "x3min"), par.getd("domain1", "x3max")
    ov = [f"domain1/Nx1={nx}", f"domain1/Nx2={nx2}", f"domain1/Nx3={nx3}"]
    if not strong:      # weak scaling: the box grows along the cut directions with the same dx
        ov.append(f"domain1/x3max={x3min + (x3max - x3min) * (world // p2)!r}")
        if p2 > 1:
            ov.append(f"domain1/x2max={x2min + (x2max - x2min) * p2!r}")
    par.cmdline(ov)
    run = aa.config.from_par(par, a.problem)
    run.integrator = a.integrator
    run.order = a.order
    t_setup = time.time()
    if nslab > 1:      # ONE process, the library cuts the Grid (csrc/slabs.hip): the path of the drop-in executables
        fac = lambda grid: c.driver.HipEngine(grid, c.local, True if a.strict else None, nslab=nslab)
        drv = c.driver.Driver(run, fac, 0, 1, c.local)
    else:
        drv = c.driver.Driver(run, None, c.rank, world, c.local, strict=True if a.strict else None, p2=p2)
    if a.ionized_slab:
        if not run.ion:
            sys.exit("--ionized-slab needs a problem with ion radiation")
        U = drv.eng.g.host_initial
        U[..., 5] = 1.0e-4 * U[..., 0]
        drv.eng.g.upload(U)
    drv.start()
    eng = drv.eng
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup
    spin_log = []
    t_spin = time.time()
    spin_up(drv, spinup if run.ion else "0", spin_log)
    torch.cuda.synchronize()
    t_spin = time.time() - t_spin
    for _ in range(warmup):
        drv.step()
    hist0 = drv.history()
    eng.g.profile_reset()
    if not a.no_kernel_times:
        eng.g.profile_enable(True)
    drv.niter_trace.clear()
    drv.host_sync_count(reset=True)
    c.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        drv.step()
    c.barrier()
    t1 = time.perf_counter()
    prof = eng.g.profile() if not a.no_kernel_times else {}
    eng.g.profile_enable(False)
    nsync = drv.host_sync_count()
    elapsed = t1 - t0
    if c.multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        c.dist.all_reduce(t, op=c.dist.ReduceOp.MAX)
        elapsed = float(t.item())
    hist1 = drv.history()      # volume integrals (dump_history.c): a NaN or Inf anywhere in the state shows up here
    w = {"run": run, "nx": nx, "nx3": nx3, "strong": strong, "elapsed": elapsed, "steps": steps, "warmup": warmup, "prof": prof,
         "nsync": nsync, "hist0": hist0, "hist1": hist1, "spin_log": spin_log, "t_spin": t_spin, "t_setup": t_setup,
         "niter": list(drv.niter_trace), "dt": drv.dt, "time": drv.time, "hbm": eng.g.device_bytes(), "spinup": spinup, "nx2": nx2, "p2": p2,
         "zones": nx * nx2 * nx3, "zones_gpu": (nx * nx2 * nx3) // (world * nslab) if (strong or nslab > 1) else nx ** 3, "nslab": nslab}
    eng.close()
    return w


KERNEL_SOURCES = ("hydro_kernels.hip", "hydro_dev.h", "ion_pass.hip", "ion_dev.h", "ion_kernels.hip", "grid.h")


def source_fingerprint():
    """sha256 over the kernel sources: a committed traffic profile carries the fingerprint it was taken on, and is stale -- not
    quoted -- once a kernel has changed (tests/test_gpu_bench_contract.py fails then: re-run profiles/prof_r04.sh)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, PKG, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


PMC = {}      # --pmc-pass: bytes per launch counted in THIS invocation (child processes under rocprofv3)
TRAFFIC_FILES = {"auto": ("r04_traffic.json", "r03_traffic.json"), "burst": ("r04_burst_traffic.json", "r03_burst_traffic.json"),
                 "19": ("r04_burst_traffic.json", "r03_burst_traffic.json")}


def chain_traffic(a, w, world, names, meta=None):
    """HBM bytes per launch of the named kernels: counted by this invocation's own rocprofv3 PMC passes (--pmc-pass), else from
    the committed passes of this same command (profiles/r04_traffic.json / r04_burst_traffic.json) -- only for the workload
    they were taken on and only while the kernel sources still have the fingerprint recorded with them."""
    plain = a.integrator == "ctu" and a.order == 2 and world == 1 and w["nslab"] == 1 and not a.ionized_slab and not a.strict
    if PMC.get("kernels") and plain and w["spinup"] == a.spinup:      # (the children ran this invocation's own window)
        if meta is not None:
            meta.update(kind="pmc-pass of this invocation", passes=PMC.get("passes"), seconds=PMC.get("seconds"))
        return {k: PMC["kernels"].get(k) for k in names}
    for tf in TRAFFIC_FILES.get(w["spinup"], ()):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
        except Exception:
            continue
        if not (tj.get("workload", "").startswith(f"{a.problem} {w['nx']}x{w['nx2']}x{w['nx3']}") and plain):
            return {}
        fp = source_fingerprint()
        stale = tj.get("source_fingerprint") != fp
        if meta is not None:
            meta.update(kind="committed profile", file="profiles/" + tf, profile_commit=tj.get("commit"),
                        profile_source_fingerprint=tj.get("source_fingerprint"), source_fingerprint=fp, stale=stale)
        return {} if stale else {k: tj["kernels"].get(k) for k in names}
    return {}


def pmc_passes(argv):
    """--pmc-pass: count the hydro chain's HBM bytes in this invocation.  Two CHILD processes, started before this one touches the
    GPU, run this same script for three timed steps under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the TCC
    counters do not fit together; the program directly behind `--`); the last three launches of every hydro kernel are averaged.
    FETCH_SIZE x 2 per the gfx950 correction of MI355X_MICROARCH.md (HBM section), WRITE_SIZE exact, KiB."""
    import csv
    import glob
    exe = shutil.which("rocprofv3")
    if not exe:
        sys.stderr.write("[bench] --pmc-pass: rocprofv3 is not on PATH; roofline.traffic stays with the committed profile\n")
        return
    keep = [x for x in argv if x not in ("--pmc-pass",)]

    def bench_name(n):
        """rocprofv3's kernel name -> the name bench.py's event profiler books the launch under (hydro chain of the CTU integrator)"""
        m = re.search(r"(k_\w+)(?:<([^>]*)>)?", n)
        if not m:
            return None
        k, targs = m.group(1), [x.strip() for x in (m.group(2) or "").split(",")]
        if k == "k_flux2_update":
            return "flux2_update"
        if k in ("k_correct_all", "k_eta_edges", "k_x1_edge_flux"):
            return "correct_all"
        if k in ("k_sweep_x1_flat", "k_sweep_x1"):         # <NS, GRAV, MODE, ORD>
            return {"0": "sweep_x1", "3": "sweep_correct_x1", "1": "correct_x1"}.get(targs[2])
        if k == "k_sweep_march":                            # <NS, D, GRAV, MODE, ORD>
            return ("sweep_x2" if targs[1] == "1" else "sweep_x3") if targs[3] == "0" else None
        if k == "k_sweep_tile":                             # <NS, D, GRAV, MODE, BT, ORD>
            return "correct_x2" if targs[1] == "1" else "correct_x3"
        if k == "k_flux2":                                  # <NS, D>
            return "flux2_x%d" % (int(targs[1]) + 1)
        if k == "k_update":
            return "update"
        return None
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    t0 = time.time()
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__)] + keep + \
                  ["--steps", "3", "--warmup", "1", "--no-burst", "--no-cpu-baseline", "--no-kernel-times", "--no-driver-window"]
            env = dict(os.environ, TMPDIR="/tmp")
            pr = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, cwd="/tmp", env=env, timeout=900)
            f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if pr.returncode != 0 or not f:
                sys.stderr.write(f"[bench] --pmc-pass: the {counter} pass failed (rc {pr.returncode}): {pr.stderr[-300:]}\n")
                return
            per = {}
            for r in csv.DictReader(open(f[0])):
                per.setdefault(r["Kernel_Name"], []).append((int(r.get("Start_Timestamp", 0) or 0), float(r["Counter_Value"])))
            acc = {}
            for n, rows in per.items():
                key = bench_name(n)
                if key is None:
                    continue
                rows.sort()
                last = [v for _, v in rows[-3:]]
                acc[key] = acc.get(key, 0.0) + sum(last) / len(last)
            got[counter] = acc
        PMC["kernels"] = {k: (2.0 * got["FETCH_SIZE"].get(k, 0.0) + got["WRITE_SIZE"].get(k, 0.0)) * 1024.0 for k in got["FETCH_SIZE"]}
        PMC["passes"] = "rocprofv3 --pmc FETCH_SIZE; --pmc WRITE_SIZE (children of this run, 3 timed steps each; fetch x2 per the gfx950 correction)"
        PMC["seconds"] = time.time() - t0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def analyse(c, w):
    """The JSON object of one window (rank 0)."""
    a, world = c.a, c.world
    run, nx, nx3, steps, elapsed, prof = w["run"], w["nx"], w["nx3"], w["steps"], w["elapsed"], w["prof"]
    zones, zones_gpu = w["zones"], w["zones_gpu"]
    niter = w["niter"]
    nsub_tot = sum(niter)
    nsub = nsub_tot / max(1, len(niter))
    value = zones * steps / elapsed
    nvar = 5 + run.nscal
    hist0, hist1 = w["hist0"], w["hist1"]
    regime = {"auto": " (stationary regime: dt at the CFL limit, sub-cycle count constant)",
              "burst": " (burst regime: the first burst of sub-cycles after the dt-doubling phase)"}.get(w["spinup"], "")
    out = {
        "metric": "cell-updates/sec (hydro+ion-rad step)", "value": value, "unit": "cell-updates/s",
        "n_gpus": world, "steps": steps, "warmup": w["warmup"], "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "strong" if w["strong"] else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (deck values on a uniform grid, generated in place"
                + ("; neutral fraction reset to 1e-4 everywhere: fully ionized slab" if a.ionized_slab else "") + ")",
        "config": {"workload": f"{a.problem} {nx}x{w['nx2']}x{nx3} single level, "
                               + ((f"CTU+{'PPM' if a.order == 3 else 'PLM'}+Roe+H-correction") if a.integrator == "ctu" else f"VL+{'PPM' if a.order == 3 else 'PLM'}+Roe")
                               + (" + static gravity + plane-parallel ion radiation" if a.problem == "ioniz_sphere"
                                  else (" + plane-parallel ion radiation" if a.problem == "ifront" else ""))
                               + (f"; timed after {len(w['spin_log'])} spin-up steps" + regime if run.ion else ""),
                   "build": "libathena_amd_strict.so (-ffp-contract=off)" if a.strict else "libathena_amd.so",
                   "zones_per_gpu": zones_gpu,
                   "partition": (f"x3 slabs x{w['nslab']} inside the library (one process, aa_params.nslab)" if w["nslab"] > 1 else
                                 (f"x2 x x3 pencils {w['p2']}x{world // w['p2']}" if w["p2"] > 1 else f"x3 slabs x{world}")
                                 + (" of one box (strong scaling)" if w["strong"] else "")),
                   "nvar": nvar, "spinup_steps": len(w["spin_log"]), "spinup_subcycle_trace": w["spin_log"], "spinup_s": w["t_spin"],
                   "radiation_subcycles_per_step": nsub, "subcycle_trace": niter,
                   "final_dt": w["dt"], "final_time": w["time"], "hbm_resident_GB": w["hbm"] / 1e9, "setup_s": w["t_setup"]},
    }
    # the state that was timed: finite everywhere, mass and energy of the box before / after (outflow
    # boundaries: not conserved to round-off), no sub-cycle loop at its iteration limit
    fin = all(math.isfinite(float(x)) for x in hist1)
    out["state_check"] = {"finite": fin, "mass_before": float(hist0[0]), "mass_after": float(hist1[0]),
                          "mass_rel_change": float(hist1[0] / hist0[0] - 1.0) if hist0[0] else None,
                          "energy_rel_change": float(hist1[1] / hist0[1] - 1.0) if hist0[1] else None,
                          "max_subcycles": max(niter) if niter else 0, "maxiter": run.maxiter,
                          "ok": bool(fin and (not run.ion or max(niter) < run.maxiter))}
    out["host_syncs_per_step"] = w["nsync"] / steps
    # new_dt costs one read-back per step (two with several ranks: the slab's own and the all-reduce's)
    out["host_syncs_per_subcycle"] = (w["nsync"] / steps - (2.0 if c.multi else 1.0)) / max(nsub, 1e-30) if run.ion else None
    if prof:
        ms_step = {k: v[0] / steps for k, v in prof.items()}
        cls = {"hydro": 0.0, "subcycle": 0.0, "ion_step": 0.0, "other": 0.0}
        for k, v in ms_step.items():
            cls[kernel_class(k)] += v
        b_h = 2 * 8 * nvar
        hydro_names = sorted((k for k in ms_step if kernel_class(k) == "hydro"), key=lambda k: -ms_step[k])
        sub_names = sorted((k for k in ms_step if kernel_class(k) == "subcycle"), key=lambda k: -ms_step[k])
        # 1. the contract's `roofline`: SURVEY 8(d)'s unit of work and its ALGORITHMIC bytes, divided by the time of the kernels
        #    that perform the unit (hipEvent pairs on the launch stream inside the timed region).  The unit that takes most of
        #    the step is reported: the hydro cell-update (2*NVAR*8 B, done by the integrator's kernel chain) or the radiation
        #    sub-cycle (64 B per cell, one k_ion_pass).  The dominant KERNEL and the bytes its own schedule must move are kept
        #    beside it (`kernel_own_bytes_frac`: round 2's `frac`).
        dom = dominant_kernel(prof)
        unit_hydro = cls["hydro"] >= cls["subcycle"]
        if unit_hydro and cls["hydro"] > 0:
            ms_unit, b_unit, names = cls["hydro"], b_h, hydro_names
            unit = f"hydro cell-update: 2*NVAR*8 = {b_h} B (SURVEY 8d), performed by " + " + ".join(names)
        elif cls["subcycle"] > 0:
            ms_unit, b_unit, names = cls["subcycle"] * steps / max(nsub_tot, 1), 64, sub_names
            unit = "radiation sub-cycle: 64 B per cell (SURVEY 8d), performed by " + " + ".join(names)
        else:
            ms_unit = 0.0
        if ms_unit > 0:
            ach = b_unit * zones_gpu / (ms_unit * 1e-3) / 1e9
            tmeta = {}
            tr = chain_traffic(a, w, world, names, tmeta)
            per_unit = steps if unit_hydro else max(nsub_tot, 1)
            traffic = (sum(tr[k] * prof[k][1] for k in names) / per_unit) if tr and all(tr.get(k) is not None for k in names) else None
            rf = {"bound": "hbm", "kernel": names[0] if len(names) == 1 else "chain(" + "+".join(names) + ")", "achieved": ach,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                  "bytes_per_launch": b_unit * zones_gpu, "bytes_basis": unit, "avg_launch_ms": ms_unit,
                  "traffic_source": tmeta or None}
            if dom:
                ms, n = prof[dom]
                scale = (nvar / 6.0) if kernel_class(dom) == "hydro" else 1.0
                kb = KERNEL_BYTES[dom]
                if dom == "correct_all" and "sweep_x3" not in prof:
                    # the x3 (and, without a sweep_x1 launch, the x1) first pass rides along: those fluxes are neither written nor read
                    kb = CORRECT_ALL_X3_BYTES if "sweep_x1" in prof else CORRECT_ALL_X1X3_BYTES
                own = kb * zones_gpu * scale / (ms / n * 1e-3) / 1e9
                rf["dominant_kernel"] = dom
                rf["dominant_kernel_avg_launch_ms"] = ms / n
                rf["kernel_own_bytes_per_zone"] = kb * scale
                rf["kernel_own_bytes_frac"] = own / HBM_PEAK_GBS
                rf["dominant_kernel_traffic"] = chain_traffic(a, w, world, [dom]).get(dom)
            out["roofline"] = rf
        # 2. the phases of the step on SURVEY 8(d)'s algorithmic bytes: hydro chain 2*NVAR*8 B per
        #    cell-update, radiation 64 B per cell and sub-cycle
        ph = {}
        if cls["hydro"] > 0:
            a_h = b_h * zones_gpu / (cls["hydro"] * 1e-3) / 1e9
            tf = HYDRO_FLOP_PER_CELL * zones_gpu / (cls["hydro"] * 1e-3) / 1e12
            ph["hydro"] = {"ms_per_step": cls["hydro"], "bytes_per_cell": b_h, "achieved_GBs": a_h, "frac_hbm": a_h / HBM_PEAK_GBS,
                           "ns_per_zone": 1e6 * cls["hydro"] / zones_gpu,
                           "fp64_TFLOPs_est": tf, "frac_fp64_peak_est": tf / FP64_PEAK_TFLOPS}
        if nsub_tot > 0 and cls["subcycle"] > 0:
            ms_sub = cls["subcycle"] * steps / nsub_tot
            a_s = 64 * zones_gpu / (ms_sub * 1e-3) / 1e9
            ph["subcycle"] = {"ms_per_subcycle": ms_sub, "bytes_per_cell": 64, "achieved_GBs": a_s, "frac_hbm": a_s / HBM_PEAK_GBS,
                              "ns_per_zone": 1e6 * ms_sub / zones_gpu, "subcycles_timed": nsub_tot,
                              "note": "all kernels of the ion step's loop / sub-cycles timed; with ONE sub-cycle per step that is the first pass "
                                      "(which also does the step's entry: floors, save_energy_and_x) plus the closing update-only pass"}
            if prof.get("ion_pass", (0, 0))[1] > 0 and nsub > 1.5:
                # the repeating unit of a long loop: one full pass = update(n-1) + sweep(n) + rates(n)
                ms_full = prof["ion_pass"][0] / prof["ion_pass"][1]
                a_f = 64 * zones_gpu / (ms_full * 1e-3) / 1e9
                ph["subcycle"]["full_pass_ms"] = ms_full
                ph["subcycle"]["full_pass_frac_hbm"] = a_f / HBM_PEAK_GBS
        ph["ion_step_overhead_ms"] = cls["ion_step"]
        ph["other_ms"] = cls["other"]
        ph["unattributed_ms"] = 1e3 * elapsed / steps - sum(cls.values())
        out["phases"] = ph
        out["kernel_ms_per_step"] = dict(sorted(ms_step.items(), key=lambda kv: -kv[1]))
        out["kernel_launches_per_step"] = {k: v[1] / steps for k, v in prof.items()}
    # 3. the whole step, the survey's definition: (2*NVAR*8 + 64*<N_sub>) B per cell-update
    bstep = (2 * 8 * nvar) + (64 * nsub if run.ion else 0)
    out["step_roofline"] = {"bytes_per_cell_update": bstep, "achieved": value / world * bstep / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                            "frac": value / world * bstep / 1e9 / HBM_PEAK_GBS}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nx", type=int, default=512, help="zones per direction per GPU (default: the 512^3 workload)")
    ap.add_argument("--problem", default="ioniz_sphere", choices=["ioniz_sphere", "ifront", "blast"])
    ap.add_argument("--integrator", default="ctu", choices=["ctu", "vl"])
    ap.add_argument("--order", type=int, default=2, choices=[2, 3], help="reconstruction: 2 PLM (default), 3 PPM (--with-order=3)")
    ap.add_argument("--spinup", default="auto",
                    help="untimed steps before the warm-up: 'auto' (until dt has stopped doubling and the sub-cycle count is "
                         "stationary, at most 320), 'burst' (until the first burst of sub-cycles after the dt-doubling phase) "
                         "or a number (0: the start-up transient; 19 at 512^3: the burst of sub-cycles)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong", "both"],
                    help="N>1: weak (nx^3 per GPU, the box grows along x3), strong (ONE nx^3 box cut into N x3 slabs: BASELINE "
                         "configs[3]) or both (default for N>1: the weak line with the strong window under `strong_scaling`)")
    ap.add_argument("--strong", action="store_true", help="same as --scaling strong")
    ap.add_argument("--p2", type=int, default=1, metavar="P2",
                    help="N>1: P2 Grids along x2 and N/P2 along x3 (x2 x x3 pencils, init_mesh.c:526-620) instead of x3 slabs")
    ap.add_argument("--inlib", type=int, default=0, metavar="N",
                    help="ONE process, the library cuts the Grid into N x3 slabs itself (aa_params.nslab; AA_SLAB_DEVICES picks the "
                         "devices): the multi-GPU path of the drop-in executables, strong scaling, no launcher involved")
    ap.add_argument("--no-burst", action="store_true", help="N=1: skip the second (burst-regime) window")
    ap.add_argument("--burst-window", action="store_true", help="N=1: time the second (burst-regime) window also with a numeric --spinup")
    ap.add_argument("--strict", action="store_true", help="time libathena_amd_strict.so (-ffp-contract=off: the bit-exact hydro build)")
    ap.add_argument("--smr", action="store_true",
                    help="BASELINE.json configs[4]: 2-level static mesh refinement, per GPU a root slab of nx^3 zones plus "
                         "nx^3 level-1 zones over the central half of the box (not the headline line)")
    ap.add_argument("--smr-deck", action="store_true",
                    help="with --smr: the deck's own root and level-1 Domains (80^3 + 52^3, the first two levels of "
                         "tst/massloss/athinput.ioniz_sphere_hires) instead of nx^3 + nx^3")
    ap.add_argument("--smr-levels", type=int, default=2,
                    help="with --smr-deck: how many of the deck's 5 nested Domains to use; more than 2 need "
                         "AA_SMR_DEEP_RADIATION=fixed (DESIGN.md section 6)")
    ap.add_argument("--ionized-slab", action="store_true",
                    help="SURVEY 8(d) worst case for the ray sweep: neutral fraction 1e-4 everywhere, so every ray crosses the whole box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-driver-window", action="store_true",
                    help="N=1: skip the windows that run the step as the RANKS of an N>1 job run it (driver.Driver.step with its collectives, "
                         "one-rank communicator) -- `driver_path`: what the host side of the multi-GPU path costs against aa_step")
    ap.add_argument("--pmc-pass", action="store_true",
                    help="count roofline.traffic in this run: two child processes under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE) before "
                         "the timed windows; otherwise the committed profile is quoted while the kernel sources are unchanged")
    ap.add_argument("--no-kernel-times", action="store_true")
    a = ap.parse_args()
    if a.strong:
        a.scaling = "strong"

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.inlib:
        self_launch(a, sys.argv[1:])        # does not return
    if a.pmc_pass and a.gpus == 1 and "WORLD_SIZE" not in os.environ and not a.smr and not a.inlib:
        pmc_passes(sys.argv[1:])            # children under rocprofv3, before this process touches the GPU

    import torch
    aa = importlib.import_module(PKG)
    driver = importlib.import_module(PKG + ".driver")
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if REHEARSAL else int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and not a.inlib:
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if not REHEARSAL and local >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} wants HIP device {local}, this node shows {torch.cuda.device_count()} "
                 f"(one rank per GPU; AA_BENCH_REHEARSAL=1 puts every rank on cuda:0 over gloo -- a rehearsal, not a measurement)")
    torch.cuda.set_device(local)
    if a.smr:
        return bench_smr(a, aa, torch, rank, world, local)
    force = bool(os.environ.get("AA_FORCE_DISTRIBUTED"))
    multi = world > 1 or force
    dist = None
    if multi:
        import torch.distributed as dist
        init_pg(dist, torch, rank, world, local)
    c = Ctx(a, aa, driver, torch, dist, rank, world, local, multi)

    if a.inlib:
        # one process, N slabs behind the C entry points (strong scaling of ONE nx^3 box; the slabs sit on the devices of
        # AA_SLAB_DEVICES, all on the current one by default = a rehearsal)
        w = run_window(c, True, a.spinup, a.steps, a.warmup, nslab=a.inlib)
        out = analyse(c, w)
        out["n_gpus"] = a.inlib
        out["config"]["slab_devices"] = os.environ.get("AA_SLAB_DEVICES", "all slabs on the current device (rehearsal, not a measurement)")
        print(json.dumps(out))
        return

    mode = a.scaling or ("both" if world > 1 else "weak")
    first_strong = (mode == "strong")
    w = run_window(c, first_strong, a.spinup, a.steps, a.warmup)
    out = analyse(c, w) if rank == 0 else None
    if mode == "both" and world > 1:
        # BASELINE's metric is also quoted on ONE 512^3 box over N GPUs (configs[3]): the same job times that window too
        ws = run_window(c, True, a.spinup, a.steps, a.warmup)
        if rank == 0:
            s = analyse(c, ws)
            out["strong_scaling"] = {k: s[k] for k in ("value", "ms_per_step", "steps", "warmup", "state_check", "host_syncs_per_subcycle",
                                                       "step_roofline") if k in s}
            out["strong_scaling"].update(zones_per_gpu=s["config"]["zones_per_gpu"], workload=s["config"]["workload"],
                                         radiation_subcycles_per_step=s["config"]["radiation_subcycles_per_step"],
                                         kernel_ms_per_step=s.get("kernel_ms_per_step"))
    if world == 1 and not multi and w["run"].ion and (a.spinup == "auto" or a.burst_window) and not a.no_burst:
        # second window: the regime in which the radiation sub-cycle dominates the step (dozens of sub-cycles), so that the
        # sub-cycle kernel's roofline fraction is a driver-timed number as well
        try:
            wb = run_window(c, False, "burst", min(6, a.steps), 0)
        except Exception as e:           # the headline window above stands on its own: say so and go on
            wb = None
            if rank == 0:
                out["regimes"] = {"burst": {"error": f"{type(e).__name__}: {e}"[:300]}}
        if rank == 0 and wb is not None:
            b = analyse(c, wb)
            sub = b.get("phases", {}).get("subcycle", {})
            out["regimes"] = {"stationary": {"nsub": out["config"]["radiation_subcycles_per_step"], "ms_per_step": out["ms_per_step"]},
                              "burst": {"nsub": b["config"]["radiation_subcycles_per_step"], "ms_per_step": b["ms_per_step"],
                                        "value": b["value"], "steps": b["steps"], "spinup_steps": b["config"]["spinup_steps"],
                                        "subcycle_trace": b["config"]["subcycle_trace"],
                                        "ms_per_subcycle": sub.get("ms_per_subcycle"), "full_pass_ms": sub.get("full_pass_ms"),
                                        "full_pass_frac_hbm": sub.get("full_pass_frac_hbm"),
                                        "hydro_ms_per_step": b.get("phases", {}).get("hydro", {}).get("ms_per_step"),
                                        "step_roofline_frac": b["step_roofline"]["frac"], "state_ok": b["state_check"]["ok"],
                                        "roofline": b.get("roofline")}}
    if world == 1 and not multi and not a.no_driver_window:
        # The same step as the RANKS of an N > 1 job run it: driver.Driver.step -- Python between the phases, the sub-cycle loop inside
        # the library with the all-gather as its callback, new_dt's all-reduce -- on a one-rank communicator.  N = 1 above is ONE C
        # call per step (aa_step); a future 1 -> N curve compares like with like only through this number.
        try:
            import torch.distributed as dist1
            os.environ["AA_FORCE_DISTRIBUTED"] = "1"
            if "MASTER_PORT" not in os.environ:      # a port of our own: another job on the host may hold the default one
                import socket
                sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1]); sk.close()
            init_pg(dist1, torch, 0, 1, local)
            cd = Ctx(a, aa, driver, torch, dist1, 0, 1, local, True)
            wd = run_window(cd, False, a.spinup, a.steps, a.warmup)
            d = analyse(cd, wd)
            dp = {"ms_per_step": d["ms_per_step"], "value": d["value"], "steps": d["steps"], "nsub": d["config"]["radiation_subcycles_per_step"],
                  "host_path_overhead_ms": d["ms_per_step"] - out["ms_per_step"], "host_syncs_per_step": d["host_syncs_per_step"],
                  "host_syncs_per_subcycle": d["host_syncs_per_subcycle"], "kernel_ms_per_step": d.get("kernel_ms_per_step"),
                  "what": "driver.Driver.step on a one-rank communicator (backend %s): the host path of every rank of an N>1 job" % dist1.get_backend()}
            if "regimes" in out and "burst" in out["regimes"] and "ms_per_step" in out["regimes"]["burst"]:
                wdb = run_window(cd, False, "burst", min(6, a.steps), 0)
                db = analyse(cd, wdb)
                dp["burst"] = {"ms_per_step": db["ms_per_step"], "nsub": db["config"]["radiation_subcycles_per_step"],
                               "host_path_overhead_ms": db["ms_per_step"] - out["regimes"]["burst"]["ms_per_step"],
                               "host_syncs_per_subcycle": db["host_syncs_per_subcycle"]}
            out["driver_path"] = dp
            dist1.barrier(); dist1.destroy_process_group()
        except Exception as e:
            out["driver_path"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            os.environ.pop("AA_FORCE_DISTRIBUTED", None)
    if rank == 0:
        if not a.no_cpu_baseline:
            try:
                cb = cpu_baseline(gpus=world)
            except Exception as e:
                cb = {"value": None, "unit": "cell-updates/s", "cores": 0, "kind": "none", "sample": f"failed: {type(e).__name__}: {e}"[:300]}
            out["cpu_baseline"] = cb
            if "phases" in out and "per_hydro_step_ns_per_zone" in cb:
                ph = out["phases"]
                out["gpu_vs_cpu"] = {"hydro_step": cb["per_hydro_step_ns_per_zone"] / ph["hydro"]["ns_per_zone"] if "hydro" in ph else None,
                                     "note": f"one MI355X against {cb['cores']} host cores, per zone and per hydro step"}
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
