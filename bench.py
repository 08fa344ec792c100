#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the hot path on N MI355X of one node.

One "step" = one pass of the reference's main loop (main.c:519-669) on synthetic input already
resident in HBM: ion-radiation step with its sub-cycles (ionrad_3d.c:862) + ghost zones +
3-D CTU hydro step (integrate_3d_ctu.c:110) + per-step core reset + new_dt + ghost zones.
Workload at N=1: BASELINE.json configs[3] on one GPU, tst/massloss/athinput.ioniz_sphere_hires
keys at 512^3 single level (hydro + static gravity + ion radiation).  With N>1 the box is cut
into x3 slabs of 512x512x512 each (weak scaling: 512 x 512 x 512*N zones, same dx), halo
exchange and scalar reductions over RCCL.

Prints ONE JSON line (rank 0).  `value` = active zones advanced per second by the whole job.
"""
import argparse
import importlib
import json
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

# Compulsory bytes per cell of each kernel = distinct doubles it must read + write per zone
# (NVAR=6): the figures are derived in DESIGN.md section 4.
KERNEL_BYTES = {
    "sweep_x1": 8 * (6 + 6), "sweep_x2": 8 * (6 + 6), "sweep_x3": 8 * (6 + 6),
    "sweep_correct_x1": 8 * (6 + 12 + 6 + 12 + 1), "correct_x1": 8 * (6 + 12 + 12 + 1), "correct_x2": 8 * (6 + 12 + 12 + 1), "correct_x3": 8 * (6 + 12 + 12 + 1),
    "vl_predict": 8 * (6 + 6 + 4),                   # U in, U^{n+1/2} out, phi x4
    "vl_flux1": 8 * (6 + 18), "vl_uhalf": 8 * (6 + 18 + 6), "vl_flux2_x1": 8 * 12, "vl_flux2_x2": 8 * 12,
    "vl_flux2_x3": 8 * 12, "flux2_x1": 8 * (12 + 3 + 6), "flux2_x2": 8 * (12 + 3 + 6),
    "flux2_x3": 8 * (12 + 3 + 6), "update": 8 * (6 + 18 + 6),
    "correct_all": 8 * (6 + 18 + 4 + 36 + 3 + 1),   # U, first-pass fluxes, phi in; L/R states x3, eta x3, d^{n+1/2} out
    "flux2_update": 8 * (36 + 3 + 6 + 5 + 6),     # L/R states of 3 directions, eta x3, U in, phi x4 + d^{n+1/2}, U out
    "ray_sweep": 8 * (1 + 2), "ray_sweep_rates": 8 * (1 + 2) + 8 * (3 + 1),   # + d, ke, E and the int2 sign word "ion_rates": 8 * (5 + 1), "ion_update": 8 * (5 + 1 + 2 + 0.5 + 2),
    "ion_begin": 8 * (6 + 6), "bvals_mhd": 0, "new_dt": 8 * 5, "pinned_cells": 0, "ppm_slopes": 3 * 8 * (6 + 6),
}


def cpu_baseline(nx=64, nlim=40):
    """The REAL reference (oracle/_ref, built from /root/reference by oracle/Makefile.ref) timed
    on one host core on a bounded sample of the same deck; falls back to the CPU restatement."""
    exe = os.path.join(ROOT, "oracle", "_ref", "athena_ioniz_sphere")
    deck = os.path.join(ROOT, PKG, "decks", "athinput.ioniz_sphere")
    if os.path.exists(exe):
        tmp = tempfile.mkdtemp(prefix="cpu_baseline_")
        try:
            t0 = time.time()
            pr = subprocess.run([exe, "-i", deck, "-d", os.path.join(tmp, "run"),
                                 f"domain1/Nx1={nx}", f"domain1/Nx2={nx}", f"domain1/Nx3={nx}", f"time/nlim={nlim}"],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, timeout=600)
            wall = time.time() - t0
            m = re.search(r"zone-cycles/wall-second = ([0-9.eE+-]+)", pr.stdout)
            its = [int(x) for x in re.findall(r"Radiation done in (\d+) iterations", pr.stderr)]
            if pr.returncode == 0 and m:
                return {"value": float(m.group(1)), "unit": "cell-updates/s", "cores": 1, "kind": "reference",
                        "sample": f"ioniz_sphere {nx}^3, {nlim} steps, sub-cycles/step {its}, {wall:.1f} s wall"}
        except Exception:
            pass
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    s = orc.make_sim("ioniz_sphere", [f"domain1/Nx{d}={nx}" for d in (1, 2, 3)]).start()
    t0 = time.time(); its = [s.step() for _ in range(nlim)]; wall = time.time() - t0
    return {"value": nx ** 3 * nlim / wall, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": f"ioniz_sphere {nx}^3, {nlim} steps, sub-cycles/step {its}, {wall:.1f} s wall"}


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
                bpl = KERNEL_BYTES.get(k, 0) * ncell
                ach = bpl / (ms / n * 1e-3) / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": f"L{l}.{k}", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": ach / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": bpl, "avg_launch_ms": ms / n}
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    else:
        m.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nx", type=int, default=512, help="zones per direction per GPU (default: the 512^3 workload)")
    ap.add_argument("--problem", default="ioniz_sphere", choices=["ioniz_sphere", "ifront", "blast"])
    ap.add_argument("--integrator", default="ctu", choices=["ctu", "vl"])
    ap.add_argument("--order", type=int, default=2, choices=[2, 3], help="reconstruction: 2 PLM (default), 3 PPM (--with-order=3)")
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
    ap.add_argument("--no-kernel-times", action="store_true")
    a = ap.parse_args()

    import torch
    aa = importlib.import_module(PKG)
    driver = importlib.import_module(PKG + ".driver")
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if REHEARSAL else int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    if a.smr:
        return bench_smr(a, aa, torch, rank, world, local)
    force = bool(os.environ.get("AA_FORCE_DISTRIBUTED"))
    if world > 1 or force:
        import torch.distributed as dist
        init_pg(dist, torch, rank, world, local)

    # weak scaling: every GPU holds nx^3 zones; the box grows along x3 with the same dx
    nx = a.nx
    deck = os.path.join(ROOT, PKG, "decks", "athinput." + a.problem)
    par = aa.athinput.ParTable.from_file(deck)
    x3min, x3max = par.getd("domain1", "x3min"), par.getd("domain1", "x3max")
    par.cmdline([f"domain1/Nx1={nx}", f"domain1/Nx2={nx}", f"domain1/Nx3={nx * world}",
                 f"domain1/x3max={x3min + (x3max - x3min) * world!r}"])
    run = aa.config.from_par(par, a.problem)
    run.integrator = a.integrator
    run.order = a.order
    t_setup = time.time()
    drv = driver.Driver(run, None, rank, world, local)
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

    def barrier():
        if world > 1 or force:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        drv.step()
    eng.g.profile_reset()
    if not a.no_kernel_times:
        eng.g.profile_enable(True)
    drv.niter_trace.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        drv.step()
    barrier()
    t1 = time.perf_counter()
    prof = eng.g.profile() if not a.no_kernel_times else {}
    eng.g.profile_enable(False)
    elapsed = t1 - t0
    if world > 1 or force:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        zones = nx * nx * nx * world
        nsub = sum(drv.niter_trace) / max(1, len(drv.niter_trace))
        value = zones * a.steps / elapsed
        out = {
            "metric": "cell-updates/sec (hydro+ion-rad step)", "value": value, "unit": "cell-updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (deck values on a uniform grid, generated in place"
                    + ("; neutral fraction reset to 1e-4 everywhere: fully ionized slab" if a.ionized_slab else "") + ")",
            "config": {"workload": f"{a.problem} {nx}x{nx}x{nx * world} single level, "
                                   + ((f"CTU+{'PPM' if a.order == 3 else 'PLM'}+Roe+H-correction") if a.integrator == "ctu" else "VL+PLM+Roe")
                                   + (" + static gravity + plane-parallel ion radiation" if a.problem == "ioniz_sphere"
                                      else (" + plane-parallel ion radiation" if a.problem == "ifront" else "")),
                       "zones_per_gpu": nx ** 3, "partition": f"x3 slabs x{world}", "nvar": 5 + run.nscal,
                       "radiation_subcycles_per_step": nsub, "subcycle_trace": drv.niter_trace,
                       "final_dt": drv.dt, "hbm_resident_GB": eng.g.device_bytes() / 1e9, "setup_s": t_setup},
        }
        # roofline of the dominant kernel: compulsory bytes of one launch / its mean duration
        # (hipEvent pairs recorded on the launch stream inside the timed region)
        if prof:
            dom = max(prof, key=lambda k: prof[k][0] if KERNEL_BYTES.get(k, 0) > 0 else -1.0)   # (boundary kernels have no per-zone figure)
            ms, n = prof[dom]
            ncell = nx ** 3
            bpl = KERNEL_BYTES.get(dom, 0) * ncell * ((5 + run.nscal) / 6.0 if not dom.startswith("ion") and dom != "ray_sweep" else 1.0)
            ach = bpl / (ms / n * 1e-3) / 1e9 if n else 0.0
            # measured HBM bytes per launch of that kernel: from the committed rocprofv3 PMC passes of
            # this same command (profiles/r01_traffic.json); only valid for the workload it was taken on
            traffic = None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
                if tj.get("workload") == f"{a.problem} {nx}x{nx}x{nx}" and a.integrator == "ctu":
                    traffic = tj["kernels"].get(dom)
            except Exception:
                pass
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "bytes_per_launch": bpl,
                               "avg_launch_ms": ms / n if n else None}
            out["kernel_ms_per_step"] = {k: v[0] / a.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
            out["kernel_launches_per_step"] = {k: v[1] / a.steps for k, v in prof.items()}
        # the survey's whole-step definition: (96 + 64*<N_sub>) B per cell-update
        bstep = (2 * 8 * (5 + run.nscal)) + (64 * nsub if run.ion else 0)
        out["step_roofline"] = {"bytes_per_cell_update": bstep, "achieved": value / world * bstep / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                                "frac": value / world * bstep / 1e9 / HBM_PEAK_GBS}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
