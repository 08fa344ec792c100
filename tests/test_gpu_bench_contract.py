"""bench.py's output contract on a small Grid: exactly ONE JSON line on stdout with the driver's keys, the
`roofline` object of the dominant kernel (it must BE the kernel with the largest total time), the per-phase
rooflines on SURVEY 8(d)'s algorithmic bytes, the state check and the `cpu_baseline` object (the reference's MPI
build on the host cores when oracle/_ref travelled, else one core / the CPU restatement).  Also the N>1 code path
of bench.py, rehearsed with two ranks on the one GPU (gloo): weak and strong scaling, and the number of host
round trips per radiation sub-cycle."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None, launcher=()):
    e = dict(os.environ); e.update(env or {})
    pr = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                        text=True, cwd=ROOT, timeout=900, env=e)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    return json.loads(lines[0])


def test_one_json_line_with_roofline_and_cpu_baseline():
    d = run_bench("--nx", "64", "--steps", "3", "--warmup", "1", "--spinup", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "phases", "step_roofline", "state_check"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "cell-updates/s" and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "64x64x64" in d["config"]["workload"] and "model" not in d["config"]
    assert "spin-up" in d["config"]["workload"] and d["config"]["spinup_steps"] == 2
    assert abs(d["value"] - 64 ** 3 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    # the contract object is on SURVEY 8(d)'s unit: here the hydro cell-update, 96 B, over the time of the integrator's chain
    sys.path.insert(0, ROOT)
    import bench
    ph = d["phases"]
    assert r["kernel"].startswith("chain(") and "correct" in r["kernel"] and "96 B" in r["bytes_basis"]
    assert abs(r["avg_launch_ms"] - ph["hydro"]["ms_per_step"]) < 1e-9 and abs(r["frac"] - ph["hydro"]["frac_hbm"]) < 1e-12
    assert r["bytes_per_launch"] == 96 * 64 ** 3
    # ... with the dominant KERNEL (largest total time among those that have a byte figure) and its own bytes beside it
    timed = {k: v for k, v in d["kernel_ms_per_step"].items() if bench.KERNEL_BYTES.get(k, 0) > 0}
    assert r["dominant_kernel"] == max(timed, key=timed.get) and r["kernel_own_bytes_frac"] > r["frac"]
    assert set(d["kernel_ms_per_step"]) <= set(bench.KERNEL_BYTES) | {"halo_pack", "halo_unpack"}, "a kernel without a byte figure"
    assert ph["hydro"]["bytes_per_cell"] == 96 and ph["subcycle"]["bytes_per_cell"] == 64
    assert ph["hydro"]["ms_per_step"] > 0 and ph["subcycle"]["ms_per_subcycle"] > 0
    assert d["state_check"]["finite"] is True and d["state_check"]["ok"] is True
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference-mpi", "reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "cell-updates/s" and c["sample"]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "athena_ioniz_sphere_mpi")) and os.path.exists("/opt/conda/bin/mpiexec"):
        assert c["kind"] == "reference-mpi" and c["cores"] > 1, c
        assert c["per_hydro_step_ns_per_zone"] > 0 and "gpu_vs_cpu" in d
        assert "per_subcycle_ns_per_zone" not in c and "subcycle" not in d["gpu_vs_cpu"]      # (a difference of two runs: noise)


def test_second_window_times_the_burst_regime():
    """N = 1: after the headline window a second Driver is spun up to the first burst of radiation sub-cycles behind the
    dt-doubling phase and a few steps are timed there, so that the sub-cycle kernel's roofline fraction on 64 B per cell is a
    number of the driver's own run (`regimes.burst`)."""
    d = run_bench("--nx", "64", "--steps", "3", "--warmup", "1", "--spinup", "2", "--burst-window", "--no-cpu-baseline", "--no-driver-window")
    b = d["regimes"]["burst"]
    assert d["regimes"]["stationary"]["ms_per_step"] == d["ms_per_step"]
    assert b["steps"] == 3 and b["spinup_steps"] >= 4 and b["nsub"] >= 2 and b["ms_per_step"] > 0
    assert b["ms_per_subcycle"] > 0 and len(b["subcycle_trace"]) == 3
    assert b["roofline"]["bound"] == "hbm"


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` as the driver types it (no torchrun around it): the program starts its ranks as child
    processes, relays their ONE JSON line and exit code.  Default for N > 1: the weak-scaling line with the strong-scaling
    window (ONE nx^3 box cut into N slabs: BASELINE configs[3]) of the same job under `strong_scaling`.  Rehearsed with both
    ranks on cuda:0 over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["AA_BENCH_REHEARSAL"] = "1"
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nx", "64", "--steps", "2", "--warmup", "1",
                         "--spinup", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=900, env=env)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "64x64x128" in d["config"]["workload"]
    s = d["strong_scaling"]
    assert s["zones_per_gpu"] == 64 ** 3 // 2 and "64x64x64" in s["workload"] and s["value"] > 0 and s["state_check"]["ok"] is True
    assert s["host_syncs_per_subcycle"] <= 1.0 + 1e-9


def test_slabs_inside_the_library_as_a_bench_mode():
    """--inlib N: one process, aa_params.nslab = N (csrc/slabs.hip), the path of the drop-in executables; rehearsed with both
    slabs on the one device.  The radiation sub-cycle may cost ONE host round trip there too."""
    d = run_bench("--inlib", "2", "--nx", "64", "--steps", "2", "--warmup", "1", "--spinup", "2", "--no-cpu-baseline")
    one = run_bench("--nx", "64", "--steps", "2", "--warmup", "1", "--spinup", "2", "--no-cpu-baseline", "--no-driver-window")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "inside the library" in d["config"]["partition"]
    assert d["config"]["subcycle_trace"] == one["config"]["subcycle_trace"] and d["state_check"]["ok"] is True
    assert abs(d["state_check"]["mass_after"] / one["state_check"]["mass_after"] - 1) < 1e-12
    assert d["host_syncs_per_subcycle"] <= 1.0 + 1e-9, d["host_syncs_per_subcycle"]


def test_other_workloads_keep_the_contract():
    for args in (("--problem", "blast", "--nx", "48"), ("--smr", "--nx", "32"), ("--integrator", "vl", "--nx", "48"), ("--order", "3", "--nx", "48"),
                 ("--strict", "--nx", "48")):
        d = run_bench(*args, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--spinup", "1", "--no-driver-window")
        assert d["value"] > 0 and d["roofline"]["achieved"] > 0 and "workload" in d["config"]


@pytest.mark.parametrize("strong", [False, True])
def test_two_ranks_rehearsed_on_one_gpu(strong):
    """The N>1 path of bench.py (torchrun, one rank per GPU) with both ranks on cuda:0 over gloo: not a measurement,
    but every collective of the real run is issued.  The radiation sub-cycle may cost ONE host round trip."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = run_bench("--gpus", "2", "--nx", "64", "--steps", "2", "--warmup", "1", "--spinup", "2", *(("--strong",) if strong else ("--no-cpu-baseline",)),
                  env={"AA_BENCH_REHEARSAL": "1"},
                  launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port)))
    assert d["n_gpus"] == 2 and d["scaling"] == ("strong" if strong else "weak")
    assert ("64x64x64" if strong else "64x64x128") in d["config"]["workload"]
    assert d["config"]["zones_per_gpu"] == (64 ** 3 // 2 if strong else 64 ** 3)
    assert d["state_check"]["ok"] is True
    assert d["host_syncs_per_subcycle"] <= 1.0 + 1e-9, d["host_syncs_per_subcycle"]
    if strong:      # N > 1: the CPU leg uses every core this process may run on, and says how many
        c = d["cpu_baseline"]
        assert c["cores"] >= 1 and c["value"] > 0 and ("cores_available" not in c or c["cores"] <= c["cores_available"])


def test_two_ranks_rehearsed_on_one_gpu_smr():
    """bench.py --smr --gpus 2 (BASELINE configs[4] across GPUs: every level cut at the same root planes, MeshDriver), both
    ranks on cuda:0 over gloo: the code path the driver would run on a multi-GPU node."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = run_bench("--gpus", "2", "--smr", "--nx", "64", "--steps", "2", "--warmup", "1", env={"AA_BENCH_REHEARSAL": "1"},
                  launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port)))
    assert d["n_gpus"] == 2 and d["value"] > 0 and "2-level SMR" in d["config"]["workload"]
    assert all(len(t) == 2 and min(t) >= 1 for t in d["config"]["subcycle_trace_per_level"])


def test_one_rank_over_rccl():
    """The multi-rank loop of bench.py on ONE rank with the real backend ("nccl" = RCCL): communicator creation with
    the device id, the barrier, new_dt's all-reduce and the sub-cycle's all_gather of the reduction words from device
    memory all go through RCCL (the point-to-point halo needs a second GPU and stays unrehearsed on this backend).
    Same answer as the plain one-GPU run."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    args = ("--nx", "64", "--steps", "2", "--warmup", "1", "--spinup", "2", "--no-cpu-baseline", "--no-driver-window")
    d = run_bench(*args, env={"AA_FORCE_DISTRIBUTED": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0",
                              "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    one = run_bench(*args)
    assert d["n_gpus"] == 1 and d["state_check"]["ok"] is True
    assert d["config"]["subcycle_trace"] == one["config"]["subcycle_trace"] and d["config"]["final_dt"] == one["config"]["final_dt"]
    assert d["state_check"]["mass_after"] == one["state_check"]["mass_after"]
    assert d["host_syncs_per_subcycle"] <= 1.0 + 1e-9


def test_two_ranks_as_pencils():
    """bench.py --gpus 2 --p2 2: the two ranks cut x2 instead of x3 (x2 x x3 pencils 2x1), rehearsed on the one GPU over gloo."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = run_bench("--gpus", "2", "--p2", "2", "--nx", "64", "--steps", "2", "--warmup", "1", "--spinup", "2", "--scaling", "strong",
                  env={"AA_BENCH_REHEARSAL": "1"},
                  launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port)))
    one = run_bench("--nx", "64", "--steps", "2", "--warmup", "1", "--spinup", "2", "--no-cpu-baseline", "--no-driver-window")
    assert d["n_gpus"] == 2 and "pencils 2x1" in d["config"]["partition"] and d["state_check"]["ok"] is True
    assert d["config"]["subcycle_trace"] == one["config"]["subcycle_trace"]
    assert abs(d["state_check"]["mass_after"] / one["state_check"]["mass_after"] - 1) < 1e-9


def test_driver_path_window_beside_the_aa_step_line():
    """N = 1 runs ONE C call per step (aa_step); the ranks of an N > 1 job run driver.Driver.step -- Python between the phases, the
    sub-cycle loop inside the library with the all-gather as its callback (aa_ion_radtransfer_3d_gather), new_dt's all-reduce.  The
    same invocation times that path too, on a one-rank RCCL communicator (`driver_path`), so that a 1 -> N curve can be read like
    for like: same sub-cycle counts, ONE host round trip per sub-cycle, and what the host side costs per step."""
    d = run_bench("--nx", "64", "--steps", "3", "--warmup", "1", "--spinup", "2", "--burst-window", "--no-cpu-baseline")
    p = d["driver_path"]
    assert "error" not in p, p
    assert p["steps"] == 3 and p["ms_per_step"] > 0 and p["nsub"] == d["config"]["radiation_subcycles_per_step"]
    assert abs(p["host_path_overhead_ms"] - (p["ms_per_step"] - d["ms_per_step"])) < 1e-9
    assert p["host_syncs_per_subcycle"] <= 1.0 + 1e-9 and "nccl" in p["what"]
    assert p["burst"]["nsub"] == d["regimes"]["burst"]["nsub"] and p["burst"]["host_syncs_per_subcycle"] <= 1.0 + 1e-9
    # the sub-cycle loop as Python wrote it before (three crossings per sub-cycle) still gives the same run
    q = run_bench("--nx", "64", "--steps", "3", "--warmup", "1", "--spinup", "2", "--no-burst", "--no-cpu-baseline",
                  env={"AA_DRIVER_PY_SUBCYCLES": "1"})["driver_path"]
    assert q["nsub"] == p["nsub"] and q["host_syncs_per_subcycle"] <= 1.0 + 1e-9


def test_traffic_is_quoted_with_its_source():
    """roofline.traffic comes from the committed PMC passes of this command only while the kernel sources still have the fingerprint
    recorded with them (`traffic_source`); --pmc-pass counts it in the run itself (children under rocprofv3)."""
    import shutil
    d = run_bench("--nx", "64", "--steps", "2", "--warmup", "1", "--spinup", "2", "--no-burst", "--no-cpu-baseline", "--no-driver-window")
    assert d["roofline"]["traffic"] is None                     # (no profile of a 64^3 workload is committed)
    if shutil.which("rocprofv3"):
        d = run_bench("--nx", "64", "--steps", "2", "--warmup", "1", "--spinup", "2", "--no-burst", "--no-cpu-baseline", "--no-driver-window",
                      "--pmc-pass")
        r = d["roofline"]
        assert r["traffic_source"]["kind"].startswith("pmc-pass") and r["traffic"] > r["bytes_per_launch"] > 0
        assert r["traffic"] < 40 * r["bytes_per_launch"]
