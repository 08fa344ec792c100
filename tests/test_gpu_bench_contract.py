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
    # the kernel reported is the one with the largest total time among those that have a byte figure
    sys.path.insert(0, ROOT)
    import bench
    timed = {k: v for k, v in d["kernel_ms_per_step"].items() if bench.KERNEL_BYTES.get(k, 0) > 0}
    assert r["kernel"] == max(timed, key=timed.get)
    assert set(d["kernel_ms_per_step"]) <= set(bench.KERNEL_BYTES) | {"halo_pack", "halo_unpack"}, "a kernel without a byte figure"
    ph = d["phases"]
    assert ph["hydro"]["bytes_per_cell"] == 96 and ph["subcycle"]["bytes_per_cell"] == 64
    assert ph["hydro"]["ms_per_step"] > 0 and ph["subcycle"]["ms_per_subcycle"] > 0
    assert d["state_check"]["finite"] is True and d["state_check"]["ok"] is True
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference-mpi", "reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "cell-updates/s" and c["sample"]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "athena_ioniz_sphere_mpi")) and os.path.exists("/opt/conda/bin/mpiexec"):
        assert c["kind"] == "reference-mpi" and c["cores"] > 1, c
        assert c["per_hydro_step_ns_per_zone"] > 0 and "per_subcycle_ns_per_zone" in c and "gpu_vs_cpu" in d


def test_other_workloads_keep_the_contract():
    for args in (("--problem", "blast", "--nx", "48"), ("--smr", "--nx", "32"), ("--integrator", "vl", "--nx", "48"), ("--order", "3", "--nx", "48"),
                 ("--strict", "--nx", "48")):
        d = run_bench(*args, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--spinup", "1")
        assert d["value"] > 0 and d["roofline"]["achieved"] > 0 and "workload" in d["config"]


@pytest.mark.parametrize("strong", [False, True])
def test_two_ranks_rehearsed_on_one_gpu(strong):
    """The N>1 path of bench.py (torchrun, one rank per GPU) with both ranks on cuda:0 over gloo: not a measurement,
    but every collective of the real run is issued.  The radiation sub-cycle may cost ONE host round trip."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = run_bench("--gpus", "2", "--nx", "64", "--steps", "2", "--warmup", "1", "--spinup", "2", *(("--strong",) if strong else ()),
                  env={"AA_BENCH_REHEARSAL": "1"},
                  launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port)))
    assert d["n_gpus"] == 2 and d["scaling"] == ("strong" if strong else "weak")
    assert ("64x64x64" if strong else "64x64x128") in d["config"]["workload"]
    assert d["config"]["zones_per_gpu"] == (64 ** 3 // 2 if strong else 64 ** 3)
    assert d["state_check"]["ok"] is True
    assert d["host_syncs_per_subcycle"] <= 1.0 + 1e-9, d["host_syncs_per_subcycle"]


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
    args = ("--nx", "64", "--steps", "2", "--warmup", "1", "--spinup", "2", "--no-cpu-baseline")
    d = run_bench(*args, env={"AA_FORCE_DISTRIBUTED": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0",
                              "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    one = run_bench(*args)
    assert d["n_gpus"] == 1 and d["state_check"]["ok"] is True
    assert d["config"]["subcycle_trace"] == one["config"]["subcycle_trace"] and d["config"]["final_dt"] == one["config"]["final_dt"]
    assert d["state_check"]["mass_after"] == one["state_check"]["mass_after"]
    assert d["host_syncs_per_subcycle"] <= 1.0 + 1e-9
