"""The reference's own MPI build on the GPU library: `mpiexec -n N athena_<cfg>_mpi_amd` (oracle/Makefile.ref dropin_mpi: the
unmodified --enable-mpi objects minus the hot path, linked on host/athena_shim.c compiled with -DAA_MPI) against
`mpiexec -n N athena_<cfg>_mpi` (the all-CPU reference) on the same deck and the same NGrid_x2 x NGrid_x3 decomposition.
Every rank drives the GPU through its own shim instance; ghost zones between the Grids travel as in bvals_mhd.c:296-493
(device pack -> MPI -> device unpack), new_dt and the radiation sub-cycle reduce with MPI_Allreduce where the reference
does.  On the one-GPU test box the ranks share the device (at most 4 of them here)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
REFBIN = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.parametrize("problem,nx,nlim,grid", [
    ("blast", (24, 16, 16), 5, (2, 2)),              # periodic in x2 and x3, two Grids each way: every rank's neighbour is the same Grid on both sides
    ("blast", (16, 12, 24), 4, (1, 3)),              # a ring of three along x3
    ("ioniz_sphere", (32, 32, 32), 3, (2, 2)),       # outflow sides + cut faces, gravity tables from each Grid's own MinX, Userwork core on two ranks' blocks
    ("ioniz_sphere", (64, 32, 32), 3, (1, 2)),       # rays of 64 zones (the two-kernel sub-cycle is forced under MPI)
])
def test_reference_mpi_ranks_on_the_gpu_library(problem, nx, nlim, grid):
    import refmpi
    cfg = "blast_mpi" if problem == "blast" else "ioniz_sphere_mpi"
    amd = os.path.join(REFBIN, f"athena_{cfg}_amd")
    cpu = os.path.join(REFBIN, f"athena_{cfg}")
    if not (os.path.exists(amd) and os.path.exists(cpu) and os.path.exists(refmpi.MPIEXEC)):
        pytest.skip("oracle/_ref MPI executables or mpiexec not on this box")
    ref = refmpi.run(nx, nlim, exe=cpu, problem=problem, grid=grid)
    gpu = refmpi.run(nx, nlim, exe=amd, problem=problem, grid=grid)
    assert gpu["stderr"].count("on HIP device") == grid[0] * grid[1]                 # every rank made its own Grid
    assert gpu["niter"] == ref["niter"]
    assert abs(gpu["time"] / ref["time"] - 1) < 1e-10 and abs(gpu["dt"] / ref["dt"] - 1) < 1e-10
    nv = 5 if problem == "blast" else 6
    a, b = gpu["U"][..., :nv], ref["U"][..., :nv]
    assert np.array_equal(np.isnan(a), np.isnan(b))
    scale = np.nanmax(np.abs(b), axis=(0, 1, 2)); scale[scale == 0] = 1
    err = np.nanmax(np.abs(a - b), axis=(0, 1, 2)) / scale
    assert err.max() < (1e-11 if problem == "blast" else 1e-8), err           # the drop-in links the default build; north_star: 1e-6


@pytest.mark.parametrize("problem,nlim,ov,nxs", [
    # the 2-level sphere of tests/golden/smr_ioniz_sphere_2lev_s4 with both Domains cut in two along x3: level 1 (root planes 10 .. 22)
    # is centred on the root's cut at plane 16, so every rank's level-1 Grid lies over its own root Grid
    ("ioniz_sphere", 4, ["domain1/Nx1=32", "domain1/Nx2=32", "domain1/Nx3=32", "domain2/Nx1=32", "domain2/Nx2=28", "domain2/Nx3=24",
                         "domain2/iDisp=16", "domain2/jDisp=18", "domain2/kDisp=20", "problem/rp=2.1e10"], [(32, 32, 32), (32, 28, 24)]),
    ("blast", 5, ["domain1/Nx1=32", "domain1/Nx2=32", "domain1/Nx3=32", "domain2/Nx1=32", "domain2/Nx2=24", "domain2/Nx3=32",
                  "domain2/iDisp=16", "domain2/jDisp=20", "domain2/kDisp=16"], [(32, 32, 32), (32, 24, 32)]),
])
def test_mpi_and_smr_together_on_the_gpu_library(problem, nlim, ov, nxs):
    """The reference's README.rst:25 configuration, --enable-mpi AND --enable-smr: `mpiexec -n 2 athena_<cfg>_smr_mpi_amd` (shim compiled
    with -DAA_MPI -DAA_SMR: every rank drives its stack of nested slabs as one aa_mesh, halo / new_dt / sub-cycle reductions through
    MPI in each Domain's communicator) against the all-CPU `athena_<cfg>_smr_mpi` on the same decomposition: identical sub-cycle
    counts on every level, time and dt, every level's state."""
    import refmpi
    cfg = "blast_smr_mpi" if problem == "blast" else "ioniz_sphere_smr_mpi"
    amd = os.path.join(REFBIN, f"athena_{cfg}_amd")
    cpu = os.path.join(REFBIN, f"athena_{cfg}")
    if not (os.path.exists(amd) and os.path.exists(cpu) and os.path.exists(refmpi.MPIEXEC)):
        pytest.skip("oracle/_ref MPI + SMR executables or mpiexec not on this box")
    ref = refmpi.run_smr(problem, ov, nlim, cpu, 2, nxs)
    gpu = refmpi.run_smr(problem, ov, nlim, amd, 2, nxs)
    assert gpu["stderr"].count("on HIP device") == 2 * len(nxs)                      # every rank made a Grid per level
    assert gpu["niter"] == ref["niter"], (gpu["niter"], ref["niter"])
    assert abs(gpu["time"] / ref["time"] - 1) < 1e-10 and abs(gpu["dt"] / ref["dt"] - 1) < 1e-10
    nv = 5 if problem == "blast" else 6
    for l in range(len(nxs)):
        a, b = gpu["U"][l][..., :nv], ref["U"][l][..., :nv]
        assert np.array_equal(np.isnan(a), np.isnan(b))
        scale = np.nanmax(np.abs(b), axis=(0, 1, 2)); scale[scale == 0] = 1
        err = np.nanmax(np.abs(a - b), axis=(0, 1, 2)) / scale
        assert err.max() < (1e-11 if problem == "blast" else 1e-8), (l, err)       # the drop-in links the default build; north_star: 1e-6
