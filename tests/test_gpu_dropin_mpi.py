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
