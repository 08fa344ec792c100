"""The N>1 path (x3 slabs, halo exchange, scalar reductions) on CPU: world_size 2 and 3 over
gloo.  The product's Driver (atmospheric-athena_amd/driver.py) is run unchanged; only the
per-slab arithmetic engine is swapped for one backed by the CPU oracle, so what is tested is
the decomposition, the halo protocol (incl. the 2-rank periodic wrap) and the reduction
rounds.  Bar: every slab equals the corresponding part of the single-Grid oracle run, bit for
bit, for position-independent problems (same property the reference has under MPI)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


class OracleEngine:
    """Engine protocol of driver.py on top of oracle/liborc.so (test infrastructure)."""

    def __init__(self, grid):
        import torch
        import orc
        self.torch = torch
        self.cfg = grid
        self.s = orc.Sim(grid).problem()
        self.N = self.s.N
        nv = 5 + grid.run.nscal
        self.nv = nv
        n = self.N[0] * self.N[1] * 4 * nv
        self.recv = [torch.empty(n, dtype=torch.float64) for _ in range(2)]
        n2 = self.N[0] * 4 * (self.N[2] - 8) * nv
        self.recv2 = [torch.empty(n2, dtype=torch.float64) for _ in range(2)]

    def bvals_local(self): self.s.bvals()
    def bvals_side(self, d, side): self.s.bvals_side(d, side)
    def bvals_ionrad(self): self.s.bvals_ionrad()
    def new_dt_local(self): return self.s.new_dt_local()
    def integrate(self): self.s.integrate()
    def userwork(self): self.s.userwork()
    def ion_begin(self): self.s.ion_begin()
    def ion_rates(self): return self.s.ion_rates()

    def ion_update(self, dt):
        self.s.ion_update(dt)
        return self.s.ion_check_range_count(), self.s.ion_dt_hydro()

    def set_mesh_state(self, time, dt, nstep):
        self.s.time = time; self.s.dt = dt; self.s.nstep = nstep

    def has_radiation(self): return bool(self.cfg.run.ion)

    def history(self):
        hist = importlib.import_module("atmospheric-athena_amd.history")
        return hist.sums_from_block(self.s.active, self.cfg.run.dx, self.cfg.run.nscal)

    def pack_x3(self, side):
        k0 = 4 if side == 0 else self.N[2] - 8
        blk = self.s.U[k0:k0 + 4, :, :, :self.nv]                  # [kk][j][i][v]
        return self.torch.from_numpy(np.ascontiguousarray(blk.transpose(3, 0, 1, 2)).reshape(-1).copy())

    def recv_buffer(self, side): return self.recv[side]

    def unpack_x3(self, side):
        k0 = 0 if side == 0 else self.N[2] - 4
        blk = self.recv[side].numpy().reshape(self.nv, 4, self.N[1], self.N[0]).transpose(1, 2, 3, 0)
        self.s.U[k0:k0 + 4, :, :, :self.nv] = blk

    # x2 x x3 pencils: the x2 halo (bvals_mhd.c:2462 pack_ix2: four rows, all i incl. ghosts, the active k-planes)
    def pack_x2(self, side):
        j0 = 4 if side == 0 else self.N[1] - 8
        blk = self.s.U[4:-4, j0:j0 + 4, :, :self.nv]               # [kk][jj][i][v]
        return self.torch.from_numpy(np.ascontiguousarray(blk.transpose(3, 0, 1, 2)).reshape(-1).copy())

    def recv_buffer_x2(self, side): return self.recv2[side]

    def unpack_x2(self, side):
        j0 = 0 if side == 0 else self.N[1] - 4
        blk = self.recv2[side].numpy().reshape(self.nv, self.N[2] - 8, 4, self.N[0]).transpose(1, 2, 3, 0)
        self.s.U[4:-4, j0:j0 + 4, :, :self.nv] = blk

    def download(self): return self.s.U.copy()


class OracleFusedEngine(OracleEngine):
    """The pass / gather / pick / fetch protocol of the one-kernel radiation sub-cycle (include/athena_amd.h, aa_ion_pass
    ...; driver.Driver._ion_radtransfer_fused) emulated on the oracle's phases: pass n applies update(n-1) with the step
    the last pick chose and -- unless that step was cut back to the limit -- runs sweep(n) + rates(n); the slab's words go
    through ONE all-gather; the pick folds them, books the applied update and chooses the next step.  What is tested is
    the driver's control flow and its collectives, not the GPU's arithmetic."""
    ion_fused = True
    BIG = 1.7976931348623157e308

    def __init__(self, grid):
        super().__init__(grid)
        t = self.torch
        self.words = t.zeros(8, dtype=t.float64)
        self.words_all = t.zeros(8 * grid.nranks, dtype=t.float64)
        self.sc = dict(dt_sel=0.0, hit=False, dt_done=0.0, dt_applied=0.0, hit_applied=False, count=0, dt_hydro=self.BIG)

    def ion_pass(self, update, sweep):
        cnt, dth, dtc, dtt = 0, self.BIG, self.BIG, self.BIG
        if update:
            cnt, dth = OracleEngine.ion_update(self, self.sc["dt_sel"])
            if self.sc["hit"]:
                sweep = False                         # (on the device: the form of the pass without a further sweep stays)
        if sweep:
            dtc, dtt = self.s.ion_rates()
        self.words[:] = self.torch.tensor([dtc, dtt, dth, float(cnt), 0, 0, 0, 0], dtype=self.torch.float64)

    def ion_pick(self, dist, first, limit):
        if dist is None:
            self.words_all[:8] = self.words
            n = 1
        else:
            dist.all_gather_into_tensor(self.words_all, self.words)
            n = self.cfg.nranks
        W = self.words_all.tolist()
        dt_chem, dt_therm, dt_hydro, count = min(W[0::8][:n]), min(W[1::8][:n]), min(W[2::8][:n]), sum(W[3::8][:n])
        sc = self.sc
        if not first:
            sc["dt_applied"], sc["hit_applied"] = sc["dt_sel"], sc["hit"]
            sc["dt_done"] = sc["dt_done"] + sc["dt_sel"]
        else:
            sc["dt_done"] = 0.0
        sc["count"], sc["dt_hydro"] = int(count), dt_hydro
        dt = dt_therm if dt_therm < dt_chem else dt_chem
        hit = False
        if sc["dt_done"] + dt > limit:
            dt = limit - sc["dt_done"]; hit = True
        sc["dt_sel"], sc["hit"] = dt, hit

    def ion_fetch(self):
        sc = self.sc
        return sc["dt_applied"], sc["hit_applied"], 0.0, 0.0, sc["count"], sc["dt_hydro"], False

    def ion_finish(self): pass


def _worker(rank, world, port, problem, overrides, nsteps, q, fused=False, p2=1):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    aa = importlib.import_module("atmospheric-athena_amd")
    driver = importlib.import_module("atmospheric-athena_amd.driver")
    import orc
    run = aa.config.load(os.path.join(orc.DECKS, "athinput." + problem), overrides, problem)
    d = driver.Driver(run, OracleFusedEngine if fused else OracleEngine, rank, world, p2=p2)
    d.start()
    its = [d.step() for _ in range(nsteps)]
    if p2 > 1:
        q.put((rank, d.grid.disp, d.grid.Nx, d.eng.download()[4:-4, 4:-4, 4:-4].copy(), its, d.time, d.dt, d.history()))
    else:
        q.put((rank, d.grid.disp[2], d.grid.Nx[2], d.eng.download()[4:-4, 4:-4, 4:-4].copy(), its, d.time, d.dt, d.history()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def run_slabs(problem, overrides, nsteps, world, fused=False, p2=1):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, problem, overrides, nsteps, q, fused, p2)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


@pytest.mark.parametrize("problem,nx,nsteps,world,fused", [
    ("blast", (12, 10, 16), 3, 2, False),        # periodic in x3: the 2-rank wrap sends both halos to one peer
    ("blast", (10, 8, 18), 2, 3, False),         # periodic ring of 3, uneven split 6/6/6
    ("ifront", (16, 8, 12), 3, 2, False),        # outflow x3 + ion radiation: reductions every sub-cycle
    ("ifront", (8, 8, 14), 2, 3, False),         # remainder cells go to the first slabs: 5/5/4
    ("ifront", (16, 8, 12), 3, 2, True),         # the same through the pass / all-gather / pick / fetch protocol
    ("ifront", (8, 8, 14), 2, 3, True),
    ("ioniz_sphere", (20, 20, 20), 2, 2, True),
])
def test_slabs_equal_single_grid(problem, nx, nsteps, world, fused):
    import orc
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
    ref = orc.make_sim(problem, ov).start()
    its_ref = [ref.step() for _ in range(nsteps)]
    res = run_slabs(problem, ov, nsteps, world, fused)
    nv = 5 + ref.grid.run.nscal
    hist = importlib.import_module("atmospheric-athena_amd.history")
    href = hist.sums_from_block(ref.active, ref.grid.run.dx, ref.grid.run.nscal)
    for rank, disp, n3, U, its, t, dt, h in res:
        assert its == its_ref
        if problem == "ioniz_sphere":
            # a slab's MinX is accumulated (init_grid.c:104-111): positions -- hence potential and the Userwork core --
            # differ in the last bit from the single Grid's, as in the reference under MPI
            assert abs(dt / ref.dt - 1) < 1e-9
            a, b = U[..., :nv], ref.active[disp:disp + n3, :, :, :nv]
            assert np.array_equal(np.isnan(a), np.isnan(b))
            scale = np.nanmax(np.abs(ref.active[..., :nv]), axis=(0, 1, 2))
            assert (np.nanmax(np.abs(a - b), axis=(0, 1, 2)) / scale).max() < 1e-8
            continue
        assert t == ref.time and dt == ref.dt
        assert np.array_equal(U[..., :nv], ref.active[disp:disp + n3, :, :, :nv]), f"slab {rank} differs"
        # history sums: SUM over slabs == the single-Grid integrals (to summation-order rounding)
        assert np.array_equal(h, res[0][7]) and np.allclose(h[[0, 1, 5, 6, 7]], href[[0, 1, 5, 6, 7]], rtol=1e-12, atol=0)


@pytest.mark.parametrize("problem,nx,nsteps,p2,p3,fused", [
    ("blast", (12, 16, 16), 3, 2, 2, False),     # periodic in x2 and x3: both wraps send both halos to one peer; corners travel
    ("blast", (10, 18, 12), 2, 3, 1, False),     # x2 only, ring of 3
    ("ifront", (16, 12, 12), 3, 2, 2, False),    # outflow sides + cut faces on every Grid, reductions every sub-cycle
    ("ifront", (16, 14, 8), 2, 3, 1, True),      # remainder rows to the first Grids (5/5/4), the one-kernel protocol
    ("ioniz_sphere", (20, 20, 20), 2, 2, 2, True),
])
def test_pencils_equal_single_grid(problem, nx, nsteps, p2, p3, fused):
    """x2 x x3 pencil decomposition (init_mesh.c:526-620 with NGrid_x1 = 1: the rays stay whole): the product's Driver with
    p2 > 1 -- x1 sides, x2 exchange, x3 exchange in the order that carries the corners (bvals_mhd.c:170) -- on the oracle
    engine over gloo.  Every pencil equals its part of the single-Grid run bit for bit for position-independent problems."""
    import orc
    ov = [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)]
    ref = orc.make_sim(problem, ov).start()
    its_ref = [ref.step() for _ in range(nsteps)]
    res = run_slabs(problem, ov, nsteps, p2 * p3, fused, p2)
    nv = 5 + ref.grid.run.nscal
    seen = np.zeros(ref.active.shape[:3], dtype=bool)
    for rank, disp, n, U, its, t, dt, h in res:
        assert its == its_ref
        sl = (slice(disp[2], disp[2] + n[2]), slice(disp[1], disp[1] + n[1]), slice(None))
        seen[sl] = True
        a, b = U[..., :nv], ref.active[sl][..., :nv]
        if problem == "ioniz_sphere":      # MinX of a Grid is accumulated (init_grid.c:104-111): last-bit positions, as under MPI
            assert abs(dt / ref.dt - 1) < 1e-9 and np.array_equal(np.isnan(a), np.isnan(b))
            scale = np.nanmax(np.abs(ref.active[..., :nv]), axis=(0, 1, 2))
            assert (np.nanmax(np.abs(a - b), axis=(0, 1, 2)) / scale).max() < 1e-8
        else:
            assert t == ref.time and dt == ref.dt
            assert np.array_equal(a, b), f"pencil {rank} differs"
    assert seen.all()


def test_pencil_geometry(aa):
    """init_mesh.c:583-620 with NGrid_x2 x NGrid_x3: ranks x2-fastest, remainder cells to the first Grids, MinX accumulated."""
    cfg = aa.config
    run = cfg.load(os.path.join(os.path.dirname(HERE), "atmospheric-athena_amd", "decks", "athinput.blast"),
                   ["domain1/Nx2=22", "domain1/Nx3=16"], "blast")
    g = [cfg.pencil(run, r, 4, 2) for r in range(8)]
    assert [x.Nx[1] for x in g[:4]] == [6, 6, 5, 5] and [x.disp[1] for x in g[:4]] == [0, 6, 12, 17]
    assert [x.disp[2] for x in g] == [0, 0, 0, 0, 8, 8, 8, 8]
    assert (g[0].lx2, g[0].rx2, g[3].lx2, g[3].rx2) == (3, 1, 2, 0)        # periodic wrap inside the x2 row
    assert (g[5].lx2, g[5].rx2, g[5].lx3, g[5].rx3) == (4, 6, 1, 1)        # two Grids along x3, periodic: the same peer twice
    assert all(x.bc[2:] == (0, 0, 0, 0) and x.p2 == 4 and x.nranks == 8 for x in g)
    m = run.xmin[1]
    for r in range(3):
        m += float(g[r].Nx[1]) * run.dx[1]
        assert g[r + 1].MinX[1] == m
    assert cfg.pencil(run, 1, 1, 2) == cfg.slab(run, 1, 2)                 # p2 = 1 is the slab decomposition
    run2 = cfg.load(os.path.join(os.path.dirname(HERE), "atmospheric-athena_amd", "decks", "athinput.ifront"), [], "ifront")
    h = cfg.pencil(run2, 0, 2, 2)
    assert (h.lx2, h.rx2, h.lx3, h.rx3) == (-1, 1, -1, 2) and h.bc[2:] == (2, 0, 2, 0)
    with pytest.raises(aa.athinput.ParError):
        cfg.pencil(run2, 0, 32, 1)


def test_slab_geometry(aa):
    """init_mesh.c:583-620 / init_grid.c:104-111: cell split, MinX accumulation, neighbours."""
    cfg = aa.config
    run = cfg.load(os.path.join(os.path.dirname(HERE), "atmospheric-athena_amd", "decks", "athinput.blast"),
                   ["domain1/Nx3=22"], "blast")
    assert cfg.split_cells(22, 4) == [6, 6, 5, 5]
    g = [cfg.slab(run, r, 4) for r in range(4)]
    assert [x.disp[2] for x in g] == [0, 6, 12, 17]
    assert (g[0].lx3, g[0].rx3, g[3].lx3, g[3].rx3) == (3, 1, 2, 0)        # periodic wrap
    assert all(x.bc[4] == 0 and x.bc[5] == 0 for x in g)
    dx3 = run.dx[2]
    m = run.xmin[2]
    for r in range(3):
        m += float(g[r].Nx[2]) * dx3
        assert g[r + 1].MinX[2] == m
    run2 = cfg.load(os.path.join(os.path.dirname(HERE), "atmospheric-athena_amd", "decks", "athinput.ifront"), [], "ifront")
    h = [cfg.slab(run2, r, 2) for r in range(2)]
    assert (h[0].lx3, h[0].rx3, h[1].lx3, h[1].rx3) == (-1, 1, 0, -1)
    assert h[0].bc[4] == 2 and h[0].bc[5] == 0 and h[1].bc[4] == 0 and h[1].bc[5] == 2
    with pytest.raises(aa.athinput.ParError):
        cfg.slab(run2, 0, 32)                                               # slabs thinner than nghost
