"""Main loop of the reference (main.c:519-669) over a root Domain cut into x3 slabs, one slab
per process / GPU, with the reference's MPI calls replaced by torch.distributed (backend
"nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests):

  * bvals_mhd's x3 Isend/Irecv pair (bvals_mhd.c:423-493)  -> batched isend/irecv of the packed
    4-plane halo (all i and j incl. ghosts so corners travel, bvals_mhd.c:170), both directions
    at once; periodic wrap = neighbour rank +-1 mod N exactly as the reference rewires lx/rx ids
  * MPI_Allreduce(MIN) of dt in new_dt (new_dt.c:177) and of dt_chem / dt_therm / dt_hydro, and
    SUM of the out-of-range cell count (ionrad_3d.c:275,399,554,672) -> all_reduce on small
    tensors; the four per-sub-cycle reductions collapse into two rounds
  * the domain is never cut along x1 (the ray direction), so the reference's rank pipeline of
    get_ph_rate_plane (ionradplane_3d.c:226-400) does not exist here.

All reductions are MIN/MAX of doubles or integer sums, hence bitwise independent of the
decomposition: an N-slab run reproduces the 1-slab run exactly for position-independent
problems (the reference has the same property).

The per-slab arithmetic is behind the small ``Engine`` interface; the product engine is
``HipEngine`` (the C-ABI library).  With one rank and no engine override the whole loop runs
inside the C library (aa_step) with no Python between kernels.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np

from .config import pencil, GridConfig, RunConfig, slab

MAXCELLCOUNT = 20   # ionrad.h:38


class HipEngine:
    """One slab on one MI355X, through include/athena_amd.h."""

    def __init__(self, grid: GridConfig, device: int = 0, strict: Optional[bool] = None, use_torch_stream: bool = True,
                 nslab: int = 1):
        import torch
        from . import lib
        self.torch = torch
        self.cfg = grid
        torch.cuda.set_device(device)
        # nslab > 1: the library cuts this Grid into x3 slabs itself (csrc/slabs.hip; one stream per slab, so the
        # caller's stream is not handed over)
        self.g = lib.setup_problem(grid, device, strict, nslab=nslab)
        if use_torch_stream and nslab == 1:
            # run the kernels on torch's current stream so that torch.distributed collectives and
            # torch.cuda.Event timing are ordered with them
            self.g.set_stream(torch.cuda.current_stream().cuda_stream)
        # Driver.step calls integrate -> userwork (pinned zones only) -> new_dt in that order: the update kernel may
        # leave new_dt's maxima behind (aa_cfl_in_update; AA_CFL_FUSED=0 keeps the separate sweep over the Grid)
        self.g.cfl_in_update(os.environ.get("AA_CFL_FUSED", "1") != "0")
        n = self.g.halo_doubles() if nslab == 1 else 0        # (slabs inside the library exchange their halos themselves)
        dev = torch.device("cuda", device)
        # the one-kernel radiation sub-cycle leaves this slab's reduction words in device memory; with several
        # ranks they are all-gathered on the stream (ONE collective per sub-cycle, no host round trip)
        self.ion_fused = bool(grid.run.ion) and self.g.ion_is_fused()
        self.words = torch.zeros(lib.ION_WORDS, dtype=torch.float64, device=dev)
        self.words_all = torch.zeros(lib.ION_WORDS * max(1, grid.nranks), dtype=torch.float64, device=dev)
        self.send = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(2)]
        self.recv = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(2)]
        n2 = self.g.halo_doubles_x2() if (nslab == 1 and grid.p2 > 1) else 0      # x2 x x3 pencils: the x2 halo as well
        self.send2 = [torch.empty(n2, dtype=torch.float64, device=dev) for _ in range(2)]
        self.recv2 = [torch.empty(n2, dtype=torch.float64, device=dev) for _ in range(2)]
        self.scalar_device = dev

    # slab arithmetic
    def bvals_local(self): self.g.bvals_mhd()
    def bvals_ionrad(self): self.g.bvals_ionrad()
    def new_dt_local(self) -> float: return self.g.new_dt_local()
    def integrate(self): self.g.integrate()
    def integrate_begin(self): self.g.integrate_begin()
    def userwork(self): self.g.apply_pinned_cells()
    def ion_begin(self): self.g.ion_begin()
    def ion_speculate(self, limit): self.g.ion_speculate(limit)
    def ion_rates(self): return self.g.ion_rates()
    def ion_update(self, dt): return self.g.ion_update(dt)
    def ion_pass(self, update, sweep): self.g.ion_pass(update, sweep, self.words.data_ptr())

    def ion_pick(self, dist, first, limit):
        """the step of the sub-cycle from the words of all slabs (ionrad_3d.c:399,:554,:275,:672 in one round)"""
        if dist is None:
            self.g.ion_pick(self.words.data_ptr(), 1, first, limit)
            return
        if dist.get_backend() == "gloo":                     # rehearsal on one GPU: gloo moves host tensors
            w = self.words.cpu(); wa = self.words_all.cpu()
            dist.all_gather_into_tensor(wa, w)
            self.words_all.copy_(wa)
        else:
            dist.all_gather_into_tensor(self.words_all, self.words)
        self.g.ion_pick(self.words_all.data_ptr(), self.cfg.nranks, first, limit)

    def ion_fetch(self): return self.g.ion_fetch()
    def ion_finish(self): self.g.ion_finish()

    def ion_radtransfer_gather(self, dist) -> int:
        """the whole sub-cycle loop inside the library (aa_ion_radtransfer_3d_gather): the only thing left to this side is the
        all-gather of the slabs' words, once per pass -- ONE crossing per sub-cycle, as a one-rank run has none"""
        if dist.get_backend() == "gloo":                     # rehearsal on one GPU: gloo moves host tensors
            def gather():
                w = self.words.cpu(); wa = self.words_all.cpu()
                dist.all_gather_into_tensor(wa, w)
                self.words_all.copy_(wa)
        else:
            def gather():
                dist.all_gather_into_tensor(self.words_all, self.words)
        return self.g.ion_radtransfer_3d_gather(self.words.data_ptr(), self.words_all.data_ptr(), self.cfg.nranks, gather)

    def host_syncs(self, reset=False): return self.g.host_syncs(reset)
    def set_mesh_state(self, time, dt, nstep): self.g.set_mesh_state(time, dt, nstep)
    def has_radiation(self) -> bool: return self.g.has_radplane()      # main.c:546, as aa_step: the ion step runs iff nradplane > 0
    def step_local(self) -> int: return self.g.step()
    def start_local(self): self.g.start()
    def mesh_state(self): return self.g.mesh_state()

    # halo
    def pack_x3(self, side: int):
        self.g.pack_x3(side, self.send[side].data_ptr()); return self.send[side]

    def recv_buffer(self, side: int): return self.recv[side]
    def unpack_x3(self, side: int): self.g.unpack_x3(side, self.recv[side].data_ptr())

    def pack_x2(self, side: int):
        self.g.pack_x2(side, self.send2[side].data_ptr()); return self.send2[side]

    def recv_buffer_x2(self, side: int): return self.recv2[side]
    def unpack_x2(self, side: int): self.g.unpack_x2(side, self.recv2[side].data_ptr())
    def bvals_side(self, dir: int, side: int): self.g.bvals_mhd_side(dir, side)
    def sync(self): self.g.sync()
    def download(self) -> np.ndarray: return self.g.download()
    def history(self) -> np.ndarray: return self.g.history()
    def close(self): self.g.close()


class Driver:
    """main() of the reference for one process of an N-process run."""

    def __init__(self, run: RunConfig, engine_factory=None, rank: int = 0, nranks: int = 1, device: int = 0,
                 strict: Optional[bool] = None, p2: int = 1):
        """p2 > 1: an NGrid_x2 x NGrid_x3 = p2 x (nranks / p2) pencil decomposition (init_mesh.c:526-620) instead of x3 slabs"""
        self.run = run
        self.rank, self.nranks = rank, nranks
        if p2 < 1 or nranks % p2:
            raise ValueError(f"{nranks} ranks cannot be dealt {p2} along x2")
        self.grid = pencil(run, rank, p2, nranks // p2)
        self.eng = engine_factory(self.grid) if engine_factory else HipEngine(self.grid, device, strict)
        self.time, self.dt, self.nstep = 0.0, 0.0, 0
        self.niter_trace: List[int] = []
        self._halo = {}           # messages in flight per axis (2: x2, 3: x3): post .. finish
        self._py_syncs = 0        # host round trips of collectives issued from here (bench: host_syncs_per_step)
        # AA_FORCE_DISTRIBUTED=1 runs the Python-orchestrated loop (with its collectives) even on one
        # rank: used to rehearse the N>1 code path on a single GPU
        self.distributed = nranks > 1 or bool(os.environ.get("AA_FORCE_DISTRIBUTED"))
        if self.distributed:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            assert dist.is_initialized() and dist.get_world_size() == nranks and dist.get_rank() == rank
            self._sdev = getattr(self.eng, "scalar_device", torch.device("cpu"))

    # ---- collectives ------------------------------------------------------------------
    def _allreduce(self, vals, op):
        if not self.distributed:
            return list(vals)
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self._sdev)
        self.dist.all_reduce(t, op=op)
        self._py_syncs += 1
        return t.tolist()

    def host_sync_count(self, reset: bool = False) -> int:
        """Times the host waited for the device to hand back scalars (library read-backs + collectives' .tolist())."""
        n = self._py_syncs + (self.eng.host_syncs(reset) if hasattr(self.eng, "host_syncs") else 0)
        if reset:
            self._py_syncs = 0
        return n

    def _min(self, *vals): return self._allreduce(vals, self.dist.ReduceOp.MIN if self.distributed else None)

    def history(self) -> np.ndarray:
        """dump_history.c:157-260: this slab's volume integrals, added over the slabs of the Domain
        (the reference's MPI_Reduce(SUM) :257; here an all-reduce so every rank may write)."""
        s = self.eng.history()
        if self.distributed:
            s = np.array(self._allreduce(s, self.dist.ReduceOp.SUM))
        return s

    def dump_history(self, writer):
        """One row of the .hst file (history.HistoryWriter); call it where main.c calls data_output."""
        vol = 1.0
        for d in range(3):
            vol *= self.run.xmax[d] - self.run.xmin[d]
        s = self.history()
        if self.rank == 0:
            writer.dump(self.time, self.dt, s, vol, self.run.nscal)

    def exchange_x3(self):
        """bvals_mhd.c:423-493 for the x3 direction."""
        self.post_x3()
        self.finish_x3()

    def post_x3(self): self._post(3)
    def finish_x3(self): self._finish(3)

    def _post(self, axis: int):
        """The sends and receives of bvals_mhd.c:296-493 along one direction (pack_i* / pack_o*, MPI_Isend / MPI_Irecv)
        without the wait; axis 2 = x2 (pencils only), 3 = x3.  With RCCL the messages travel on the communicator's stream
        from here on; _finish() makes the kernel stream wait for them and unpacks.  What is queued in between must not
        touch the ghost zones of that direction or the buffers."""
        assert axis not in self._halo
        g, dist = self.grid, self.dist if self.distributed else None
        lo, hi = (g.lx2, g.rx2) if axis == 2 else (g.lx3, g.rx3)
        if not self.distributed or (lo < 0 and hi < 0):
            return
        e = self.eng
        pack = e.pack_x2 if axis == 2 else e.pack_x3
        recvb = e.recv_buffer_x2 if axis == 2 else e.recv_buffer
        host_stage = (dist.get_backend() == "gloo")   # gloo moves host tensors only

        def out(side):
            t = pack(side)
            return t.cpu() if (host_stage and t.is_cuda) else t

        rbuf = {}

        def inn(side):
            t = recvb(side)
            rbuf[side] = (self.torch.empty(t.shape, dtype=t.dtype) if (host_stage and t.is_cuda) else t)
            return rbuf[side]

        tag0 = 10 * axis
        if lo == hi and lo >= 0:
            # two Grids along this direction with periodic wrap: both messages go to the same peer; post them so that the
            # peer's inner planes (its first send) land in my OUTER ghosts (my first receive)
            ops = [dist.P2POp(dist.isend, out(0), lo, tag=tag0),
                   dist.P2POp(dist.isend, out(1), hi, tag=tag0 + 1),
                   dist.P2POp(dist.irecv, inn(1), hi, tag=tag0),
                   dist.P2POp(dist.irecv, inn(0), lo, tag=tag0 + 1)]
        else:
            ops = []
            # my inner planes fill the lower neighbour's outer ghosts, and vice versa
            if lo >= 0:
                ops.append(dist.P2POp(dist.isend, out(0), lo, tag=tag0))
                ops.append(dist.P2POp(dist.irecv, inn(0), lo, tag=tag0 + 1))
            if hi >= 0:
                ops.append(dist.P2POp(dist.isend, out(1), hi, tag=tag0 + 1))
                ops.append(dist.P2POp(dist.irecv, inn(1), hi, tag=tag0))
        self._halo[axis] = (dist.batch_isend_irecv(ops), rbuf)

    def _finish(self, axis: int):
        """MPI_Waitall + unpack_i* / unpack_o* of bvals_mhd.c:296-493."""
        if axis not in self._halo:
            return
        works, rbuf = self._halo.pop(axis)
        e = self.eng
        recvb = e.recv_buffer_x2 if axis == 2 else e.recv_buffer
        unpack = e.unpack_x2 if axis == 2 else e.unpack_x3
        for w in works:
            w.wait()
        for side in (0, 1):
            if side in rbuf:
                dst = recvb(side)
                if rbuf[side] is not dst:
                    dst.copy_(rbuf[side])
                unpack(side)

    # ---- the reference's call sites ---------------------------------------------------------
    def bvals_mhd(self, exchange: bool = True, wait: bool = True):
        if self.grid.p2 == 1:
            self.eng.bvals_local()      # x1, x2 and physical x3 faces
        else:
            # pencils: x1, then x2 (physical sides here, cut faces by messages), then x3 -- the order that carries the
            # corners (bvals_mhd.c:170); the x2 messages must have landed before x3 is packed
            for side in (0, 1):
                self.eng.bvals_side(0, side)
            for side in (0, 1):
                self.eng.bvals_side(1, side)
            if exchange:
                self._post(2)
                self._finish(2)
            for side in (0, 1):
                self.eng.bvals_side(2, side)
        if exchange:
            self.post_x3()
            if wait:
                self.finish_x3()

    def new_dt(self):               # new_dt.c:169-185
        dtc = self._min(self.eng.new_dt_local())[0] if self.distributed else self.eng.new_dt_local()
        self.dt = dtc if self.nstep == 0 else min(2.0 * self.dt, dtc)
        if self.time < self.run.tlim and (self.run.tlim - self.time) < self.dt:
            self.dt = self.run.tlim - self.time
        self.eng.set_mesh_state(self.time, self.dt, self.nstep)

    def _ion_radtransfer_fused(self) -> int:
        """ionrad_3d.c:862-1047 with the one-kernel sub-cycle (include/athena_amd.h, aa_ion_pass): the loop is cut at
        the reduction that yields the step, so a sub-cycle is one pass, ONE all-gather of the slabs' words and one
        read-back; the sweep after a data-dependent stop is speculative and dropped by ion_finish."""
        e = self.eng
        dist = self.dist if self.distributed else None
        if dist is not None and hasattr(e, "ion_radtransfer_gather") and not os.environ.get("AA_DRIVER_PY_SUBCYCLES"):
            # the loop below inside the library, the all-gather as its callback (AA_DRIVER_PY_SUBCYCLES=1: the loop as written here)
            niter = e.ion_radtransfer_gather(dist)
            self.dt = e.mesh_state()[1]
            return niter
        dt_done, niter = 0.0, 0
        e.ion_begin()
        if hasattr(e, "ion_speculate"):
            e.ion_speculate(self.dt)        # the first pass may already apply the first update with the whole step
        e.ion_pass(False, True)
        e.ion_pick(dist, True, self.dt)
        while True:
            e.ion_pass(True, True)          # the pass skips its sweep by itself once the step was cut back to the limit
            e.ion_pick(dist, False, self.dt)
            dt, hit, dt_chem, dt_therm, cellcount, dt_hydro, neg = e.ion_fetch()       # the one read-back
            if neg:
                raise RuntimeError("[compute_chem_rates]: negative dt_chem")           # ionrad_3d.c:389-391
            dt_done += dt
            niter += 1
            if cellcount > MAXCELLCOUNT:
                self.dt = dt_done
                break
            if hit:
                break
            if dt_hydro < dt_done:
                self.dt = dt_done
                break
        e.ion_finish()
        if niter == self.run.maxiter:
            self.dt = dt_done
        if self.dt < 0:
            raise RuntimeError(f"[ion_radtransfer_3d]: dt = {self.dt}")
        e.set_mesh_state(self.time, self.dt, self.nstep)
        return niter

    def ion_radtransfer(self) -> int:   # ionrad_3d.c:862-1047, root level
        e = self.eng
        if getattr(e, "ion_fused", False):
            return self._ion_radtransfer_fused()
        dt_done, niter, hydro_done = 0.0, 0, False
        e.ion_begin()
        while not hydro_done:
            dt_chem, dt_therm = e.ion_rates()
            if self.distributed:
                dt_chem, dt_therm = self._allreduce((dt_chem, dt_therm), self.dist.ReduceOp.MIN)
            dt = min(dt_therm, dt_chem)
            if dt_done + dt > self.dt:
                dt = self.dt - dt_done
                hydro_done = True
            cellcount, dt_hydro = e.ion_update(dt)
            if self.distributed:
                # one round: SUM of the count and MIN of dt_hydro (as MAX of its negative)
                t = self.torch.tensor([float(cellcount), 0.0], dtype=self.torch.float64, device=self._sdev)
                h = self.torch.tensor([dt_hydro], dtype=self.torch.float64, device=self._sdev)
                w1 = self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=True)
                w2 = self.dist.all_reduce(h, op=self.dist.ReduceOp.MIN, async_op=True)
                w1.wait(); w2.wait()
                cellcount, dt_hydro = int(t[0].item()), float(h[0].item())
            dt_done += dt
            niter += 1
            if cellcount > MAXCELLCOUNT:
                self.dt = dt_done
                break
            if hydro_done:
                break
            if dt_hydro < dt_done:
                self.dt = dt_done
                break
        if niter == self.run.maxiter:
            self.dt = dt_done
        if self.dt < 0:
            raise RuntimeError(f"[ion_radtransfer_3d]: dt = {self.dt}")
        e.set_mesh_state(self.time, self.dt, self.nstep)
        return niter

    # ---- main.c ---------------------------------------------------------------------------------
    def start(self):                # main.c:412-451
        if not self.distributed and hasattr(self.eng, "start_local"):
            self.eng.start_local()
            self.time, self.dt, self.nstep = self.eng.mesh_state()
            return
        self.eng.set_mesh_state(self.time, self.dt, self.nstep)
        self.bvals_mhd()
        self.eng.bvals_ionrad()
        self.new_dt()

    def step(self) -> int:          # main.c:519-669
        if not self.distributed and hasattr(self.eng, "step_local"):
            niter = self.eng.step_local()
            self.time, self.dt, self.nstep = self.eng.mesh_state()
            self.niter_trace.append(niter)
            return niter
        niter = 0
        if self.eng.has_radiation():
            niter = self.ion_radtransfer()
            # the x3 halo travels while the first-pass x1 / x2 sweeps of the planes ks..ke run (they read no
            # neighbour data: aa_integrate_begin); the rest of the integrator follows the unpack
            self.bvals_mhd(wait=False)
            if hasattr(self.eng, "integrate_begin"):
                self.eng.integrate_begin()
            self.finish_x3()
        self.eng.integrate()
        self.eng.userwork()
        self.nstep += 1
        self.time += self.dt
        self.new_dt()
        # main.c:635-644.  With radiation on, nothing reads the neighbour's planes before the bvals_mhd
        # that follows the next ion step (the ion kernels, new_dt and the dumps touch active zones only,
        # and that call re-sends every field anyway): the x3 halo of this call is left to it.
        self.bvals_mhd(exchange=not self.eng.has_radiation())
        self.niter_trace.append(niter)
        return niter


# ==================================================================================================
# Static mesh refinement over several GPUs
# ==================================================================================================
class HipMeshEngine:
    """One rank's stack of x3 slabs (one per level present on the rank) on one MI355X."""

    def __init__(self, cfg, device: int = 0, strict: Optional[bool] = None, use_torch_stream: bool = True):
        import torch
        from . import lib
        self.torch = torch
        self.cfg = cfg
        torch.cuda.set_device(device)
        self.mesh = lib.Mesh(cfg.levels, device, strict, links=cfg.links)
        if use_torch_stream:
            self.mesh.set_stream(torch.cuda.current_stream().cuda_stream)
        self.lev = self.mesh.lev
        dev = torch.device("cuda", device)
        self.scalar_device = dev
        self.send = [[torch.empty(g.halo_doubles(), dtype=torch.float64, device=dev) for _ in range(2)] for g in self.lev]
        self.recv = [[torch.empty(g.halo_doubles(), dtype=torch.float64, device=dev) for _ in range(2)] for g in self.lev]
        self._fbuf = {}
        # one-kernel radiation sub-cycle (levels with rays of 48 zones or more): this rank's reduction words of the level
        # being stepped, and those of all ranks, in device memory
        self.words = torch.zeros(lib.ION_WORDS, dtype=torch.float64, device=dev)
        self.words_all = torch.zeros(lib.ION_WORDS * max(1, cfg.nranks), dtype=torch.float64, device=dev)
        self.neutral = torch.tensor([1.7976931348623157e308, 1.7976931348623157e308, 0, 0, 0, 0, 0, 0], dtype=torch.float64, device=dev)

    nlev = property(lambda s: len(s.lev))

    # per-level slab arithmetic
    def bvals_local(self, l): self.lev[l].bvals_mhd()
    def bvals_ionrad(self, l): self.lev[l].bvals_ionrad()
    def integrate(self, l): self.lev[l].integrate()
    def userwork(self, l): self.lev[l].apply_pinned_cells()
    def ion_begin(self, l): self.lev[l].ion_begin()
    def ion_speculate(self, l, limit): self.lev[l].ion_speculate(limit)
    def ion_rates(self, l): return self.lev[l].ion_rates()
    def ion_update(self, l, dt): return self.lev[l].ion_update(dt)
    def ion_is_fused(self, l): return self.lev[l].ion_is_fused()
    def ion_pass(self, l, update, sweep): self.lev[l].ion_pass(update, sweep, self.words.data_ptr())
    def ion_neutral_words(self): self.words.copy_(self.neutral)          # a rank that holds no zones of the level

    def ion_gather(self, dist):
        """all ranks' words of the level being stepped -> self.words_all (device); one collective per sub-cycle"""
        if dist is None:
            self.words_all[:self.words.numel()].copy_(self.words)
        elif dist.get_backend() == "gloo":                       # rehearsal on one GPU: gloo moves host tensors
            w = self.words.cpu(); wa = self.words_all.cpu()
            dist.all_gather_into_tensor(wa, w)
            self.words_all.copy_(wa)
        else:
            dist.all_gather_into_tensor(self.words_all, self.words)

    def ion_pick(self, l, nranks, first, limit): self.lev[l].ion_pick(self.words_all.data_ptr(), nranks, first, limit)
    def ion_fetch(self, l): return self.lev[l].ion_fetch()
    def ion_finish(self, l): self.lev[l].ion_finish()
    def set_level_state(self, l, time, dt, nstep): self.lev[l].set_mesh_state(time, dt, nstep)
    def cfl_max_v(self, l): return self.lev[l].cfl_max_v()
    def has_radiation(self) -> bool: return bool(self.cfg.levels[0].run.ion)

    # x3 halo of level l
    def pack_x3(self, l, side):
        self.lev[l].pack_x3(side, self.send[l][side].data_ptr()); return self.send[l][side]

    def recv_buffer(self, l, side): return self.recv[l][side]
    def unpack_x3(self, l, side): self.lev[l].unpack_x3(side, self.recv[l][side].data_ptr())

    # level coupling inside the rank
    def restrict_correct_pair(self, l): self.mesh.restrict_correct_pair(l)
    def ion_restrict_correct(self): self.mesh.ionradRestrictCorrect()
    def prolongate(self): self.mesh.Prolongate()
    def ionflux_prolong(self, l): self.mesh.ionflux_prolong(l)

    # flux correction across a cut
    def flux_buffer(self, n1, n2, side=0):
        key = (n1, n2, side)
        if key not in self._fbuf:
            self._fbuf[key] = self.torch.empty(n1 * n2 * 6, dtype=self.torch.float64, device=self.scalar_device)
        return self._fbuf[key]

    def flux_x3_export(self, l, side):
        g = self.lev[l]; nx = self.cfg.levels[l].Nx
        t = self.torch.empty((nx[0] // 2) * (nx[1] // 2) * 6, dtype=self.torch.float64, device=self.scalar_device)
        g.flux_x3_export(side, t.data_ptr()); return t

    def flux_x3_apply(self, l, side, i0, j0, n1, n2, t): self.lev[l].flux_x3_apply(side, i0, j0, n1, n2, t.data_ptr())

    def download(self, l) -> np.ndarray: return self.lev[l].download()
    def sync(self): self.lev[0].sync()
    def close(self): self.mesh.close()


class MeshDriver:
    """main() of the reference built with STATIC_MESH_REFINEMENT (main.c:395-447, :519-669) for one
    process of an N-process run in which every level is cut into x3 slabs at the same root planes
    (config.mesh_slabs).  Inside a rank the level coupling is the local aa_mesh; across ranks go the
    x3 halo of every level, the flux correction of a parent plane that lies across a cut from the
    child's boundary, and the scalar reductions (the MPI calls of smr.c, bvals_mhd.c, new_dt.c and
    ionrad_3d.c in the reference)."""

    def __init__(self, par, run: RunConfig, engine_factory=None, rank: int = 0, nranks: int = 1, device: int = 0,
                 strict: Optional[bool] = None, cuts=None):
        from .config import mesh_slabs
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.run, self.rank, self.nranks = run, rank, nranks
        from .config import levels as _levels
        self.cfg = mesh_slabs(par, run, rank, nranks, cuts)
        self.level_nx1 = [g.Nx[0] for g in _levels(par, run)]     # zones along the rays of every level (x3 slabs keep them)
        self.eng = engine_factory(self.cfg) if engine_factory else HipMeshEngine(self.cfg, device, strict)
        self.NL = len(self.cfg.table[0])                  # levels of the Mesh
        self.nl = len(self.cfg.levels)                    # levels present on this rank
        self.time, self.dt, self.nstep = 0.0, 0.0, 0
        self.dtl = [0.0] * self.NL                        # pGrid->dt of every level
        self.tcoarse = 0.0
        # AA_FORCE_DISTRIBUTED=1: issue the collectives even on one rank (rehearsal of the N>1 path)
        self.distributed = nranks > 1 or bool(os.environ.get("AA_FORCE_DISTRIBUTED"))
        if self.distributed:
            assert dist.is_initialized() and dist.get_world_size() == nranks and dist.get_rank() == rank
        self._sdev = getattr(self.eng, "scalar_device", torch.device("cpu"))
        self.niter_trace: List[List[int]] = []
        self._fused_cache = {}

    def has(self, l: int) -> bool: return l < self.nl

    def _allreduce(self, vals, op):
        if not self.distributed:
            return list(vals)
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self._sdev)
        self.dist.all_reduce(t, op=op)
        return t.tolist()

    def _p2p(self, sends, recvs):
        """sends: [(tensor, peer)], recvs: [(tensor, peer)]; gloo moves host tensors only."""
        dist = self.dist
        host_stage = (dist.get_backend() == "gloo")
        ops, staged = [], []
        for t, peer in sends:
            ops.append(dist.P2POp(dist.isend, t.cpu() if (host_stage and t.is_cuda) else t, peer))
        for t, peer in recvs:
            if host_stage and t.is_cuda:
                h = self.torch.empty(t.shape, dtype=t.dtype); staged.append((t, h)); t = h
            ops.append(dist.P2POp(dist.irecv, t, peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for t, h in staged:
            t.copy_(h)

    # ---- ghost zones ---------------------------------------------------------------------------
    def exchange_x3(self, l: int):
        if not self.distributed or not self.has(l):
            return
        g, e = self.cfg.levels[l], self.eng
        if g.lx3 < 0 and g.rx3 < 0:
            return
        sends, recvs, sides = [], [], []
        if g.lx3 == g.rx3 and g.lx3 >= 0:
            # two slabs with periodic wrap: post so that the peer's inner planes land in my OUTER ghosts
            sends = [(e.pack_x3(l, 0), g.lx3), (e.pack_x3(l, 1), g.rx3)]
            recvs = [(e.recv_buffer(l, 1), g.rx3), (e.recv_buffer(l, 0), g.lx3)]
            sides = [0, 1]
        else:
            if g.lx3 >= 0:
                sends.append((e.pack_x3(l, 0), g.lx3)); recvs.append((e.recv_buffer(l, 0), g.lx3)); sides.append(0)
            if g.rx3 >= 0:
                sends.append((e.pack_x3(l, 1), g.rx3)); recvs.append((e.recv_buffer(l, 1), g.rx3)); sides.append(1)
        self._p2p(sends, recvs)
        for side in sides:
            e.unpack_x3(l, side)

    def bvals_mhd(self, l: int):
        if self.has(l):
            self.eng.bvals_local(l)
        self.exchange_x3(l)

    # ---- level coupling ------------------------------------------------------------------------
    def restrict_correct(self):
        """smr.c:1207: finest pair first; a parent plane that lies across a cut from the child's
        boundary is corrected by the rank that owns it, with the child's restricted flux as message."""
        for l in range(self.NL - 2, -1, -1):
            if self.has(l + 1):
                self.eng.restrict_correct_pair(l)
            if not self.distributed:
                continue
            sends, recvs, todo = [], [], []
            if self.has(l + 1):
                for side, peer in enumerate(self.cfg.links[l].corr_to):
                    if peer >= 0:
                        sends.append((self.eng.flux_x3_export(l + 1, side), peer))
            for (lp, side, src, i0, j0, n1, n2) in self.cfg.corr_in:
                if lp == l:
                    buf = self.eng.flux_buffer(n1, n2, side)
                    recvs.append((buf, src)); todo.append((side, i0, j0, n1, n2, buf))
            self._p2p(sends, recvs)
            for side, i0, j0, n1, n2, buf in todo:
                self.eng.flux_x3_apply(l, side, i0, j0, n1, n2, buf)

    # ---- time step -------------------------------------------------------------------------------
    def new_dt(self):
        """new_dt.c:32 over all levels: max(|v|+a) per level (MAX over its slabs), carried from level to
        level (:33), one dt for the Mesh."""
        v = []
        for l in range(self.NL):
            v += self.eng.cfl_max_v(l) if self.has(l) else [0.0, 0.0, 0.0]
        if self.distributed:
            v = self._allreduce(v, self.dist.ReduceOp.MAX)
        cum, max_dti = [0.0, 0.0, 0.0], 0.0
        for l in range(self.NL):
            for d in range(3):
                cum[d] = cum[d] if cum[d] > v[3 * l + d] else v[3 * l + d]
            for d in range(3):
                q = cum[d] / (self.run.dx[d] / float(1 << l))
                max_dti = max_dti if max_dti > q else q
        dtc = self.run.cour_no / max_dti
        self.dt = dtc if self.nstep == 0 else min(2.0 * self.dt, dtc)
        if self.time < self.run.tlim and (self.run.tlim - self.time) < self.dt:
            self.dt = self.run.tlim - self.time
        for l in range(self.NL):
            self.dtl[l] = self.dt
            if self.has(l):
                self.eng.set_level_state(l, self.time, self.dt, self.nstep)

    # ---- radiation ---------------------------------------------------------------------------------
    def _level_fused(self, l: int) -> bool:
        """whether level l runs the one-kernel sub-cycle: asked of the LIBRARY (aa_ion_is_fused: it knows aa_params.ion_path,
        AA_ION_FUSED as it parses it, the ray direction) on the ranks that hold zones of the level, and agreed on by all
        ranks once -- ranks without zones of the level have no Grid to ask, and a rank that disagreed would issue the
        other protocol's collectives."""
        if l in self._fused_cache:
            return self._fused_cache[l]
        if not hasattr(self.eng, "ion_pass") or not hasattr(self.eng, "ion_is_fused"):
            ans = False
        else:
            mine = (1.0 if self.eng.ion_is_fused(l) else 0.0) if self.has(l) else -1.0          # -1: no opinion
            if self.distributed:
                hi = self._allreduce((mine,), self.dist.ReduceOp.MAX)[0]
                lo = self._allreduce((mine if mine >= 0 else 2.0,), self.dist.ReduceOp.MIN)[0]
                if hi >= 0 and lo <= 1.0 and hi != lo:
                    raise RuntimeError(f"[MeshDriver]: the ranks disagree on the radiation sub-cycle path of level {l}")
                ans = hi > 0.5
            else:
                ans = mine > 0.5
        self._fused_cache[l] = ans
        return ans

    def _ion_radtransfer_fused(self, l: int) -> int:
        """ionrad_3d.c:862 on level l with the one-kernel sub-cycle (see Driver._ion_radtransfer_fused): one pass, ONE
        all-gather of the ranks' reduction words and one read-back per sub-cycle.  A rank without zones of the level
        contributes neutral words and follows the control flow from the gathered words (the arithmetic of k_ion_pick2)."""
        e, has = self.eng, self.has(l)
        dist = self.dist if self.distributed else None
        nr = self.nranks if self.distributed else 1
        fine = l != 0
        if fine:
            if has:
                e.ionflux_prolong(l)
        else:
            self.tcoarse = 0.0
        limit = self.tcoarse if fine else self.dtl[0]
        if has:
            e.set_level_state(l, self.time, self.dtl[l], self.nstep)
            e.ion_begin(l)
            if hasattr(e, "ion_speculate"):
                e.ion_speculate(l, limit)      # (see Driver._ion_radtransfer_fused)
        mir = {"dt_sel": 0.0, "hit": False, "neg": False, "dt_done": 0.0}

        def one_pass(update, first):
            if has:
                e.ion_pass(l, update, True)
            else:
                e.ion_neutral_words()
            e.ion_gather(dist)
            if has:
                e.ion_pick(l, nr, first, limit)
                return e.ion_fetch(l) if not first else None
            W = e.words_all.tolist()                                   # this rank's one wait of the sub-cycle
            n = len(W) // nr
            dt_chem = min(W[0::n]); dt_therm = min(W[1::n]); max_dti = max(W[2::n]); count = sum(W[3::n]); neg = max(W[4::n]) != 0.0
            out = None
            if not first:
                out = (mir["dt_sel"], mir["hit"], dt_chem, dt_therm, int(count), self.run.cour_no / max_dti if max_dti > 0 else float("inf"), mir["neg"])
                mir["dt_done"] = mir["dt_done"] + mir["dt_sel"]
            else:
                mir["dt_done"] = 0.0
            dt = dt_therm if dt_therm < dt_chem else dt_chem
            hit = False
            if mir["dt_done"] + dt > limit:
                dt = limit - mir["dt_done"]; hit = True
            mir["dt_sel"], mir["hit"], mir["neg"] = dt, hit, neg
            return out

        one_pass(False, True)
        dt_done, niter = 0.0, 0
        while True:
            dt, hit, dt_chem, dt_therm, cellcount, dt_hydro, neg = one_pass(True, False)
            if neg:
                raise RuntimeError("[compute_chem_rates]: negative dt_chem")
            dt_done += dt
            niter += 1
            if not fine:
                if cellcount > MAXCELLCOUNT:
                    self.dtl[0] = dt_done; break
                if hit:
                    break
                if dt_hydro < dt_done:
                    self.dtl[0] = dt_done; break
            elif hit:
                self.dtl[l] = dt_done; break
        if has:
            e.ion_finish(l)
        if not fine:
            if niter == self.run.maxiter:
                self.dtl[0] = dt_done
            self.tcoarse = dt_done
        self.dt = self.dtl[l]                                   # ionrad_3d.c:1030 pMesh->dt = pGrid->dt
        if has:
            e.set_level_state(l, self.time, self.dtl[l], self.nstep)
        return niter

    def ion_radtransfer(self, l: int) -> int:
        """ionrad_3d.c:862 on level l (all ranks take part in the reductions, with neutral values where
        the level is absent)."""
        if self._level_fused(l):
            return self._ion_radtransfer_fused(l)
        e, has = self.eng, self.has(l)
        fine = l != 0
        INF = float("inf")
        if fine:
            if has:
                e.ionflux_prolong(l)
        else:
            self.tcoarse = 0.0
        if has:
            e.set_level_state(l, self.time, self.dtl[l], self.nstep)
            e.ion_begin(l)
        dt_done, niter, hydro_done, coarse_done = 0.0, 0, False, False
        while fine or not hydro_done:
            dt_chem, dt_therm = e.ion_rates(l) if has else (INF, INF)
            if self.distributed:
                dt_chem, dt_therm = self._allreduce((dt_chem, dt_therm), self.dist.ReduceOp.MIN)
            dt = min(dt_therm, dt_chem)
            if not fine:
                if dt_done + dt > self.dtl[0]:
                    dt = self.dtl[0] - dt_done; hydro_done = True
            elif dt_done + dt > self.tcoarse:
                dt = self.tcoarse - dt_done; coarse_done = True
            cellcount, dt_hydro = e.ion_update(l, dt) if has else (0, INF)
            if self.distributed and not fine:
                t = self._allreduce((float(cellcount),), self.dist.ReduceOp.SUM)
                h = self._allreduce((dt_hydro,), self.dist.ReduceOp.MIN)
                cellcount, dt_hydro = int(t[0]), h[0]
            dt_done += dt
            niter += 1
            if not fine:
                if cellcount > MAXCELLCOUNT:
                    self.dtl[0] = dt_done; break
                if hydro_done:
                    break
                if dt_hydro < dt_done:
                    self.dtl[0] = dt_done; break
            elif coarse_done:
                self.dtl[l] = dt_done; break
        if not fine:
            if niter == self.run.maxiter:
                self.dtl[0] = dt_done
            self.tcoarse = dt_done
        self.dt = self.dtl[l]                                   # ionrad_3d.c:1030 pMesh->dt = pGrid->dt
        if has:
            e.set_level_state(l, self.time, self.dtl[l], self.nstep)
        return niter

    # ---- main.c ----------------------------------------------------------------------------------------
    def start(self):
        for l in range(self.nl):
            self.eng.set_level_state(l, self.time, 0.0, self.nstep)
        self.restrict_correct()
        for l in range(self.NL):
            self.bvals_mhd(l)
            if self.has(l):
                self.eng.bvals_ionrad(l)
        self.eng.prolongate()
        self.new_dt()
        return self

    def step(self) -> List[int]:
        niter = [0] * self.NL
        if self.eng.has_radiation():                              # main.c:546-562
            for l in range(self.NL):
                niter[l] = self.ion_radtransfer(l)
                self.bvals_mhd(l)
            self.eng.ion_restrict_correct()
        for l in range(self.nl):                                  # :572-585
            self.eng.set_level_state(l, self.time, self.dtl[l], self.nstep)
            self.eng.integrate(l)
        self.restrict_correct()                                   # :591
        for l in range(self.nl):                                  # :597
            self.eng.userwork(l)
        self.nstep += 1
        self.time += self.dt                                      # :618-626
        self.new_dt()                                             # :629
        for l in range(self.NL):                                  # :635-644
            self.bvals_mhd(l)
        self.eng.prolongate()                                     # :647
        self.niter_trace.append(niter)
        return niter
