"""TEST INFRASTRUCTURE: how far the ORACLE moves when its start state is changed by one unit in the last place.

Two parity cases sit on knife edges (a planet two or three zones in radius beside a 1e5 density jump: negative face
pressures, NaN etas, Roe->HLLE switches, limiter decisions): any arithmetic that differs from the reference's in the last
bit (fused multiply-adds, device exp / log) lands a few zones on the other side of a decision.  The spread of the oracle's
own 1-ulp twins -- the largest relative difference and the number of zones that move at all -- is what such arithmetic can
be held to there; the GPU tests assert their error against these numbers, computed in the same session (tests/
test_gpu_parity.py), instead of against a constant."""
import functools
import os

import numpy as np

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEEDS = (1, 2, 3, 4, 5, 6)


def perturb(U, seed):
    """every conserved variable of every zone one unit in the last place up or down"""
    rng = np.random.default_rng(seed)
    for c in range(U.shape[-1]):
        A = U[..., c]
        A[...] = np.where(rng.integers(0, 2, A.shape) == 1, np.nextafter(A, np.inf), np.nextafter(A, -np.inf))


def spread(a, b):
    """|a - b| / max|b| per field, fields that vanish identically left out"""
    scale = np.nanmax(np.abs(b), axis=(0, 1, 2))
    return np.abs(a - b)[..., scale > 0] / scale[scale > 0]


@functools.lru_cache(maxsize=None)
def sphere_24x16x12():
    """-> (largest spread, largest number of zones beyond 1e-8) over the twins of the 2-step run"""
    ov = ["domain1/Nx1=24", "domain1/Nx2=16", "domain1/Nx3=12"]
    base = orc.make_sim("ioniz_sphere", ov).start()
    for _ in range(2):
        base.step()
    worst, nflip = 0.0, 0
    for seed in SEEDS:
        p = orc.make_sim("ioniz_sphere", ov)
        perturb(p.active, seed)
        p.start()
        for _ in range(2):
            p.step()
        assert np.array_equal(np.isnan(p.active), np.isnan(base.active))
        e = spread(p.active, base.active)
        worst = max(worst, float(np.nanmax(e)))
        nflip = max(nflip, int((e > 1e-8).any(axis=-1).sum()))
    return worst, nflip


@functools.lru_cache(maxsize=None)
def developed_sphere_32():
    """-> (largest spread, largest number of zones beyond 1e-9) over the twins of the 3 steps from the reference's step 12"""
    g = np.load(os.path.join(GOLD, "dev_ioniz_sphere_32x32x32_s12_s15.npz"))
    nx = g["nx"]
    ov = [f"domain1/Nx{d + 1}={int(nx[d])}" for d in range(3)]

    def advance(seed):
        s = orc.make_sim("ioniz_sphere", ov)
        s.active[...] = g["UA"]
        if seed:
            perturb(s.active, seed)
        s.time = float(g["timeA"]); s.dt = float(g["dtA"]); s.nstep = int(g["nstepA"])
        s.bvals(); s.bvals_ionrad()
        for _ in range(int(g["nstepB"]) - int(g["nstepA"])):
            s.step()
        return s.active.copy()

    base = advance(0)
    assert np.array_equal(base, g["UB"], equal_nan=True)          # (the oracle is pinned to the reference on this pair)
    worst, nflip = 0.0, 0
    for seed in SEEDS:
        e = spread(advance(seed), base)
        worst = max(worst, float(np.nanmax(e)))
        nflip = max(nflip, int((e > 1e-9).any(axis=-1).sum()))
    return worst, nflip
