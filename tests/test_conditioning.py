"""How ill-conditioned are the two cases whose default-build GPU tolerance sits above north_star's 1e-6
(tests/test_gpu_parity.py: the 24x16x12 sphere at 2e-5, the developed 32^3 sphere at 1e-3 on <= 0.5 % of the zones)?

Measured with the ORACLE alone (CPU, no GPU): the same run is repeated with the energy of every zone of the start
state moved by ONE unit in the last place (random sign).  A planet two or three zones in radius beside a 1e5 density
jump produces negative face pressures (NaN etas), Roe->HLLE switches and limiter decisions on a knife edge; a 1-ulp
perturbation moves those decisions, and the spread between the two oracle runs is the accuracy ANY arithmetic that
differs from the reference's in the last bit (fused multiply-adds, device exp/log) can be held to on these cases.
The GPU tests' tolerances must not be tighter than this spread allows, nor much looser."""
import os

import numpy as np
import pytest

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _perturb(U, seed):
    """every conserved variable of every zone one unit in the last place up or down"""
    rng = np.random.default_rng(seed)
    for c in range(U.shape[-1]):
        A = U[..., c]
        A[...] = np.where(rng.integers(0, 2, A.shape) == 1, np.nextafter(A, np.inf), np.nextafter(A, -np.inf))


def _spread(a, b):
    scale = np.nanmax(np.abs(b), axis=(0, 1, 2))
    err = np.abs(a - b)[..., scale > 0] / scale[scale > 0]
    return err


def test_sphere_24x16x12_spread_of_a_one_ulp_perturbation():
    ov = ["domain1/Nx1=24", "domain1/Nx2=16", "domain1/Nx3=12"]
    base = orc.make_sim("ioniz_sphere", ov).start()
    worst = 0.0
    for _ in range(2):
        base.step()
    for seed in (1, 2, 3):
        p = orc.make_sim("ioniz_sphere", ov)
        _perturb(p.active, seed)
        p.start()
        for _ in range(2):
            p.step()
        assert np.array_equal(np.isnan(p.active), np.isnan(base.active))
        worst = max(worst, float(np.nanmax(_spread(p.active, base.active))))
    # the GPU test allows 2e-5 on this case; a 1-ulp change of the input already moves the answer by more than 1e-8
    # (what every well-conditioned case is held to) -- and stays below the GPU tolerance
    print("24x16x12 sphere, 2 steps: spread of a 1-ulp perturbation", worst)
    assert 1e-8 < worst < 2e-5


def test_developed_sphere_spread_of_a_one_ulp_perturbation():
    g = np.load(os.path.join(GOLD, "dev_ioniz_sphere_32x32x32_s12_s15.npz"))
    nx = g["nx"]
    ov = [f"domain1/Nx{d + 1}={int(nx[d])}" for d in range(3)]

    def advance(seed):
        s = orc.make_sim("ioniz_sphere", ov)
        s.active[...] = g["UA"]
        if seed:
            _perturb(s.active, seed)
        s.time = float(g["timeA"]); s.dt = float(g["dtA"]); s.nstep = int(g["nstepA"])
        s.bvals(); s.bvals_ionrad()
        for _ in range(int(g["nstepB"]) - int(g["nstepA"])):
            s.step()
        return s.active.copy()

    base = advance(0)
    assert np.array_equal(base, g["UB"], equal_nan=True)          # (the oracle is pinned to the reference on this pair)
    worst, frac = 0.0, 0.0
    for seed in (1, 2, 3):
        err = _spread(advance(seed), base)
        worst = max(worst, float(np.nanmax(err)))
        frac = max(frac, float((err > 1e-9).any(axis=-1).mean()))
    # the GPU test of the default build allows 1e-3 on at most 0.5 % of the zones: the oracle itself, fed a state that
    # differs in the last bit, spreads as far on a comparable share of the zones
    print("developed 32^3 sphere, 3 steps: spread of a 1-ulp perturbation", worst, "share of zones above 1e-9:", frac)
    assert 1e-6 < worst < 1e-2 and frac < 0.02
