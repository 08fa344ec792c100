"""How ill-conditioned are the two parity cases that cannot be held to 1e-8 (tests/test_gpu_parity.py: the 24x16x12 sphere,
the developed 32^3 sphere)?

Measured with the ORACLE alone (CPU, no GPU; tests/twins.py): the same run is repeated with every conserved variable of
every zone of the start state moved by ONE unit in the last place (random sign).  A planet two or three zones in radius
beside a 1e5 density jump produces negative face pressures (NaN etas), Roe->HLLE switches and limiter decisions on a knife
edge; a 1-ulp perturbation moves those decisions, and the spread between the oracle runs is the accuracy ANY arithmetic
that differs from the reference's in the last bit (fused multiply-adds, device exp/log) can be held to on these cases.
The GPU tests assert their error against these same numbers (computed in their own session): <= 2x the twins' spread and
<= 2x their count of moved zones on the 24x16x12 sphere; <= the twins' spread, <= 3x their count AND north_star's 1e-6
on the developed sphere.  Here: the numbers are what the comments say they are."""
import twins


def test_sphere_24x16x12_spread_of_a_one_ulp_perturbation():
    worst, nflip = twins.sphere_24x16x12()
    print("24x16x12 sphere, 2 steps: spread of a 1-ulp perturbation", worst, "zones beyond 1e-8:", nflip)
    # a 1-ulp change of the input moves the answer by more than 1e-8 (what every well-conditioned case is held to) and by
    # more than north_star's 1e-6: no implementation can meet 1e-6 here; the GPU (3.9e-6 on 40-46 zones) sits inside 2x this
    assert 2e-6 < worst < 2e-5 and 20 <= nflip <= 100


def test_developed_sphere_spread_of_a_one_ulp_perturbation():
    worst, nflip = twins.developed_sphere_32()
    print("developed 32^3 sphere, 3 steps: spread of a 1-ulp perturbation", worst, "zones beyond 1e-9:", nflip)
    assert 1e-6 < worst < 1e-2 and 15 <= nflip <= 200
