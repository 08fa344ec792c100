"""The layouts adopted in round 3 against the ones they replaced: the x1 first-pass sweep with a plane's rows laid end to end
(AA_X1_FLAT=0: a block per row piece) and the PPM slope arrays along x2 / x3 by a march (AA_SLOPES_MARCH=0: one zone per thread).
Strict build: the same bits.  Default build: the flat x1 sweep changes no bit either (the same inlined arithmetic per zone); the
marching slope kernel converts a cell with ONE instance of cons_to_prim where k_slopes has three, which hipcc may contract
differently: rounding level.  Odd sizes, so that rows, blocks and chunks end in the middle of wavefronts; the library reads the
switches once per process, hence the child processes."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOOL = os.path.join(HERE, "tools", "layout_run.py")

CASES = [("blast", 37, 19, 23, 2, "ctu", 3), ("ioniz_sphere", 70, 21, 18, 3, "ctu", 2), ("blast", 130, 18, 17, 3, "vl", 3),
         ("ioniz_sphere", 128, 128, 128, 2, "ctu", 2)]        # the last one takes the big-Grid kernels (2^21 zones)


def _run(case, env, out, strict):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, TOOL] + [str(x) for x in case] + [out] + (["strict"] if strict else []), env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("strict", [True, False], ids=["strict", "default"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{c[1]}x{c[2]}x{c[3]}-o{c[4]}-{c[5]}")
def test_old_and_new_layouts(case, strict, tmp_path):
    new = _run(case, {}, str(tmp_path / "new.npy"), strict)
    assert np.isfinite(new).all()
    old_x1 = _run(case, {"AA_X1_FLAT": "0"}, str(tmp_path / "old_x1.npy"), strict)
    assert np.array_equal(new, old_x1)
    if case[4] == 3:
        old_sl = _run(case, {"AA_SLOPES_MARCH": "0"}, str(tmp_path / "old_sl.npy"), strict)
        if strict:
            assert np.array_equal(new, old_sl)
        else:
            scale = np.abs(old_sl).max(axis=tuple(range(old_sl.ndim - 1)), keepdims=True)
            err = float(np.max(np.abs(new - old_sl)/scale))
            print("default build, marching slopes against k_slopes: max error / field scale", err)
            assert err < 1e-11


@pytest.mark.parametrize("strict", [True, False], ids=["strict", "default"])
@pytest.mark.parametrize("case,ov", [(CASES[0], ""), (CASES[1], ""), (CASES[3], ""),
                                     (("blast", 40, 21, 18, 2, "ctu", 3), "domain1/bc_ix1=1 domain1/bc_ox1=2 domain1/bc_ix2=2 domain1/bc_ox2=1 domain1/bc_ix3=1 domain1/bc_ox3=1")],
                         ids=["blast-periodic", "sphere-outflow", "sphere-128", "blast-reflect-outflow-mix"])
def test_fewer_launches_same_bits(case, ov, strict, tmp_path):
    """Round 4, for the small Grids of the reference's own decks (a step there is a few dozen launches and 2 + N_sub read-backs):
    bvals_mhd's three passes as ONE launch over the ghost shell (k_bc_shell; AA_BC_ONE=0: the passes), the scalars back through a
    polled mailbox in pinned memory (AA_MAILBOX=0: copy + stream wait), the sub-cycle's fold and pick in one launch where one rank
    reduces alone (AA_ION_FUSE_PICK=0), the pinned zones' share of new_dt's maxima taken by the kernel that pins them (AA_PIN_ONE=0).  Copies, sign flips and the same arithmetic: the whole block incl. every ghost zone and corner
    must come out bit for bit, in both builds -- with periodic, outflow and reflecting sides mixed."""
    new = _run(case, {"LAYOUT_OV": ov}, str(tmp_path / "new.npy"), strict)
    old = _run(case, {"LAYOUT_OV": ov, "AA_BC_ONE": "0", "AA_MAILBOX": "0", "AA_ION_FUSE_PICK": "0", "AA_PIN_ONE": "0"}, str(tmp_path / "old.npy"), strict)
    assert np.isfinite(new).all() or case[0] == "ioniz_sphere"
    assert np.array_equal(new, old, equal_nan=True)
