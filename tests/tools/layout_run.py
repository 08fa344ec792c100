"""Helper of tests/test_gpu_layout_switches.py: a few steps of a deck on the GPU, the state written to an .npy file.  Run as a child
process because the library reads its layout switches (AA_X1_FLAT, AA_SLOPES_MARCH, AA_CORRECT_ALL ...) once per process.
usage: layout_run.py <problem> <nx1> <nx2> <nx3> <order> <integrator> <steps> <out.npy> [strict]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
aa = importlib.import_module("atmospheric-athena_amd")
lib = importlib.import_module("atmospheric-athena_amd.lib")
problem, n1, n2, n3, order, integ, steps, out = sys.argv[1:9]
ov = [f"domain1/Nx1={n1}", f"domain1/Nx2={n2}", f"domain1/Nx3={n3}"] + os.environ.get("LAYOUT_OV", "").split()      # (LAYOUT_OV: further deck overrides)
run = aa.config.load(os.path.join(ROOT, "atmospheric-athena_amd", "decks", f"athinput.{problem}"), ov, problem, integ)
run.order = int(order)
g = lib.setup_problem(aa.config.slab(run), 0, len(sys.argv) > 9 and sys.argv[9] == "strict")
g.start()
for _ in range(int(steps)):
    g.step()
np.save(out, g.download())
print("dt", g.dt)
