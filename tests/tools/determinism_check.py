"""Run-to-run determinism of the DEFAULT build (one process, every case twice, whole blocks compared bit for bit): Mesh fixtures with and
without radiation on both kernel chains, single Grids on both chains.  usage (GPU box, repo root): python tests/tools/determinism_check.py"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import orc
aa = importlib.import_module("atmospheric-athena_amd"); lib = importlib.import_module("atmospheric-athena_amd.lib")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
def mesh(name, problem, strict):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    par = aa.athinput.ParTable.from_file(orc.deck_for(problem, g)).cmdline([str(o) for o in g["overrides"]])
    run = aa.config.from_par(par, problem)
    m = lib.Mesh(aa.config.levels(par, run), 0, strict)
    m.start(); its = []
    for _ in range(int(g["nstep"])): its += (m.step() or [])
    out = [lev.download() for lev in m.lev]; m.close()
    return out, its
def single(problem, n, nstep, strict):
    run = aa.config.load(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput." + problem), [f"domain1/Nx{d}={n}" for d in (1, 2, 3)], problem)
    g = lib.setup_problem(aa.config.slab(run), 0, strict); g.start()
    its = [g.step() for _ in range(nstep)]
    U = g.download(); g.close()
    return [U], its
bad = 0
for ca in ("0", "1"):
    os.environ["AA_CORRECT_ALL"] = ca
    for label, fn in (("mesh blast tree", lambda: mesh("smr_blast_tree_s5", "blast", False)), ("mesh blast 3lev", lambda: mesh("smr_blast_3lev_edge_s8", "blast", False)),
                      ("mesh sphere 2lev", lambda: mesh("smr_ioniz_sphere_2lev_s4", "ioniz_sphere", False)),
                      ("ioniz_sphere 48^3 x8", lambda: single("ioniz_sphere", 48, 8, False)), ("ifront 40^3 x6", lambda: single("ifront", 40, 6, False)),
                      ("blast 70^3 x5", lambda: single("blast", 70, 5, False))):
        a, ia = fn(); b, ib = fn(); c, ic = fn()
        same = ia == ib == ic and all(np.array_equal(x, y, equal_nan=True) and np.array_equal(x, z, equal_nan=True) for x, y, z in zip(a, b, c))
        bad += not same
        print(f"AA_CORRECT_ALL={ca} {label}: three runs identical: {same}", flush=True)
print("nondeterministic cases:", bad)
sys.exit(1 if bad else 0)
