"""Does a VALU-bound kernel pair (first-pass sweeps x1 + x2) overlap usefully with the byte-bound pair (k_correct_all +
k_flux2_update) when they run on two HIP streams?  Two resident 512^3 Grids A and B (173 GB), A: aa_integrate_begin (the
sweeps of the planes ks..ke) issued n times, B: aa_integrate_3d_ctu behind its own aa_integrate_begin (ghost-plane sweeps +
correct_all + flux2_update).  Serial = the two timed one after the other; concurrent = both issued back to back, one sync.
A measurement of what a k-chunk pipeline of the integrator (sweeps of chunk c+1 beside correct_all of chunk c) could gain.
usage (GPU box, repo root): python profiles/microbench/overlap.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
aa = importlib.import_module("atmospheric-athena_amd")
lib = importlib.import_module("atmospheric-athena_amd.lib")
nx = int(os.environ.get("NX", "512"))
ov = [f"domain1/Nx{d + 1}={nx}" for d in range(3)]
run = aa.config.load(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput.ioniz_sphere"), ov, "ioniz_sphere")
A = lib.setup_problem(aa.config.slab(run), 0, False)
B = lib.setup_problem(aa.config.slab(run), 0, False)
for g in (A, B):
    g.start()
    for _ in range(3): g.step()
    g.sync()
import statistics
def prep():
    A.integrate_3d_ctu() if primed[0] else None     # consumes A's pending sweeps (untimed)
    primed[0] = False
    B.integrate_begin(); A.sync(); B.sync()
primed = [False]
def timed(f):
    ts = []
    for _ in range(6):
        prep(); t0 = time.perf_counter(); f(); A.sync(); B.sync(); ts.append((time.perf_counter() - t0)*1e3)
    return statistics.median(ts[1:])
def a_sweeps(): A.integrate_begin(); primed[0] = True
def b_rest(): B.integrate_3d_ctu()
def both(): B.integrate_3d_ctu(); A.integrate_begin(); primed[0] = True
def both2(): A.integrate_begin(); B.integrate_3d_ctu(); primed[0] = True
ta, tb = timed(a_sweeps), timed(b_rest)
print("A: sweeps x1 + x2 of the planes ks..ke              %.2f ms" % ta)
print("B: ghost-plane sweeps + correct_all + flux2_update  %.2f ms" % tb)
print("serial sum                                          %.2f ms" % (ta + tb))
print("both streams at once (B issued first)               %.2f ms" % timed(both))
print("both streams at once (A issued first)               %.2f ms" % timed(both2))
