"""k_correct_all on the slab of an 8-way (4-way, 2-way) strong-scaling run -- 512 x 512 x 64 (128, 256) zones on one GPU -- with the
default chunk length (64 planes) and with AA_CA_KC set by the caller: how much the tail of a launch of few, long blocks costs.
usage (GPU box, repo root): [AA_CA_KC=22] python profiles/microbench/slab_chunks.py 64"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
aa = importlib.import_module("atmospheric-athena_amd")
lib = importlib.import_module("atmospheric-athena_amd.lib")
n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ov = ["domain1/Nx1=512", "domain1/Nx2=512", f"domain1/Nx3={n3}"]
run = aa.config.load(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput.ioniz_sphere"), ov, "ioniz_sphere")
g = lib.setup_problem(aa.config.slab(run), 0, False)
g.start()
for _ in range(3): g.step()
ts = []
for _ in range(8):
    g.sync(); t0 = time.perf_counter(); g.integrate_3d_ctu(); g.sync(); ts.append((time.perf_counter() - t0)*1e3)
ts.sort()
print(f"Nx3={n3} AA_CA_KC={os.environ.get('AA_CA_KC', 'default')}: integrate_3d_ctu {ts[len(ts)//2]:.3f} ms (median of 8)")
