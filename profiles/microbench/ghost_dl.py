import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
aa = importlib.import_module("atmospheric-athena_amd"); lib = importlib.import_module("atmospheric-athena_amd.lib")
for n in (128, 256, 384):
    run = aa.config.load(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput.blast"), [f"domain1/Nx{d}={n}" for d in (1, 2, 3)], "blast")
    g = lib.setup_problem(aa.config.slab(run), 0, False)
    g.start(); g.step()
    U = g.download()
    for name, f in (("full block", lambda: g.download()), ("ghost shell", lambda: g.download_ghost_zones(U)), ("upload", lambda: g.upload(U))):
        f(); t0 = time.perf_counter()
        for _ in range(5): f()
        dt = (time.perf_counter() - t0)/5
        print(f"{n}^3 {name}: {dt*1e3:.2f} ms  ({U.nbytes/1e6:.0f} MB block)", flush=True)
    g.close()
