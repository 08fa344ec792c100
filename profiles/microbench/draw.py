"""Where does the run-to-run spread of the big kernels come from?  One process, the 512^3 Grid created / destroyed several times
(a fresh hipMalloc of the 87 GB pool each time; with `ballast` a block of BALLAST_GB is allocated first and kept, so that the pool
lands elsewhere): per-kernel hipEvent times of 6 steps after 2 warm-up steps, from the start of the deck.
usage (GPU box, repo root): python profiles/microbench/draw.py [rounds] [ballast_gb ...]"""
import importlib, os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
aa = importlib.import_module("atmospheric-athena_amd")
lib = importlib.import_module("atmospheric-athena_amd.lib")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ballast = [float(x) for x in sys.argv[2:]] or [0.0]
hip = ctypes.CDLL("libamdhip64.so")
run = aa.config.load(os.path.join(ROOT, "atmospheric-athena_amd", "decks", "athinput.ioniz_sphere"),
                     ["domain1/Nx1=512", "domain1/Nx2=512", "domain1/Nx3=512"], "ioniz_sphere")
for r in range(rounds):
    for b in ballast:
        p = ctypes.c_void_p()
        if b > 0: assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(b*2**30))) == 0
        g = lib.setup_problem(aa.config.slab(run), 0, False)
        g.start()
        for _ in range(2): g.step()
        g.profile_reset(); g.profile_enable(True)
        for _ in range(6): g.step()
        g.sync()
        pr = {k: ms/6 for k, (ms, n) in g.profile().items() if ms/6 > 1.0 and k != "ion_pass"}
        g.profile_enable(False)
        print(f"round {r} ballast {b:5.1f} GB:", {k: round(v, 2) for k, v in pr.items()}, flush=True)
        g.close(); del g
        if b > 0: hip.hipFree(p)
