import importlib, os, sys, numpy as np
sys.path.insert(0, "/root/repo")
aa = importlib.import_module("atmospheric-athena_amd"); lib = importlib.import_module("atmospheric-athena_amd.lib")
ov = ["domain1/Nx1=128", "domain1/Nx2=24", "domain1/Nx3=20"]
run = aa.config.load("/root/repo/atmospheric-athena_amd/decks/athinput.ioniz_sphere", ov, "ioniz_sphere")
g = lib.setup_problem(aa.config.slab(run), 0, False)
g.start()
for _ in range(3): g.step()
U = g.download(); np.save(sys.argv[1], U); print("dt", g.dt, "nan", int(np.isnan(U).sum()))
