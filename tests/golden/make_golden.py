#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference and oracle/_ref, built by
`make -C oracle ref`).  It executes the unmodified reference executables on tiny grids,
reads the full-precision restart dumps they write (restart.c:531-770: labelled raw double
blocks over active zones) and stores inputs + expected outputs as compact .npz files.  It
also drives the reference's own kernels (fluxes, lr_states, Cons1D_to_Prim1D, cfast)
through oracle/_ref/libref_<cfg>.so to produce function-level known-answer vectors.

Fixtures are DATA (arrays + the scalar trace of each run); no reference text is stored.

usage: python tests/golden/make_golden.py [whole] [shk] [dev] [kernels] [ppm] [smr] [hst] [ray] [cool] [vlppm] [noh]
"""
import ctypes as C
import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
REFBIN = os.path.join(ROOT, "oracle", "_ref")

LABELS = ["DENSITY", "1-MOMENTUM", "2-MOMENTUM", "3-MOMENTUM", "ENERGY"]


def read_rst(path, nx, nscal, ion):
    b = open(path, "rb").read()
    pos = b.index(b"N_STEP\n") + len(b"N_STEP\n")
    nstep = struct.unpack_from("<i", b, pos)[0]
    pos = b.index(b"\nTIME\n", pos) + len(b"\nTIME\n")
    time = struct.unpack_from("<d", b, pos)[0]
    pos = b.index(b"\nTIME_STEP\n", pos) + len(b"\nTIME_STEP\n")
    dt = struct.unpack_from("<d", b, pos)[0]
    n = nx[0] * nx[1] * nx[2]
    out = np.zeros((nx[2], nx[1], nx[0], 6))
    for c, lab in enumerate(LABELS):
        tag = b"\n" + lab.encode() + b"\n"
        pos = b.index(tag, pos) + len(tag)
        out[..., c] = np.frombuffer(b, dtype="<f8", count=n, offset=pos).reshape(nx[2], nx[1], nx[0])
        pos += 8 * n
    ef = None
    if ion:
        tag = b"\nEDGEFLUX\n"
        pos = b.index(tag, pos) + len(tag)
        ne = (nx[0] + 1) * (nx[1] + 1) * (nx[2] + 1)
        ef = np.frombuffer(b, dtype="<f8", count=ne, offset=pos).reshape(nx[2] + 1, nx[1] + 1, nx[0] + 1).copy()
        pos += 8 * ne
    if nscal:
        tag = b"\nSCALAR 0\n"
        pos = b.index(tag, pos) + len(tag)
        out[..., 5] = np.frombuffer(b, dtype="<f8", count=n, offset=pos).reshape(nx[2], nx[1], nx[0])
    return dict(nstep=nstep, time=time, dt=dt, U=out, edgeflux=ef)


def run_reference(cfg, deck, nx, nlim, extra, pid, nscal, ion):
    exe = os.path.join(REFBIN, "athena_" + cfg)
    tmp = tempfile.mkdtemp(prefix="golden_")
    rundir = os.path.join(tmp, "run")
    args = [exe, "-i", deck, "-d", rundir,
            f"domain1/Nx1={nx[0]}", f"domain1/Nx2={nx[1]}", f"domain1/Nx3={nx[2]}",
            f"time/nlim={nlim}"] + extra
    pr = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp)
    if pr.returncode != 0:
        raise RuntimeError(pr.stdout[-2000:] + pr.stderr[-2000:])
    niter = [int(m) for m in re.findall(r"Radiation done in (\d+) iterations", pr.stderr)]
    rsts = sorted(f for f in os.listdir(rundir) if f.endswith(".rst"))
    first = read_rst(os.path.join(rundir, rsts[0]), nx, nscal, ion)
    last = read_rst(os.path.join(rundir, rsts[-1]), nx, nscal, ion)
    shutil.rmtree(tmp)
    return first, last, niter


def save(name, first, last, niter, nx, overrides):
    d = dict(nx=np.array(nx), U0=first["U"], U=last["U"], nstep=last["nstep"], time=last["time"],
             dt=last["dt"], dt0=first["dt"], niter=np.array(niter, dtype=np.int64),
             overrides=np.array(overrides))
    if last["edgeflux"] is not None:
        d["edgeflux"] = last["edgeflux"]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(f"{name}: nstep={last['nstep']} time={last['time']:.17g} dt={last['dt']:.17g} niter={niter}")


def whole_runs():
    ifront = os.path.join(REF, "tst/ionradiation/athinput.ifront")
    sphere = os.path.join(REF, "tst/massloss/athinput.ioniz_sphere_hires")
    blast = os.path.join(REF, "tst/3D-hydro/athinput.blast")
    # (name, cfg, deck, nx, nlim, extra reference-cmdline, nscal, ion, overrides for OUR decks)
    for nx, nlims in (((16, 8, 8), (1, 3, 6)), ((8, 12, 16), (4,))):
        for nlim in nlims:
            f, l, it = run_reference("ifront", ifront, nx, nlim,
                                     ["job/maxout=3", "output3/out_fmt=rst", "output3/dt=1e300",
                                      "output1/dt=1e300", "output2/dt=1e300"], "ifront", 1, True)
            save(f"ifront_{nx[0]}x{nx[1]}x{nx[2]}_n{nlim}", f, l, it, nx, [])
    for nx, nlims in (((20, 20, 20), (1, 3)), ((24, 16, 12), (2,))):
        for nlim in nlims:
            f, l, it = run_reference("ioniz_sphere", sphere, nx, nlim,
                                     ["job/num_domains=1", "job/maxout=1", "output1/dt=1e300"],
                                     "ioniz_sphere", 1, True)
            save(f"ioniz_sphere_{nx[0]}x{nx[1]}x{nx[2]}_n{nlim}", f, l, it, nx, [])
    # van Leer integrator (NO_H_CORRECTION: the only VL build of the reference that compiles)
    f, l, it = run_reference("blast_vl", blast, (16, 12, 20), 4,
                             ["job/num_domains=1", "job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], "Blast", 0, False)
    save("vl_blast_16x12x20_n4", f, l, it, (16, 12, 20), [])
    f, l, it = run_reference("ioniz_sphere_vl", sphere, (20, 16, 12), 2,
                             ["job/num_domains=1", "job/maxout=1", "output1/dt=1e300"], "ioniz_sphere", 1, True)
    save("vl_ioniz_sphere_20x16x12_n2", f, l, it, (20, 16, 12), [])
    f, l, it = run_reference("ifront_vl", ifront, (16, 8, 8), 3,
                             ["job/maxout=3", "output3/out_fmt=rst", "output3/dt=1e300", "output1/dt=1e300",
                              "output2/dt=1e300"], "ifront", 1, True)
    save("vl_ifront_16x8x8_n3", f, l, it, (16, 8, 8), [])
    for nx, nlims in (((16, 16, 16), (1, 5)), ((12, 20, 16), (4,))):
        for nlim in nlims:
            f, l, it = run_reference("blast", blast, nx, nlim,
                                     ["job/num_domains=1", "job/maxout=1", "output1/out_fmt=rst",
                                      "output1/dt=1e300"], "Blast", 0, False)
            save(f"blast_{nx[0]}x{nx[1]}x{nx[2]}_n{nlim}", f, l, it, nx, [])


def rayplane_runs():
    """Radiation planes along -x1 ... no: dir=-1 (rays along +x1) and dir=-2 (rays along +x2) with our own problem file
    tests/fixtures/rayplane_dir.c (a density pattern that depends on the zone indices only) linked into the reference's
    ifront configuration.  dir=-3 is not pinned: ionradplane_3d.c:136-145 leaves cell_len uninitialised for it."""
    deck0 = open(os.path.join(REF, "tst/ionradiation/athinput.ifront")).read()
    tmp = tempfile.mkdtemp(prefix="golden_deck_")
    deck = os.path.join(tmp, "athinput.rayplane")
    open(deck, "w").write(re.sub(r"(?m)^(nradplanes\s*=.*)$", r"\1\nraydir = -1", deck0, count=1))
    for d, nx, nlim in ((-1, (12, 10, 8), 3), (-2, (12, 10, 8), 3), (-2, (6, 70, 5), 2)):
        f, l, it = run_reference("rayplane", deck, nx, nlim,
                                 ["job/maxout=3", "output3/out_fmt=rst", "output3/dt=1e300", "output1/dt=1e300", "output2/dt=1e300",
                                  f"problem/raydir={d}"], "ifront", 1, True)
        save(f"rayplane_dir{-d}_{nx[0]}x{nx[1]}x{nx[2]}_n{nlim}", f, l, it, nx, [f"raydir={d}"])
    shutil.rmtree(tmp)


def vl_ppm_runs():
    """--with-integrator=vl --with-order=3: the van Leer integrator on piecewise parabolic states without tracing
    (lr_states_ppm.c:502-507, the branch for integrators other than CTU)."""
    sphere = os.path.join(REF, "tst/massloss/athinput.ioniz_sphere_hires")
    blast = os.path.join(REF, "tst/3D-hydro/athinput.blast")
    f, l, it = run_reference("blast_vl_ppm", blast, (16, 12, 20), 4,
                             ["job/num_domains=1", "job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], "Blast", 0, False)
    save("vl_ppm_blast_16x12x20_n4", f, l, it, (16, 12, 20), [])
    f, l, it = run_reference("ioniz_sphere_vl_ppm", sphere, (20, 16, 12), 2,
                             ["job/num_domains=1", "job/maxout=1", "output1/dt=1e300"], "ioniz_sphere", 1, True)
    save("vl_ppm_ioniz_sphere_20x16x12_n2", f, l, it, (20, 16, 12), [])


def no_h_correction_runs():
    """CTU without --enable-h-correction (the reference's configure default: no eta arrays, roe.c without etah)."""
    sphere = os.path.join(REF, "tst/massloss/athinput.ioniz_sphere_hires")
    blast = os.path.join(REF, "tst/3D-hydro/athinput.blast")
    f, l, it = run_reference("blast_noh", blast, (16, 12, 20), 4,
                             ["job/num_domains=1", "job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], "Blast", 0, False)
    save("noh_blast_16x12x20_n4", f, l, it, (16, 12, 20), [])
    f, l, it = run_reference("ioniz_sphere_noh", sphere, (20, 16, 12), 2,
                             ["job/num_domains=1", "job/maxout=1", "output1/dt=1e300"], "ioniz_sphere", 1, True)
    save("noh_ioniz_sphere_20x16x12_n2", f, l, it, (20, 16, 12), [])


def cooling_runs():
    """Optically thin cooling (integrate_3d_ctu.c Steps 1c-3c, 8b, 11c with CoolingFunc = KoyInut, microphysics/cool.c:48): our own
    problem file tests/fixtures/cool_pattern.c (diffuse gas in cgs units, density / temperature / velocity by integer patterns of
    the zone indices, 150 K ... 6000 K) linked into the reference's blast configuration; the same state without cooling beside
    it, so that the tests can show the terms matter."""
    deck0 = open(os.path.join(REF, "tst/3D-hydro/athinput.blast")).read()
    tmp = tempfile.mkdtemp(prefix="golden_deck_")
    deck = os.path.join(tmp, "athinput.coolpat")
    open(deck, "w").write(re.sub(r"(?m)^(radius\s*=.*)$", r"\1\nn0 = 30.0\nT0 = 2500.0\nv0 = 2.0e4\ncool = 1", deck0, count=1))
    box = []
    for e in (1, 2, 3):
        box += [f"domain1/x{e}min=-1.0e18", f"domain1/x{e}max=1.0e18"]
    for cool, nx, nlim in ((1, (16, 12, 10), 4), (0, (16, 12, 10), 4), (1, (12, 8, 20), 3)):
        over = box + ["time/tlim=1.0e30", f"problem/cool={cool}"]
        f, l, it = run_reference("coolpat", deck, nx, nlim,
                                 ["job/num_domains=1", "job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"] + over, "Blast", 0, False)
        save(f"coolpat_c{cool}_{nx[0]}x{nx[1]}x{nx[2]}_n{nlim}", f, l, it, nx, over + ["n0=30.0", "T0=2500.0", "v0=2.0e4"])
    shutil.rmtree(tmp)


def shock_tubes():
    """Sod's shock tube (tst/1D-hydro/athinput.sod, BASELINE configs[0]) on 3-D grids, along x1, x2 and
    x3 (prob/shkset1d.c rotates the state), reference built with the 3-D CTU integrator + H-correction."""
    sod = os.path.join(REF, "tst/1D-hydro/athinput.sod")
    for d, nx in ((1, (48, 8, 6)), (2, (6, 48, 8)), (3, (8, 6, 48))):
        over = [f"problem/shk_dir={d}", "time/cour_no=0.4"]
        for e in (1, 2, 3):
            over += [f"domain1/x{e}min=-0.5", f"domain1/x{e}max=0.5"]
        f, l, it = run_reference("shk3d", sod, nx, 12, ["job/maxout=1", "output1/out_fmt=rst", "output1/out=cons", "output1/dt=1e300"] + over,
                                 "Sod", 0, False)
        save(f"shkset1d_d{d}_{nx[0]}x{nx[1]}x{nx[2]}_n12", f, l, it, nx, over)


def ppm_runs():
    """--with-order=3 (piecewise parabolic reconstruction, lr_states_ppm.c): function-level vectors on
    the pencils of kernels_nscal*.npz through libref_<cfg>_ppm.so, and whole runs."""
    for cfg, nscal in (("ifront_ppm", 1), ("blast_ppm", 0)):
        L = C.CDLL(os.path.join(REFBIN, f"libref_{cfg}.so"))
        k = np.load(os.path.join(HERE, f"kernels_nscal{nscal}.npz"))
        L.ref_set_gamma.argtypes = [C.c_double]
        L.ref_set_gamma(float(k["gamma"]))
        Wp = np.ascontiguousarray(k["Wp"]); m = Wp.shape[0]
        il, iu = 3, m - 4                                   # needs W over [il-3, iu+3]
        Wl = np.zeros_like(Wp); Wr = np.zeros_like(Wp)
        L.ref_lr_states.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int, C.c_int,
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ref_lr_states(m, dp(Wp), float(k["dt"]), float(k["dx"]), il, iu, dp(Wl), dp(Wr))
        np.savez_compressed(os.path.join(HERE, f"kernels_ppm_nscal{nscal}.npz"), gamma=k["gamma"], Wp=Wp, dt=k["dt"], dx=k["dx"],
                            il=il, iu=iu, Wl=Wl, Wr=Wr)
        print(f"kernels_ppm_nscal{nscal}: pencil of {m}")
    ifront = os.path.join(REF, "tst/ionradiation/athinput.ifront")
    sphere = os.path.join(REF, "tst/massloss/athinput.ioniz_sphere_hires")
    blast = os.path.join(REF, "tst/3D-hydro/athinput.blast")
    f, l, it = run_reference("blast_ppm", blast, (16, 12, 20), 5,
                             ["job/num_domains=1", "job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], "Blast", 0, False)
    save("ppm_blast_16x12x20_n5", f, l, it, (16, 12, 20), [])
    f, l, it = run_reference("ifront_ppm", ifront, (16, 8, 8), 4,
                             ["job/maxout=3", "output3/out_fmt=rst", "output3/dt=1e300", "output1/dt=1e300", "output2/dt=1e300"],
                             "ifront", 1, True)
    save("ppm_ifront_16x8x8_n4", f, l, it, (16, 8, 8), [])
    f, l, it = run_reference("ioniz_sphere_ppm", sphere, (32, 32, 32), 2,
                             ["job/num_domains=1", "job/maxout=1", "output1/dt=1e300", "problem/rp=2.1e10"], "ioniz_sphere", 1, True)
    save("ppm_ioniz_sphere_32x32x32_n2", f, l, it, (32, 32, 32), ["problem/rp=2.1e10"])


def developed_states():
    """Pairs of reference states (step A, step B > A) of developed flows: the tests load state A
    (U, time, dt, nstep -- everything a restart carries, restart.c:531-560) and must arrive at B."""
    sphere = os.path.join(REF, "tst/massloss/athinput.ioniz_sphere_hires")
    blast = os.path.join(REF, "tst/3D-hydro/athinput.blast")
    ifront = os.path.join(REF, "tst/ionradiation/athinput.ifront")
    cases = [("blast", "blast", blast, (24, 24, 24), 30, 34,
              ["job/num_domains=1", "job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], 0, False),
             ("ioniz_sphere", "ioniz_sphere", sphere, (32, 32, 32), 12, 15,    # NaN-free at this resolution
              ["job/num_domains=1", "job/maxout=1", "output1/dt=1e300"], 1, True),
             ("ifront", "ifront", ifront, (24, 8, 8), 40, 44,
              ["job/maxout=3", "output3/out_fmt=rst", "output3/dt=1e300", "output1/dt=1e300", "output2/dt=1e300"], 1, True)]
    # The regime the headline benchmark is quoted in: dt at the CFL limit, ONE radiation sub-cycle per step (ionrad_3d.c:919-1012
    # with the loop body executed once).  At the deck's own box size that state is only NaN-free from about 256^3 up (the planet's
    # 1e5 density jump); a box of +-1.5e10 cm around the planet at 36^3 (dx = 8.3e8 cm, between the 128^3 and 256^3 decks)
    # reaches it after 27 steps and holds it for ten: sub-cycle counts 11, 5, 22 x 4, 2, 2, 2, then 1 per step.
    zoom = [f"domain1/x{d}{m}={s}1.5e10" for d in (1, 2, 3) for m, s in (("min", "-"), ("max", ""))]
    cases.append(("ioniz_sphere", "ioniz_sphere", sphere, (36, 36, 36), 27, 33,
                  ["job/num_domains=1", "job/maxout=1", "output1/dt=1e300"] + zoom, 1, True))
    for name, cfg, deck, nx, A, B, extra, nscal, ion in cases:
        _, a, ita = run_reference(cfg, deck, nx, A, extra, name, nscal, ion)
        _, b, itb = run_reference(cfg, deck, nx, B, extra, name, nscal, ion)
        d = dict(nx=np.array(nx), UA=a["U"], nstepA=a["nstep"], timeA=a["time"], dtA=a["dt"],
                 UB=b["U"], nstepB=b["nstep"], timeB=b["time"], dtB=b["dt"], niter=np.array(itb[A:], dtype=np.int64),
                 overrides=np.array([e for e in extra if e.startswith("domain1/x")]))
        if b["edgeflux"] is not None:
            d["edgefluxB"] = b["edgeflux"]
        np.savez_compressed(os.path.join(HERE, f"dev_{name}_{nx[0]}x{nx[1]}x{nx[2]}_s{A}_s{B}.npz"), **d)
        print(f"dev_{name}: steps {A}->{B}, t {a['time']:.6g}->{b['time']:.6g}, niter {itb[A:]}")


def read_rst_levels(path, nxs, nscal, ion):
    """Restart dump of a static-mesh-refinement run: one header, then the Domains' blocks one after
    the other, root first (restart.c:531-770 loops over levels)."""
    b = open(path, "rb").read()
    pos = b.index(b"N_STEP\n") + 7; nstep = struct.unpack_from("<i", b, pos)[0]
    pos = b.index(b"\nTIME\n", pos) + 6; time = struct.unpack_from("<d", b, pos)[0]
    pos = b.index(b"\nTIME_STEP\n", pos) + 11; dt = struct.unpack_from("<d", b, pos)[0]
    levels = []
    for nx in nxs:
        n = nx[0] * nx[1] * nx[2]
        U = np.zeros((nx[2], nx[1], nx[0], 6)); ef = None
        for c, lab in enumerate(LABELS):
            tag = b"\n" + lab.encode() + b"\n"; pos = b.index(tag, pos) + len(tag)
            U[..., c] = np.frombuffer(b, "<f8", n, pos).reshape(nx[2], nx[1], nx[0]); pos += 8 * n
        if ion:
            tag = b"\nEDGEFLUX\n"; pos = b.index(tag, pos) + len(tag)
            ne = (nx[0] + 1) * (nx[1] + 1) * (nx[2] + 1)
            ef = np.frombuffer(b, "<f8", ne, pos).reshape(nx[2] + 1, nx[1] + 1, nx[0] + 1).copy(); pos += 8 * ne
        if nscal:
            tag = b"\nSCALAR 0\n"; pos = b.index(tag, pos) + len(tag)
            U[..., 5] = np.frombuffer(b, "<f8", n, pos).reshape(nx[2], nx[1], nx[0]); pos += 8 * n
        levels.append((U, ef))
    return dict(nstep=nstep, time=time, dt=dt, levels=levels)


def smr_runs():
    """Static mesh refinement (reference built with STATIC_MESH_REFINEMENT, serial): nested levels,
    one Domain each, from the reference's own decks (they carry <domain2..> blocks) plus overrides.
    The fixture keeps the overrides: the tests apply the same ones to our decks."""
    sphere = os.path.join(REF, "tst/massloss/athinput.ioniz_sphere_hires")
    blast = os.path.join(REF, "tst/3D-hydro/athinput.blast")

    def dom(n, nx, disp=None):
        o = [f"domain{n}/Nx{d + 1}={nx[d]}" for d in range(3)]
        if disp:
            o += [f"domain{n}/{k}Disp={disp[d]}" for d, k in enumerate("ijk")]
        return o

    cases = [
        # 2 levels, radiation + gravity + Userwork; planet large enough to stay NaN-free
        ("smr_ioniz_sphere_2lev_s4", "ioniz_sphere_smr", sphere, ["job/maxout=1", "output1/dt=1e300"], 4, 1, True,
         ["job/num_domains=2"] + dom(1, (32, 32, 32)) + dom(2, (32, 28, 24), (16, 18, 20)) + ["problem/rp=2.1e10"]),
        # the deck's own planet at this resolution: NaN zones appear in step 2 (order-sensitive MAX chains)
        ("smr_ioniz_sphere_2lev_nan_s2", "ioniz_sphere_smr", sphere, ["job/maxout=1", "output1/dt=1e300"], 2, 1, True,
         ["job/num_domains=2"] + dom(1, (32, 32, 32)) + dom(2, (24, 24, 24), (20, 20, 20))),
        # 3 levels, hydro only, every level displaced, periodic root
        ("smr_blast_3lev_s6", "blast_smr", blast, ["job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], 6, 0, False,
         ["job/num_domains=3"] + dom(1, (16, 24, 16)) + dom(2, (12, 16, 20), (8, 20, 6)) + dom(3, (12, 8, 16), (20, 48, 16))),
        # 3 levels touching the root boundary (outflow / reflecting), level 2 two zones from level 1's edge
        ("smr_blast_3lev_edge_s8", "blast_smr", blast, ["job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], 8, 0, False,
         ["job/num_domains=3"] + dom(1, (16, 16, 16)) + dom(2, (20, 16, 20), (0, 16, 6)) + dom(3, (16, 12, 16), (0, 36, 16))
         + ["domain1/bc_ix1=2", "domain1/bc_ox1=2", "domain1/bc_ix2=2", "domain1/bc_ox2=2", "domain1/bc_ix3=1",
            "domain1/bc_ox3=1", "domain1/x2min=-0.5", "domain1/x2max=0.5", "problem/radius=0.3"]),
    ]
    # several Domains on a level (MeshS.Domain[nl][nd]): two level-1 patches one root zone apart in x1 with overlapping x2 / x3
    # ranges -- the root zones between them are flux-corrected from both -- and two patches that both take the radiation
    # from the root
    cases += [
        ("smr_blast_2dom_s6", "blast_smr", blast, ["job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], 6, 0, False,
         ["job/num_domains=3"] + dom(1, (16, 24, 16)) + dom(2, (12, 16, 20), (8, 20, 6)) + ["domain3/level=1"] + dom(3, (6, 8, 8), (22, 24, 10))),
        ("smr_ioniz_sphere_2dom_s3", "ioniz_sphere_smr", sphere, ["job/maxout=1", "output1/dt=1e300"], 3, 1, True,
         ["job/num_domains=3"] + dom(1, (32, 32, 32)) + dom(2, (32, 28, 24), (16, 18, 20)) + ["domain3/level=1"] + dom(3, (16, 16, 12), (8, 20, 48))
         + ["problem/rp=2.1e10"]),
    ]
    # a tree: two level-1 Domains and a level-2 Domain under the SECOND of them (the decks stop at <domain3>: the fourth block
    # travels with the fixture as text to append)
    extra4 = "\n<domain4>\nlevel = 2\nNx1 = 12\nNx2 = 12\nNx3 = 12\niDisp = 48\njDisp = 44\nkDisp = 40\n"
    cases += [
        ("smr_blast_tree_s5", "blast_smr", blast, ["job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], 5, 0, False,
         ["job/num_domains=4"] + dom(1, (20, 20, 20)) + dom(2, (12, 12, 12), (4, 4, 4)) + ["domain3/level=1"] + dom(3, (12, 12, 12), (20, 18, 16))
         + dom(4, (12, 12, 12), (48, 44, 40)) + ["problem/radius=0.25"]),
    ]
    # the same 2-level blast with the van Leer integrator and with third-order reconstruction
    two = ["job/num_domains=2"] + dom(1, (16, 24, 16)) + dom(2, (12, 16, 20), (8, 20, 6))
    cases += [("smr_vl_blast_2lev_s5", "blast_smr_vl", blast, ["job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], 5, 0, False, two),
              ("smr_ppm_blast_2lev_s5", "blast_smr_ppm", blast, ["job/maxout=1", "output1/out_fmt=rst", "output1/dt=1e300"], 5, 0, False, two)]
    for name, cfg, deck, refextra, nlim, nscal, ion, over in cases:
        if os.environ.get("GOLDEN_ONLY") and os.environ["GOLDEN_ONLY"] not in name:
            continue                                   # (regenerate a subset: GOLDEN_ONLY=2dom python make_golden.py smr)
        nlev = int(over[0].split("=")[1])
        nxs = [tuple(int(next(o for o in over if o.startswith(f"domain{n}/Nx{d}=")).split("=")[1]) for d in (1, 2, 3))
               for n in range(1, nlev + 1)]
        tmp = tempfile.mkdtemp(prefix="golden_")
        rundir = os.path.join(tmp, "run")
        extra = extra4 if "tree" in name else ""
        if extra:                                                  # a deck with the extra <domainN> block
            open(os.path.join(tmp, "deck"), "w").write(open(deck).read() + extra)
            deck = os.path.join(tmp, "deck")
        pr = subprocess.run([os.path.join(REFBIN, "athena_" + cfg), "-i", deck, "-d", rundir, f"time/nlim={nlim}"] + refextra + over,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp)
        if pr.returncode != 0:
            raise RuntimeError(pr.stdout[-2000:] + pr.stderr[-2000:])
        niter = [int(m) for m in re.findall(r"Radiation done in (\d+) iterations", pr.stderr)]
        rsts = sorted(f for f in os.listdir(rundir) if f.endswith(".rst"))
        first = read_rst_levels(os.path.join(rundir, rsts[0]), nxs, nscal, ion)
        last = read_rst_levels(os.path.join(rundir, rsts[-1]), nxs, nscal, ion)
        shutil.rmtree(tmp)
        d = dict(overrides=np.array(over), extra_deck=np.array(extra), nlevels=nlev, nstep=last["nstep"], time=last["time"], dt=last["dt"],
                 dt0=first["dt"], niter=np.array(niter, dtype=np.int64))
        for l, (U, ef) in enumerate(last["levels"]):
            d[f"U{l}"] = U
            if ef is not None:
                d[f"edgeflux{l}"] = ef
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(f"{name}: nstep={last['nstep']} time={last['time']:.17g} dt={last['dt']:.17g} niter={niter} "
              f"nan={[int(np.isnan(U).sum()) for U, _ in last['levels']]}")


def history_runs():
    """.hst files written by the reference (dump_history.c): the text is the expected output of our
    history writer for the same run."""
    sphere = os.path.join(REF, "tst/massloss/athinput.ioniz_sphere_hires")
    blast = os.path.join(REF, "tst/3D-hydro/athinput.blast")
    for name, cfg, deck, nx, nlim, extra in (
            ("hst_blast_16x16x16_s3", "blast", blast, (16, 16, 16), 3,
             ["job/num_domains=1", "job/maxout=1", "output1/out_fmt=hst", "output1/dt=1e-9"]),
            ("hst_ioniz_sphere_20x20x20_s2", "ioniz_sphere", sphere, (20, 20, 20), 2,
             ["job/num_domains=1", "job/maxout=1", "output1/out_fmt=hst", "output1/dt=1e-9"])):
        tmp = tempfile.mkdtemp(prefix="golden_")
        rundir = os.path.join(tmp, "run")
        pr = subprocess.run([os.path.join(REFBIN, "athena_" + cfg), "-i", deck, "-d", rundir, f"time/nlim={nlim}"]
                            + [f"domain1/Nx{d + 1}={nx[d]}" for d in range(3)] + extra,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp)
        if pr.returncode != 0:
            raise RuntimeError(pr.stdout[-2000:] + pr.stderr[-2000:])
        hst = [f for f in os.listdir(rundir) if f.endswith(".hst")]
        text = open(os.path.join(rundir, hst[0])).read()
        shutil.rmtree(tmp)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), nx=np.array(nx), nstep=nlim, text=np.array(text),
                            basename=np.array(hst[0][:-4]))
        print(f"{name}: {len(text.splitlines())} lines in {hst[0]}")


# ------------------------------------------------------------------------------------
def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def kernel_vectors():
    rng = np.random.default_rng(20261004)
    gamma = 1.666666666667
    for cfg, nscal in (("ifront", 1), ("blast", 0)):
        L = C.CDLL(os.path.join(REFBIN, f"libref_{cfg}.so"))
        nv = 5 + nscal
        assert L.ref_nvar() == nv
        L.ref_set_gamma.argtypes = [C.c_double]
        L.ref_set_gamma(gamma)
        # ---- Riemann states: smooth, strong jumps, supersonic both ways, near-vacuum (HLLE) ----
        n = 4096

        def state(n, dlo, dhi, vscale, plo, phi):
            d = np.exp(rng.uniform(np.log(dlo), np.log(dhi), n))
            v = rng.normal(0.0, vscale, (n, 3))
            p = np.exp(rng.uniform(np.log(plo), np.log(phi), n))
            U = np.zeros((n, nv))
            U[:, 0] = d
            U[:, 1:4] = d[:, None] * v
            U[:, 4] = p / (gamma - 1.0) + 0.5 * d * (v ** 2).sum(1)
            if nscal:
                U[:, 5] = d * rng.uniform(1e-4, 1.0, n)
            return U

        Ul = state(n, 0.1, 10.0, 1.0, 0.05, 20.0)
        Ur = Ul + 0.01 * (state(n, 0.1, 10.0, 1.0, 0.05, 20.0) - Ul)     # near-smooth first quarter
        Ur[n // 4:] = state(n - n // 4, 0.1, 10.0, 1.0, 0.05, 20.0)       # independent jumps
        q = n // 8
        Ul[4 * q:5 * q, 1] += 30.0 * Ul[4 * q:5 * q, 0]; Ur[4 * q:5 * q, 1] += 30.0 * Ur[4 * q:5 * q, 0]
        Ul[5 * q:6 * q, 1] -= 30.0 * Ul[5 * q:6 * q, 0]; Ur[5 * q:6 * q, 1] -= 30.0 * Ur[5 * q:6 * q, 0]
        for U in (Ul, Ur):   # keep E consistent after the boosts
            pass
        Ul[4 * q:6 * q, 4] = 1.0 / (gamma - 1) + 0.5 * (Ul[4 * q:6 * q, 1:4] ** 2).sum(1) / Ul[4 * q:6 * q, 0]
        Ur[4 * q:6 * q, 4] = 1.5 / (gamma - 1) + 0.5 * (Ur[4 * q:6 * q, 1:4] ** 2).sum(1) / Ur[4 * q:6 * q, 0]
        # strong rarefactions: receding flows over a big density/pressure contrast -> HLLE fallback
        Ul[6 * q:7 * q] = state(q, 1.0, 5.0, 0.1, 1.0, 5.0); Ur[6 * q:7 * q] = state(q, 1e-6, 1e-4, 0.1, 1e-8, 1e-6)
        Ul[6 * q:7 * q, 1] -= 8.0 * Ul[6 * q:7 * q, 0]; Ur[6 * q:7 * q, 1] += 8.0 * Ur[6 * q:7 * q, 0]
        Ul[7 * q:, :] = state(n - 7 * q, 1e-3, 1e3, 3.0, 1e-4, 1e2); Ur[7 * q:, :] = state(n - 7 * q, 1e-3, 1e3, 3.0, 1e-4, 1e2)
        Ul[6 * q:7 * q, 4] = 1.0 / (gamma - 1) + 0.5 * (Ul[6 * q:7 * q, 1:4] ** 2).sum(1) / Ul[6 * q:7 * q, 0]
        Ur[6 * q:7 * q, 4] = 1e-7 / (gamma - 1) + 0.5 * (Ur[6 * q:7 * q, 1:4] ** 2).sum(1) / Ur[6 * q:7 * q, 0]
        eta = np.where(rng.uniform(size=n) < 0.5, 0.0, rng.uniform(0.0, 3.0, n))
        F = np.zeros((n, nv))
        L.ref_fluxes(n, dp(Ul), dp(Ur), dp(eta), dp(F))
        W = np.zeros((n, nv)); L.ref_cons_to_prim(n, dp(Ul), dp(W))
        cf = np.zeros(n); L.ref_cfast(n, dp(Ul), dp(cf))
        # ---- reconstruction pencils: smooth, shocks, extrema, both flow signs ----
        m = 2048
        x = np.linspace(0, 1, m)
        Wp = np.zeros((m, nv))
        Wp[:, 0] = 1.0 + 0.5 * np.sin(14 * np.pi * x) + (x > 0.5) * 2.0
        Wp[:, 1] = 1.5 * np.sin(6 * np.pi * x) + rng.normal(0, 0.05, m)
        Wp[:, 2] = rng.normal(0, 0.5, m)
        Wp[:, 3] = np.cos(9 * np.pi * x)
        Wp[:, 4] = 1.0 + 0.3 * np.cos(22 * np.pi * x) + (x > 0.25) * 3.0 + rng.uniform(0, 0.05, m)
        Wp[m // 2:m // 2 + 200, 1] += 6.0          # supersonic to the right
        Wp[m // 4:m // 4 + 200, 1] -= 6.0          # supersonic to the left
        Wp[100:140, :] = Wp[100, :]                # flat stretch: zero slopes
        if nscal:
            Wp[:, 5] = np.clip(0.5 + 0.5 * np.sin(31 * np.pi * x) + rng.normal(0, 0.02, m), 1e-4, 1.0)
        dtp, dxp = 0.004, 0.01
        il, iu = 3, m - 4
        Wl = np.zeros((m, nv)); Wr = np.zeros((m, nv))
        L.ref_lr_states.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int, C.c_int,
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ref_lr_states(m, dp(Wp), dtp, dxp, il, iu, dp(Wl), dp(Wr))
        np.savez_compressed(os.path.join(HERE, f"kernels_nscal{nscal}.npz"), gamma=gamma, Ul=Ul, Ur=Ur, eta=eta,
                            F=F, W=W, cfast=cf, Wp=Wp, dt=dtp, dx=dxp, il=il, iu=iu, Wl=Wl, Wr=Wr)
        print(f"kernels_nscal{nscal}: {n} Riemann problems, pencil of {m}")


if __name__ == "__main__":
    if not os.path.isdir(REF) or not os.path.isdir(REFBIN):
        sys.exit("needs /root/reference and oracle/_ref (make -C oracle ref)")
    which = sys.argv[1:] or ["whole", "shk", "dev", "kernels", "ppm", "smr", "hst", "ray", "cool", "vlppm", "noh"]
    if "cool" in which:
        cooling_runs()
    if "vlppm" in which:
        vl_ppm_runs()
    if "noh" in which:
        no_h_correction_runs()
    if "ray" in which:
        rayplane_runs()
    if "hst" in which:
        history_runs()
    if "whole" in which:
        whole_runs()
    if "ppm" in which:
        ppm_runs()
    if "shk" in which:
        shock_tubes()
    if "dev" in which:
        developed_states()
    if "kernels" in which:
        kernel_vectors()
    if "smr" in which:
        smr_runs()
