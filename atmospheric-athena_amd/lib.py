"""ctypes binding of the C-ABI (include/athena_amd.h) and of the host-side C problem files.

There is no CPU fallback: if libathena_amd.so is missing or no MI355X is visible the calls
raise.  ``Grid`` mirrors the reference's per-step call sites one-to-one (bvals_mhd, new_dt,
integrate_3d_ctu, ion_radtransfer_3d; main.c:519-669) on a device-resident Grid.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .config import GridConfig, NGHOST

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class AthenaError(RuntimeError):
    """ath_error() of the reference (utils.c:118): every failure is fatal to the run."""


class aa_params(C.Structure):
    _fields_ = [
        ("Nx", C.c_int * 3), ("rootNx", C.c_int * 3),
        ("xmin", C.c_double * 3), ("xmax", C.c_double * 3), ("MinX", C.c_double * 3),
        ("bc", C.c_int * 6), ("nscal", C.c_int), ("ion", C.c_int),
        ("gamma", C.c_double), ("cour_no", C.c_double), ("tlim", C.c_double),
        ("sigma_ph", C.c_double), ("m_H", C.c_double), ("mu", C.c_double), ("e_gamma", C.c_double),
        ("alpha_C", C.c_double), ("k_B", C.c_double), ("time_unit", C.c_double),
        ("max_de_iter", C.c_double), ("max_de_therm_iter", C.c_double), ("max_dx_iter", C.c_double),
        ("max_de_step", C.c_double), ("max_de_therm_step", C.c_double), ("max_dx_step", C.c_double),
        ("tfloor", C.c_double), ("tceil", C.c_double),
        ("maxiter", C.c_int), ("device", C.c_int), ("integrator", C.c_int), ("level", C.c_int),
        ("order", C.c_int), ("ion_path", C.c_int), ("nslab", C.c_int),
    ]


ION_WORDS = 8     # AA_ION_WORDS of include/athena_amd.h

GRAVPOT = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_double)
GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p)      # aa_gather_fn of include/athena_amd.h

_libs = {}
_host = None


def build(force: bool = False) -> None:
    """Compile the HIP libraries and the host C library in-tree (hipcc cross-compiles gfx950)."""
    import subprocess
    args = ["make", "-C", os.path.join(HERE, "csrc"), "-j8"] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host")] + (["-B"] if force else []),
                          stdout=subprocess.DEVNULL)


def strict_default() -> bool:
    return os.environ.get("ATHENA_AMD_STRICT", "0") not in ("", "0")


def load(strict: bool | None = None) -> C.CDLL:
    strict = strict_default() if strict is None else strict
    if strict in _libs:
        return _libs[strict]
    name = "libathena_amd_strict.so" if strict else "libathena_amd.so"
    if os.environ.get("ATHENA_AMD_VARIANT") and not strict:        # A/B experiment builds (csrc/Makefile `variant`)
        name = f"libathena_amd_{os.environ['ATHENA_AMD_VARIANT']}.so"
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        raise AthenaError(f"{path} is missing: run __graft_entry__.build() (no CPU fallback exists)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.  If this library pulled
    # in the system runtime first, a later `import torch` in the same process would find "No HIP
    # GPUs".  Loading torch first makes both share torch's runtime (same SONAME).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    P = C.c_void_p; D = C.c_double; I = C.c_int; LL = C.c_longlong
    dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int); llp = C.POINTER(C.c_longlong)
    sig = {
        "aa_create": (I, [C.POINTER(aa_params), C.POINTER(P)]),
        "aa_destroy": (None, [P]),
        "aa_last_error": (C.c_char_p, []),
        "aa_set_stream": (I, [P, P]),
        "aa_sync": (I, [P]),
        "aa_device_bytes": (LL, [P]),
        "aa_upload_cons": (I, [P, dp]), "aa_download_cons": (I, [P, dp]), "aa_download_ghost_zones": (I, [P, dp]),
        "aa_upload_edgeflux": (I, [P, dp]), "aa_download_edgeflux": (I, [P, dp]),
        "aa_get_mesh_state": (I, [P, dp, dp, ip]), "aa_set_mesh_state": (I, [P, D, D, I]),
        "aa_set_static_grav_pot": (I, [P, GRAVPOT]),
        "aa_set_cooling": (I, [P, I]),
        "aa_set_static_grav_tables": (I, [P, dp, dp, dp, dp]),
        "aa_set_pinned_cells": (I, [P, LL, llp, dp]), "aa_apply_pinned_cells": (I, [P]),
        "aa_add_radplane_3d": (I, [P, I, D]), "aa_has_radplane": (I, [P]),
        "aa_bvals_mhd": (I, [P]), "aa_bvals_mhd_side": (I, [P, I, I]), "aa_bvals_ionrad": (I, [P]), "aa_new_dt": (I, [P]),
        "aa_integrate_3d_ctu": (I, [P]), "aa_integrate_begin": (I, [P]), "aa_cfl_in_update": (I, [P, I]), "aa_integrate_3d_vl": (I, [P]), "aa_ion_radtransfer_3d": (I, [P, ip]),
        "aa_start": (I, [P]), "aa_step": (I, [P, ip]),
        "aa_new_dt_local": (I, [P, dp]), "aa_ion_begin": (I, [P]), "aa_ion_rates": (I, [P, dp, dp]),
        "aa_ion_update": (I, [P, D, llp, dp]),
        "aa_ion_is_fused": (I, [P]), "aa_ion_speculate": (I, [P, D]), "aa_ion_pass": (I, [P, I, I, P]), "aa_ion_pick": (I, [P, P, I, I, D]),
        "aa_ion_fetch": (I, [P, dp, ip, dp, dp, llp, dp, ip]), "aa_ion_finish": (I, [P]),
        "aa_ion_radtransfer_3d_gather": (I, [P, P, P, I, GATHER, P, ip]), "aa_host_syncs": (I, [P, I]),
        "aa_halo_doubles": (LL, [P]), "aa_pack_x3": (I, [P, I, P]), "aa_unpack_x3": (I, [P, I, P]),
        "aa_halo_doubles_x2": (LL, [P]), "aa_pack_x2": (I, [P, I, P]), "aa_unpack_x2": (I, [P, I, P]),
        "aa_halo_doubles_dir": (LL, [P, I]), "aa_halo_get": (I, [P, I, I, dp]), "aa_halo_put": (I, [P, I, I, dp]), "aa_device_count": (I, []),
        "aa_mesh_create": (I, [I, C.POINTER(P), ip, C.POINTER(P)]), "aa_mesh_destroy": (None, [P]),
        "aa_mesh_get_state": (I, [P, dp, dp, ip]), "aa_mesh_set_state": (I, [P, D, D, I]),
        "aa_mesh_set_stream": (I, [P, P]), "aa_mesh_restrict_correct_pair": (I, [P, I]),
        "aa_mesh_restrict_correct": (I, [P]), "aa_mesh_ionrad_restrict_correct": (I, [P]),
        "aa_mesh_prolongate": (I, [P]), "aa_mesh_new_dt": (I, [P]), "aa_mesh_ion_radtransfer": (I, [P, I, ip]),
        "aa_mesh_start": (I, [P]), "aa_mesh_step": (I, [P, ip]),
        "aa_mesh_ionflux_prolong": (I, [P, I]), "aa_mesh_create_local": (I, [I, C.POINTER(P), ip, C.POINTER(P)]),
        "aa_flux_x3_export": (I, [P, I, P]), "aa_flux_x3_apply": (I, [P, I, I, I, I, I, P]), "aa_cfl_max_v": (I, [P, dp]),
        "aa_test_fluxes": (I, [I, D, I, dp, dp, dp, dp]),
        "aa_test_lr_states": (I, [I, D, I, dp, D, D, I, I, dp, dp]),
        "aa_test_lr_states_ppm": (I, [I, D, I, dp, D, D, I, I, dp, dp]),
        "aa_test_explog": (I, [I, dp, dp, dp]),
        "aa_test_xdiv": (I, [I, dp, dp, dp]),
        "aa_history": (I, [P, dp]),
        "aa_profile_enable": (I, [P, I]), "aa_profile_reset": (I, [P]), "aa_profile_count": (I, [P]),
        "aa_profile_name": (C.c_char_p, [P, I]), "aa_profile_get": (I, [P, I, dp, llp]),
    }
    for name_, (res, args) in sig.items():
        f = getattr(L, name_)          # AttributeError here = header and library disagree
        f.restype = res; f.argtypes = args
    L._sig = sig
    _libs[strict] = L
    return L


def host() -> C.CDLL:
    global _host
    if _host is None:
        path = os.path.join(HERE, "libathena_amd_host.so")
        if not os.path.exists(path):
            raise AthenaError(f"{path} is missing: run __graft_entry__.build()")
        H = C.CDLL(path)
        dp = C.POINTER(C.c_double); pp = C.POINTER(aa_params); D = C.c_double
        H.aa_problem_ifront.argtypes = [pp, D, D, dp]; H.aa_problem_ifront.restype = C.c_int
        H.aa_problem_ioniz_sphere.argtypes = [pp, D, D, D, D, dp]; H.aa_problem_ioniz_sphere.restype = C.c_int
        H.aa_problem_blast.argtypes = [pp, D, D, D, D, D, dp]; H.aa_problem_blast.restype = C.c_int
        H.aa_problem_shkset1d.argtypes = [pp, dp, dp, C.c_int, dp]; H.aa_problem_shkset1d.restype = C.c_int
        H.aa_planet_pot.argtypes = [D, D, D]; H.aa_planet_pot.restype = D
        H.aa_ioniz_sphere_pinned.argtypes = [pp, C.POINTER(C.c_longlong), dp]
        H.aa_ioniz_sphere_pinned.restype = C.c_longlong
        _host = H
    return _host


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def params_from_grid(g: GridConfig, device: int = 0, ion_path: int = 0, nslab: int = 1) -> aa_params:
    r = g.run
    p = aa_params()
    for d in range(3):
        p.Nx[d] = g.Nx[d]; p.rootNx[d] = r.rootNx[d]
        p.xmin[d] = r.xmin[d]; p.xmax[d] = r.xmax[d]; p.MinX[d] = g.MinX[d]
    for b in range(6):
        p.bc[b] = g.bc[b]
    p.nscal = r.nscal; p.ion = 1 if r.ion else 0
    p.gamma = r.gamma; p.cour_no = r.cour_no; p.tlim = r.tlim
    for k, v in r.ionp.items():
        setattr(p, k, v)
    p.maxiter = r.maxiter
    p.device = device
    p.integrator = {"ctu": 0, "vl": 1, "ctu-noh": 2}[r.integrator]
    p.level = g.level
    p.order = getattr(r, "order", 2)
    p.ion_path = ion_path
    p.nslab = nslab
    return p


class Grid:
    """Device-resident Grid.  Method names follow the reference's call sites."""

    def __init__(self, grid: GridConfig, device: int = 0, strict: bool | None = None, ion_path: int = 0, nslab: int = 1):
        """nslab > 1: the Grid is cut into x3 slabs inside the library (aa_params.nslab), one per GPU."""
        self.cfg = grid
        self.L = load(strict)
        self.params = params_from_grid(grid, device, ion_path, nslab)
        self.nvar = 5 + grid.run.nscal
        self.N = tuple(n + 2 * NGHOST for n in grid.Nx)          # (N1, N2, N3)
        h = C.c_void_p()
        self._h = None
        self._chk(self.L.aa_create(C.byref(self.params), C.byref(h)))
        self._h = h
        self._keep = []

    def _chk(self, rc: int):
        if rc != 0:
            raise AthenaError(self.L.aa_last_error().decode() or f"athena_amd error {rc}")

    def close(self):
        if self._h is not None:
            self.L.aa_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state --------------------------------------------------------------------
    def new_host_block(self) -> np.ndarray:
        """A zeroed host ConsS block [N3][N2][N1][nvar] (GridS.U of the reference)."""
        return np.zeros((self.N[2], self.N[1], self.N[0], self.nvar), dtype=np.float64)

    def upload(self, U: np.ndarray):
        assert U.shape == (self.N[2], self.N[1], self.N[0], self.nvar) and U.dtype == np.float64
        U = np.ascontiguousarray(U)
        self._chk(self.L.aa_upload_cons(self._h, _dp(U)))

    def download(self) -> np.ndarray:
        U = self.new_host_block()
        self._chk(self.L.aa_download_cons(self._h, _dp(U)))
        return U

    def download_ghost_zones(self, U: np.ndarray):
        """refresh only the ghost zones of a host block whose active zones are current"""
        self._chk(self.L.aa_download_ghost_zones(self._h, _dp(U)))

    def download_edgeflux(self) -> np.ndarray:
        nx = self.cfg.Nx
        ef = np.zeros((nx[2] + 1, nx[1] + 1, nx[0] + 1))
        self._chk(self.L.aa_download_edgeflux(self._h, _dp(ef)))
        return ef

    def mesh_state(self):
        t = C.c_double(); dt = C.c_double(); n = C.c_int()
        self.L.aa_get_mesh_state(self._h, C.byref(t), C.byref(dt), C.byref(n))
        return t.value, dt.value, n.value

    def set_mesh_state(self, time: float, dt: float, nstep: int):
        self.L.aa_set_mesh_state(self._h, time, dt, nstep)

    time = property(lambda s: s.mesh_state()[0])
    dt = property(lambda s: s.mesh_state()[1])
    nstep = property(lambda s: s.mesh_state()[2])

    def set_stream(self, stream_ptr: int): self._chk(self.L.aa_set_stream(self._h, C.c_void_p(stream_ptr)))
    def sync(self): self._chk(self.L.aa_sync(self._h))
    def device_bytes(self) -> int: return int(self.L.aa_device_bytes(self._h))

    # ---- hooks --------------------------------------------------------------------
    def set_static_grav_pot(self, cfunc):
        """cfunc: a C function pointer double(double,double,double) (or None)."""
        cb = GRAVPOT(0) if cfunc is None else C.cast(cfunc, GRAVPOT)
        self._keep.append(cb)
        self._chk(self.L.aa_set_static_grav_pot(self._h, cb))

    def set_pinned_cells(self, index: np.ndarray, values: np.ndarray):
        index = np.ascontiguousarray(index, dtype=np.int64); values = np.ascontiguousarray(values, dtype=np.float64)
        self._chk(self.L.aa_set_pinned_cells(self._h, len(index),
                                             index.ctypes.data_as(C.POINTER(C.c_longlong)), _dp(values)))

    def add_radplane_3d(self, dir: int, flux: float): self._chk(self.L.aa_add_radplane_3d(self._h, dir, flux))
    def has_radplane(self) -> bool: return bool(self.L.aa_has_radplane(self._h))

    # ---- call sites of the main loop ------------------------------------------------
    def bvals_mhd(self): self._chk(self.L.aa_bvals_mhd(self._h))
    def bvals_ionrad(self): self._chk(self.L.aa_bvals_ionrad(self._h))
    def new_dt(self): self._chk(self.L.aa_new_dt(self._h))
    def set_cooling(self, kind: int = 1):
        """CoolingFunc = KoyInut (kind 1, AA_COOL_KOYINUT) or NULL (0); CTU integrator only."""
        self._chk(self.L.aa_set_cooling(self._h, int(kind)))

    def integrate_3d_ctu(self): self._chk(self.L.aa_integrate_3d_ctu(self._h))
    def integrate_3d_vl(self): self._chk(self.L.aa_integrate_3d_vl(self._h))
    def integrate_begin(self): self._chk(self.L.aa_integrate_begin(self._h))
    def ion_speculate(self, limit: float): self._chk(self.L.aa_ion_speculate(self._h, float(limit)))
    def cfl_in_update(self, on: bool = True): self._chk(self.L.aa_cfl_in_update(self._h, 1 if on else 0))

    def integrate(self):
        """(*Integrate)(pD): the function pointer integrate_init() selected (integrate.c:63-75)."""
        if self.cfg.run.integrator == "vl":
            self.integrate_3d_vl()
        else:
            self.integrate_3d_ctu()
    def apply_pinned_cells(self): self._chk(self.L.aa_apply_pinned_cells(self._h))
    def start(self): self._chk(self.L.aa_start(self._h))

    def ion_radtransfer_3d(self) -> int:
        n = C.c_int(); self._chk(self.L.aa_ion_radtransfer_3d(self._h, C.byref(n))); return n.value

    def step(self) -> int:
        n = C.c_int(); self._chk(self.L.aa_step(self._h, C.byref(n))); return n.value

    # ---- phases ----------------------------------------------------------------------
    def new_dt_local(self) -> float:
        v = C.c_double(); self._chk(self.L.aa_new_dt_local(self._h, C.byref(v))); return v.value

    def ion_begin(self): self._chk(self.L.aa_ion_begin(self._h))

    def ion_rates(self):
        a = C.c_double(); b = C.c_double()
        self._chk(self.L.aa_ion_rates(self._h, C.byref(a), C.byref(b))); return a.value, b.value

    def ion_update(self, dt: float):
        c = C.c_longlong(); h = C.c_double()
        self._chk(self.L.aa_ion_update(self._h, dt, C.byref(c), C.byref(h))); return c.value, h.value

    # the one-kernel sub-cycle (aa_ion_is_fused): pass / [all-gather of the words] / pick / fetch
    def ion_is_fused(self) -> bool: return bool(self.L.aa_ion_is_fused(self._h))

    def ion_pass(self, update: bool, sweep: bool, dev_words: int = 0):
        self._chk(self.L.aa_ion_pass(self._h, int(update), int(sweep), C.c_void_p(dev_words or None)))

    def ion_pick(self, dev_words_all: int, nranks: int, first: bool, limit: float):
        self._chk(self.L.aa_ion_pick(self._h, C.c_void_p(dev_words_all or None), nranks, int(first), limit))

    def ion_fetch(self):
        """-> (dt, limit_hit, dt_chem, dt_therm, cellcount, dt_hydro, neg_dt_chem) of the update the last pass applied:
        the one read-back of a sub-cycle"""
        dt = C.c_double(); a = C.c_double(); b = C.c_double(); h = C.c_double(); hit = C.c_int(); neg = C.c_int(); n = C.c_longlong()
        self._chk(self.L.aa_ion_fetch(self._h, C.byref(dt), C.byref(hit), C.byref(a), C.byref(b), C.byref(n), C.byref(h), C.byref(neg)))
        return dt.value, bool(hit.value), a.value, b.value, n.value, h.value, bool(neg.value)

    def ion_finish(self): self._chk(self.L.aa_ion_finish(self._h))

    def ion_radtransfer_3d_gather(self, dev_words: int, dev_words_all: int, nranks: int, gather) -> int:
        """ion_radtransfer_3d of one rank of a multi-rank run in ONE call: `gather()` is the driver's all-gather of the
        AA_ION_WORDS doubles at dev_words into dev_words_all, called once per pass on this Grid's stream."""
        err = []

        def cb(_ctx):
            try:
                gather(); return 0
            except BaseException as e:      # an exception must not cross the C frames: hand it over afterwards
                err.append(e); return 1
        n = C.c_int()
        rc = self.L.aa_ion_radtransfer_3d_gather(self._h, C.c_void_p(dev_words), C.c_void_p(dev_words_all), nranks, GATHER(cb), None, C.byref(n))
        if err:
            raise err[0]
        self._chk(rc)
        return n.value

    def host_syncs(self, reset: bool = False) -> int: return int(self.L.aa_host_syncs(self._h, int(reset)))

    def cfl_max_v(self):
        v = (C.c_double * 3)(); self._chk(self.L.aa_cfl_max_v(self._h, v)); return list(v)

    def flux_x3_export(self, side: int, dev_ptr: int): self._chk(self.L.aa_flux_x3_export(self._h, side, C.c_void_p(dev_ptr)))

    def flux_x3_apply(self, side: int, i0: int, j0: int, n1: int, n2: int, dev_ptr: int):
        self._chk(self.L.aa_flux_x3_apply(self._h, side, i0, j0, n1, n2, C.c_void_p(dev_ptr)))

    def halo_doubles(self) -> int: return int(self.L.aa_halo_doubles(self._h))
    def pack_x3(self, side: int, dev_ptr: int): self._chk(self.L.aa_pack_x3(self._h, side, C.c_void_p(dev_ptr)))
    def unpack_x3(self, side: int, dev_ptr: int): self._chk(self.L.aa_unpack_x3(self._h, side, C.c_void_p(dev_ptr)))
    def halo_doubles_x2(self) -> int: return int(self.L.aa_halo_doubles_x2(self._h))
    def pack_x2(self, side: int, dev_ptr: int): self._chk(self.L.aa_pack_x2(self._h, side, C.c_void_p(dev_ptr)))
    def unpack_x2(self, side: int, dev_ptr: int): self._chk(self.L.aa_unpack_x2(self._h, side, C.c_void_p(dev_ptr)))
    def bvals_mhd_side(self, dir: int, side: int): self._chk(self.L.aa_bvals_mhd_side(self._h, dir, side))

    def history(self) -> np.ndarray:
        """Volume integrals of this Grid in .hst column order (dump_history.c:157-200)."""
        s = np.zeros(9); self._chk(self.L.aa_history(self._h, _dp(s))); return s

    # ---- measurement -------------------------------------------------------------------
    def profile_enable(self, on: bool = True): self.L.aa_profile_enable(self._h, 1 if on else 0)
    def profile_reset(self): self.L.aa_profile_reset(self._h)

    def profile(self):
        out = {}
        for i in range(self.L.aa_profile_count(self._h)):
            ms = C.c_double(); n = C.c_longlong()
            self._chk(self.L.aa_profile_get(self._h, i, C.byref(ms), C.byref(n)))
            out[self.L.aa_profile_name(self._h, i).decode()] = (ms.value, n.value)
        return out


def setup_problem(grid: GridConfig, device: int = 0, strict: bool | None = None, ion_path: int = 0, nslab: int = 1) -> Grid:
    """problem(DomainS*) of the reference for the shipped decks: fill the host block with the
    C problem generator, upload it, register the hooks (main.c:393)."""
    r = grid.run; pr = r.prob
    g = Grid(grid, device, strict, ion_path, nslab)
    H = host()
    U = g.new_host_block()
    if r.problem == "ifront":
        rc = H.aa_problem_ifront(C.byref(g.params), pr["n_H"], pr["cs"], _dp(U))
    elif r.problem == "ioniz_sphere":
        rc = H.aa_problem_ioniz_sphere(C.byref(g.params), pr["cs"], pr.get("rp", 1.2e10),
                                       pr.get("mp", 1.0e30), pr.get("np", 6.0e8), _dp(U))
    elif r.problem == "blast":
        rc = H.aa_problem_blast(C.byref(g.params), pr["radius"], pr["pamb"], pr.get("damb", 1.0),
                                pr.get("drat", 1.0), pr["prat"], _dp(U))
    elif r.problem == "shkset1d":
        wl = np.array([pr["dl"], pr["pl"], pr["v1l"], pr["v2l"], pr["v3l"]])
        wr = np.array([pr["dr"], pr["pr"], pr["v1r"], pr["v2r"], pr["v3r"]])
        rc = H.aa_problem_shkset1d(C.byref(g.params), _dp(wl), _dp(wr), int(pr["shk_dir"]), _dp(U))
    else:
        raise AthenaError(f"unknown problem {r.problem}")
    if rc != 0:
        raise AthenaError(f"problem generator {r.problem} rejected the configuration")
    g.upload(U)
    if r.problem == "ioniz_sphere":
        g.set_static_grav_pot(H.aa_planet_pot)                           # StaticGravPot = PlanetPot
        n = H.aa_ioniz_sphere_pinned(C.byref(g.params), None, None)
        idx = np.zeros(max(n, 1), dtype=np.int64); val = np.zeros((max(n, 1), 6))
        H.aa_ioniz_sphere_pinned(C.byref(g.params), idx.ctypes.data_as(C.POINTER(C.c_longlong)), _dp(val))
        g.set_pinned_cells(idx[:n], val[:n])
    if r.ion:
        if pr.get("trad", 0.0) >= 1.0e-20:                               # ioniz_sphere.c:168-170
            g.close()
            raise AthenaError("Delaying ionization not currently enabled for use with SMR + MPI")
        g.add_radplane_3d(-1, pr["flux"])
    g.host_initial = U
    return g


class Mesh:
    """Nested static-mesh-refinement levels on one GPU (MeshS; one or several Domains per level: grids level by level
    from the root, deck order inside a level, as config.levels() returns them).  Method
    names follow the reference: RestrictCorrect, Prolongate (smr.c), new_dt, and the per-level
    ion_radtransfer_3d with its coarse -> fine EdgeFlux hand-off (ionrad_smr.c)."""

    def __init__(self, grids, device: int = 0, strict: bool | None = None, links=None, ion_path: int = 0):
        """links: config.LinkConfig list for one rank's stack of slabs (multi-GPU SMR); None = the whole Mesh."""
        self.lev = [setup_problem(g, device, strict, ion_path) for g in grids]
        self.L = self.lev[0].L
        n = len(grids)
        hs = (C.c_void_p * n)(*[g._h for g in self.lev])
        h = C.c_void_p()
        self._h = None
        if links is None:
            disp = (C.c_int * (3 * n))(*[g.disp[d] if g.level else 0 for g in grids for d in range(3)])
            self._chk(self.L.aa_mesh_create(n, hs, disp, C.byref(h)))
        else:
            flat = [v for L_ in links for v in (*L_.cs, *L_.n, *L_.prol, *L_.corr, *L_.cdisp)]
            arr = (C.c_int * max(1, len(flat)))(*flat)
            self._chk(self.L.aa_mesh_create_local(n, hs, arr, C.byref(h)))
        self._h = h

    def _chk(self, rc: int):
        if rc != 0:
            raise AthenaError(self.L.aa_last_error().decode() or f"athena_amd error {rc}")

    def close(self):
        if self._h is not None:
            self.L.aa_mesh_destroy(self._h); self._h = None
        for g in self.lev:
            g.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def state(self):
        t = C.c_double(); dt = C.c_double(); n = C.c_int()
        self.L.aa_mesh_get_state(self._h, C.byref(t), C.byref(dt), C.byref(n))
        return t.value, dt.value, n.value

    time = property(lambda s: s.state()[0])
    dt = property(lambda s: s.state()[1])
    nstep = property(lambda s: s.state()[2])

    def set_stream(self, stream_ptr: int): self._chk(self.L.aa_mesh_set_stream(self._h, C.c_void_p(stream_ptr)))
    def RestrictCorrect(self): self._chk(self.L.aa_mesh_restrict_correct(self._h))
    def restrict_correct_pair(self, l: int): self._chk(self.L.aa_mesh_restrict_correct_pair(self._h, l))
    def ionradRestrictCorrect(self): self._chk(self.L.aa_mesh_ionrad_restrict_correct(self._h))
    def Prolongate(self): self._chk(self.L.aa_mesh_prolongate(self._h))
    def new_dt(self): self._chk(self.L.aa_mesh_new_dt(self._h))

    def ion_radtransfer_3d(self, level: int) -> int:
        n = C.c_int(); self._chk(self.L.aa_mesh_ion_radtransfer(self._h, level, C.byref(n))); return n.value

    def ionflux_prolong(self, level: int): self._chk(self.L.aa_mesh_ionflux_prolong(self._h, level))

    def start(self):
        self._chk(self.L.aa_mesh_start(self._h)); return self

    def step(self):
        it = (C.c_int * len(self.lev))()
        self._chk(self.L.aa_mesh_step(self._h, it))
        return list(it)
