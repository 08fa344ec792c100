"""Reader / writer of the reference's restart dumps (`out_fmt = rst`, src/restart.c), the only
full-precision output format of the reference (bin and vtk are single precision).

Layout (restart.c:463-983, read back by restart_grids :52-456): the parameter dump as text,
terminated by a line ``<par_end>``; then ``N_STEP\\n`` + int32, ``\\nTIME\\n`` + double,
``\\nTIME_STEP\\n`` + double; then for every Grid the labelled blocks ``\\nDENSITY\\n``,
``\\n1-MOMENTUM\\n``, ``\\n2-MOMENTUM\\n``, ``\\n3-MOMENTUM\\n``, ``\\nENERGY\\n`` (Nx1*Nx2*Nx3 doubles
each over ACTIVE zones, [k][j][i]), with ion radiation ``\\nEDGEFLUX\\n`` ((Nx1+1)(Nx2+1)(Nx3+1)
doubles), then ``\\nSCALAR n\\n`` per passive scalar; finally ``\\nUSER_DATA\\n`` followed by whatever
the problem file writes (nothing for ifront / ioniz_sphere / blast).  With static mesh refinement
the Grid blocks of all Domains follow each other, root first (the loop over levels of
restart.c:531-770), under the one header: `write_rst_levels` / `read_rst_levels`.

With these two functions a run of this package can be continued by the reference
(``athena -r file.rst``) and vice versa, and the parity tests can start from a developed
reference state instead of the initial condition.
"""
from __future__ import annotations

import struct
from typing import Dict, Optional, Sequence

import numpy as np

from .athinput import ParTable

_LABELS = ("DENSITY", "1-MOMENTUM", "2-MOMENTUM", "3-MOMENTUM", "ENERGY")


def par_dump(par: ParTable) -> str:
    """par_dump(2, fp) of the reference (par.c:370-410): blocks, aligned `name = value` lines."""
    out = []
    for block, items in par.blocks.items():
        out.append(f"<{block}>")
        width = max((len(k) for k in items), default=0)
        for k, v in items.items():
            out.append(f"{k:<{width}} = {v}")
        out.append("")
    out.append("<par_end>")
    return "\n".join(out) + "\n"


def write_rst(path: str, par_text: str, nstep: int, time: float, dt: float, U: np.ndarray,
              edgeflux: Optional[np.ndarray] = None) -> None:
    """U: active zones [Nx3][Nx2][Nx1][nvar] (nvar = 5 or 6)."""
    if not par_text.rstrip().endswith("<par_end>"):
        par_text = par_text.rstrip("\n") + "\n<par_end>\n"
    nvar = U.shape[-1]
    with open(path, "wb") as f:
        f.write(par_text.encode())
        f.write(b"N_STEP\n" + struct.pack("<i", int(nstep)))
        f.write(b"\nTIME\n" + struct.pack("<d", float(time)))
        f.write(b"\nTIME_STEP\n" + struct.pack("<d", float(dt)))
        for c, lab in enumerate(_LABELS):
            f.write(b"\n" + lab.encode() + b"\n")
            f.write(np.ascontiguousarray(U[..., c], dtype="<f8").tobytes())
        if edgeflux is not None:
            f.write(b"\nEDGEFLUX\n")
            f.write(np.ascontiguousarray(edgeflux, dtype="<f8").tobytes())
        for n in range(nvar - 5):
            f.write(f"\nSCALAR {n}\n".encode())
            f.write(np.ascontiguousarray(U[..., 5 + n], dtype="<f8").tobytes())
        f.write(b"\nUSER_DATA\n")


def read_rst(path: str, nx: Sequence[int], nscal: int, ion: bool) -> Dict:
    b = open(path, "rb").read()
    end = b.index(b"<par_end>")
    end = b.index(b"\n", end) + 1
    header = b[:end].decode(errors="replace")
    pos = end
    if b[pos:pos + 7] != b"N_STEP\n":
        raise ValueError("[restart_grids]: Expected N_STEP")
    pos += 7
    nstep = struct.unpack_from("<i", b, pos)[0]; pos += 4

    def expect(label: bytes):
        nonlocal pos
        tag = b"\n" + label + b"\n"
        if b[pos:pos + len(tag)] != tag:
            raise ValueError(f"[restart_grids]: Expected {label.decode()}, found {b[pos:pos + 24]!r}")
        pos += len(tag)

    expect(b"TIME"); time = struct.unpack_from("<d", b, pos)[0]; pos += 8
    expect(b"TIME_STEP"); dt = struct.unpack_from("<d", b, pos)[0]; pos += 8
    n = int(nx[0]) * int(nx[1]) * int(nx[2])
    U = np.zeros((nx[2], nx[1], nx[0], 5 + nscal))
    for c, lab in enumerate(_LABELS):
        expect(lab.encode())
        U[..., c] = np.frombuffer(b, dtype="<f8", count=n, offset=pos).reshape(nx[2], nx[1], nx[0]); pos += 8 * n
    ef = None
    if ion:
        expect(b"EDGEFLUX")
        ne = (nx[0] + 1) * (nx[1] + 1) * (nx[2] + 1)
        ef = np.frombuffer(b, dtype="<f8", count=ne, offset=pos).reshape(nx[2] + 1, nx[1] + 1, nx[0] + 1).copy(); pos += 8 * ne
    for s in range(nscal):
        expect(f"SCALAR {s}".encode())
        U[..., 5 + s] = np.frombuffer(b, dtype="<f8", count=n, offset=pos).reshape(nx[2], nx[1], nx[0]); pos += 8 * n
    expect(b"USER_DATA")
    return dict(header=header, par=ParTable.from_text(header), nstep=nstep, time=time, dt=dt, U=U, edgeflux=ef)


def write_rst_levels(path: str, par_text: str, nstep: int, time: float, dt: float,
                     levels: Sequence[Sequence[Optional[np.ndarray]]]) -> None:
    """levels: [(U_active, edgeflux or None), ...] root first."""
    if not par_text.rstrip().endswith("<par_end>"):
        par_text = par_text.rstrip("\n") + "\n<par_end>\n"
    with open(path, "wb") as f:
        f.write(par_text.encode())
        f.write(b"N_STEP\n" + struct.pack("<i", int(nstep)))
        f.write(b"\nTIME\n" + struct.pack("<d", float(time)))
        f.write(b"\nTIME_STEP\n" + struct.pack("<d", float(dt)))
        for U, edgeflux in levels:
            for c, lab in enumerate(_LABELS):
                f.write(b"\n" + lab.encode() + b"\n")
                f.write(np.ascontiguousarray(U[..., c], dtype="<f8").tobytes())
            if edgeflux is not None:
                f.write(b"\nEDGEFLUX\n")
                f.write(np.ascontiguousarray(edgeflux, dtype="<f8").tobytes())
            for n in range(U.shape[-1] - 5):
                f.write(f"\nSCALAR {n}\n".encode())
                f.write(np.ascontiguousarray(U[..., 5 + n], dtype="<f8").tobytes())
        f.write(b"\nUSER_DATA\n")


def read_rst_levels(path: str, nxs: Sequence[Sequence[int]], nscal: int, ion: bool) -> Dict:
    """nxs: active zones (Nx1, Nx2, Nx3) of every level, root first."""
    b = open(path, "rb").read()
    end = b.index(b"<par_end>")
    end = b.index(b"\n", end) + 1
    header = b[:end].decode(errors="replace")
    pos = end

    def expect(label: bytes):
        nonlocal pos
        tag = (b"" if label == b"N_STEP" else b"\n") + label + b"\n"
        if b[pos:pos + len(tag)] != tag:
            raise ValueError(f"[restart_grids]: Expected {label.decode()}, found {b[pos:pos + 24]!r}")
        pos += len(tag)

    expect(b"N_STEP"); nstep = struct.unpack_from("<i", b, pos)[0]; pos += 4
    expect(b"TIME"); time = struct.unpack_from("<d", b, pos)[0]; pos += 8
    expect(b"TIME_STEP"); dt = struct.unpack_from("<d", b, pos)[0]; pos += 8
    levels = []
    for nx in nxs:
        n = int(nx[0]) * int(nx[1]) * int(nx[2])
        U = np.zeros((nx[2], nx[1], nx[0], 5 + nscal)); ef = None
        for c, lab in enumerate(_LABELS):
            expect(lab.encode())
            U[..., c] = np.frombuffer(b, dtype="<f8", count=n, offset=pos).reshape(nx[2], nx[1], nx[0]); pos += 8 * n
        if ion:
            expect(b"EDGEFLUX")
            ne = (nx[0] + 1) * (nx[1] + 1) * (nx[2] + 1)
            ef = np.frombuffer(b, dtype="<f8", count=ne, offset=pos).reshape(nx[2] + 1, nx[1] + 1, nx[0] + 1).copy(); pos += 8 * ne
        for sc in range(nscal):
            expect(f"SCALAR {sc}".encode())
            U[..., 5 + sc] = np.frombuffer(b, dtype="<f8", count=n, offset=pos).reshape(nx[2], nx[1], nx[0]); pos += 8 * n
        levels.append((U, ef))
    expect(b"USER_DATA")
    return dict(header=header, par=ParTable.from_text(header), nstep=nstep, time=time, dt=dt, levels=levels)
