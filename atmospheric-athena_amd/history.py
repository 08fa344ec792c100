"""History dumps (.hst) in the reference's format (dump_history.c): one formatted row of volume
averages per dump, column headers on the first one.

Columns for HYDRO / ADIABATIC: time, dt, mass, total E, x1/x2/x3 Mom., x1/x2/x3-KE, then one
column per passive scalar (dump_history.c:361-400).  Every quantity is the sum over active zones of
``dVol*q`` (:157-200), summed over the Grids of the Domain (MPI_Reduce :257) and divided by the
Domain volume (:271-279).  Levels > 0 write to ``lev<N>/<basename>-lev<N>.hst`` (:300-330).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np

LABELS = ["time   ", "dt      ", "mass    ", "total E ", "x1 Mom. ", "x2 Mom. ", "x3 Mom. ",
          "x1-KE   ", "x2-KE   ", "x3-KE   "]


def sums_from_block(U: np.ndarray, dx: Sequence[float], nscal: int) -> np.ndarray:
    """The 9 volume integrals from a host block of ACTIVE zones [k][j][i][var] (same order as
    aa_history): used for hosts that already hold the state (restart files, the CPU checker)."""
    dVol = dx[0] * dx[1] * dx[2]
    d, M1, M2, M3, E = (U[..., c] for c in range(5))
    d1 = 1.0 / d
    out = [d.sum(), E.sum(), M1.sum(), M2.sum(), M3.sum(),
           (0.5 * M1 * M1 * d1).sum(), (0.5 * M2 * M2 * d1).sum(), (0.5 * M3 * M3 * d1).sum(),
           U[..., 5].sum() if nscal else 0.0]
    return dVol * np.array(out)


def header(level: int, domain: int, volume: float, nscal: int) -> str:
    s = "# Athena history dump for level=%i domain=%i volume=%e\n" % (level, domain, volume)
    s += "#   [1]=" + LABELS[0]
    for n, lab in enumerate(LABELS[1:], start=2):
        s += "   [%i]=%s" % (n, lab)
    for n in range(nscal):
        s += "  [%i]=scalar %i" % (len(LABELS) + 1 + n, n)
    return s + "\n#\n"


def format_row(values: Sequence[float], dat_fmt: Optional[str] = None) -> str:
    fmt = " %14.6e" if dat_fmt is None else " " + dat_fmt          # dump_history.c:143-148
    return "".join(fmt % v for v in values) + "\n"


class HistoryWriter:
    """dump_history(pM, pOut) for one Domain; appends like the reference (fopen "a", :353)."""

    def __init__(self, rundir: str, basename: str, level: int = 0, domain: int = 0,
                 dat_fmt: Optional[str] = None):
        d = os.path.join(rundir, f"lev{level}") if level > 0 else rundir
        os.makedirs(d, exist_ok=True)
        name = basename + (f"-lev{level}" if level > 0 else "") + (f"-dom{domain}" if domain > 0 else "")
        self.path = os.path.join(d, name + ".hst")
        self.level, self.domain, self.dat_fmt, self.num = level, domain, dat_fmt, 0

    def dump(self, time: float, dt: float, sums: np.ndarray, volume: float, nscal: int):
        """sums: aa_history() of the Domain (already added over its Grids)."""
        row = [time, dt] + [float(v) / volume for v in sums[:8 + nscal]]
        with open(self.path, "a") as f:
            if self.num == 0:
                f.write(header(self.level, self.domain, volume, nscal))
            f.write(format_row(row, self.dat_fmt))
        self.num += 1
