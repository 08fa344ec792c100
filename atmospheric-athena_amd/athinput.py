"""athinput reader: the run-time parameter tier of the reference (src/par.c).

Mirrors the behaviour the hot path depends on (reference file:line for each rule):
  * ``<block>`` headers, ``name = value # comment`` lines         (par.c:60-120, :560-640)
  * command-line overrides ``block/name=value`` may only change a key that already
    exists in the deck; an unknown block or name is a fatal error  (par.c:194, :700-760)
  * ``par_getd/par_geti/par_gets`` abort on a missing key, the ``_def`` forms insert the
    default into the table                                         (par.c:254-420)

Only the key set of SURVEY.md section 8(b) is interpreted by the rest of the package; all
other blocks (outputs, log) are carried but ignored.
"""
from __future__ import annotations

import re
from collections import OrderedDict


class ParError(RuntimeError):
    """Equivalent of the reference's ``ath_error`` for the parameter tier (utils.c:118)."""


class ParTable:
    def __init__(self):
        self.blocks: "OrderedDict[str, OrderedDict[str, str]]" = OrderedDict()

    # -- construction ---------------------------------------------------------------
    @classmethod
    def from_text(cls, text: str) -> "ParTable":
        t = cls()
        cur = None
        for raw in text.splitlines():
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            m = re.match(r"^<\s*([^>]+?)\s*>", line)
            if m:
                cur = m.group(1)
                if cur.startswith("par_end"):
                    break
                t.blocks.setdefault(cur, OrderedDict())
                continue
            if cur is None:
                raise ParError("[par_open]: Drop a block name here")
            if "=" not in line:
                raise ParError(f"[par_open]: No '=' found in line \"{raw}\"")
            name, rest = line.split("=", 1)
            value = rest.split("#", 1)[0].strip()
            t.blocks[cur][name.strip()] = value
        return t

    @classmethod
    def from_file(cls, path: str) -> "ParTable":
        with open(path, "r") as f:
            return cls.from_text(f.read())

    def cmdline(self, args) -> "ParTable":
        """``block/name=value`` overrides; both block and name must already exist."""
        for a in args or ():
            m = re.match(r"^([^/=]+)/([^=]+)=(.*)$", a)
            if not m:
                raise ParError(f"[par_cmdline]: unrecognised argument \"{a}\"")
            block, name, value = m.group(1).strip(), m.group(2).strip(), m.group(3).strip()
            if block not in self.blocks:
                raise ParError(f"[par_cmdline]: Block \"{block}\" on command line not found")
            if name not in self.blocks[block]:
                raise ParError(f"[par_cmdline]: Name \"{name}\" in block \"{block}\" not found")
            self.blocks[block][name] = value
        return self

    # -- queries --------------------------------------------------------------------
    def exist(self, block: str, name: str) -> bool:
        return block in self.blocks and name in self.blocks[block]

    def _get(self, block: str, name: str) -> str:
        if not self.exist(block, name):
            raise ParError(f"[par_get]: Block \"{block}\", name \"{name}\" not found")
        return self.blocks[block][name]

    def gets(self, block: str, name: str) -> str:
        return self._get(block, name)

    def getd(self, block: str, name: str) -> float:
        return float(self._get(block, name))

    def geti(self, block: str, name: str) -> int:
        # the reference reads ints with atoi(): leading integer part of e.g. "100000000." or "1e3"
        s = self._get(block, name)
        m = re.match(r"^\s*([+-]?\d+)", s)
        return int(m.group(1)) if m else 0

    def getd_def(self, block: str, name: str, default: float) -> float:
        if not self.exist(block, name):
            self.blocks.setdefault(block, OrderedDict())[name] = repr(float(default))
        return self.getd(block, name)

    def geti_def(self, block: str, name: str, default: int) -> int:
        if not self.exist(block, name):
            self.blocks.setdefault(block, OrderedDict())[name] = str(int(default))
        return self.geti(block, name)
