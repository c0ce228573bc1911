"""Text <-> integer PRG helpers used by the oracle-side tests (TEST INFRASTRUCTURE).

Restates, for test construction only:
  * ``encode_prg``           libgramtools/src/prg/linearised_prg.cpp:241-265
  * ``prg_string_to_ints``   libgramtools/src/prg/linearised_prg.cpp:166-213
  * ``encode_dna_bases``     libgramtools/src/common/utils.cpp:73-92
"""
import numpy as np

_BASE = {"A": 1, "C": 2, "G": 3, "T": 4, "a": 1, "c": 2, "g": 3, "t": 4}


def encode_prg(prg_raw: str):
    """Legacy numbered PRG text ("gct5c6g6t6ag") -> list of ints. Non-nested only."""
    out, digits = [], []

    def flush():
        if digits:
            m = 0
            for d in digits:
                m = m * 10 + d
            out.append(m)
            digits.clear()

    for c in prg_raw:
        if c in _BASE:
            flush()
            out.append(_BASE[c])
        else:
            digits.append(ord(c) - ord("0"))
    flush()
    return out


def prg_string_to_ints(s: str):
    """Bracketed PRG text ("a[c,g[ct,t]a]c") -> list of ints; sites numbered by '[' order."""
    out, stack, max_marker = [], [], 3
    for c in s:
        if c == "[":
            max_marker += 2
            stack.append(max_marker)
            out.append(max_marker)
        elif c == "]":
            out.append(stack.pop() + 1)
        elif c == ",":
            out.append(stack[-1] + 1)
        else:
            out.append(_BASE[c])
    return out


def encode_dna_bases(s: str):
    """ACGT text -> uint8 array of 1..4; any other character yields an empty read."""
    out = []
    for c in s:
        b = _BASE.get(c, 0)
        if b == 0:
            return np.zeros(0, dtype=np.uint8)
        out.append(b)
    return np.asarray(out, dtype=np.uint8)


def ints_to_prg_bytes(ints) -> bytes:
    """The ``gram_dir/prg`` on-disk form: little-endian uint32 per symbol
    (libgramtools/src/prg/linearised_prg.cpp:8-45,82-115)."""
    return np.asarray(ints, dtype="<u4").tobytes()


def ints_to_prg_string(ints) -> str:
    """Integer PRG -> bracketed text (libgramtools/src/prg/linearised_prg.cpp:133-164): odd markers open a site, even ones
    separate alleles, the last even marker of a site closes it."""
    out, last = [], {}
    for pos, s in enumerate(ints):
        s = int(s)
        if s > 4:
            if s % 2 == 1:
                out.append("[")
            else:
                out.append(",")
                last[s] = pos
        else:
            out.append("ACGT"[s - 1])
    for pos in last.values():
        out[pos] = "]"
    return "".join(out)
