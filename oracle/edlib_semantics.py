"""oracle/edlib_semantics.py -- TEST INFRASTRUCTURE ONLY (parity oracle, never shipped).

CPU restatement of the one third-party primitive on the reference's hot path:
``edlib.align(q, t, mode, 'locations', k, additionalEqualities=IUPAC_EQUIV)``
(call site /root/reference/src/specimux/alignment.py:42; threshold call site
src/specimux/orchestration.py:552).  edlib (Martinsos/edlib, ``edlib>=1.1.2`` in
pyproject.toml:28) is not vendored in the reference and not installed here, so this
module restates its published behaviour twice:

* ``align_py``  - pure-Python O(m*n) DP (small cases, obviously correct);
* ``align_c``   - the same DP in plain C (oracle/align_oracle.c) through ctypes,
                  fast enough for thousands of reads.

Both return the *raw* edlib-style dict ``{'editDistance': d, 'locations': [(s, e), ...]}``;
the wrapper that clamps and shifts (alignment.py:37-50) lives in specimux_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

# /root/reference/src/specimux/constants.py:13-20 (data, 28 symmetric pairs)
IUPAC_PAIRS = ("YC YT RA RG NA NC NG NT WA WT MA MC SC SG KG KT "
               "BC BG BT DA DG DT HA HC HT VA VC VG").split()
_EQ = set()
for _p in IUPAC_PAIRS:
    _EQ.add((_p[0], _p[1]))
    _EQ.add((_p[1], _p[0]))

HW, SHW, NW = "HW", "SHW", "NW"
_MODE_ID = {HW: 0, SHW: 1, NW: 2}


def eq(a, b, iupac=True):
    """A.1: equal chars, or one of the 28 (symmetric, NOT transitive) IUPAC pairs."""
    return a == b or (iupac and (a, b) in _EQ)


def _last_row(q, t, mode, iupac):
    m, n = len(q), len(t)
    prev = list(range(m + 1))
    last = [prev[m]]
    for j in range(1, n + 1):
        cur = [0 if mode == HW else j] + [0] * m
        tj = t[j - 1]
        for i in range(1, m + 1):
            sub = prev[i - 1] + (0 if eq(q[i - 1], tj, iupac) else 1)
            cur[i] = min(sub, prev[i] + 1, cur[i - 1] + 1)
        last.append(cur[m])
        prev = cur
    return last


def align_py(q, t, mode=HW, k=-1, iupac=True):
    m, n = len(q), len(t)
    if m == 0 or n == 0:  # A.4
        if mode == NW:
            return {"editDistance": max(m, n), "locations": [(0, n - 1)]}
        return {"editDistance": m, "locations": [(None, -1)]}
    last = _last_row(q, t, mode, iupac)
    best = last[n] if mode == NW else min(last[1:])
    if k >= 0 and best > k:
        return {"editDistance": -1, "locations": []}
    locs = []
    for j in (range(n, n + 1) if mode == NW else range(1, n + 1)):
        if last[j] != best:
            continue
        e, s = j - 1, 0
        if mode == HW:  # A.3: smallest start with NW(q, t[s..e]) == best
            rl = _last_row(q[::-1], t[e::-1], SHW, iupac)
            p = max(c - 1 for c in range(1, e + 2) if rl[c] == best)
            s = e - p
        locs.append((s, e))
    return {"editDistance": best, "locations": locs}


# ----------------------------------------------------------------------------- C twin
_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_SO = os.path.join(_BUILD, "liboracle_align.so")
_lib = None
_NONE = -2147483648


def build_c(force=False):
    """gcc -O2 -shared oracle/align_oracle.c -> oracle/_build/liboracle_align.so"""
    src = os.path.join(_HERE, "align_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(_BUILD, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src])
    return _SO


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build_c())
        lib.oracle_align.restype = ctypes.c_int
        lib.oracle_align.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int)]
        _lib = lib
    return _lib


def align_c(q, t, mode=HW, k=-1, iupac=True):
    lib = _load()
    qb, tb = q.encode("latin-1"), t.encode("latin-1")
    cap = max(len(tb), 1)
    starts = (ctypes.c_int * cap)()
    ends = (ctypes.c_int * cap)()
    dist = ctypes.c_int()
    nloc = ctypes.c_int()
    lib.oracle_align(qb, len(qb), tb, len(tb), k, _MODE_ID[mode], 1 if iupac else 0,
                     ctypes.byref(dist), starts, ends, cap, ctypes.byref(nloc))
    locs = [(None if starts[i] == _NONE else starts[i], ends[i]) for i in range(nloc.value)]
    return {"editDistance": dist.value, "locations": locs}


def align(q, t, mode=HW, k=-1, iupac=True):
    """Default entry: the C twin (falls back to nothing -- gcc is part of the image)."""
    return align_c(q, t, mode, k, iupac)
