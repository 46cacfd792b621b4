"""align_seq: the reference's single-alignment primitive (alignment.py:21-50), executed on the GPU.

The reference calls edlib here; this module calls `smx_align` / `smx_align_batch` (include/smx.h), the device Myers
bit-vector kernels with the same semantics (all optimal end positions, edlib's start rule, the 28 IUPAC
equalities of constants.py:13-20).  This is the unit-parity / diagnostic primitive, not the batch path --
`process_sequences` never goes through it.  Callers that need many alignments (trace level 3, --color) collect their
requests, run them in ONE launch (`align_batch`) and replay through `align_seq`, which finds them in the active
`AlignCache`.  There is no CPU implementation: without libsmx.so and a GPU the call raises."""
import ctypes as C
from typing import List, Optional, Tuple

from . import _lib
from .constants import AlignMode

_MODE = {AlignMode.INFIX: 0, AlignMode.PREFIX: 1}
_CAP = 512


class AlignmentResult:
    """models.py:34-69 without the edlib dict: distance() == -1 means "no match within max_distance"."""

    __slots__ = ("_distance", "_locations")

    def __init__(self, distance: int, locations: List[Tuple[Optional[int], int]]):
        self._distance = distance
        self._locations = list(locations)

    def matched(self) -> bool:
        return self._distance > -1

    def distance(self) -> int:
        return self._distance

    def location(self):
        return self._locations[0]

    def locations(self):
        return self._locations

    def reversed(self, seq_length: int) -> "AlignmentResult":   # models.py:52-63
        if self._distance == -1:
            return AlignmentResult(self._distance, self._locations)
        return AlignmentResult(self._distance, [(seq_length - b - 1, seq_length - a - 1) for a, b in self._locations])

    def adjust_start(self, s: int):   # models.py:65-69 (in place)
        if self._distance != -1:
            self._locations = [(a + s, b + s) for a, b in self._locations]


_active_cache = None   # AlignCache of the batch being replayed (set by AlignCache.__enter__)


def align_batch(requests):
    """requests: list of (query str, target str (already sliced, non-empty), max_distance, mode) -> list of
    (distance, [(start, end), ...]) in target coordinates.  One kernel launch for all of them."""
    import numpy as np
    if not requests:
        return []
    lib = _lib.load()
    qtab, qidx = {}, np.empty(len(requests), dtype=np.uint32)
    for i, (q, _t, _k, _m) in enumerate(requests):
        if len(q) < 1 or len(q) > 64:
            raise ValueError("query length must be 1..64 for the device aligner")
        qidx[i] = qtab.setdefault(q, len(qtab))
    queries = list(qtab)
    qoff = np.zeros(len(queries) + 1, dtype=np.uint32)
    qoff[1:] = np.cumsum([len(q) for q in queries])
    qblob = "".join(queries).encode("ascii")
    toff = np.zeros(len(requests) + 1, dtype=np.uint64)
    toff[1:] = np.cumsum([len(t) for _q, t, _k, _m in requests], dtype=np.uint64)
    tblob = "".join(t for _q, t, _k, _m in requests).encode("latin-1", "replace")
    k = np.array([int(r[2]) for r in requests], dtype=np.int32)
    mode = np.array([_MODE[r[3]] for r in requests], dtype=np.uint8)
    n = len(requests)
    cap = 16
    while True:
        dist, nloc = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        starts, ends = np.empty((n, cap), dtype=np.int32), np.empty((n, cap), dtype=np.int32)
        _lib.check(lib.smx_align_batch(qblob, _lib.ptr(qoff), len(queries), tblob, _lib.ptr(toff), _lib.ptr(qidx), _lib.ptr(k),
                                       _lib.ptr(mode), n, _lib.ptr(dist), _lib.ptr(nloc), _lib.ptr(starts), _lib.ptr(ends), cap))
        if int(nloc.max()) <= cap:
            break
        cap = int(nloc.max())   # rare: more optimal locations than the first guess
    return [(int(dist[i]), [(int(starts[i, j]), int(ends[i, j])) for j in range(int(nloc[i]))] if dist[i] != -1 else [])
            for i in range(n)]


class AlignCache:
    """Results of one align_batch call, found again by align_seq while the cache is active (`with cache:`)."""

    def __init__(self):
        self.table = {}

    def fill(self, requests):
        todo = [r for r in dict.fromkeys(requests) if r not in self.table]
        for r, res in zip(todo, align_batch(todo)):
            self.table[r] = res

    def __enter__(self):
        global _active_cache
        self._prev, _active_cache = _active_cache, self
        return self

    def __exit__(self, *exc):
        global _active_cache
        _active_cache = self._prev


def align_seq(query, target, max_distance: int, start: int, end: int, mode: str = AlignMode.INFIX) -> AlignmentResult:
    """Same signature and slice semantics as the reference (negative starts wrap like Python slices, -1 means
    "from the beginning" / "to the end")."""
    query = str(getattr(query, "seq", query))
    target_seq = str(getattr(target, "seq", target))
    s = 0 if start == -1 else start
    e = len(target_seq) if end == -1 else min(end, len(target_seq))
    t = target_seq[s:e]
    if mode not in _MODE:
        raise NotImplementedError(f"alignment mode {mode!r} is not on the device path (HW and SHW are)")
    if len(query) < 1 or len(query) > 64:
        raise ValueError("query length must be 1..64 for the device aligner")
    if not t:
        # edlib on an empty target reports editDistance = len(query); align_seq then clamps it
        d = len(query) if len(query) <= max_distance else -1
        return AlignmentResult(d, [(None, -1)] if d != -1 else [])
    if mode == AlignMode.PREFIX:
        t = t[:len(query) + int(max_distance)]   # exact: a prefix alignment within k ends inside the first len + k bases
    if _active_cache is not None:
        hit = _active_cache.table.get((query, t, int(max_distance), mode))
        if hit is not None:
            m = AlignmentResult(hit[0], list(hit[1]))
            m.adjust_start(s)
            return m
    lib = _lib.load()
    dist, nloc = C.c_int(), C.c_int()
    starts, ends = (C.c_int * _CAP)(), (C.c_int * _CAP)()
    _lib.check(lib.smx_align(query.encode("ascii"), len(query), t.encode("latin-1", "replace"), len(t), int(max_distance),
                             _MODE[mode], C.byref(dist), starts, ends, _CAP, C.byref(nloc)))
    if nloc.value > _CAP:
        raise RuntimeError("more optimal locations than the binding's buffer holds")
    m = AlignmentResult(dist.value, [(starts[i], ends[i]) for i in range(nloc.value)] if dist.value != -1 else [])
    m.adjust_start(s)
    return m


def color_sequence(seq: str, quality_scores, p1_location, p2_location, b1_location, b2_location) -> str:
    """`--color` (alignment.py:59-104): barcodes blue, primers green, bases below Q10 in lower case."""
    blue, green, reset = "\033[0;34m", "\033[0;32m", "\033[0m"
    n = len(seq)
    out = [""] * n

    def paint(location, color):
        if location is None:
            return
        a, e = location
        if a < 0 or e < 0:
            return
        for i in range(a, e + 1):
            if i < n:
                out[i] = color + (seq[i].lower() if quality_scores[i] < 10 else seq[i]) + reset

    paint(b1_location, blue)
    paint(p1_location, green)
    paint(p2_location, green)
    paint(b2_location, blue)
    for i in range(n):
        if out[i] == "":
            out[i] = seq[i].lower() if quality_scores[i] < 10 else seq[i]
    return "".join(out)
