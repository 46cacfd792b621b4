"""align_seq: the reference's single-alignment primitive (alignment.py:21-50), executed on the GPU.

The reference calls edlib here; this module calls `smx_align` (include/smx.h), the device Myers
bit-vector kernel with the same semantics (all optimal end positions, edlib's start rule, the 28 IUPAC
equalities of constants.py:13-20).  One kernel launch per call: this is the unit-parity / diagnostic
primitive (trace level 3 uses it), not the batch path -- `process_sequences` never goes through it.
There is no CPU implementation: without libsmx.so and a GPU the call raises."""
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
