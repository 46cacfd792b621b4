"""Enumerations and constant tables of the drop-in interface.

Same names and values as the reference's src/specimux/constants.py (AlignMode :26, SampleId :33,
TrimMode :40, MultipleMatchStrategy :48, ResolutionType :54, Barcode :91, Primer :106,
Orientation :121, IUPAC_EQUIV :13) so user code written against specimux keeps working."""
from enum import Enum

# the 28 symmetric, non-transitive equalities every hot-path alignment uses (constants.py:13-20)
IUPAC_EQUIV = [(a, b) for a, bs in (("Y", "CT"), ("R", "AG"), ("N", "ACGT"), ("W", "AT"), ("M", "AC"),
                                    ("S", "CG"), ("K", "GT"), ("B", "CGT"), ("D", "AGT"), ("H", "ACT"),
                                    ("V", "ACG")) for b in bs]
IUPAC_CODES = {a for a, _ in IUPAC_EQUIV} | set("ACGT")


class AlignMode:
    GLOBAL, INFIX, PREFIX = "NW", "HW", "SHW"


class SampleId:
    UNKNOWN = "unknown"
    PREFIX_FWD_MATCH = "barcode_fwd_"
    PREFIX_REV_MATCH = "barcode_rev_"


class TrimMode:
    PRIMERS, BARCODES, TAILS, NONE = "primers", "barcodes", "tails", "none"


class MultipleMatchStrategy:
    NONE, BEST = "none", "best"


class ResolutionType(Enum):
    FULL_MATCH = 1
    PARTIAL_FORWARD = 2
    PARTIAL_REVERSE = 3
    MULTIPLE_SPECIMENS = 4
    UNKNOWN = 5
    DEREPLICATED_FULL = 6

    def to_string(self):
        return {1: "full_match", 2: "partial_forward", 3: "partial_reverse", 4: "multiple_specimens",
                6: "dereplicated_full"}.get(self.value, "unknown")

    def is_full_match(self):
        return self in (ResolutionType.FULL_MATCH, ResolutionType.DEREPLICATED_FULL)

    def is_partial_match(self):
        return self in (ResolutionType.PARTIAL_FORWARD, ResolutionType.PARTIAL_REVERSE)

    def is_unknown(self):
        return self is ResolutionType.UNKNOWN


class Barcode(Enum):
    B1 = 1
    B2 = 2

    def to_string(self):
        return "forward" if self is Barcode.B1 else "reverse"


class Primer(Enum):
    FWD = 3
    REV = 4

    def to_string(self):
        return "forward" if self is Primer.FWD else "reverse"


class Orientation(Enum):
    FORWARD = 1
    REVERSE = 2
    UNKNOWN = 3

    def to_string(self):
        return self.name.lower()
