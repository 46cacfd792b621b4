"""Barcode prefilter of the drop-in interface (reference: src/specimux/bloom_filter.py).

The reference inserts `barcode + variant[:L-k]` for every string within k edits of each barcode into an
mmap Bloom filter (5 % false positives) and probes `barcode + target[:L-k]` before calling edlib.  On
the GPU the same rule is evaluated exactly, in-kernel and without a hash table:
on ACGT targets the filter is transparent (it never rejects something the aligner would accept), and a
non-ACGT character inside the first L-k target bases rejects.  This class is the on/off switch that
process_sequences recognises, plus a host-side `match` with the exact-set rule for API compatibility."""
from typing import List

from .constants import Primer
from .models import reverse_complement


class BloomPrefilter:
    smx_exact_set = True   # recognised by specimux_amd.demultiplex._prefilter_enabled

    def __init__(self, barcodes: List[str], max_distance: int, error_rate: float = 0.05, filename=None):
        if not barcodes:
            raise ValueError("Must provide at least one barcode")
        self.barcodes = list(dict.fromkeys(barcodes))
        self.barcode_length = len(self.barcodes[0])
        self.max_distance = max_distance
        self.min_length = self.barcode_length - max_distance

    @classmethod
    def create_filter(cls, barcode_rcs, max_distance, error_rate=0.05):
        return None   # nothing to cache: no filter file exists on this path

    @classmethod
    def load_readonly(cls, filename, barcodes, max_distance):
        return cls(barcodes, max_distance)

    def match(self, barcode: str, sequence: str) -> bool:
        """Exact membership: is sequence[:L-k] the [:L-k] truncation of a string within k edits (ACGT)?"""
        if barcode not in self.barcodes:
            return True
        x = sequence[:self.min_length]
        if len(x) < self.min_length or any(c not in "ACGT" for c in x):
            return False
        # min_j edit(x, barcode[:j]) <= k  (the unaligned barcode tail completes the variant for free)
        prev = list(range(len(barcode) + 1))
        for i, cx in enumerate(x, start=1):
            cur = [i] + [0] * len(barcode)
            for j, cb in enumerate(barcode, start=1):
                cur[j] = min(prev[j - 1] + (cx != cb), prev[j] + 1, cur[j - 1] + 1)
            prev = cur
        return min(prev) <= self.max_distance

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def barcodes_for_bloom_prefilter(specimens) -> List[str]:
    fwd, rev = [], []
    for primer in specimens.get_primers(Primer.FWD):
        fwd.extend(b for b in primer.barcodes if b not in fwd)
    for primer in specimens.get_primers(Primer.REV):
        rev.extend(b for b in primer.barcodes if b not in rev)
    return [reverse_complement(b) for b in fwd + rev]
