"""Data model of the drop-in interface (reference: src/specimux/models.py).

Only what crosses the process_sequences boundary is kept as Python objects: PrimerInfo (:20),
MatchParameters (:331), WriteOperation (:341), SequenceBatch (:360), WorkerException (:368).
AlignmentResult / CandidateMatch (:34-328) have no Python counterpart here: that state lives in
LDS inside the demux kernel (specimux_amd/csrc/smx_kernels.hip)."""
from typing import Dict, List, NamedTuple, Optional, Tuple

from .constants import Primer, ResolutionType

_COMPLEMENT = bytes.maketrans(b"ACGTMRWSYKVHDBXNUacgtmrwsykvhdbxnu", b"TGCAKYWSRMBDHVXNAtgcakywsrmbdhvxna")


def reverse_complement(seq: str) -> str:
    """Bio.Seq.reverse_complement for DNA: ambiguity codes, case kept, U -> A, others unchanged."""
    return seq.encode("latin-1").translate(_COMPLEMENT)[::-1].decode("latin-1")


class PrimerInfo:
    """One primer of primers.fasta.  `barcodes` is an insertion-ordered list (the reference uses a
    set whose iteration order depends on PYTHONHASHSEED; first appearance in specimens.txt is the
    canonical order here -- SURVEY Q4)."""

    def __init__(self, name: str, seq: str, direction: Primer, pools: List[str], file_index: int = 0):
        self.name = name
        self.primer = seq.upper()
        self.primer_rc = reverse_complement(self.primer)
        self.direction = direction
        self.pools = pools
        self.file_index = file_index
        self.barcodes: List[str] = []
        self.specimens = set()

    def add_barcode(self, barcode: str):
        if barcode not in self.barcodes:
            self.barcodes.append(barcode)

    def __repr__(self):
        return f"PrimerInfo({self.name!r}, {self.direction.to_string()})"


class MatchParameters:
    def __init__(self, max_dist_primers: Dict[str, int], max_dist_index: int, search_len: int, preorient: bool):
        self.max_dist_primers = max_dist_primers   # keyed by primer SEQUENCE, like the reference
        self.max_dist_index = max_dist_index
        self.search_len = search_len
        self.preorient = preorient


class WriteOperation(NamedTuple):
    sample_id: str
    seq_id: str
    distance_code: str
    sequence: str
    quality_sequence: str
    quality_scores: List[int]
    p1_location: Optional[Tuple[int, int]]
    p2_location: Optional[Tuple[int, int]]
    b1_location: Optional[Tuple[int, int]]
    b2_location: Optional[Tuple[int, int]]
    primer_pool: str
    p1_name: str
    p2_name: str
    resolution_type: ResolutionType
    trace_sequence_id: Optional[str] = None


class SequenceBatch(NamedTuple):
    seq_number: int
    seq_records: list
    parameters: MatchParameters
    start_idx: int


class WorkerException(Exception):
    pass
