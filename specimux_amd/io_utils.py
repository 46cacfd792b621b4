"""File formats and the output tree of the drop-in interface (reference: src/specimux/io_utils.py).

Inputs:  primers.fasta (`>name pool=a,b position=forward|reverse`, :270-322), specimens.txt (TSV with
SampleID PrimerPool FwIndex FwPrimer RvIndex RvPrimer, :324-377), reads as FASTQ/FASTA, plain or gzip
(:380-450).  Output: `{full|partial|unknown}/{pool}/{p1}-{p2}/{prefix}{sample}.{fastq|fasta}` with the header
`{id} {p1d,b1d,b2d,p2d} pool={pool} primers={p1}+{p2} {sample}`, full matches duplicated at pool level
(:197-268).  No Biopython / cachetools: the parsers and the writer are this package's own."""
import csv
import gzip
import logging
import os
import shutil
import sys
from collections import defaultdict
from typing import Dict, Iterator, List, Optional

from .constants import Primer, ResolutionType, SampleId
from .databases import PrimerDatabase, Specimens
from .models import PrimerInfo, WriteOperation, reverse_complement


class SeqRecord:
    """Minimal stand-in for Bio.SeqRecord: what process_sequences reads (id, description, seq,
    letter_annotations['phred_quality'])."""
    __slots__ = ("id", "description", "seq", "_qual", "_phred")

    def __init__(self, seq: str, id: str = "", description: str = "", quality: Optional[str] = None):
        self.seq = seq
        self.id = id
        self.description = description
        self._qual = quality      # Phred+33 string, None for FASTA
        self._phred = None

    def __len__(self):
        return len(self.seq)

    @property
    def quality_string(self) -> Optional[str]:
        return self._qual

    @property
    def letter_annotations(self) -> Dict[str, List[int]]:
        if self._qual is None:
            return {}
        if self._phred is None:
            self._phred = [ord(c) - 33 for c in self._qual]
        return {"phred_quality": self._phred}

    def reverse_complement(self) -> "SeqRecord":
        return SeqRecord(reverse_complement(self.seq), self.id, self.description,
                         None if self._qual is None else self._qual[::-1])


def _open_text(filename: str):
    return gzip.open(filename, "rt") if filename.endswith((".gz", ".gzip")) else open(filename, "rt")


def parse_fasta(handle) -> Iterator[SeqRecord]:
    title, parts = None, []
    for line in handle:
        if line.startswith(">"):
            if title is not None:
                yield _fasta_record(title, parts)
            title, parts = line[1:].rstrip("\r\n"), []
        elif title is not None:
            parts.append(line.strip())
    if title is not None:
        yield _fasta_record(title, parts)


def _fasta_record(title: str, parts: List[str]) -> SeqRecord:
    words = title.split(None, 1)
    return SeqRecord("".join(parts).replace(" ", ""), words[0] if words else "", title)


def parse_fastq(handle) -> Iterator[SeqRecord]:
    """FASTQ with the tolerance of Biopython's FastqGeneralIterator: wrapped sequence/quality lines,
    '@' allowed as first quality character; id = first whitespace-delimited word of the title."""
    line = handle.readline()
    while line:
        if not line.strip():
            line = handle.readline()
            continue
        if line[0] != "@":
            raise ValueError("Records in Fastq files should start with '@' character")
        title = line[1:].rstrip("\r\n")
        seq_parts = []
        line = handle.readline()
        while line and line[0] != "+":
            seq_parts.append(line.strip())
            line = handle.readline()
        if not line:
            raise ValueError("End of file without quality information.")
        seq = "".join(seq_parts)
        qual = handle.readline().strip()
        line = handle.readline()
        while line and not (line[0] == "@" and len(qual) >= len(seq)):
            qual += line.strip()
            line = handle.readline()
        if len(qual) != len(seq):
            raise ValueError(f"Lengths of sequence and quality values differs for {title} ({len(seq)} and {len(qual)}).")
        words = title.split(None, 1)
        yield SeqRecord(seq, words[0] if words else "", title, qual)


def detect_file_format(filename: str) -> str:
    base = os.path.basename(filename)
    root, ext = os.path.splitext(base)
    while ext.lower() in (".gz", ".gzip", ".bz2", ".zip"):
        base = root
        root, ext = os.path.splitext(base)
    low = base.lower()
    if low.endswith((".fastq", ".fq")):
        return "fastq"
    if low.endswith((".fasta", ".fa", ".fna")):
        return "fasta"
    try:
        with _open_text(filename) as fh:
            first = fh.read(1)
        if first == "@":
            return "fastq"
        if first == ">":
            return "fasta"
    except Exception:
        pass
    return "fasta"


def open_sequence_file(filename: str, args) -> Iterator[SeqRecord]:
    """Iterator over the reads; sets args.isfastq like the reference (io_utils.py:429-450)."""
    fmt = detect_file_format(filename)
    args.isfastq = fmt == "fastq"

    def gen():
        with _open_text(filename) as fh:
            yield from (parse_fastq(fh) if fmt == "fastq" else parse_fasta(fh))
    return gen()


def native_sequence_records(filename: str, args, batch_reads: int = 65536) -> Iterator[SeqRecord]:
    """The same iterator as open_sequence_file, parsed by the native streaming reader of libsmx.so (multi-threaded FASTQ
    engine, gzip, FASTA): the record path of `-F -d` (trace diagnostics) reads its input through it."""
    from .native_io import Reader
    reader = Reader(filename)
    args.isfastq = reader.is_fastq

    def gen():
        try:
            while True:
                b = reader.next_batch(batch_reads)
                if b is None:
                    return
                for i in range(len(b)):
                    rid, seq, qual = b.record(i)
                    yield SeqRecord(seq, rid, rid, qual)
                b.close()
        finally:
            reader.close()
    return gen()


def read_primers_file(filename: str) -> PrimerDatabase:
    registry = PrimerDatabase()
    with _open_text(filename) as fh:
        for index, rec in enumerate(parse_fasta(fh)):
            pools, position = [], None
            for word in rec.description.split():
                if word.startswith("pool="):
                    pools = [p.strip() for p in word[len("pool="):].replace(";", ",").split(",")]
                elif word.startswith("position="):
                    position = word[len("position="):]
            if not pools:
                raise ValueError(f"Missing pool specification for primer {rec.id}")
            if not position:
                raise ValueError(f"Missing position specification for primer {rec.id}")
            if position not in ("forward", "reverse"):
                raise ValueError(f"Invalid primer position '{position}' for {rec.id}")
            direction = Primer.FWD if position == "forward" else Primer.REV
            registry.add_primer(PrimerInfo(rec.id, rec.seq, direction, pools, file_index=index), pools)
    registry.validate_pools()
    stats = registry.get_pool_stats()
    logging.info(f"Loaded {stats['total_primers']} primers in {stats['total_pools']} pools")
    for pool, st in stats["pools"].items():
        logging.info(f"Pool {pool}: {st['forward_primers']} forward, {st['reverse_primers']} reverse primers")
    return registry


_SPECIMEN_COLUMNS = ("SampleID", "PrimerPool", "FwIndex", "FwPrimer", "RvIndex", "RvPrimer")


def read_specimen_file(filename: str, primer_registry: PrimerDatabase) -> Specimens:
    specimens = Specimens(primer_registry)
    with open(filename, "r", newline="") as fh:
        reader = csv.DictReader(fh, delimiter="\t")
        missing = set(_SPECIMEN_COLUMNS) - set(reader.fieldnames or [])
        if missing:
            raise ValueError(f"Missing required columns in specimen file: {missing}")
        empty = []
        for row_num, row in enumerate(reader, start=1):
            b1, b2 = row["FwIndex"].upper(), row["RvIndex"].upper()
            if not b1.strip() or not b2.strip():
                empty.append(f"Row {row_num} ({row['SampleID']}): "
                             f"{'FwIndex is empty' if not b1.strip() else 'RvIndex is empty'}")
                continue
            try:
                specimens.add_specimen(row["SampleID"], row["PrimerPool"], b1, row["FwPrimer"], b2, row["RvPrimer"])
            except (KeyError, ValueError) as e:
                raise ValueError(f"Error processing row {row_num}: {e}")
    if empty:
        more = f"\n... and {len(empty) - 10} more" if len(empty) > 10 else ""
        raise ValueError(f"Empty barcodes found in {len(empty)} specimen(s). "
                         "Single-indexed demultiplexing is not supported.\n" + "\n".join(empty[:10]) + more)
    if not specimens._specimens:
        raise ValueError("No valid data found in the specimen file")
    return specimens


# ------------------------------------------------------------------------------------ output
def _safe(sample_id: str) -> str:
    return "".join(c if c.isalnum() or c in "._-$#" else "_" for c in sample_id)


class OutputManager:
    """Buffered append-only writer for the output tree.  One process owns one instance; files are
    opened lazily in append mode and at most `max_open_files` handles stay open."""

    def __init__(self, output_dir: str, prefix: str, is_fastq: bool, max_open_files: int = 200, buffer_size: int = 500):
        self.output_dir, self.prefix, self.is_fastq = output_dir, prefix, is_fastq
        self.max_open_files, self.buffer_size = max_open_files, buffer_size
        self._buffers: Dict[str, List[str]] = defaultdict(list)
        self._handles: Dict[str, object] = {}

    def __enter__(self):
        os.makedirs(self.output_dir, exist_ok=True)
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _make_filename(self, sample_id, pool, p1, p2, resolution_type: ResolutionType) -> str:
        ext = ".fastq" if self.is_fastq else ".fasta"
        top = "unknown" if resolution_type.is_unknown() else ("partial" if resolution_type.is_partial_match() else "full")
        return os.path.join(self.output_dir, top, pool or "unknown", f"{p1 or 'unknown'}-{p2 or 'unknown'}",
                            f"{self.prefix}{_safe(sample_id or SampleId.UNKNOWN)}{ext}")

    def record_text(self, op: WriteOperation) -> str:
        header = f"{op.seq_id} {op.distance_code} pool={op.primer_pool} primers={op.p1_name}+{op.p2_name} {op.sample_id}"
        if self.is_fastq:
            return f"@{header}\n{op.sequence}\n+\n{op.quality_sequence}\n"
        return f">{header}\n{op.sequence}\n"

    def write_sequence(self, write_op: WriteOperation, trace_logger=None):
        text = self.record_text(write_op)
        filename = self._make_filename(write_op.sample_id, write_op.primer_pool, write_op.p1_name, write_op.p2_name,
                                       write_op.resolution_type)
        if trace_logger:   # io_utils.py:227-233: the path relative to the output directory
            trace_logger.log_sequence_output(write_op.trace_sequence_id, write_op.sample_id, write_op.primer_pool,
                                             f"{write_op.p1_name}-{write_op.p2_name}",
                                             os.path.relpath(filename, self.output_dir))
        self.write(filename, text)
        if write_op.resolution_type.is_full_match():   # pool-level aggregate (io_utils.py:256-268)
            ext = ".fastq" if self.is_fastq else ".fasta"
            self.write(os.path.join(self.output_dir, "full", write_op.primer_pool,
                                    f"{self.prefix}{_safe(write_op.sample_id)}{ext}"), text)

    def write(self, filename: str, data: str):
        buf = self._buffers[filename]
        buf.append(data)
        if len(buf) >= self.buffer_size:
            self.flush_buffer(filename)

    def flush_buffer(self, filename: str):
        buf = self._buffers.get(filename)
        if not buf:
            return
        fh = self._handles.get(filename)
        if fh is None:
            if len(self._handles) >= self.max_open_files:
                old, oldfh = next(iter(self._handles.items()))
                oldfh.close()
                del self._handles[old]
            os.makedirs(os.path.dirname(filename), exist_ok=True)
            fh = self._handles[filename] = open(filename, "a")
        fh.write("".join(buf))
        buf.clear()

    def flush_all(self):
        for filename in list(self._buffers):
            self.flush_buffer(filename)
        for fh in self._handles.values():
            fh.flush()

    def close(self):
        self.flush_all()
        for fh in self._handles.values():
            fh.close()
        self._handles.clear()


def output_write_operation(write_op: WriteOperation, output_manager: Optional[OutputManager], args, trace_logger=None):
    if args.output_to_files:
        output_manager.write_sequence(write_op, trace_logger)
        return
    seq = write_op.sequence
    if getattr(args, "color", False):
        from .alignment import color_sequence
        seq = color_sequence(seq, write_op.quality_scores, write_op.p1_location, write_op.p2_location,
                             write_op.b1_location, write_op.b2_location)
    mark = "@" if args.isfastq else ">"
    sys.stdout.write(f"{mark}{write_op.seq_id} {write_op.distance_code} {write_op.sample_id}\n{seq}\n")
    if args.isfastq:
        sys.stdout.write(f"+\n{write_op.quality_sequence}\n")


def cleanup_locks(output_dir: str):
    shutil.rmtree(os.path.join(output_dir, ".specimux_locks"), ignore_errors=True)


def cleanup_empty_directories(output_dir: str):
    """Bottom-up removal of directories holding nothing but primers.fasta / primers.txt (io_utils.py:521-568)."""
    if not os.path.exists(output_dir):
        return
    meta = {"primers.fasta", "primers.txt"}
    for dirpath, _dirnames, filenames in os.walk(output_dir, topdown=False):
        if dirpath == output_dir:
            continue
        try:
            if any(f not in meta for f in filenames):   # the cheap test first: a directory of 768 sample files is not stat'ed
                continue
            entries = os.listdir(dirpath)
            if any(os.path.isdir(os.path.join(dirpath, e)) for e in entries):
                continue
            for f in filenames:
                os.remove(os.path.join(dirpath, f))
            os.rmdir(dirpath)
        except OSError:
            pass
