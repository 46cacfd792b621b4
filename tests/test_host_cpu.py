"""CPU-side tests (no GPU needed): the C-ABI library loads and exports every symbol include/smx.h declares,
host logic (parsers, thresholds, panel compiler, window packer, generator) agrees with the oracle, and the
product path refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

from conftest import GOLDEN, REPO
from oracle import edlib_semantics as E
from oracle import specimux_oracle as O
from parity_utils import Both, make_args

P, S = f"{GOLDEN}/primers.fasta", f"{GOLDEN}/specimens.txt"


def test_library_exports_every_declared_symbol():
    from specimux_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "smx.h")).read()
    declared = set(re.findall(r"\b(smx_[a-z_0-9]+)\s*\(", header))
    bound = {name for name, _r, _a in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.smx_abi_version() == _lib.ABI_VERSION


def test_record_layouts_match_header():
    from specimux_amd import _lib
    assert _lib.OP_DTYPE.itemsize == 32 and _lib.HIT_DTYPE.itemsize == 24
    assert _lib.OP_DTYPE.fields["dist"][1] == 20 and _lib.OP_DTYPE.fields["read"][1] == 28


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from specimux_amd import _lib
    both = Both(P, S)
    with pytest.raises(_lib.SmxError) as ei:
        both.product_ops([("x", "ACGT" * 30, "I" * 120)])
    assert ei.value.code == _lib.ERR_DEVICE and "no CPU path" in str(ei.value)
    d, n = C.c_int(), C.c_int()
    rc = _lib.load().smx_align(b"ACGT", 4, b"ACGTACGT", 8, 1, 0, C.byref(d), None, None, 0, C.byref(n))
    assert rc == _lib.ERR_DEVICE


def test_thresholds_and_registration_order_match_oracle():
    both = Both(P, S)
    assert both.parameters.max_dist_index == 3
    assert [p.name for p in both.specimens._primers.values()] == ["gITS7", "ITS4", "ITS1F"]
    assert [p.barcodes for p in both.specimens._primers.values()] == [p.barcodes for p in both.opanel.primers.values()]
    for flags in (dict(index_edit_distance=2), dict(primer_edit_distance=3)):
        b = Both(P, S, **flags)
        assert b.parameters.max_dist_primers == b.opar.max_dist_primers


def test_edit_distance_equals_oracle_nw():
    from specimux_amd.orchestration import edit_distance
    rnd = random.Random(3)
    for _ in range(500):
        a = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 30)))
        b = "".join(rnd.choice("ACGTN") for _ in range(rnd.randint(0, 30)))
        assert edit_distance(a, b) == E.align(a, b, E.NW, -1, iupac=False)["editDistance"], (a, b)


def test_native_min_pairwise_distance_equals_python_loop():
    """setup_match_parameters takes its barcode distances from libsmx (smx_min_pairwise_distance); the Python bit-parallel
    edit_distance stays as the checker: random sets of equal and ragged lengths, duplicates (distance 0), the empty
    string, a set large enough for the threaded path, and non-ASCII text (falls back to the Python loop)."""
    import itertools
    from specimux_amd.orchestration import _native_min_pairwise, edit_distance
    rng = np.random.default_rng(11)

    def rand_seq(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    cases = [[rand_seq(13) for _ in range(40)], [rand_seq(int(rng.integers(5, 20))) for _ in range(30)],
             ["ACGT", "ACGT", "TTTT"], ["", "ACG"], [rand_seq(12) for _ in range(100)], ["AÇGT", "ACGT", "AGGT"]]
    for seqs in cases:
        exp = min(edit_distance(x, y) for x, y in itertools.combinations(seqs, 2))
        assert _native_min_pairwise(seqs) == exp, seqs[:3]


def test_panel_compiler_flattening():
    from specimux_amd.demultiplex import compiled_panel
    both = Both(P, S)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    assert cp.primer_names == ["gITS7", "ITS4", "ITS1F"]
    assert cp.pairs == [(0, 1, cp.pools.index("ITS2")), (2, 1, cp.pools.index("ITS"))]   # Q5 / Q6
    assert cp.specimen_ids == ["TEST_SPECIMEN_001", "TEST_SPECIMEN_002", "TEST_SPECIMEN_003"]
    assert cp.counts_len == 8 + 3 and cp.window_stride == 160 and cp.hits_per_read == 6
    assert cp.desc.prefilter_min_len == 10 and cp.desc.k_index == 3
    d = cp._keep
    assert list(d["primer_k"]) == [6, 6, 7] and list(d["primer_file_index"]) == [2, 0, 1]
    assert list(d["spec_p1mask"]) == [1, 1, 4] and list(d["spec_p2mask"]) == [2, 2, 2]


def test_panel_limits_fail_loudly(tmp_path):
    from specimux_amd import _lib
    from specimux_amd.demultiplex import compiled_panel
    pf = tmp_path / "p.fasta"
    pf.write_text(">F pool=X position=forward\n" + "ACGT" * 17 + "\n>R pool=X position=reverse\nTCCTCCGCTTATTGATATGC\n")
    sf = tmp_path / "s.txt"
    sf.write_text("SampleID\tPrimerPool\tFwIndex\tFwPrimer\tRvIndex\tRvPrimer\n"
                  "a\tX\tACGTACGTACGTA\tF\tTTGCAAGGTCAAC\tR\nb\tX\tGGATCCAATTGCA\tF\tCATGCATTTGGAC\tR\n")
    both = Both(os.fspath(pf), os.fspath(sf))
    with pytest.raises(_lib.SmxError) as ei:
        compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    assert ei.value.code == _lib.ERR_UNSUPPORTED and "length 68" in str(ei.value)


def test_pack_windows():
    from specimux_amd.demultiplex import compiled_panel, concat_records
    from specimux_amd.io_utils import SeqRecord
    both = Both(P, S)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    rnd = random.Random(1)
    seqs = ["", "A", "ACGT" * 10, "".join(rnd.choice("ACGT") for _ in range(80)),
            "".join(rnd.choice("ACGT") for _ in range(81)), "".join(rnd.choice("ACGTN") for _ in range(500))]
    bases, offsets, _ = concat_records([SeqRecord(s, f"r{i}") for i, s in enumerate(seqs)])
    w, lens = cp.pack_windows(bases, offsets)
    assert w.shape == (len(seqs), 160) and list(lens) == [len(s) for s in seqs]
    for i, s in enumerate(seqs):
        sp = min(80, len(s))
        assert w[i, :sp].tobytes().decode() == s[:sp] and not w[i, sp:80].any()
        assert w[i, 80:80 + sp].tobytes().decode() == s[len(s) - sp:] and not w[i, 80 + sp:].any()


def test_fastq_parser_matches_oracle_reader(tmp_path):
    from specimux_amd.io_utils import open_sequence_file
    args = make_args()
    mine = [(r.id, r.seq, r.quality_string) for r in open_sequence_file(f"{GOLDEN}/sequences.fastq", args)]
    assert args.isfastq and mine == list(O.read_fastq(f"{GOLDEN}/sequences.fastq"))
    assert mine[0][0] == "d5a99b43-8367-4a95-9143-12c3f62f07f1"       # id = first whitespace token (tab here)
    wrapped = tmp_path / "w.fq"
    wrapped.write_text("@r1 x\nACGT\nAC\n+\n@III\nII\n@r2\nGG\n+r2\n@@\n")
    got = [(r.id, r.seq, r.quality_string) for r in open_sequence_file(os.fspath(wrapped), args)]
    assert got == [("r1", "ACGTAC", "@IIIII"), ("r2", "GG", "@@")]


def test_generator_windows_equal_full_reads():
    from specimux_amd import synth
    pan = synth.panel_c1()
    rs = synth.make_reads(pan, 500, 9, windows_only=False)
    ws = synth.make_reads(pan, 500, 9)
    assert np.array_equal(rs.head, ws.head) and np.array_equal(rs.tail, ws.tail) and np.array_equal(rs.lens, ws.lens)
    for i, s in enumerate(rs.reads):
        sp = min(80, len(s))
        assert len(s) == rs.lens[i]
        assert rs.head[i, :sp].tobytes().decode() == s[:sp] and rs.tail[i, :sp].tobytes().decode() == s[len(s) - sp:]
    assert set(np.unique(rs.truth["category"])) == set(range(6))
    # chunk independence: whole chunks of a longer set are the same reads
    old = synth.CHUNK
    try:
        synth.CHUNK = 128
        a, b = synth.make_reads(pan, 256, 9), synth.make_reads(pan, 300, 9)
        assert np.array_equal(b.head[:256], a.head) and np.array_equal(b.lens[:256], a.lens)
    finally:
        synth.CHUNK = old


def test_bloom_prefilter_exact_set_rule_matches_oracle():
    from specimux_amd.bloom_filter import BloomPrefilter
    mine = BloomPrefilter(["ACGTACGTTGCAA", "TTGACCATGCATG"], 3)
    theirs = O.ExactPrefilter(["ACGTACGTTGCAA", "TTGACCATGCATG"], 3)
    rnd = random.Random(2)
    for _ in range(400):
        t = "".join(rnd.choice("ACGTN") for _ in range(rnd.randint(0, 16)))
        if rnd.random() < 0.5:
            t = "ACGTACGTTGCAA"[rnd.randint(0, 3):] + t
        for b in mine.barcodes:
            assert mine.match(b, t) == theirs.match(b, t), (b, t)


def test_subsample_top_quality(tmp_path):
    from specimux_amd.orchestration import subsample_top_quality
    d = tmp_path / "out" / "full" / "P" / "F-R"
    d.mkdir(parents=True)
    (d / "primers.fasta").write_text(">F\nACGT\n")
    recs = [("a", "ACGT", "IIII"), ("b", "ACGT", "####"), ("c", "AC", "5I"), ("d", "ACGT", "IIII")]
    (d / "s.fastq").write_text("".join(f"@{i} 0,0,0,0 pool=P primers=F+R s\n{s}\n+\n{q}\n" for i, s, q in recs))
    subsample_top_quality(str(tmp_path / "out"), 2)
    got = (tmp_path / "out" / "subsample" / "P" / "F-R" / "s.fastq").read_text()
    assert got == "@a 0,0,0,0 pool=P primers=F+R s\nACGT\n+\nIIII\n@d 0,0,0,0 pool=P primers=F+R s\nACGT\n+\nIIII\n"
    assert (tmp_path / "out" / "subsample" / "P" / "F-R" / "primers.fasta").exists()


def test_trace_logger_file_format(tmp_path):
    """TraceLogger (trace.py:24-107): file name, header, 1-based event counter, verbosity gates, sequence ids."""
    import csv
    from specimux_amd.trace import TraceLogger

    class Rec:
        id = "read/1"

    with TraceLogger(True, 2, str(tmp_path), "worker_7", "20260101_120000", buffer_size=2) as tl:
        sid = tl.get_sequence_id(Rec(), 41)
        assert sid == "read/1#00000041#worker_7" and tl.get_sequence_id(Rec()) == "read/1#00000001#worker_7"
        tl.log_sequence_received(sid, 123, "read/1")
        tl.log_primer_search(sid, "ITS4", "reverse", 43, 123, False, -1, -1)      # level 2: failed searches are dropped
        tl.log_primer_search(sid, "ITS4", "reverse", 43, 123, True, 2, 99)
        tl.log_barcode_search(sid, "ACGT", "reverse", "ITS4", 1, 2, True, 0, 1)   # level 3 only
        tl.log_orientation_detected(sid, "forward", 2, 0, 1.0)
        tl.log_sequence_output(sid, "S1", "ITS", "ITS1F-ITS4", "full/ITS/ITS1F-ITS4/S1.fastq")
    path = tmp_path / "trace" / "specimux_trace_20260101_120000_worker_7.tsv"
    rows = list(csv.reader(open(path, newline=""), delimiter="\t"))
    assert rows[0] == ["timestamp", "worker_id", "event_seq", "sequence_id", "event_type"]
    assert [r[1:] for r in rows[1:]] == [
        ["worker_7", "1", sid, "SEQUENCE_RECEIVED", "123", "read/1"],
        ["worker_7", "2", sid, "PRIMER_SEARCH", "ITS4", "reverse", "43", "123", "true", "2", "99"],
        ["worker_7", "3", sid, "ORIENTATION_DETECTED", "forward", "2", "0", "1.000"],
        ["worker_7", "4", sid, "SEQUENCE_OUTPUT", "S1", "ITS", "ITS1F-ITS4", "full/ITS/ITS1F-ITS4/S1.fastq"]]


def test_oracle_trace_on_golden_reads():
    """The oracle's trace restatement is self-consistent on the 40 golden reads: one RECEIVED / OUTPUT pair per
    record, candidate ids dense, level 1 is a subsequence of level 2 is a subsequence of level 3."""
    from oracle import specimux_oracle as O
    g = os.path.join(os.path.dirname(__file__), "golden", "integration_test_suite")
    panel = O.load_panel(f"{g}/primers.fasta", f"{g}/specimens.txt")
    par = O.setup_params(panel)
    reads = list(O.read_fastq(f"{g}/sequences.fastq"))
    streams = {}
    for level in (1, 2, 3):
        tr = O.Tracer(level, "main")
        ops, total, matched = O.process_sequences(reads, par, panel, tr=tr)
        O.trace_outputs(tr, ops)
        streams[level] = [tuple(r[2:]) for r in tr.rows]
        kinds = [r[3] for r in tr.rows]
        assert (total, matched) == (40, 6)
        assert kinds.count("SEQUENCE_RECEIVED") == 40 and kinds.count("SEQUENCE_OUTPUT") == len(ops) == 40
    for lo, hi in ((1, 2), (2, 3)):
        it = iter(streams[hi])
        assert all(ev in it for ev in streams[lo]), f"level {lo} is not a subsequence of level {hi}"
    assert not any(r[1] in ("PRIMER_SEARCH", "BARCODE_SEARCH") for r in streams[1])
    assert any(r[1] == "BARCODE_SEARCH" for r in streams[3])
