"""Native streaming helpers (libsmx.so host code, no GPU): reader vs the Python parser, window packer, and the
output writer vs the oracle's tree -- the writer is fed smx_op records built from the ORACLE's write operations,
so formatting / orientation / trimming / paths are checked without a device."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN, read_expected_tree
from oracle import specimux_oracle as O
from parity_utils import Both

P, S = f"{GOLDEN}/primers.fasta", f"{GOLDEN}/specimens.txt"


def _all_records(path, chunk=1000):
    from specimux_amd.native_io import Reader
    r = Reader(path)
    out = []
    while True:
        b = r.next_batch(chunk)
        if b is None:
            break
        out += [b.record(i) for i in range(len(b))]
    return out, r.is_fastq


def test_reader_equals_python_parser_on_golden():
    from specimux_amd.io_utils import open_sequence_file
    from parity_utils import make_args
    for name in ("sequences.fastq", "sequences_rc.fastq"):
        got, fq = _all_records(f"{GOLDEN}/{name}", chunk=7)
        exp = [(r.id, r.seq, r.quality_string) for r in open_sequence_file(f"{GOLDEN}/{name}", make_args())]
        assert fq and got == exp and len(got) == 40


def test_reader_formats(tmp_path):
    from specimux_amd.io_utils import open_sequence_file
    from parity_utils import make_args
    cases = {
        "wrapped.fq": "@r1 x y\nACGT\nAC\n+\n@III\nII\n@r2\nGG\n+r2\n@@\n\n@r3\n\n+\n\n",
        "crlf.fastq": "@a b\r\nACGT\r\n+\r\nIIII\r\n@c\r\nTT\r\n+\r\n##\r\n",
        "plain.fasta": ">a desc\nACG T\nTT\n\n>b\nGGA\n>empty\n>c\nA\n",
        "noext": "@q\nAC\n+\nII\n",
        "noext_fa": ">q\nAC\n",
    }
    for name, text in cases.items():
        path = tmp_path / name
        path.write_text(text)
        gz = tmp_path / (name + ".gz")
        with gzip.open(gz, "wt") as fh:
            fh.write(text)
        for pth in (path, gz):
            got, fq = _all_records(os.fspath(pth), chunk=2)
            a = make_args()
            exp = [(r.id, r.seq, r.quality_string) for r in open_sequence_file(os.fspath(pth), a)]
            assert got == exp, (name, got, exp)
            assert fq == a.isfastq
    bad = tmp_path / "bad.fastq"
    bad.write_text("@r\nACGT\n+\nII\n")
    from specimux_amd import _lib
    with pytest.raises(_lib.SmxError):
        _all_records(os.fspath(bad))


@pytest.mark.parametrize("engine", ["parallel", "serial"])
@pytest.mark.parametrize("kind", ["fastq", "fasta"])
def test_reader_rejects_truncated_and_corrupt_gzip(tmp_path, monkeypatch, kind, engine):
    """A gzip stream that ends inside a member, or fails its integrity check, is an error (the reference's gzip
    module raises EOFError / BadGzipFile in open_sequence_file): never a short but 'successful' read."""
    from specimux_amd import _lib
    if engine == "serial":
        monkeypatch.setenv("SMX_IO_SERIAL", "1")
    rng = np.random.default_rng(3)
    recs = []
    for i in range(4000):
        s = "".join("ACGT"[c] for c in rng.integers(0, 4, 300))
        recs.append(f"@r{i}\n{s}\n+\n{'I' * 300}\n" if kind == "fastq" else f">r{i}\n{s[:150]}\n{s[150:]}\n")
    raw = gzip.compress("".join(recs).encode(), 6)
    good = tmp_path / f"good.{kind}.gz"
    good.write_bytes(raw)
    got, _fq = _all_records(os.fspath(good), chunk=1500)
    assert len(got) == 4000
    cut = tmp_path / f"cut.{kind}.gz"
    cut.write_bytes(raw[: len(raw) * 2 // 3])                     # truncated mid-stream
    flipped = bytearray(raw)
    for k in range(len(raw) // 2, len(raw) // 2 + 8):
        flipped[k] ^= 0x5A                                       # corrupt deflate data mid-stream
    flip = tmp_path / f"flip.{kind}.gz"
    flip.write_bytes(bytes(flipped))
    notrailer = tmp_path / f"notrailer.{kind}.gz"
    notrailer.write_bytes(raw[:-6])                               # only the CRC/size trailer is incomplete
    for bad in (cut, flip, notrailer):
        with pytest.raises(_lib.SmxError):
            _all_records(os.fspath(bad), chunk=1500)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_byte_ranges_partition_the_records(tmp_path, world):
    """Multi-GPU file sharding: rank k reads the records that START inside its byte range; over all ranks every record
    appears exactly once, in file order, whatever the cut points hit (header, sequence, '+' or quality line)."""
    from specimux_amd.distributed import shard_range
    from specimux_amd.native_io import Reader
    rng = np.random.default_rng(world)
    recs = []
    for i in range(3000):
        n = int(rng.integers(1, 400))
        s = "".join("ACGT"[c] for c in rng.integers(0, 4, n))
        q = "".join(chr(33 + int(c)) for c in rng.integers(0, 42, n))   # qualities include '@' (31) and '+' (10)
        recs.append((f"r{i}", s, q))
    path = tmp_path / "reads.fastq"
    path.write_text("".join(f"@{i} x\n{s}\n+\n{q}\n" for i, s, q in recs))
    size = os.path.getsize(path)
    got = []
    for rank in range(world):
        lo, hi = shard_range(size, rank, world)
        r = Reader(os.fspath(path), byte_range=(lo, hi))
        while True:
            b = r.next_batch(700)
            if b is None:
                break
            got += [b.record(i) for i in range(len(b))]
    assert got == recs
    # degenerate ranges: empty, beyond the end
    for lo, hi in ((10, 10), (size, size + 100), (size - 1, size)):
        r = Reader(os.fspath(path), byte_range=(lo, hi))
        assert r.next_batch(10) is None
    from specimux_amd import _lib
    gz = tmp_path / "reads.fastq.gz"
    with gzip.open(gz, "wt") as fh:
        fh.write(path.read_text())
    with pytest.raises(_lib.SmxError):
        Reader(os.fspath(gz), byte_range=(0, 100))


def test_pack_windows_batch_equals_pack_windows():
    from specimux_amd.demultiplex import compiled_panel, concat_records
    from specimux_amd.io_utils import SeqRecord
    from specimux_amd.native_io import Reader
    both = Both(P, S)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    b = Reader(f"{GOLDEN}/sequences.fastq").next_batch(100)
    w1, l1 = b.pack_windows(cp.search_len, cp.window_stride)
    recs = [SeqRecord(s, i, i, q) for i, s, q in (b.record(k) for k in range(len(b)))]
    bases, offsets, _ = concat_records(recs)
    w2, l2 = cp.pack_windows(bases, offsets)
    assert np.array_equal(w1, w2) and np.array_equal(l1, l2)


def _pack4_reference(seqs, S, pstride):
    """4-bit windows restated in numpy (include/smx.h smx_pack_windows4): base j of a window in byte j / 2, low nibble
    first, code = index in ACGTNRYKMSWBDHV or 15; -> (packed, number of reads with a 'U' inside a window)."""
    lut = np.full(256, 15, dtype=np.uint8)
    lut[np.frombuffer(b"ACGTNRYKMSWBDHV", dtype=np.uint8)] = np.arange(15, dtype=np.uint8)
    hb = (S + 1) // 2
    out = np.full((len(seqs), pstride), 0xFF, dtype=np.uint8)
    special = 0
    for i, s in enumerate(seqs):
        a = np.frombuffer(s.encode("latin-1"), dtype=np.uint8)
        Sp = min(len(a), S)
        for e, w in enumerate((a[:Sp], a[len(a) - Sp:])):
            codes = np.full(2 * hb, 15, dtype=np.uint8)
            codes[:Sp] = lut[w]
            out[i, e * hb:(e + 1) * hb] = codes[0::2] | (codes[1::2] << 4)
        special += int((a[:Sp] == 85).any() or (a[len(a) - Sp:] == 85).any())
    return out, special


@pytest.mark.parametrize("search_len", [80, 81, 7, 1, 160])
def test_pack_windows4_equals_numpy_restatement(tmp_path, search_len):
    """The 4-bit packer (both entry points: parsed batch and flat arrays) against a numpy restatement: every IUPAC letter,
    lower case, other characters, 'U', reads shorter than the window, empty-ish reads."""
    import ctypes as C
    from specimux_amd import _lib
    from specimux_amd.native_io import Reader
    lib = _lib.load()
    rng = np.random.default_rng(search_len)
    alphabet = "ACGTACGTACGTACGTNRYKMSWBDHVacgtnXU?-"
    lens = rng.integers(1, 3 * search_len + 6, 400)
    seqs = ["".join(alphabet[k] for k in rng.integers(0, len(alphabet), L)) for L in lens]
    pstride = (2 * ((search_len + 1) // 2) + 15) & ~15
    exp, n_u = _pack4_reference(seqs, search_len, pstride)
    assert n_u > 0
    # flat arrays
    bases = np.frombuffer("".join(seqs).encode("latin-1"), dtype=np.uint8)
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    packed, ol, nsp = np.zeros((len(seqs), pstride), dtype=np.uint8), np.zeros(len(seqs), dtype=np.int32), C.c_uint32()
    _lib.check(lib.smx_pack_windows4(_lib.ptr(bases), _lib.ptr(off), len(seqs), search_len, _lib.ptr(packed), _lib.ptr(ol), C.byref(nsp)))
    assert np.array_equal(packed, exp) and np.array_equal(ol, lens) and nsp.value == n_u
    # parsed batch (threaded)
    fq = tmp_path / "r.fastq"
    fq.write_text("".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(seqs)))
    b = Reader(str(fq)).next_batch(1000)
    packed2, ol2 = np.zeros_like(packed), np.zeros_like(ol)
    assert b.pack_windows4_into(search_len, packed2, ol2) == n_u
    assert np.array_equal(packed2, exp) and np.array_equal(ol2, lens)


def _ops_from_oracle(cp, oracle_ops, n_reads, id_to_index):
    """Oracle write operations -> (ops[n_reads], extra[]) smx_op arrays."""
    from specimux_amd import _lib
    ops = np.zeros(n_reads, dtype=_lib.OP_DTYPE)
    ops["rtype"] = _lib.R_FILTERED
    extra = []
    seen = set()
    for op in oracle_ops:
        rec = np.zeros(1, dtype=_lib.OP_DTYPE)[0]
        i = id_to_index[op.seq_id]
        rec["read"] = i
        rec["sample"] = cp.specimen_ids.index(op.sample_id) if op.sample_id in cp.specimen_ids else -1
        rec["trim_start"], rec["trim_end"] = op.trim
        rec["pool"] = cp.pools.index(op.pool) if op.pool in cp.pools else -1
        rec["p1"] = cp.primer_names.index(op.p1) if op.p1 in cp.primer_names else -1
        rec["p2"] = cp.primer_names.index(op.p2) if op.p2 in cp.primer_names else -1
        rec["barcode"] = cp.barcodes.index(op.sample_id.split("_")[-1]) if op.sample_id.startswith("barcode_") else -1
        rec["dist"] = [(-1 if d == "X" else int(d)) for d in op.code.split(",")]
        rec["rtype"] = op.rtype
        rec["flags"] = _lib.OPF_REVERSE if op.reverse else 0
        if i in seen:
            extra.append(rec)
        else:
            seen.add(i)
            ops[i] = rec
    return ops, (np.array(extra, dtype=_lib.OP_DTYPE) if extra else np.zeros(0, dtype=_lib.OP_DTYPE))


@pytest.mark.parametrize("seqfile", ["sequences.fastq", "sequences_rc.fastq"])
def test_writer_tree_from_oracle_ops_equals_expected_output(tmp_path, seqfile):
    from specimux_amd.demultiplex import compiled_panel
    from specimux_amd.native_io import Reader, Writer
    both = Both(P, S)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    reads, _ = O.read_sequences(f"{GOLDEN}/{seqfile}")
    oracle_ops, _, _ = O.process_sequences(reads, both.opar, both.opanel)
    batch = Reader(f"{GOLDEN}/{seqfile}").next_batch(1000)
    ops, extra = _ops_from_oracle(cp, oracle_ops, len(reads), {r[0]: i for i, r in enumerate(reads)})
    w = Writer(os.fspath(tmp_path / "out"), "", True, cp)
    w.write(batch, ops, extra)
    w.close()
    got = read_expected_tree(os.fspath(tmp_path / "out"))
    tree = {}
    for op in oracle_ops:
        for p in O.op_path(op):
            tree.setdefault(p, []).append(O.op_record(op))
    assert got == {k: sorted(v) for k, v in tree.items()}
    if seqfile == "sequences.fastq":
        assert got == read_expected_tree(f"{GOLDEN}/expected_output")


def test_writer_prefix_fasta_and_safe_names(tmp_path):
    from specimux_amd import _lib
    from specimux_amd.native_io import Reader, Writer

    class FakePanel:
        specimen_ids = ["we ird/na:me", "ok-1"]
        pools, primer_names, barcodes = ["P1"], ["F", "R"], ["ACGT"]
    fa = tmp_path / "in.fasta"
    fa.write_text(">r1 d\nAACCGGTT\n>r2\nACGTACGT\n")
    batch = Reader(os.fspath(fa)).next_batch(10)
    ops = np.zeros(2, dtype=_lib.OP_DTYPE)
    ops[0] = (0, 2, 6, 0, 0, 1, -1, [0, 1, 2, 0], _lib.R_DEREP_FULL, _lib.OPF_REVERSE, 1, 0)
    ops[1] = (-1, 0, 8, -1, -1, 1, 0, [-1, -1, 3, 1], _lib.R_PARTIAL_REV, 0, 1, 1)
    w = Writer(os.fspath(tmp_path / "o"), "pre_", False, FakePanel)
    w.write(batch, ops, np.zeros(0, dtype=_lib.OP_DTYPE))
    w.close()
    # reverse complement of AACCGGTT is AACCGGTT; [2:6] = CCGG
    assert (tmp_path / "o/full/P1/F-R/pre_we_ird_na_me.fasta").read_text() == ">r1 0,1,2,0 pool=P1 primers=F+R we ird/na:me\nCCGG\n"
    assert (tmp_path / "o/full/P1/pre_we_ird_na_me.fasta").exists()
    assert (tmp_path / "o/partial/unknown/unknown-R/pre_barcode_rev_ACGT.fasta").read_text() == \
        ">r2 X,X,3,1 pool=unknown primers=unknown+R barcode_rev_ACGT\nACGTACGT\n"


def test_fast_engine_blocks_and_fallback(tmp_path):
    """Uncompressed strict FASTQ goes through the parallel zero-copy engine: small byte budgets force many blocks
    (records cut by block ends), small read budgets force truncation + re-reads; an irregular record in the middle
    of the file switches to the general engine without losing or duplicating records."""
    from specimux_amd import synth
    from specimux_amd.native_io import Reader
    pan = synth.panel_c1()
    rs = synth.make_reads(pan, 6000, 5, windows_only=False)
    fq = tmp_path / "big.fastq"
    rs.write_fastq(os.fspath(fq))
    exp = [(f"read{i:07d}", s, q) for i, (s, q) in enumerate(zip(rs.reads, rs.quals))]
    for max_reads, max_bytes in ((100000, 0), (1000, 0), (100000, 300000), (777, 150000)):
        r = Reader(os.fspath(fq))
        got = []
        while True:
            b = r.next_batch(max_reads, max_bytes)
            if b is None:
                break
            assert len(b) <= max_reads
            got += [b.record(i) for i in range(len(b))]
        assert got == exp, (max_reads, max_bytes, len(got))
    # irregular record (wrapped sequence) after 4000 regular ones
    lines = open(fq).read().split("\n")
    k = 4 * 4000
    seq = lines[k + 1]
    lines[k + 1:k + 2] = [seq[:50], seq[50:]]
    wrapped = tmp_path / "wrapped_mid.fastq"
    wrapped.write_text("\n".join(lines))
    for max_bytes in (0, 200000):
        r = Reader(os.fspath(wrapped))
        got = []
        while True:
            b = r.next_batch(1500, max_bytes)
            if b is None:
                break
            got += [b.record(i) for i in range(len(b))]
        assert got == exp


def test_threaded_writer_equals_oracle_tree(tmp_path):
    """> 512 reads: the writer shards the output files over threads; the tree must equal the oracle's."""
    from specimux_amd import synth
    from specimux_amd.demultiplex import compiled_panel
    from specimux_amd.native_io import Reader, Writer
    pan = synth.panel_c1()
    pf, sf = pan.write(os.fspath(tmp_path / "panel"))
    rs = synth.make_reads(pan, 3000, 8, windows_only=False)
    fq = tmp_path / "r.fastq"
    rs.write_fastq(os.fspath(fq))
    both = Both(pf, sf)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    reads, _ = O.read_sequences(os.fspath(fq))
    oracle_ops, _, _ = O.process_sequences(reads, both.opar, both.opanel)
    batch = Reader(os.fspath(fq)).next_batch(10000)
    assert len(batch) == 3000
    ops, extra = _ops_from_oracle(cp, oracle_ops, len(reads), {r[0]: i for i, r in enumerate(reads)})
    w = Writer(os.fspath(tmp_path / "out"), "", True, cp)
    w.write(batch, ops, extra)
    w.close()
    got = read_expected_tree(os.fspath(tmp_path / "out"))
    tree = {}
    for op in oracle_ops:
        for p in O.op_path(op):
            tree.setdefault(p, []).append(O.op_record(op))
    assert got == {k: sorted(v) for k, v in tree.items()}
    # per-file record order = input order (the reference makes no promise here; this implementation does)
    for path, recs in tree.items():
        text = open(os.path.join(tmp_path, "out", path)).read()
        assert text == "".join(recs), path


def test_writer_reverse_complement_every_length_and_letter(tmp_path):
    """The writer reverse-complements 16 bytes per step (smx_io.cpp revcomp_copy / reverse_copy): every record length
    0..70 with every trim offset 0..3, over IUPAC letters in both cases, 'U', and bytes outside the alphabet, against
    str.translate of the reference's table (models.py: Bio.Seq.reverse_complement semantics) and a reversed quality."""
    from specimux_amd import _lib
    from specimux_amd.models import reverse_complement
    from specimux_amd.native_io import Reader, Writer

    class FakePanel:
        specimen_ids, pools, primer_names, barcodes = ["s"], ["P"], ["F", "R"], ["ACGT"]
    rng = np.random.default_rng(5)
    letters = "ACGTMRWSYKVHDBXNUacgtmrwsykvhdbxnu*-.?"
    recs = []
    for L in range(1, 71):
        for k in range(4):
            seq = "".join(letters[j] for j in rng.integers(0, len(letters), L))
            qual = "".join(chr(33 + int(j)) for j in rng.integers(0, 60, L))
            recs.append((f"r{L}_{k}", seq, qual, k if k < L else 0))
    fq = tmp_path / "in.fastq"
    fq.write_text("".join(f"@{i}\n{s}\n+\n{q}\n" for i, s, q, _ in recs))
    batch = Reader(os.fspath(fq)).next_batch(len(recs) + 1)
    assert len(batch) == len(recs)
    ops = np.zeros(len(recs), dtype=_lib.OP_DTYPE)
    for i, (_id, s, _q, k) in enumerate(recs):
        ops[i] = (0, k, max(k, len(s) - (k + 1) // 2), 0, 0, 1, -1, [0, 0, 0, 0], _lib.R_DEREP_FULL, _lib.OPF_REVERSE, 1, i)
    w = Writer(os.fspath(tmp_path / "o"), "", True, FakePanel)
    w.write(batch, ops, np.zeros(0, dtype=_lib.OP_DTYPE))
    w.close()
    exp = ""
    for (rid, s, q, k), op in zip(recs, ops):
        a, e = int(op["trim_start"]), int(op["trim_end"])
        exp += f"@{rid} 0,0,0,0 pool=P primers=F+R s\n{reverse_complement(s)[a:e]}\n+\n{q[::-1][a:e]}\n"
    assert (tmp_path / "o/full/P/F-R/s.fastq").read_text() == exp
    assert (tmp_path / "o/full/P/s.fastq").read_text() == exp


@pytest.mark.parametrize("pos", [3, 15, 16, 17, 40, 63])
@pytest.mark.parametrize("ch", [" ", "\t"])
def test_white_space_inside_a_sequence_line_leaves_the_fast_engine(tmp_path, pos, ch, monkeypatch):
    """The strict parser checks every sequence line for white space 16 bytes at a time (smx_io.cpp has_space): a blank
    anywhere in the line must hand the file to the general engine, i.e. give what the copying path gives."""
    from specimux_amd.native_io import Reader
    seq = "ACGTTGCAAC" * 7
    recs = [("a", seq, "I" * 70), ("b", seq[:pos] + ch + seq[pos + 1:], "J" * 70), ("c", seq[::-1], "K" * 70)]
    fq = tmp_path / "blank.fastq"
    fq.write_text("".join(f"@{i}\n{s}\n+\n{q}\n" for i, s, q in recs))

    def read_all():
        r = Reader(os.fspath(fq))
        out = []
        while True:
            b = r.next_batch(100)
            if b is None:
                return out
            out += [b.record(i) for i in range(len(b))]
    got = read_all()
    monkeypatch.setenv("SMX_IO_NO_MMAP", "1")
    assert got == read_all()
    exp, _ = O.read_sequences(os.fspath(fq))     # the oracle's line-based parser (Bio.SeqIO semantics)
    assert got == [tuple(r) for r in exp]
    assert [g[0] for g in got] == ["a", "b", "c"] and got[0] == recs[0] and got[2] == recs[2]


def test_writer_many_tiny_records_flush_by_piece_count(tmp_path, monkeypatch):
    """A flush is ONE writev (ranks of a multi-GPU run append to the same files: specimux_amd/distributed.py), so a file's
    pending pieces are capped near 1000 whatever their bytes: 6000 trimmed 12-base records for one file are some 24 000
    pieces, far below the 64 KB byte threshold per flush at a time -- every record must arrive, in order, untorn."""
    from specimux_amd import _lib
    from specimux_amd.native_io import Reader, Writer

    class FakePanel:
        specimen_ids, pools, primer_names, barcodes = ["s"], ["P"], ["F", "R"], ["ACGT"]
    monkeypatch.setenv("SMX_IO_THREADS", "4")
    n = 6000
    rng = np.random.default_rng(9)
    seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, 12)) for _ in range(n)]
    fq = tmp_path / "tiny.fastq"
    fq.write_text("".join(f"@t{i}\n{s}\n+\n{'I' * 12}\n" for i, s in enumerate(seqs)))
    batch = Reader(os.fspath(fq)).next_batch(n + 1)
    assert len(batch) == n
    ops = np.zeros(n, dtype=_lib.OP_DTYPE)
    for i in range(n):
        ops[i] = (0, 2, 10, 0, 0, 1, -1, [0, 0, 0, 0], _lib.R_DEREP_FULL, 0, 1, i)
    w = Writer(os.fspath(tmp_path / "o"), "", True, FakePanel)
    w.write(batch, ops, np.zeros(0, dtype=_lib.OP_DTYPE))
    w.close()
    exp = "".join(f"@t{i} 0,0,0,0 pool=P primers=F+R s\n{s[2:10]}\n+\nIIIIIIII\n" for i, s in enumerate(seqs))
    assert (tmp_path / "o/full/P/F-R/s.fastq").read_text() == exp
    assert (tmp_path / "o/full/P/s.fastq").read_text() == exp
