"""GPU parity tests (run on the MI355X box: `pytest -m gpu`).  Everything goes through the C ABI of
include/smx.h (ctypes -> libsmx.so -> HIP kernels) and is compared bit-exactly with the CPU oracle
(oracle/), which itself is pinned to the reference's golden suite by tests/test_oracle_golden.py."""
import ctypes as C
import random

import numpy as np
import pytest

from conftest import GOLDEN, read_expected_tree
from parity_utils import Both, reads_from_set, tmp_panel

pytestmark = pytest.mark.gpu

P, S = f"{GOLDEN}/primers.fasta", f"{GOLDEN}/specimens.txt"


@pytest.fixture(scope="module")
def lib():
    from specimux_amd import _lib
    lib = _lib.load()
    n = C.c_int(0)
    _lib.check(lib.smx_device_init(0, C.byref(n)))
    assert n.value >= 1
    return lib


def golden_reads(name):
    from oracle import specimux_oracle as O
    recs, _ = O.read_sequences(f"{GOLDEN}/{name}")
    return recs


# ------------------------------------------------------------------ the alignment primitive
def test_smx_align_matches_oracle(lib):
    from oracle import edlib_semantics as E
    from specimux_amd import _lib
    rnd = random.Random(11)
    alpha = "ACGT" * 8 + "NRYKMSWBDHV" + "ax"
    qalpha = "ACGT" * 6 + "NRYKMSWBDHV"
    n_checked = 0
    for it in range(300):
        m = rnd.choice([1, 5, 13, 20, 23, 31, 32, 33, 47, 64])
        n = rnd.randint(1, 120)
        q = "".join(rnd.choice(qalpha) for _ in range(m))
        t = "".join(rnd.choice(alpha) for _ in range(n))
        if it % 2 == 0 and n > m:   # plant a mutated copy so that matches exist
            pos = rnd.randint(0, n - m)
            copy = list(q)
            for _ in range(rnd.randint(0, 4)):
                copy[rnd.randrange(m)] = rnd.choice("ACGT")
            t = t[:pos] + "".join(copy) + t[pos + m:]
        for mode, mid in ((E.HW, 0), (E.SHW, 1)):
            k = rnd.choice([0, 1, 3, 7, m - 1]) if m > 1 else 0
            k = min(k, m - 1)
            exp = E.align(q, t, mode, k)
            cap = len(t)
            starts, ends = (C.c_int * cap)(), (C.c_int * cap)()
            dist, nloc = C.c_int(), C.c_int()
            _lib.check(lib.smx_align(q.encode(), len(q), t.encode("latin-1"), len(t), k, mid, C.byref(dist), starts,
                                     ends, cap, C.byref(nloc)))
            got = {"editDistance": dist.value, "locations": [(starts[i], ends[i]) for i in range(nloc.value)]}
            assert got == exp, (q, t, mode, k)
            n_checked += 1
    assert n_checked == 600


def test_smx_align_batch_matches_oracle(lib):
    """smx_align_batch (one launch for thousands of alignments; what trace level 3 and --color use through
    alignment.AlignCache) against the oracle's DP: distances and ALL optimal (start, end) locations, HW and SHW."""
    from oracle import edlib_semantics as E
    from specimux_amd.alignment import AlignCache, align_batch, align_seq
    from specimux_amd.constants import AlignMode
    rnd = random.Random(23)
    alpha = "ACGT" * 8 + "NRYKMSWBDHV" + "ax"
    qalpha = "ACGT" * 6 + "NRYKMSWBDHV"
    reqs, exp = [], []
    queries = ["".join(rnd.choice(qalpha) for _ in range(m)) for m in (1, 5, 13, 13, 20, 22, 31, 32, 33, 47, 64)]
    for it in range(3000):
        q = rnd.choice(queries)
        m = len(q)
        n = rnd.randint(1, 150)
        t = "".join(rnd.choice(alpha) for _ in range(n))
        if it % 2 == 0 and n > m:
            pos = rnd.randint(0, n - m)
            copy = list(q)
            for _ in range(rnd.randint(0, 4)):
                copy[rnd.randrange(m)] = rnd.choice("ACGT")
            t = t[:pos] + "".join(copy) + t[pos + m:]
        k = min(rnd.choice([0, 1, 3, 7, m - 1]), m - 1) if m > 1 else 0
        mode = rnd.choice([AlignMode.INFIX, AlignMode.PREFIX])
        reqs.append((q, t, k, mode))
        e = E.align(q, t, E.HW if mode == AlignMode.INFIX else E.SHW, k)
        exp.append((e["editDistance"], e["locations"]))
    got = align_batch(reqs)
    for r, g, e in zip(reqs, got, exp):
        assert g[0] == e[0] and [tuple(x) for x in g[1]] == [tuple(x) for x in e[1]], (r, g, e)
    # the cache path of align_seq returns what the single-call path returns
    cache = AlignCache()
    cache.fill([(q, (t[:len(q) + k] if mode == AlignMode.PREFIX else t), k, mode) for q, t, k, mode in reqs[:200]])
    for q, t, k, mode in reqs[:200]:
        direct = align_seq(q, t, k, 0, len(t), mode)
        with cache:
            cached = align_seq(q, t, k, 0, len(t), mode)
        assert (direct.distance(), direct.locations()) == (cached.distance(), cached.locations())


# ------------------------------------------------------------------ golden suite, through the GPU
@pytest.mark.parametrize("seqfile", ["sequences.fastq", "sequences_rc.fastq"])
def test_golden_hits_and_ops(lib, seqfile):
    both = Both(P, S)
    reads = golden_reads(seqfile)
    assert both.assert_hits_equal(reads, seqfile) > 20
    both.assert_ops_equal(reads, seqfile)


def test_golden_tree_equals_reference_expected_output(lib, tmp_path):
    """The reference's own integration test (tests/test_integration.py:75-114), through the CLI entry point,
    comparing FULL records and the primers.fasta / primers.txt side files."""
    import os
    from specimux_amd import cli
    out = tmp_path / "out"
    cli.main(["specimux", P, S, f"{GOLDEN}/sequences.fastq", "-F", "-O", str(out)])
    exp = read_expected_tree(f"{GOLDEN}/expected_output")
    got = read_expected_tree(str(out))
    assert got == exp
    log = (out / "log.txt").read_text()
    assert "Processed 40 sequences, match rate: 15.0%" in log and "Elapsed time" in log
    for dirpath, _d, files in os.walk(f"{GOLDEN}/expected_output"):
        for fn in files:
            if fn.startswith("primers."):
                rel = os.path.relpath(os.path.join(dirpath, fn), f"{GOLDEN}/expected_output")
                assert (out / rel).read_text() == open(os.path.join(dirpath, fn)).read(), rel
    assert not (out / "partial" / "unknown").exists()   # empty directories are pruned


def test_reference_integration_command_with_trace(lib, tmp_path):
    """tests/test_integration.py:75-114 verbatim: `-F -O out -d` must produce the expected tree AND a trace
    directory (`_validate_output_structure` :141-148); the trace file carries one SEQUENCE_OUTPUT per record."""
    import csv
    from specimux_amd import cli
    out = tmp_path / "out"
    cli.main(["specimux", P, S, f"{GOLDEN}/sequences.fastq", "-F", "-O", str(out), "-d"])
    assert read_expected_tree(str(out)) == read_expected_tree(f"{GOLDEN}/expected_output")
    files = sorted((out / "trace").glob("specimux_trace_*_worker_1.tsv"))
    assert len(files) == 1
    rows = list(csv.reader(open(files[0], newline=""), delimiter="\t"))
    assert rows[0] == ["timestamp", "worker_id", "event_seq", "sequence_id", "event_type"]
    kinds = [r[4] for r in rows[1:]]
    assert kinds.count("SEQUENCE_RECEIVED") == 40 and kinds.count("SEQUENCE_OUTPUT") == 40
    assert [int(r[2]) for r in rows[1:]] == list(range(1, len(rows)))
    assert "Processed 40 sequences, match rate: 15.0%" in (out / "log.txt").read_text()


# ------------------------------------------------------------------ synthetic configs
@pytest.fixture(scope="module")
def c1(tmp_path_factory):
    from specimux_amd import synth
    pan = synth.panel_c1()
    return pan, tmp_panel(tmp_path_factory, pan, "c1")


@pytest.fixture(scope="module")
def c2(tmp_path_factory):
    from specimux_amd import synth
    pan = synth.panel_c2()
    return pan, tmp_panel(tmp_path_factory, pan, "c2")


@pytest.fixture(scope="module")
def c3(tmp_path_factory):
    from specimux_amd import synth
    pan = synth.panel_c3()
    return pan, tmp_panel(tmp_path_factory, pan, "c3")


def test_c1_plumbing_1k_reads(lib, c1):
    from specimux_amd import synth
    pan, (pf, sf) = c1
    rs = synth.make_reads(pan, 1000, 1001, windows_only=False)
    reads = reads_from_set(rs, range(1000), 80)
    both = Both(pf, sf)
    both.assert_hits_equal(reads[:300], "c1")
    got = both.assert_ops_equal(reads, "c1")
    assert sum(1 for k in got if k[6] == "DEREP") > 300      # most intact reads resolve to a specimen


def test_c2_768_specimens(lib, c2):
    from specimux_amd import synth
    pan, (pf, sf) = c2
    rs = synth.make_reads(pan, 1500, 2002, windows_only=False)
    reads = reads_from_set(rs, range(1500), 80)
    both = Both(pf, sf)
    assert both.parameters.max_dist_index == 3
    both.assert_hits_equal(reads[:250], "c2")
    got = both.assert_ops_equal(reads, "c2")
    # truth recovery (sanity, not exactness): intact reads get their own specimen
    by_id = {k[0]: k for k in got if k[6] == "DEREP"}
    ok = tot = 0
    for i in range(1500):
        if rs.truth["category"][i] == 0:
            tot += 1
            k = by_id.get(f"r{i}")
            ok += bool(k and k[1] == f"ITS_F{rs.truth['fwd'][i]:02d}_R{rs.truth['rev'][i]:02d}")
    assert ok / tot > 0.6


def test_c3_four_pools_shared_primer_iupac(lib, c3):
    from specimux_amd import synth
    pan, (pf, sf) = c3
    rs = synth.make_reads(pan, 1000, 3003, insert_mean=900, insert_sd=250, windows_only=False)
    reads = reads_from_set(rs, range(1000), 80)
    both = Both(pf, sf)
    both.assert_hits_equal(reads[:120], "c3")
    both.assert_ops_equal(reads, "c3")


def test_c5_wide_window_high_error(lib, c3):
    from specimux_amd import synth
    pan, (pf, sf) = c3
    rs = synth.make_reads(pan, 400, 5005, search_len=160, error_rate=0.15, windows_only=False)
    reads = reads_from_set(rs, range(400), 160)
    both = Both(pf, sf, search_len=160)
    both.assert_hits_equal(reads[:80], "c5")
    both.assert_ops_equal(reads, "c5")


FLAG_SETS = [
    dict(trim="none"), dict(trim="tails"), dict(trim="primers"),
    dict(dereplicate="none"), dict(disable_preorient=True), dict(disable_prefilter=True),
    dict(index_edit_distance=2, primer_edit_distance=4), dict(index_edit_distance=5),
    dict(search_len=40), dict(search_len=120), dict(search_len=256), dict(min_length=300, max_length=900),
    dict(trim="tails", dereplicate="none", disable_preorient=True, disable_prefilter=True),
    # k = 4 selects the 9-row padded bit-sliced scan (demux_kernel<u32,256,2> / bitsliced_shw_pad<4,13>), k = 6, 7 the
    # generic banded one (bitsliced_shw<8>); search_len 50 / 81 / 1..3 leave the unrolled primer loop ((S & 3) != 0)
    # and the 16-byte encode path ((S & 15) != 0)
    dict(index_edit_distance=4), dict(index_edit_distance=6), dict(index_edit_distance=7),
    dict(index_edit_distance=4, trim="tails"), dict(index_edit_distance=7, trim="primers", dereplicate="none"),
    dict(search_len=50), dict(search_len=81), dict(search_len=81, trim="tails"), dict(search_len=3), dict(search_len=2),
    dict(search_len=1),
]


@pytest.mark.parametrize("flags", FLAG_SETS, ids=lambda f: ",".join(f"{k}={v}" for k, v in f.items()))
def test_flag_matrix(lib, c2, flags):
    from specimux_amd import synth
    pan, (pf, sf) = c2
    S_ = flags.get("search_len", 80)
    rs = synth.make_reads(pan, 400, 77, search_len=S_, windows_only=False)
    reads = reads_from_set(rs, range(400), S_)
    both = Both(pf, sf, **flags)
    both.assert_hits_equal(reads[:60], str(flags))
    got = both.assert_ops_equal(reads, str(flags))
    if flags.get("index_edit_distance", 0) >= 6 and flags.get("dereplicate", "best") == "best":
        # k ~ half the barcode length: a few reads tie with dozens of barcodes and the reference writes one record per
        # tied specimen (demultiplex.py:434-436, Q9) -- far more than 16 per read.  The kernel keeps one running trim shift
        # per candidate instead of a bounded emission log, so those reads come out record for record like any other.
        from collections import Counter
        assert max(Counter(k[0] for k in got).values()) > 16
    # the golden reads too (real ONT data, gITS7 has IUPAC R)
    gold = Both(P, S, **flags)
    gold.assert_ops_equal(golden_reads("sequences.fastq"), f"golden {flags}")


# ------------------------------------------------------------------ edge cases the domain has
def _edge_reads(pan):
    """Hand-built reads around the reference's quirks."""
    from specimux_amd.synth import ITS1F, ITS4, revcomp
    f0, f1, r0 = pan.fwd[0], pan.fwd[1], pan.rev[0]
    ins = "ACGGTTCAGGCTAACGTTAGC" * 12
    good = "TTTT" + f0 + ITS1F + ins + revcomp(ITS4) + revcomp(r0) + "GGGG"
    reads = [
        ("empty", ""), ("one", "A"), ("short_noprimer", "ACGT" * 10),
        ("good", good), ("good_rc", revcomp(good)),
        ("lower", good.lower()),                                   # Q15: reads are not upper-cased
        ("primer_at_very_end", ins + revcomp(ITS4)),               # Q2: empty barcode target
        ("primer_at_very_start", ITS1F + ins),
        ("n_in_barcode", "TT" + f0[:4] + "N" + f0[5:] + ITS1F + ins + revcomp(ITS4) + revcomp(r0)),   # Q7
        ("n_in_primer", f0 + ITS1F[:6] + "N" + ITS1F[7:] + ins + revcomp(ITS4) + revcomp(r0)),
        ("iupac_read", f0 + ITS1F[:3] + "R" + ITS1F[4:] + ins + revcomp(ITS4) + revcomp(r0)),
        ("both_ends_rev_primer", ITS4 + ins + revcomp(ITS4) + revcomp(r0)),      # ambiguous orientation
        ("both_ends_fwd_primer", f0 + ITS1F + ins + revcomp(ITS1F) + revcomp(f1)),
        ("chimera_two_fwd_barcodes", f0 + ITS1F + ins[:40] + f1 + ITS1F + ins + revcomp(ITS4) + revcomp(r0)),
        ("u_base", "U" + good[1:]),
    ]
    # Q1: reads shorter than search_len that DO contain primer + barcode (every length 20..82)
    core = f0 + ITS1F
    for L in range(20, 83):
        s = (core + ins)[:L]
        reads.append((f"shortF{L}", s))
        reads.append((f"shortR{L}", revcomp(s)))
        t = (ins[: max(0, L - len(core) - 3)] + revcomp(ITS4) + revcomp(r0) + "ACG")[-L:]
        reads.append((f"shortT{L}", t))
    # tied barcodes: a target equidistant from two barcodes (Q9) -- mutate f0 towards f1 half way
    diff = [i for i in range(len(f0)) if f0[i] != f1[i]]
    for cut in range(1, len(diff)):
        mid = list(f0)
        for i in diff[:cut]:
            mid[i] = f1[i]
        reads.append((f"tie{cut}", "".join(mid) + ITS1F + ins + revcomp(ITS4) + revcomp(r0)))
        reads.append((f"tie{cut}_nob2", "".join(mid) + ITS1F + ins + revcomp(ITS4)))
    rng = np.random.default_rng(5)
    return [(rid, s, (rng.integers(3, 41, len(s)) + 33).astype(np.uint8).tobytes().decode()) for rid, s in reads]


@pytest.mark.parametrize("flags", [dict(), dict(trim="tails"), dict(trim="primers"), dict(dereplicate="none"),
                                   dict(disable_prefilter=True), dict(disable_preorient=True, index_edit_distance=5)],
                         ids=lambda f: ",".join(f"{k}={v}" for k, v in f.items()) or "default")
def test_edge_cases(lib, c2, flags):
    pan, (pf, sf) = c2
    reads = [r for r in _edge_reads(pan) if r[0] != "u_base"]
    both = Both(pf, sf, **flags)
    both.assert_hits_equal(reads, f"edge {flags}")
    both.assert_ops_equal(reads, f"edge {flags}")


def test_length_filter_emits_nothing(lib, c2):
    pan, (pf, sf) = c2
    reads = _edge_reads(pan)[:8]
    both = Both(pf, sf, min_length=50)
    got = both.assert_ops_equal(reads, "minlen")
    assert not any(k[0] in ("empty", "one", "short_noprimer") for k in got)


# ------------------------------------------------------------------ trace TSV (-d): events == oracle's restatement
@pytest.mark.parametrize("level", [1, 2, 3])
def test_trace_events_golden(lib, tmp_path, level):
    """Every event row (worker, sequence number, sequence id, type, fields) of the 40 golden reads, in order.
    The reference holds no trace fixture (its expected_output/trace files are empty placeholders): parity of the
    event stream is against the oracle's line-by-line restatement of trace.py / demultiplex.py ("unpinned")."""
    both = Both(P, S)
    n = both.assert_trace_equal(golden_reads("sequences.fastq"), level, tmp_path / f"g{level}", "golden")
    assert n >= {1: 500, 2: 650, 3: 1300}[level]


@pytest.mark.parametrize("flags", [dict(), dict(trim="tails"), dict(trim="primers"), dict(trim="none"),
                                   dict(dereplicate="none"), dict(disable_prefilter=True),
                                   dict(disable_preorient=True, index_edit_distance=5), dict(min_length=60)],
                         ids=lambda f: ",".join(f"{k}={v}" for k, v in f.items()) or "default")
def test_trace_events_flags_and_edge_cases(lib, c2, tmp_path, flags):
    from specimux_amd import synth
    pan, (pf, sf) = c2
    rs = synth.make_reads(pan, 150, 99, windows_only=False)
    reads = reads_from_set(rs, range(150), 80) + [r for r in _edge_reads(pan) if r[0] != "u_base"]
    both = Both(pf, sf, **flags)
    both.assert_trace_equal(reads, 2, tmp_path / "l2", f"edge {flags}")
    both.assert_trace_equal(reads[:40] + reads[150:190], 3, tmp_path / "l3", f"edge {flags}")


def test_trace_multi_pool_panel(lib, c3, tmp_path):
    from specimux_amd import synth
    pan, (pf, sf) = c3
    rs = synth.make_reads(pan, 120, 3003, windows_only=False)
    both = Both(pf, sf)
    both.assert_trace_equal(reads_from_set(rs, range(120), 80), 2, tmp_path / "c3", "c3")


@pytest.mark.parametrize("flags", [dict(), dict(trim="primers"), dict(trim="tails"), dict(trim="none")],
                         ids=lambda f: ",".join(f"{k}={v}" for k, v in f.items()) or "default")
def test_color_locations(lib, c2, flags):
    """--color: the primer / barcode locations painted on the (trimmed, oriented) record."""
    from specimux_amd import synth
    from specimux_amd.alignment import color_sequence
    pan, (pf, sf) = c2
    rs = synth.make_reads(pan, 120, 41, windows_only=False)
    reads = reads_from_set(rs, range(120), 80) + [r for r in _edge_reads(pan) if r[0] != "u_base"][:60]
    assert Both(pf, sf, **flags).assert_locations_equal(reads, f"c2 {flags}") > 200
    assert Both(P, S, **flags).assert_locations_equal(golden_reads("sequences.fastq"), f"golden {flags}") > 40
    painted = color_sequence("ACGTACGT", [40, 5, 40, 40, 40, 40, 5, 40], (1, 2), None, (0, 0), (6, 9))
    assert painted == "\033[0;34mA\033[0m\033[0;32mc\033[0m\033[0;32mG\033[0mTAC\033[0;34mg\033[0m\033[0;34mT\033[0m"


def test_cli_stdout_color(lib, capsys):
    """`specimux primers specimens reads -n 8 --color` (stdout mode, orchestration.py:458-545): four lines per
    record, header "<id> <distance code> <sample>", the matched regions wrapped in ANSI colours."""
    from specimux_amd import cli
    cli.main(["specimux", P, S, f"{GOLDEN}/sequences.fastq", "-n", "8", "--color"])
    out = capsys.readouterr().out.splitlines()
    assert len(out) == 8 * 4 and all(line.startswith("@") for line in out[0::4])
    assert out[4].split()[2] == "TEST_SPECIMEN_001"            # second golden read is a full match
    assert "\033[0;32m" in out[5] and "\033[0;34m" not in out[5]   # trim=barcodes keeps the primers, cuts the barcodes


# ------------------------------------------------------------------ BASELINE-size properties
def test_full_size_properties_765k(lib, c2):
    """configs[1]: 768 specimens, 765k reads.  Size-independent properties + an oracle spot check."""
    from specimux_amd import _lib, synth
    from specimux_amd.demultiplex import compiled_panel
    pan, (pf, sf) = c2
    both = Both(pf, sf)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    n = 765000
    rs = synth.make_reads(pan, n, 2002)
    windows = rs.windows(cp.window_stride)
    ops, extra, counts = cp.run(windows, rs.lens)
    # (1) counter consistency: totals, one primary record per read, specimen counters == full records
    assert counts[_lib.CNT_TOTAL] == n and counts[_lib.CNT_FILTERED] == 0 and counts[_lib.CNT_OVERFLOW] == 0
    allops = np.concatenate([ops, extra])
    assert len(allops) == counts[_lib.CNT_OPS_FULL] + counts[_lib.CNT_OPS_PARTIAL] + counts[_lib.CNT_OPS_UNKNOWN]
    assert int(ops["n_ops"].sum()) == len(allops)
    full = allops[(allops["rtype"] == _lib.R_DEREP_FULL) & ((allops["flags"] & _lib.OPF_TRIM_EMPTY) == 0)]
    assert np.array_equal(np.bincount(full["sample"], minlength=768), counts[_lib.CNT_SPECIMEN0:].astype(np.int64))
    assert counts[_lib.CNT_SPECIMEN0:].sum() == counts[_lib.CNT_OPS_FULL]
    matched_reads = np.unique(np.concatenate([ops["read"][ops["rtype"] == _lib.R_DEREP_FULL],
                                              extra["read"][extra["rtype"] == _lib.R_DEREP_FULL]]))
    assert len(matched_reads) == counts[_lib.CNT_MATCHED]
    assert 0.5 < counts[_lib.CNT_MATCHED] / n < 0.9
    # (2) determinism / idempotence
    ops2, extra2, counts2 = cp.run(windows, rs.lens)
    assert np.array_equal(ops, ops2) and np.array_equal(counts, counts2)
    # (3) strand symmetry: reverse-complementing every read keeps each read's set of specimens
    comp = np.zeros(256, dtype=np.uint8)
    comp[[65, 67, 71, 84]] = (84, 71, 67, 65)
    Sp = np.minimum(rs.lens, 80)
    j = np.arange(80)[None, :]
    src = np.clip(Sp[:, None] - 1 - j, 0, 79)
    rc = synth.ReadSet()
    rc.lens = rs.lens
    rc.head = comp[np.take_along_axis(rs.tail, src, axis=1)]
    rc.tail = comp[np.take_along_axis(rs.head, src, axis=1)]
    rc.head[j >= Sp[:, None]] = 0
    rc.tail[j >= Sp[:, None]] = 0
    ops_rc, extra_rc, counts_rc = cp.run(rc.windows(cp.window_stride), rc.lens)
    assert np.array_equal(counts[_lib.CNT_SPECIMEN0:], counts_rc[_lib.CNT_SPECIMEN0:])
    assert counts[_lib.CNT_MATCHED] == counts_rc[_lib.CNT_MATCHED]
    single = (ops["n_ops"] == 1) & (ops_rc["n_ops"] == 1)
    assert np.array_equal(ops["sample"][single], ops_rc["sample"][single])
    assert np.array_equal(ops["dist"][single], ops_rc["dist"][single])
    # (4) oracle spot check on a seeded sample, every category represented
    rng = np.random.default_rng(99)
    idx = np.sort(rng.choice(n, 1200, replace=False))
    reads = reads_from_set(rs, idx, 80)
    both.assert_ops_equal(reads, "765k sample")


def _full_size_properties(cp, both, rs, S, n_specimens, label, sample=1500):
    """Size-independent properties of one big batch + an oracle spot check (shared by the full-size config tests)."""
    from specimux_amd import _lib
    n = len(rs.lens)
    windows = rs.windows(cp.window_stride)
    ops, extra, counts = cp.run(windows, rs.lens)
    assert counts[_lib.CNT_TOTAL] == n and counts[_lib.CNT_OVERFLOW] == 0
    allops = np.concatenate([ops, extra])
    assert len(allops) == counts[_lib.CNT_OPS_FULL] + counts[_lib.CNT_OPS_PARTIAL] + counts[_lib.CNT_OPS_UNKNOWN]
    assert int(ops["n_ops"].sum()) == len(allops)
    full = allops[(allops["rtype"] == _lib.R_DEREP_FULL) & ((allops["flags"] & _lib.OPF_TRIM_EMPTY) == 0)]
    assert np.array_equal(np.bincount(full["sample"], minlength=n_specimens), counts[_lib.CNT_SPECIMEN0:].astype(np.int64))
    assert counts[_lib.CNT_SPECIMEN0:].sum() == counts[_lib.CNT_OPS_FULL]
    ops2, extra2, counts2 = cp.run(windows, rs.lens)                       # determinism / idempotence
    assert np.array_equal(ops, ops2) and np.array_equal(counts, counts2)
    rng = np.random.default_rng(99)
    idx = np.sort(rng.choice(n, sample, replace=False))
    both.assert_ops_equal(reads_from_set(rs, idx, S), f"{label} sample")
    return counts


def test_full_size_c3_5M(lib, c3):
    """configs[2]: 3072 specimens over 4 pools (shared reverse primer, degenerate primers), 5 M reads in one batch on one
    GPU: counter consistency, idempotence, and 1 500 seeded reads against the oracle record by record."""
    from specimux_amd import _lib, synth
    from specimux_amd.demultiplex import compiled_panel
    pan, (pf, sf) = c3
    both = Both(pf, sf)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    rs = synth.make_reads(pan, 5_000_000, 3003, workers=16, insert_mean=900, insert_sd=250)
    counts = _full_size_properties(cp, both, rs, 80, 3072, "c3 5M")
    assert 0.5 < counts[_lib.CNT_MATCHED] / 5_000_000 < 0.9
    per_pool = counts[_lib.CNT_SPECIMEN0:].reshape(4, 768).sum(axis=1)
    assert per_pool.min() > 0.15 * per_pool.sum()                          # all four pools demultiplex


def test_full_size_c5_stress_shape(lib, c3):
    """configs[4]-shaped on one GPU: the 3072-specimen panel with degenerate primers, -l 160, 15 % error reads of 2-5 kb,
    2 % of the reads with N inside the end windows: 1 M reads, properties + oracle sample."""
    from specimux_amd import _lib, synth
    from specimux_amd.demultiplex import compiled_panel
    pan, (pf, sf) = c3
    both = Both(pf, sf, search_len=160)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    rs = synth.make_reads(pan, 1_000_000, 5005, workers=16, search_len=160, error_rate=0.15, insert_mean=3500, insert_sd=800,
                          insert_min=2000, insert_max=5000, n_frac=0.02)
    has_n = ((rs.head == 78) | (rs.tail == 78)).any(axis=1)
    assert 0.02 < has_n.mean() < 0.06 and 2000 <= np.percentile(rs.lens, 10) and np.percentile(rs.lens, 90) <= 5600
    counts = _full_size_properties(cp, both, rs, 160, 3072, "c5 1M", sample=800)
    assert counts[_lib.CNT_MATCHED] / 1_000_000 > 0.2
    # the N reads take the scalar primer scan (prescan fallback queue): a sample of them alone against the oracle
    idx = np.nonzero(has_n)[0][:300]
    both.assert_ops_equal(reads_from_set(rs, idx, 160), "c5 N reads")


def test_counts_allreduce_single_rank(lib):
    """RCCL wrapper of the C ABI with a world of one (the multi-rank path is exercised by bench.py --gpus N)."""
    import torch
    from specimux_amd import _lib
    uid = (C.c_uint8 * 128)()
    _lib.check(lib.smx_comm_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(lib.smx_comm_init(uid, 1, 0, C.byref(comm)))
    t = torch.arange(20, dtype=torch.int64, device="cuda")
    _lib.check(lib.smx_counts_allreduce(C.c_void_p(t.data_ptr()), 20, comm, None))
    torch.cuda.synchronize()
    assert t.cpu().tolist() == list(range(20))
    lib.smx_comm_destroy(comm)


# ------------------------------------------------------------------ end to end through the CLI (native reader/writer)
def _oracle_tree(pf, sf, seqfile, **kw):
    tree, total, matched = __import__("oracle.specimux_oracle", fromlist=["x"]).run_files(pf, sf, seqfile, **kw)
    return {k: sorted(v) for k, v in tree.items()}, total, matched


@pytest.mark.parametrize("variant", ["plain", "gz", "window", "fasta", "tails_prefix"])
def test_cli_end_to_end_synthetic(lib, c2, tmp_path, variant):
    import gzip
    import shutil
    from specimux_amd import cli, synth
    pan, (pf, sf) = c2
    rs = synth.make_reads(pan, 1200, 4242, windows_only=False)
    fq = tmp_path / "reads.fastq"
    rs.write_fastq(str(fq))
    seqfile, extra_args, okw = str(fq), [], {}
    if variant == "gz":
        seqfile = str(tmp_path / "reads.fastq.gz")
        with open(fq, "rb") as a, gzip.open(seqfile, "wb") as b:
            shutil.copyfileobj(a, b)
    elif variant == "window":
        extra_args = ["-n", "101,500"]
    elif variant == "fasta":
        seqfile = str(tmp_path / "reads.fasta")
        with open(seqfile, "w") as fh:
            for i, s in enumerate(rs.reads):
                fh.write(f">read{i:07d} synthetic\n{s[:70]}\n{s[70:]}\n")
    elif variant == "tails_prefix":
        extra_args = ["--trim", "tails", "-P", "x_"]
        okw = {"trim": "tails"}
    out = tmp_path / "out"
    cli.main(["specimux", pf, sf, seqfile, "-F", "-O", str(out)] + extra_args)
    got = read_expected_tree(str(out)) if variant != "fasta" else None
    exp, total, matched = _oracle_tree(pf, sf, seqfile, **okw)
    if variant == "window":
        from oracle import specimux_oracle as O
        panel = O.load_panel(pf, sf)
        recs, _ = O.read_sequences(seqfile)
        ops, total, matched = O.process_sequences(recs[100:600], O.setup_params(panel), panel)
        exp = {}
        for op in ops:
            for p in O.op_path(op):
                exp.setdefault(p, []).append(O.op_record(op))
        exp = {k: sorted(v) for k, v in exp.items()}
    if variant == "tails_prefix":
        exp = {"/".join(k.split("/")[:-1] + ["x_" + k.split("/")[-1]]): v for k, v in exp.items()}
    if variant == "fasta":
        import os
        got = {}
        for dirpath, _d, files in os.walk(out):
            for fn in files:
                if fn.endswith(".fasta") and not fn.startswith("primers"):
                    lines = open(os.path.join(dirpath, fn)).read().split("\n")
                    got[os.path.relpath(os.path.join(dirpath, fn), out)] = sorted(
                        "\n".join(lines[i:i + 2]) + "\n" for i in range(0, len(lines) - 1, 2))
    assert got == exp
    log = (out / "log.txt").read_text()
    assert f"Processed {total:,} sequences, match rate: {matched / total:.1%}" in log


@pytest.mark.parametrize("mode", ["append", "merge"])
def test_cli_two_rank_launch_equals_single_process(lib, c2, tmp_path, mode):
    """The product multi-GPU path for files, rehearsed with two ranks on this one GPU (gloo for the counts, since RCCL
    refuses two ranks per device): `python -m torch.distributed.run --nproc-per-node 2 -m specimux_amd.cli ... -F`.
    Byte-range shards; default: both ranks append to the one tree (same records in every file as the single-process run,
    none torn, order free as in the reference's worker pool); SMX_RANK_MERGE=1: per-rank trees, merged -- the tree is
    identical, byte for byte, to the single-process run."""
    import os
    import subprocess
    import sys
    from specimux_amd import cli, synth
    from conftest import REPO
    pan, (pf, sf) = c2
    rs = synth.make_reads(pan, 3000, 515, windows_only=False)
    fq = tmp_path / "reads.fastq"
    rs.write_fastq(str(fq))
    one = tmp_path / "one"
    cli.main(["specimux", pf, sf, str(fq), "-F", "-O", str(one)])
    two = tmp_path / "two"
    env = dict(os.environ, SMX_DIST_BACKEND="gloo", PYTHONPATH=REPO)
    env.pop("SMX_RANK_MERGE", None)
    if mode == "merge":
        env["SMX_RANK_MERGE"] = "1"
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                    "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300), "-m", "specimux_amd.cli", pf, sf, str(fq),
                    "-F", "-O", str(two)], check=True, env=env, timeout=600, cwd=REPO)

    def tree(root):
        out = {}
        for dirpath, _d, files in os.walk(root):
            for fn in files:
                if fn != "log.txt":
                    text = open(os.path.join(dirpath, fn)).read()
                    out[os.path.relpath(os.path.join(dirpath, fn), root)] = text if mode == "merge" else sorted(text.split("@read"))
        return out
    a, b = tree(one), tree(two)
    assert a == b and sum(1 for k in a if k.startswith("full/")) > 100
    assert not any(n.startswith(".smx_rank_") for n in os.listdir(two))
    log = (two / "log.txt").read_text()
    assert "Demultiplexed on 2 GPUs" in log and "Processed 3,000 sequences" in log


# ------------------------------------------------------------------ panel shapes that select other kernel paths
def _custom_panel(tmp_path_factory, name, n_fwd, n_rev, bc_len, min_dist, fwd_primer=None, mixed=False, seed=7, rev_primer=None):
    from specimux_amd import synth
    f, r = synth.make_barcodes(n_fwd, n_rev, length=bc_len, min_dist=min_dist, seed=seed)
    pools = [("ITS", "FWD", fwd_primer or synth.ITS1F, "ITS4", rev_primer or synth.ITS4)]
    pan = synth.Panel(pools, f, r)          # the reads are generated from the uniform panel
    files = pan
    if mixed:   # the panel FILE gets one shorter forward barcode: mixed lengths (the prefilter must then be off)
        files = synth.Panel(pools, [f[0][:-2]] + f[1:], r)
    return pan, tmp_panel(tmp_path_factory, files, name)


# primer lengths that select the other instantiations of the prescan DP kernel (prescan_dp_kernel<rows, extra symbols, *>,
# smx_prescan.hip SMX_PRE_VARIANTS): rows 22 (<= 22 nt), 24 (23-24 nt), 31 (25-31 nt) x all-ACGT / degenerate letters
PRIMER_SHAPES = {
    "24nt_acgt_primer": "CTTGGTCATTTAGAGGAAGTAAAA",              # -> <24, 0>
    "28nt_acgt_primer": "CTTGGTCATTTAGAGGAAGTAAAAGTCG",          # -> <31, 0>
    "31nt_acgt_primer": "CTTGGTCATTTAGAGGAAGTAAAAGTCGTAA",       # -> <31, 0>, every row live
    "28nt_degenerate_primer": "CTTGGTCATYTAGAGGARGTAAAAGTCG",    # -> <31, 4>
    "20nt_degenerate_primer": "GAYGAYMGWGATCAYTTYGG",            # -> <22, 4>
}
# the same with the degenerate primer SECOND in the panel (primer index 1): the DP kernel reads its letter sets from the
# kernel-argument struct at a primer-dependent offset, and a scalar load with a misaligned base register silently fetches
# another primer's (seen in a variant of the kernel for every primer index that is not a multiple of 4)
REVERSE_PRIMER_SHAPES = {
    "20nt_degenerate_reverse_primer": "TCCTCCGCTTATTGATRTGY",          # -> <22, 4>, degenerate letters at primer 1
    "26nt_degenerate_reverse_primer": "TCCTCCGCTTATTGATATGCRYAAGT",    # -> <31, 4>, degenerate letters at primer 1
}


@pytest.mark.parametrize("shape", ["96x4_multiword", "24nt_barcodes", "40nt_primer_64bit", "mixed_lengths",
                                   "8nt_barcodes", "10nt_barcodes", "16nt_barcodes", "16nt_barcodes_k4",
                                   "16nt_barcodes_k6", "8nt_barcodes_k4", "96x4_multiword_k5", "40nt_primer_64bit_k5"]
                         + sorted(PRIMER_SHAPES) + sorted(REVERSE_PRIMER_SHAPES))
def test_panel_shapes(lib, tmp_path_factory, monkeypatch, shape):
    from specimux_amd import synth
    flags = {}
    name = shape
    if "_k" in shape:   # bitsliced_shw_pad<4,16> (16 nt, k = 4), the generic scan at other heights / word counts
        flags = dict(index_edit_distance=int(shape.rsplit("_k", 1)[1]))
        shape = shape.rsplit("_k", 1)[0]
    if shape == "96x4_multiword":      # > 64 barcodes on one primer: 3-word tie masks, G = 128 slots
        pan, (pf, sf) = _custom_panel(tmp_path_factory, shape, 96, 4, 13, 5)
    elif shape == "24nt_barcodes":     # longer than the bit-sliced path's 16 rows: per-barcode lean path, larger k
        pan, (pf, sf) = _custom_panel(tmp_path_factory, shape, 12, 8, 24, 10)
        flags = dict(disable_prefilter=True)
    elif shape == "40nt_primer_64bit":  # primer longer than 32 nt: 64-bit primer words
        pan, (pf, sf) = _custom_panel(tmp_path_factory, shape, 8, 6, 13, 6,
                                      fwd_primer="CTTGGTCATTTAGAGGAAGTAAAAGTCGTAACAAGGTTTCC")
    elif shape in PRIMER_SHAPES:
        pan, (pf, sf) = _custom_panel(tmp_path_factory, shape, 8, 6, 13, 6, fwd_primer=PRIMER_SHAPES[shape])
    elif shape in REVERSE_PRIMER_SHAPES:
        pan, (pf, sf) = _custom_panel(tmp_path_factory, shape, 8, 6, 13, 6, rev_primer=REVERSE_PRIMER_SHAPES[shape])
    elif shape.endswith("nt_barcodes"):   # the padded bit-sliced scan's other heights: M = 8, 12, 16 rows
        n = int(shape.split("nt")[0])
        pan, (pf, sf) = _custom_panel(tmp_path_factory, shape, 8, 6, n, {8: 4, 10: 5, 16: 7}[n])
    else:
        pan, (pf, sf) = _custom_panel(tmp_path_factory, shape, 8, 6, 13, 6, mixed=True)
        flags = dict(disable_prefilter=True)
    rs = synth.make_reads(pan, 500, 99, windows_only=False)
    reads = reads_from_set(rs, range(500), 80)
    for fl in (flags, dict(flags, trim="tails")):
        both = Both(pf, sf, **fl)
        both.assert_hits_equal(reads[:80], f"{name} {fl}")
        got = both.assert_ops_equal(reads, f"{name} {fl}")
        assert sum(1 for k in got if k[6] == "DEREP") > 100
    if name in PRIMER_SHAPES or name in REVERSE_PRIMER_SHAPES:
        # the same DP kernel with the match words compiled in (<rows, symbols, 1>: what panels with compact demux tiles run),
        # compact tiles forced on, with a capacity that sends part of the tiles through the overflow list; short reads too
        monkeypatch.setenv("SMX_COMPACT_ITEMS", "120")
        short = synth.make_reads(pan, 100, 98, insert_mean=40, insert_sd=30, windows_only=False)
        reads2 = reads[:300] + reads_from_set(short, range(100), 80, prefix="s")
        both = Both(pf, sf, **flags)
        both.assert_hits_equal(reads2[:60] + reads2[-40:], f"{name} compact", lean=True)
        both.assert_ops_equal(reads2, f"{name} compact")


def test_forced_small_barcode_rounds(c2, monkeypatch):
    """SMX_TEST_CAPS shrinks the per-round capacities (hits, entries) so that every tile needs many barcode
    rounds: the multi-round bookkeeping must give the same records as one round."""
    from specimux_amd import synth
    pan, (pf, sf) = c2
    rs = synth.make_reads(pan, 700, 31, windows_only=False)
    reads = reads_from_set(rs, range(700), 80) + [r for r in _edge_reads(pan) if r[0] != "u_base"]
    monkeypatch.setenv("SMX_TEST_CAPS", "5,80")
    for fl in (dict(), dict(trim="tails"), dict(disable_preorient=True)):
        both = Both(pf, sf, **fl)
        both.assert_hits_equal(reads[:100], f"caps {fl}")
        both.assert_ops_equal(reads, f"caps {fl}")


@pytest.mark.parametrize("env", [dict(), dict(SMX_COMPACT_ITEMS="40"), dict(SMX_COMPACT_ITEMS="16", SMX_COMPACT_R="24"),
                                 dict(SMX_COMPACT="0"), dict(SMX_NO_TABLE_SHARING="1")],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()) or "default")
def test_compact_tiles_many_primer_panel(lib, c3, monkeypatch, env):
    """The lean kernel's compact mode (panels with many primers: per-alignment records only for the alignments the
    prescan's match words flag) and the dense redo launch behind it.  Default sizing, a record capacity small enough
    that most tiles overflow into the redo launch, a tile size that is not a power of two, compact mode off, and one
    barcode table per primer instead of one per distinct barcode list (this panel's eight primers share two): the
    same records and hit tables as the oracle every time -- including reads with N (not covered by the prescan: every
    alignment of such a read takes a record) and short reads."""
    from specimux_amd import synth
    pan, (pf, sf) = c3
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rs = synth.make_reads(pan, 1800, 909, insert_mean=900, insert_sd=250, windows_only=False, n_frac=0.04)
    reads = reads_from_set(rs, range(1800), 80)
    short = synth.make_reads(pan, 120, 910, insert_mean=40, insert_sd=30, windows_only=False)
    reads += reads_from_set(short, range(120), 80, prefix="s")
    for fl in (dict(), dict(trim="tails"), dict(dereplicate="none", trim="primers")):
        both = Both(pf, sf, **fl)
        both.assert_hits_equal(reads[:200] + reads[-60:], f"compact {env} {fl}", lean=True)
        both.assert_ops_equal(reads, f"compact {env} {fl}")


def test_compact_tiles_forced_on_two_primer_panel(lib, c2, monkeypatch):
    """Compact mode forced onto the 2-primer panel (where the default never uses it), with the edge-case reads."""
    from specimux_amd import synth
    pan, (pf, sf) = c2
    monkeypatch.setenv("SMX_COMPACT_ITEMS", "96")
    rs = synth.make_reads(pan, 900, 77, windows_only=False)
    reads = reads_from_set(rs, range(900), 80) + [r for r in _edge_reads(pan) if r[0] != "u_base"]
    for fl in (dict(), dict(disable_preorient=True)):
        both = Both(pf, sf, **fl)
        both.assert_hits_equal(reads[:150], f"compact c2 {fl}", lean=True)
        both.assert_ops_equal(reads, f"compact c2 {fl}")


@pytest.mark.parametrize("env", [dict(SMX_NO_SPECIALISE="1"), dict(SMX_NO_SPECIALISE_NP="1")], ids=lambda e: ",".join(e))
def test_generic_kernel_on_default_flags(lib, c2, c3, monkeypatch, env):
    """The reference's default flags normally run the specialised instantiations of the lean kernel (search_len 80, index
    distance 3, trim barcodes, ... as compile-time constants; a further one for two-primer panels).  With the
    specialisation switched off the generic instantiation serves the same panels: same records, same hit tables."""
    from specimux_amd import synth
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name, (pan, (pf, sf)), S, kw in (("c2", c2, 80, {}), ("c3", c3, 80, dict(insert_mean=900, insert_sd=250)),
                                       ("c5", c3, 160, dict(error_rate=0.15))):   # (the -l 160 shape has its own instantiation)
        rs = synth.make_reads(pan, 1200 if S == 80 else 500, 4242, search_len=S, windows_only=False, **kw)
        reads = reads_from_set(rs, range(len(rs.lens)), S) + ([r for r in _edge_reads(pan) if r[0] != "u_base"] if name == "c2" else [])
        both = Both(pf, sf, **({} if S == 80 else dict(search_len=S)))
        both.assert_hits_equal(reads[:150], f"generic {env} {name}", lean=True)
        both.assert_ops_equal(reads, f"generic {env} {name}")


@pytest.mark.parametrize("shape", ["mixed_lengths", "24nt_barcodes", "96x4_multiword", "16nt_barcodes_k5"])
def test_compact_tiles_other_scan_variants(lib, tmp_path_factory, monkeypatch, shape):
    """Compact tiles under the barcode-scan variants the default panels do not reach: the per-barcode lean scan (mixed
    lengths, 24-nt barcodes), three tie-mask words per hit, the k 4..7 bit-sliced scan -- forced on, with a record
    capacity that every tile fits and one that sends most tiles through the overflow list and the redo launch."""
    from specimux_amd import synth
    flags = {}
    if shape == "mixed_lengths":
        pan, (pf, sf) = _custom_panel(tmp_path_factory, "cm_" + shape, 8, 6, 13, 6, mixed=True)
        flags = dict(disable_prefilter=True)
    elif shape == "24nt_barcodes":
        pan, (pf, sf) = _custom_panel(tmp_path_factory, "cm_" + shape, 12, 8, 24, 10)
        flags = dict(disable_prefilter=True)
    elif shape == "96x4_multiword":
        pan, (pf, sf) = _custom_panel(tmp_path_factory, "cm_" + shape, 96, 4, 13, 5)
    else:
        pan, (pf, sf) = _custom_panel(tmp_path_factory, "cm_" + shape, 8, 6, 16, 7)
        flags = dict(index_edit_distance=5)
    rs = synth.make_reads(pan, 600, 123, windows_only=False, n_frac=0.03)
    reads = reads_from_set(rs, range(600), 80)
    for items in ("176", "120"):   # every tile fits / most tiles go through the overflow list (measured with SMX_DEBUG_OVERFLOW)
        monkeypatch.setenv("SMX_COMPACT_ITEMS", items)
        for fl in (flags, dict(flags, trim="primers")):
            both = Both(pf, sf, **fl)
            both.assert_hits_equal(reads[:100], f"compact {shape} {fl} {items}", lean=True)
            got = both.assert_ops_equal(reads, f"compact {shape} {fl} {items}")
            assert sum(1 for k in got if k[6] == "DEREP") > 100


# ------------------------------------------------------------------ randomised evidence inside the suite
from parity_utils import FUZZ_FLAG_SETS  # noqa: E402  (shared with the manual loop, tests/fuzz_parity.py)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_bounded_fuzz(lib, c1, c2, c3, seed):
    """The manual fuzz loop (tests/fuzz_parity.py), bounded: fixed seeds, the three panels in rotation over 14 flag sets
    (search lengths on both sides of the prescan's multiple-of-16 rule, N reads, high error rates, every trim / dereplicate
    mode), records of every read and hit tables of a sample against the oracle."""
    from specimux_amd import synth
    panels = {"c1": c1, "c2": c2, "c3": c3}
    checked = 0
    for fi, flags in enumerate(FUZZ_FLAG_SETS):
        name = ("c2", "c3", "c1")[(seed + fi) % 3]
        pan, (pf, sf) = panels[name]
        flags = dict(flags)
        gen = {k: flags.pop(k) for k in ("error_rate", "n_frac") if k in flags}   # generator-only knobs
        S_ = flags.get("search_len", 80)
        rs = synth.make_reads(pan, 400, 7000 + 131 * seed + fi, search_len=S_, windows_only=False, **gen)
        reads = reads_from_set(rs, range(400), S_)
        both = Both(pf, sf, **flags)
        both.assert_hits_equal(reads[:40], f"fuzz {name} seed {seed} {flags}")
        both.assert_ops_equal(reads, f"fuzz {name} seed {seed} {flags}")
        checked += len(reads)
    assert checked == 400 * len(FUZZ_FLAG_SETS)


# ------------------------------------------------------------------ the 8-GPU configs at per-rank shard size, one shard after another
def _shards_on_one_gpu(cp, both, pan, n_shards, reads_per_shard, seed0, S_, n_specimens, label, sample, **gen):
    """What rank r of an 8-GPU run does, for r = 0..7 in turn on this GPU: its own shard (seed0 + r, the bench's
    shard_seed rule), properties + an oracle sample per shard; the counts vectors summed like the all-reduce would."""
    from specimux_amd import _lib, synth
    from specimux_amd.distributed import shard_seed
    total = np.zeros(cp.counts_len, dtype=np.uint64)
    for r in range(n_shards):
        rs = synth.make_reads(pan, reads_per_shard, shard_seed(seed0, r), workers=16, search_len=S_, **gen)
        total += _full_size_properties(cp, both, rs, S_, n_specimens, f"{label} shard {r}", sample=sample)
        del rs
    assert total[_lib.CNT_TOTAL] == n_shards * reads_per_shard and total[_lib.CNT_OVERFLOW] == 0
    assert total[_lib.CNT_SPECIMEN0:].sum() == total[_lib.CNT_OPS_FULL]
    return total


def test_config4_shards_768_specimens_50M(lib, c2):
    """configs[3]: the 768-specimen panel, 50 M reads over 8 GPUs = 6.25 M reads per rank.  The eight shards run one after
    another on this one GPU (seed 4004 + rank); per shard: counter consistency, idempotence and 1 500 seeded reads against
    the oracle record by record; the summed counts vector is what the RCCL reduce would deliver."""
    from specimux_amd import _lib
    from specimux_amd.demultiplex import compiled_panel
    pan, (pf, sf) = c2
    both = Both(pf, sf)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    total = _shards_on_one_gpu(cp, both, pan, 8, 6_250_000, 4004, 80, 768, "c4", 1500)
    assert 0.5 < total[_lib.CNT_MATCHED] / 50_000_000 < 0.9
    assert total[_lib.CNT_SPECIMEN0:].min() > 0.5 * total[_lib.CNT_SPECIMEN0:].mean()   # every specimen of the grid demultiplexes


def test_config5_shards_stress_10M(lib, c3):
    """configs[4]: 3072 specimens, degenerate primers, -l 160, 15 % error reads of 2-5 kb, 10 M reads over 8 GPUs = 1.25 M
    per rank, shard after shard on this GPU (seed 5005 + rank); 600 oracle reads per shard."""
    from specimux_amd import _lib
    from specimux_amd.demultiplex import compiled_panel
    pan, (pf, sf) = c3
    both = Both(pf, sf, search_len=160)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    total = _shards_on_one_gpu(cp, both, pan, 8, 1_250_000, 5005, 160, 3072, "c5", 600, error_rate=0.15, insert_mean=3500,
                               insert_sd=800, insert_min=2000, insert_max=5000, n_frac=0.02)
    assert total[_lib.CNT_MATCHED] / 10_000_000 > 0.2


# ------------------------------------------------------------------ long-lived panels, multi-rank record path, -n under sharding
def test_panel_reused_by_many_pipeline_runs(lib, c2, tmp_path):
    """One CompiledPanel through the streaming pipeline eight times: every run creates three lanes (own streams) and
    destroys them; the panel's per-stream slots (16) must be given back, and every run must write the same tree."""
    from specimux_amd import synth
    from specimux_amd.demultiplex import compiled_panel
    from specimux_amd.pipeline import run_streaming
    pan, (pf, sf) = c2
    both = Both(pf, sf)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    rs = synth.make_reads(pan, 2000, 616, windows_only=False)
    fq = tmp_path / "reads.fastq"
    rs.write_fastq(str(fq))
    first = None
    for i in range(8):
        out = tmp_path / f"out{i}"
        total, matched, counts, _fq = run_streaming(str(fq), cp, str(out), "")
        got = (total, matched, counts.tolist(), read_expected_tree(str(out)))
        assert total == 2000
        if first is None:
            first = got
        assert got == first


def _tree_text(root, sort_records=False):
    import os
    out = {}
    for dirpath, _d, files in os.walk(root):
        for fn in files:
            if fn == "log.txt" or "trace" in os.path.relpath(dirpath, root).split(os.sep):
                continue
            text = open(os.path.join(dirpath, fn)).read()
            out[os.path.relpath(os.path.join(dirpath, fn), root)] = sorted(text.split("@read")) if sort_records else text
    return out


def _two_rank_cli(pf, sf, fq, out, extra_args, tmp_path):
    import os
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, SMX_DIST_BACKEND="gloo", PYTHONPATH=REPO)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                    "127.0.0.1", "--master-port", str(29900 + os.getpid() % 300), "-m", "specimux_amd.cli", pf, sf, str(fq),
                    "-F", "-O", str(out)] + extra_args, check=True, env=env, timeout=600, cwd=REPO)


def test_cli_two_rank_trace_run_is_rank0_only(lib, c2, tmp_path):
    """`-F -d 1` under a two-process launch: the record path is one process's job -- rank 0 runs it, rank 1 leaves -- so the
    tree equals the single-process tree (no duplicated records) and there is one trace file per batch, not two."""
    import glob
    from specimux_amd import cli, synth
    pan, (pf, sf) = c2
    rs = synth.make_reads(pan, 600, 517, windows_only=False)
    fq = tmp_path / "reads.fastq"
    rs.write_fastq(str(fq))
    one, two = tmp_path / "one", tmp_path / "two"
    cli.main(["specimux", pf, sf, str(fq), "-F", "-O", str(one), "-d", "1"])
    _two_rank_cli(pf, sf, fq, two, ["-d", "1"], tmp_path)
    assert _tree_text(one) == _tree_text(two)
    assert len(glob.glob(str(two / "trace" / "*.tsv"))) == len(glob.glob(str(one / "trace" / "*.tsv"))) == 1


def test_cli_two_rank_record_window(lib, c2, tmp_path):
    """`-n start,num` under a two-process launch (the reference's -F path takes it too, cli.py:54-68): every rank reads the
    file, applies the window and keeps every second batch; same records, file by file, as the single-process run."""
    from specimux_amd import cli, synth
    pan, (pf, sf) = c2
    rs = synth.make_reads(pan, 3000, 518, windows_only=False)
    fq = tmp_path / "reads.fastq"
    rs.write_fastq(str(fq))
    one, two = tmp_path / "one", tmp_path / "two"
    cli.main(["specimux", pf, sf, str(fq), "-F", "-O", str(one), "-n", "101,2000"])
    _two_rank_cli(pf, sf, fq, two, ["-n", "101,2000"], tmp_path)
    a, b = _tree_text(one, True), _tree_text(two, True)
    assert a == b and sum(len(v) - 1 for k, v in a.items() if k.startswith(("full/ITS/ITS1F", "partial", "unknown"))) >= 2000
    assert "Processed 2,000 sequences" in (two / "log.txt").read_text()


# ------------------------------------------------------------------ 4-bit windows across the boundary
@pytest.mark.parametrize("search_len", [80, 81, 160, 24, 3])
def test_packed_windows_unpack_on_device(lib, c2, search_len):
    """smx_pack_windows4 (host) -> smx_unpack_windows_device == smx_pack_windows for every letter of the 4-bit alphabet;
    characters outside it come back as the padding byte 0 (both are code 15 to every kernel LUT).  Fast path (search_len a
    multiple of 8) and the byte-wise path."""
    import torch
    from specimux_amd import _lib
    from specimux_amd.demultiplex import compiled_panel
    pan, (pf, sf) = c2
    both = Both(pf, sf, search_len=search_len)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    rng = np.random.default_rng(search_len)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTNRYKMSWBDHVacgtn?X", dtype=np.uint8)
    n = 5000
    lens = rng.integers(1, 3 * search_len + 6, n)
    bases = alphabet[rng.integers(0, len(alphabet), int(lens.sum()))]
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    windows, wl = cp.pack_windows(bases, off)
    ps = int(lib.smx_packed_stride(cp.handle))
    packed, pl = np.zeros((n, ps), dtype=np.uint8), np.zeros(n, dtype=np.int32)
    _lib.check(lib.smx_pack_windows4(_lib.ptr(bases), _lib.ptr(off), n, search_len, _lib.ptr(packed), _lib.ptr(pl), None))
    d_packed = torch.from_numpy(packed).cuda()
    d_windows = torch.full((n, cp.window_stride), 0x55, dtype=torch.uint8, device="cuda")
    _lib.check(lib.smx_unpack_windows_device(cp.handle, None, C.c_void_p(d_packed.data_ptr()), n, C.c_void_p(d_windows.data_ptr())))
    torch.cuda.synchronize()
    got = d_windows.cpu().numpy()
    inside = np.isin(windows, np.frombuffer(b"ACGTNRYKMSWBDHV", dtype=np.uint8))
    assert np.array_equal(got, np.where(inside, windows, 0)) and np.array_equal(pl, wl)


def test_pipeline_packed_lanes_equal_ascii_lanes(lib, c2, tmp_path, monkeypatch):
    """The streaming pipeline ships 4-bit windows by default; SMX_LANES_ASCII=1 ships 8-bit ones.  Same tree, same counts --
    also for reads with N, lower-case bases and other characters in the windows, and for a batch with a 'U' (which falls
    back to ASCII by itself); the tree equals the oracle's."""
    from specimux_amd import synth
    from specimux_amd.demultiplex import compiled_panel
    from specimux_amd.pipeline import run_streaming
    pan, (pf, sf) = c2
    both = Both(pf, sf)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
    rs = synth.make_reads(pan, 3000, 717, windows_only=False, n_frac=0.05)
    reads = list(rs.reads)
    rng = random.Random(5)
    for i in range(0, 3000, 37):   # lower case / junk inside the windows of some reads
        s = list(reads[i])
        for pos in (rng.randrange(0, 60), len(s) - 1 - rng.randrange(0, 60)):
            s[pos] = rng.choice("acgtn?X-")
        reads[i] = "".join(s)
    for with_u in (False, True):
        if with_u:
            reads[11] = reads[11][:30] + "U" + reads[11][31:]
        fq = tmp_path / f"reads{int(with_u)}.fastq"
        fq.write_text("".join(f"@read{i:07d} x\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(zip(reads, rs.quals))))
        trees = []
        for ascii_lanes in (False, True):
            if ascii_lanes:
                monkeypatch.setenv("SMX_LANES_ASCII", "1")
            else:
                monkeypatch.delenv("SMX_LANES_ASCII", raising=False)
            out = tmp_path / f"out{int(with_u)}{int(ascii_lanes)}"
            total, matched, counts, _fq = run_streaming(str(fq), cp, str(out), "")
            trees.append((total, matched, counts.tolist(), read_expected_tree(str(out))))
        assert trees[0] == trees[1] and trees[0][0] == 3000
        if not with_u:
            exp_tree, total, matched = _oracle_tree(pf, sf, str(fq))
            assert (total, matched) == trees[0][:2]
            assert {k: sorted(v) for k, v in exp_tree.items()} == trees[0][3]


# ------------------------------------------------------------------ user-supplied barcode prefilters (databases.py:311-323)
class _SuffixGatePrefilter:
    """A BarcodePrefilter a user might write: lets a barcode through only if its last 4 letters occur in the first 24
    letters of the target (cheap seed test).  Unlike the Bloom filter it rejects true matches now and then, so the records
    differ from both the prefilter-off and the Bloom run: a real test of the host-evaluated path."""

    def __init__(self):
        self.calls = 0

    def match(self, barcode: str, sequence: str) -> bool:
        self.calls += 1
        return barcode[-4:] in sequence[:24]


@pytest.mark.parametrize("flags", [dict(), dict(trim="tails"), dict(trim="primers", dereplicate="none"), dict(disable_preorient=True)],
                         ids=lambda f: ",".join(f"{k}={v}" for k, v in f.items()) or "default")
def test_user_prefilter_matches_oracle(lib, c2, c3, flags):
    """process_sequences with an arbitrary BarcodePrefilter object: primer alignments from the kernel, the barcode
    alignments the callback lets through from the device aligner, selection replayed on the host -- record for record what
    the oracle (the reference's loop with the same callback, demultiplex.py:796) produces.  Edge-case reads included."""
    from specimux_amd import synth
    from specimux_amd.demultiplex import process_sequences
    from specimux_amd.io_utils import SeqRecord
    from oracle import specimux_oracle as O
    from parity_utils import RT
    for name, (pan, (pf, sf)) in (("c2", c2), ("c3", c3)):
        rs = synth.make_reads(pan, 500, 321, windows_only=False, n_frac=0.03, **(dict(insert_mean=900, insert_sd=250) if name == "c3" else {}))
        reads = reads_from_set(rs, range(500), 80) + ([r for r in _edge_reads(pan) if r[0] != "u_base"] if name == "c2" else [])
        both = Both(pf, sf, **flags)
        gate, ogate = _SuffixGatePrefilter(), _SuffixGatePrefilter()
        recs = [SeqRecord(s, rid, rid, q) for rid, s, q in reads]
        ops, total, matched = process_sequences(recs, both.parameters, both.specimens, both.args, gate)
        got = [(op.seq_id, op.sample_id, op.distance_code, op.primer_pool, op.p1_name, op.p2_name,
                RT[op.resolution_type.value], op.sequence, op.quality_sequence) for op in ops]
        oops, ototal, omatched = O.process_sequences(reads, both.opar, both.opanel, prefilter=ogate)
        exp = [(op.seq_id, op.sample_id, op.code, op.pool, op.p1, op.p2, RT[op.rtype], op.sequence, op.quality) for op in oops]
        assert (total, matched) == (ototal, omatched), name
        assert got == exp, f"{name} {flags}: first difference {next((g, e) for g, e in zip(got, exp) if g != e)}"
        assert gate.calls == ogate.calls > 1000
        # and it is not the Bloom / no-prefilter answer by accident
        plain, _t, _m = both.product_ops(reads)
        assert [k[:3] for k in plain] != [k[:3] for k in got]


def test_batches_in_flight_on_two_streams(lib, c2, c3):
    """smx_panel_set_streams(2): two batches in flight on two HIP streams, every persistent demux launch sized to half of the
    CUs' workgroup slots, kernels of the two batches running side by side.  Records, extra records and counts of every batch
    equal those of the same batch run alone through the host-buffer path (which the other tests pin to the oracle) -- for
    the dense two-primer kernel and the compact + redo launches of the 8-primer panel."""
    import torch
    from specimux_amd import _lib, synth
    from specimux_amd.demultiplex import compiled_panel
    dev = torch.device("cuda", 0)
    for name, (pan, (pf, sf)), n in (("c2", c2, 150_000), ("c3", c3, 100_000)):
        both = Both(pf, sf)
        cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)
        sets = [synth.make_reads(pan, n, 6100 + b, workers=8, n_frac=0.01) for b in range(2)]
        alone = []
        for rs in sets:
            ops, extra, counts = cp.run(rs.windows(cp.window_stride), rs.lens)
            alone.append((ops.copy(), np.sort(extra, order=["read", "sample", "trim_start", "p1", "p2", "barcode", "rtype"]), counts.copy()))
        cp.set_streams(2)
        try:
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            bufs = []
            for rs in sets:
                bufs.append(dict(w=torch.from_numpy(rs.windows(cp.window_stride)).to(dev), l=torch.from_numpy(rs.lens).to(dev),
                                 ops=torch.zeros(n * 32, dtype=torch.uint8, device=dev), extra=torch.zeros(n * 32, dtype=torch.uint8, device=dev),
                                 ne=torch.zeros(4, dtype=torch.int32, device=dev), counts=torch.zeros(cp.counts_len, dtype=torch.int64, device=dev)))
            torch.cuda.synchronize()
            for rep in range(4):   # interleaved launches: batch 0 on stream 0, batch 1 on stream 1, four rounds
                for k, b in enumerate(bufs):
                    b["counts"].zero_()
                torch.cuda.synchronize()
                for _round in range(2):
                    for k, b in enumerate(bufs):
                        _lib.check(lib.smx_batch_run_device(cp.handle, C.c_void_p(streams[k].cuda_stream), C.c_void_p(b["w"].data_ptr()),
                                                            C.c_void_p(b["l"].data_ptr()), n, C.c_void_p(b["ops"].data_ptr()),
                                                            C.c_void_p(b["extra"].data_ptr()), n, C.c_void_p(b["ne"].data_ptr()),
                                                            C.c_void_p(b["counts"].data_ptr()), None, None))
                torch.cuda.synchronize()
                for k, b in enumerate(bufs):
                    ops = b["ops"].cpu().numpy().view(_lib.OP_DTYPE)
                    ne = int(b["ne"][0].item())
                    extra = np.sort(b["extra"].cpu().numpy().view(_lib.OP_DTYPE)[:ne], order=["read", "sample", "trim_start", "p1", "p2", "barcode", "rtype"])
                    counts = b["counts"].cpu().numpy().astype(np.uint64)
                    assert np.array_equal(ops, alone[k][0]), (name, rep, k)
                    assert np.array_equal(extra, alone[k][1]), (name, rep, k)
                    assert np.array_equal(counts, 2 * alone[k][2]), (name, rep, k)   # (two launches per batch and round)
        finally:
            cp.set_streams(1)
