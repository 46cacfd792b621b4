"""Helpers shared by the parity tests: build the same panel for the product (specimux_amd, GPU) and for
the oracle (oracle/, CPU), run both on the same reads, and diff hit tables and write operations."""
import argparse
import os

import numpy as np

from oracle import specimux_oracle as O

RT = {1: "FULL", 2: "PFWD", 3: "PREV", 4: "MULTI", 5: "UNKNOWN", 6: "DEREP"}

# flag sets of the randomised parity loops (tests/fuzz_parity.py, test_gpu_parity.py::test_bounded_fuzz); error_rate / n_frac
# are knobs of the read generator, not specimux flags
FUZZ_FLAG_SETS = [dict(), dict(trim="tails"), dict(trim="primers"), dict(dereplicate="none"), dict(disable_prefilter=True),
                  dict(disable_preorient=True), dict(search_len=64), dict(search_len=120), dict(index_edit_distance=2),
                  dict(primer_edit_distance=4), dict(index_edit_distance=4, disable_prefilter=True),
                  dict(search_len=160, error_rate=0.12, n_frac=0.05), dict(search_len=48, n_frac=0.1), dict(search_len=96, trim="tails")]


def make_args(**kw):
    a = argparse.Namespace(index_edit_distance=-1, primer_edit_distance=-1, search_len=80, disable_preorient=False,
                           disable_prefilter=False, dereplicate="best", trim="barcodes", diagnostics=None,
                           min_length=-1, max_length=-1, output_to_files=False, color=False, isfastq=True)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


class Both:
    """Product and oracle views of one panel + flag set."""

    def __init__(self, primer_file, specimen_file, **flags):
        import specimux_amd as sa
        from specimux_amd.bloom_filter import BloomPrefilter, barcodes_for_bloom_prefilter
        self.args = make_args(**flags)
        a = self.args
        reg = sa.read_primers_file(primer_file)
        self.specimens = sa.read_specimen_file(specimen_file, reg)
        self.specimens.validate()
        self.parameters = sa.setup_match_parameters(a, self.specimens)
        self.prefilter = None if a.disable_prefilter else BloomPrefilter(
            barcodes_for_bloom_prefilter(self.specimens), self.parameters.max_dist_index)
        # oracle
        self.opanel = O.load_panel(primer_file, specimen_file)
        self.opar = O.setup_params(self.opanel, search_len=a.search_len, index_edit_distance=a.index_edit_distance,
                                   primer_edit_distance=a.primer_edit_distance, preorient=not a.disable_preorient,
                                   prefilter=not a.disable_prefilter, trim=a.trim, dereplicate=a.dereplicate,
                                   min_length=a.min_length, max_length=a.max_length)
        assert self.opar.max_dist_index == self.parameters.max_dist_index
        assert self.opar.max_dist_primers == self.parameters.max_dist_primers

    # reads: list of (id, bases, quality)
    def product_ops(self, reads):
        from specimux_amd.demultiplex import process_sequences
        from specimux_amd.io_utils import SeqRecord
        recs = [SeqRecord(s, rid, rid, q) for rid, s, q in reads]
        ops, total, matched = process_sequences(recs, self.parameters, self.specimens, self.args, self.prefilter)
        keys = [(op.seq_id, op.sample_id, op.distance_code, op.primer_pool, op.p1_name, op.p2_name,
                 RT[op.resolution_type.value], op.sequence, op.quality_sequence) for op in ops]
        return keys, total, matched

    def oracle_ops(self, reads):
        ops, total, matched = O.process_sequences(reads, self.opar, self.opanel)
        keys = [(op.seq_id, op.sample_id, op.code, op.pool, op.p1, op.p2, RT[op.rtype], op.sequence, op.quality)
                for op in ops]
        return keys, total, matched

    def assert_ops_equal(self, reads, label=""):
        got, gt, gm = self.product_ops(reads)
        exp, et, em = self.oracle_ops(reads)
        assert (gt, gm) == (et, em), f"{label}: totals {gt, gm} != oracle {et, em}"
        if got != exp:
            by_g, by_e = {}, {}
            for k in got:
                by_g.setdefault(k[0], []).append(k)
            for k in exp:
                by_e.setdefault(k[0], []).append(k)
            bad = [r for r in by_e if by_g.get(r) != by_e[r]] + [r for r in by_g if r not in by_e]
            rid = bad[0]
            seq = next(s for i, s, _ in reads if i == rid)
            raise AssertionError(f"{label}: {len(bad)} read(s) differ; first {rid} (len {len(seq)}):\n"
                                 f"  gpu   : {[k[1:7] + (len(k[7]),) for k in by_g.get(rid, [])]}\n"
                                 f"  oracle: {[k[1:7] + (len(k[7]),) for k in by_e.get(rid, [])]}\n  seq: {seq}")
        return got

    # ------------------------------------------------------------------ match locations (--color)
    def assert_locations_equal(self, reads, label=""):
        """p1/p2/b1/b2 locations of every record (what --color paints) against the oracle's CandidateMatch state.
        Trim-to-empty fallback records are skipped: the kernel's record no longer names their candidate."""
        from specimux_amd.demultiplex import process_sequences
        from specimux_amd.io_utils import SeqRecord
        recs = [SeqRecord(s, rid, rid, q) for rid, s, q in reads]
        self.args.color, self.args.output_to_files = True, False
        try:
            ops, _t, _m = process_sequences(recs, self.parameters, self.specimens, self.args, self.prefilter)
        finally:
            self.args.color = False
        oops, _t, _m = O.process_sequences(reads, self.opar, self.opanel)
        assert len(ops) == len(oops)
        n = 0
        for g, e in zip(ops, oops):
            assert g.seq_id == e.seq_id and g.sequence == e.sequence
            if e.sample_id == "unknown" and e.pool == "unknown" and e.p1 == "unknown" and e.p2 == "unknown" and \
                    (e.p1_loc or e.p2_loc):
                continue   # fallback record
            got = (g.p1_location, g.p2_location, g.b1_location, g.b2_location)
            exp = (e.p1_loc, e.p2_loc, e.b1_loc, e.b2_loc)
            assert got == exp, f"{label}: read {g.seq_id} ({e.sample_id}, {e.code}): gpu {got} oracle {exp}"
            n += sum(1 for x in exp if x is not None)
        return n

    # ------------------------------------------------------------------ trace events (-d)
    def product_trace(self, reads, level, out_dir, worker_id="main", record_offset=0):
        """Rows of the TSV the product writes for this batch, without the timestamp column."""
        import csv
        from specimux_amd.demultiplex import process_sequences
        from specimux_amd.io_utils import OutputManager, SeqRecord, output_write_operation
        from specimux_amd.trace import TraceLogger
        recs = [SeqRecord(s, rid, rid, q) for rid, s, q in reads]
        self.args.output_dir, self.args.output_to_files, self.args.output_file_prefix = str(out_dir), True, ""
        tl = TraceLogger(True, level, str(out_dir), worker_id, "20260101_000000")
        ops, _t, _m = process_sequences(recs, self.parameters, self.specimens, self.args, self.prefilter, tl, record_offset)
        with OutputManager(str(out_dir), "", True) as om:
            for op in ops:
                output_write_operation(op, om, self.args, tl)
        tl.close()
        with open(tl.filepath, newline="") as fh:
            rows = list(csv.reader(fh, delimiter="\t"))
        assert rows[0] == ["timestamp", "worker_id", "event_seq", "sequence_id", "event_type"]
        return [r[1:] for r in rows[1:]]

    def oracle_trace(self, reads, level, worker_id="main", record_offset=0):
        tr = O.Tracer(level, worker_id)
        ops, _t, _m = O.process_sequences(reads, self.opar, self.opanel, tr=tr, record_offset=record_offset)
        O.trace_outputs(tr, ops)
        return tr.rows

    def assert_trace_equal(self, reads, level, out_dir, label=""):
        got = self.product_trace(reads, level, out_dir)
        exp = self.oracle_trace(reads, level)
        for i, (g, e) in enumerate(zip(got, exp)):
            assert g == e, f"{label} level {level}: event {i + 1} differs\n  gpu   : {g}\n  oracle: {e}"
        assert len(got) == len(exp), f"{label} level {level}: {len(got)} events != oracle {len(exp)}"
        return len(got)

    # ------------------------------------------------------------------ hit tables
    def assert_hits_equal(self, reads, label="", lean=None):
        """Hit tables vs the oracle.  Default: both dumps -- the per-barcode ("slots") kernel with every barcode's
        distance, and the hit table of the kernel the flags select (the lean, bit-sliced one that is benchmarked)."""
        if lean is None:
            n = self.assert_hits_equal(reads, label + " [slots]", lean=False)
            self.assert_hits_equal(reads, label + " [lean]", lean=True)
            return n
        from specimux_amd.demultiplex import compiled_panel, concat_records
        from specimux_amd.io_utils import SeqRecord
        cp = compiled_panel(self.specimens, self.parameters, self.args, self.prefilter)
        recs = [SeqRecord(s, rid, rid, q) for rid, s, q in reads]
        bases, offsets, _ = concat_records(recs)
        windows, lens = cp.pack_windows(bases, offsets)
        ops, extra, counts, hits, bdist = cp.run(windows, lens, want_hits="lean" if lean else True)
        need_starts = self.args.trim in ("primers", "tails")
        names = cp.primer_names
        checked = 0
        for i, rec in enumerate(reads):
            table = O.hit_table(self.opar, self.opanel, rec)
            L = len(rec[1])
            filtered = (self.args.min_length != -1 and L < self.args.min_length) or \
                       (self.args.max_length != -1 and L > self.args.max_length)
            for p, name in enumerate(names):
                for e, end in enumerate("AB"):
                    h = hits[i, p * 2 + e]
                    exp = table[(name, end)]
                    ctx = f"{label} read {rec[0]} (len {L}) primer {name} end {end}: gpu {h} oracle {exp}"
                    assert int(h["pdist"]) == exp["pdist"], ctx
                    if exp["pdist"] < 0:
                        continue
                    assert int(h["nloc"]) == len(exp["locs"]), ctx
                    assert int(h["first_end"]) == exp["locs"][0][1], ctx
                    if need_starts:
                        assert int(h["first_start"]) == exp["locs"][0][0], ctx
                    if int(h["bbest"]) == -2:      # orientation-pruned (or filtered): the GPU never searched
                        assert filtered or self.parameters.preorient, ctx
                        continue
                    bcs = cp.primers[p].barcodes
                    dists = {bc: d for bc, (d, _l) in exp["barcodes"].items()}
                    if not lean:
                        for bi, bc in enumerate(bcs):
                            assert int(bdist[i, p * 2 + e, bi]) == dists.get(bc, -1), f"{ctx} barcode {bc}"
                    if dists:
                        best = min(dists.values())
                        tied = [bc for bc in bcs if dists.get(bc) == best]
                        assert int(h["bbest"]) == best and int(h["ntied"]) == len(tied), ctx
                        assert cp.barcodes[int(h["first_tied"])] == tied[0], ctx
                        tail = max(l[1] for _d, locs in exp["barcodes"].values() for l in locs)
                        if not lean or self.args.trim == "tails":
                            assert int(h["tail_end"]) == tail, ctx
                    else:
                        assert int(h["bbest"]) == -1, ctx
                    checked += 1
        return checked


from specimux_amd.synth import rebuild_read  # noqa: E402,F401


def reads_from_set(rs, idx, S, prefix="r"):
    out = []
    rng = np.random.default_rng(12345)
    for i in idx:
        if rs.reads is not None:
            s, q = rs.reads[i], rs.quals[i]
        else:
            s = rebuild_read(rs.head[i], rs.tail[i], int(rs.lens[i]), S)
            q = (rng.integers(3, 41, len(s)) + 33).astype(np.uint8).tobytes().decode()
        out.append((f"{prefix}{i}", s, q))
    return out


def tmp_panel(tmp_path_factory, panel, name):
    d = tmp_path_factory.mktemp(name)
    return panel.write(os.fspath(d))
