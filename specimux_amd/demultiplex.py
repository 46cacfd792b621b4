"""process_sequences: the batch operator of the drop-in interface, executed on the GPU.

Reference boundary: src/specimux/demultiplex.py:108-212 (callers multiprocessing_utils.py:89 and
orchestration.py:513).  Same signature, same return value (write_ops, total_count, matched_count); the
whole per-read pipeline below it runs inside one HIP kernel (specimux_amd/csrc/smx_kernels.hip) behind
the C ABI of include/smx.h.  There is no CPU implementation here: without libsmx.so or without a GPU
this raises."""
import logging
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .constants import MultipleMatchStrategy, ResolutionType, SampleId, TrimMode
from .databases import PassthroughPrefilter
from .models import WriteOperation, reverse_complement
from .panel import CompiledPanel


def _is_user_prefilter(prefilter) -> bool:
    """Anything that implements the reference's BarcodePrefilter protocol (databases.py:311-316: `match(barcode,
    sequence) -> bool`) other than the two built-ins."""
    return not (prefilter is None or isinstance(prefilter, PassthroughPrefilter) or getattr(prefilter, "smx_exact_set", False))


def _prefilter_enabled(prefilter) -> bool:
    """None / PassthroughPrefilter -> off.  A BloomPrefilter (bloom_filter.py) -> the exact-set rule in
    the kernel.  Arbitrary user prefilters are Python callables: process_sequences evaluates them on the host
    (_process_with_user_prefilter); the kernel then runs with its own rule off."""
    if prefilter is None or isinstance(prefilter, PassthroughPrefilter) or _is_user_prefilter(prefilter):
        return False
    return True


def compiled_panel(specimens, parameters, args, prefilter, want_starts=False) -> CompiledPanel:
    """One CompiledPanel per distinct (panel, thresholds, flags); cached on the Specimens object."""
    key = (parameters.max_dist_index, parameters.search_len, parameters.preorient,
           tuple(sorted(parameters.max_dist_primers.items())), getattr(args, "trim", TrimMode.BARCODES),
           getattr(args, "dereplicate", MultipleMatchStrategy.BEST), _prefilter_enabled(prefilter),
           getattr(args, "min_length", -1), getattr(args, "max_length", -1), len(specimens._specimens),
           bool(want_starts))
    cache = specimens.__dict__.setdefault("_smx_panels", {})
    if key not in cache:
        cache[key] = CompiledPanel(specimens, parameters, trim=key[4], dereplicate=key[5], prefilter=key[6],
                                   min_length=key[7], max_length=key[8], want_starts=key[10])
    return cache[key]


def concat_records(seq_records):
    """-> (uint8 bases, uint64 offsets).  Characters outside latin-1 cannot be DNA; they become '?'."""
    seqs = [str(r.seq) for r in seq_records]
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        offsets[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    blob = "".join(seqs).encode("latin-1", "replace")
    return np.frombuffer(blob, dtype=np.uint8), offsets, seqs


def order_ops(ops: np.ndarray, extra: np.ndarray):
    """Yield (read_index, record) in the reference's write-op order: reads in input order, a read's
    records in emission order (primary first, then its extra records in buffer order)."""
    by_read = {}
    for j in range(len(extra)):
        by_read.setdefault(int(extra["read"][j]), []).append(j)
    for i in range(len(ops)):
        if ops["rtype"][i] == _lib.R_FILTERED:
            continue
        yield i, ops[i]
        for j in by_read.get(i, ()):
            yield i, extra[j]


def op_names(panel: CompiledPanel, rec):
    """(sample_id, pool, p1_name, p2_name, distance_code, ResolutionType) of one smx_op record."""
    rtype = int(rec["rtype"])
    if rec["sample"] >= 0:
        sample = panel.specimen_ids[int(rec["sample"])]
    elif rtype == _lib.R_PARTIAL_FWD:
        sample = SampleId.PREFIX_FWD_MATCH + panel.barcodes[int(rec["barcode"])]
    elif rtype == _lib.R_PARTIAL_REV:
        sample = SampleId.PREFIX_REV_MATCH + panel.barcodes[int(rec["barcode"])]
    else:
        sample = SampleId.UNKNOWN
    pool = panel.pools[int(rec["pool"])] if rec["pool"] >= 0 else "unknown"
    p1 = panel.primer_names[int(rec["p1"])] if rec["p1"] >= 0 else "unknown"
    p2 = panel.primer_names[int(rec["p2"])] if rec["p2"] >= 0 else "unknown"
    code = ",".join(str(int(d)) if d >= 0 else "X" for d in rec["dist"])
    return sample, pool, p1, p2, code, ResolutionType(rtype)


class _Locator:
    """Match locations for `--color` (WriteOperation.p1/p2/b1/b2_location, models.py:262-276): primer locations come
    from the kernel's hit table, the best barcode's location from the device aligner (the kernel keeps distances, not
    where each barcode aligned).  Coordinates are those of the candidate's orientation after the trim shift (Q8)."""

    def __init__(self, panel, parameters, prefilter_on):
        self.panel, self.par, self.prefilter_on = panel, parameters, prefilter_on
        self.pf_min = (len(panel.barcodes[0]) - parameters.max_dist_index) if prefilter_on else 0

    def _barcode(self, primer, bd_row, sequence, reversed_sequence, L):
        from .alignment import align_seq
        from .constants import AlignMode
        best_d = min((int(d) for d in bd_row[:len(primer.barcodes)] if d >= 0), default=-1)
        if best_d < 0:
            return None
        b = next(b for i, b in enumerate(primer.barcodes) if bd_row[i] == best_d)   # first of the stable sort
        b_rc = reverse_complement(b)
        pm = align_seq(primer.primer_rc, sequence, self.par.max_dist_primers[primer.primer], L - self.par.search_len, L)
        best = None
        for loc in pm.locations():
            start = loc[1] + 1
            if self.prefilter_on:
                x = sequence[start:][:self.pf_min]
                if len(x) < self.pf_min or any(ch not in "ACGT" for ch in x):
                    continue
            bm = align_seq(b_rc, sequence, self.par.max_dist_index, start, L, AlignMode.PREFIX)
            if bm.matched() and (best is None or bm.distance() < best.distance()):
                best = bm
        if best is None:
            return None
        a, e = best.location()
        return (L - e - 1, L - a - 1) if reversed_sequence else (a, e)

    def _ends(self, seq, rseq, rec, hits, bdist):
        """(primer, hit index, searched string, bdist row) of the record's two ends, as locate() walks them."""
        o = 1 if rec["flags"] & _lib.OPF_REVERSE else 0
        for which, pi in ((1, int(rec["p1"])), (2, int(rec["p2"]))):
            if pi < 0:
                continue
            primer = self.panel.primers[pi]
            if which == 1:
                h, sequence = pi * 2 + (0 if o == 0 else 1), (rseq if o == 0 else seq)
            else:
                h, sequence = pi * 2 + (1 if o == 0 else 0), (seq if o == 0 else rseq)
            yield primer, sequence, bdist[h]

    def prefetch(self, cache, seqs, ops, extra, hits, bdist):
        """Collect the alignments locate() will request for this batch and run them in two launches."""
        from .constants import AlignMode
        S, stage1, pend = self.par.search_len, [], []
        for i, rec in order_ops(ops, extra):
            if (rec["flags"] & _lib.OPF_TRIM_EMPTY) or (rec["p1"] < 0 and rec["p2"] < 0):
                continue
            seq = seqs[i]
            rseq = reverse_complement(seq)
            L = len(seq)
            for primer, sequence, bd_row in self._ends(seq, rseq, rec, hits[i], bdist[i]):
                best_d = min((int(d) for d in bd_row[:len(primer.barcodes)] if d >= 0), default=-1)
                if best_d < 0:
                    continue
                b = next(b for k_, b in enumerate(primer.barcodes) if bd_row[k_] == best_d)
                s0 = L - S
                s_ = 0 if s0 == -1 else s0
                t = sequence[s_:L]
                if not t:
                    continue
                req = (primer.primer_rc, t, int(self.par.max_dist_primers[primer.primer]), AlignMode.INFIX)
                stage1.append(req)
                pend.append((sequence, s_, reverse_complement(b), req))
        cache.fill(stage1)
        stage2 = []
        for sequence, s_, b_rc, req in pend:
            dist, locs = cache.table[req]
            if dist < 0:
                continue
            L = len(sequence)
            for _a, e in locs:
                start = e + s_ + 1
                if self.prefilter_on:
                    x = sequence[start:][:self.pf_min]
                    if len(x) < self.pf_min or any(ch not in "ACGT" for ch in x):
                        continue
                s2 = 0 if start == -1 else start
                t = sequence[s2:L][:len(b_rc) + int(self.par.max_dist_index)]
                if t:
                    stage2.append((b_rc, t, int(self.par.max_dist_index), AlignMode.PREFIX))
        cache.fill(stage2)

    def locate(self, seq, rseq, rec, hits, bdist, shift):
        """-> (p1, p2, b1, b2) locations of one record; `shift` = what trim_locations has subtracted so far."""
        L = len(seq)
        o = 1 if rec["flags"] & _lib.OPF_REVERSE else 0
        out = [None, None, None, None]
        for which, pi in ((1, int(rec["p1"])), (2, int(rec["p2"]))):
            if pi < 0:
                continue
            primer = self.panel.primers[pi]
            if which == 1:
                h, sequence, rev = pi * 2 + (0 if o == 0 else 1), (rseq if o == 0 else seq), True
            else:
                h, sequence, rev = pi * 2 + (1 if o == 0 else 0), (seq if o == 0 else rseq), False
            a, e = int(hits[h]["first_start"]), int(hits[h]["first_end"])
            out[which - 1] = (L - e - 1, L - a - 1) if rev else (a, e)
            out[which + 1] = self._barcode(primer, bdist[h], sequence, rev, L)
        return tuple(None if loc is None else (loc[0] - shift, loc[1] - shift) for loc in out)


def _process_with_user_prefilter(seq_records, parameters, specimens, args, prefilter, trace_logger, record_offset):
    """process_sequences for a user-supplied BarcodePrefilter (reference boundary: databases.py:311-323; the call it
    guards: demultiplex.py:796).  A Python callable cannot run inside the kernel, so this path splits the work: the kernel
    (its own prefilter rule off, dump mode) does the primer alignments and the orientation votes for the whole batch; the
    barcode alignments -- only those (barcode, primer location) pairs the user's `match(b_rc, sequence[start:])` lets
    through, exactly as match_one_end walks them -- go to the device aligner in two batched launches (alignment.AlignCache);
    selection, dereplication and specimen resolution are replayed on the host by the same code that checks the kernel's
    records under -d (trace.BatchReplayer).  A slow path by construction (Python per read): the price of an arbitrary
    callback, paid only by callers who pass one."""
    from . import trace as _trace
    from .alignment import AlignCache, align_seq
    from .constants import AlignMode
    if getattr(args, "color", False) and not getattr(args, "output_to_files", False):
        raise NotImplementedError("--color is not available together with a user-supplied barcode prefilter")
    if trace_logger is not None and getattr(trace_logger, "verbosity", 1) >= 3:
        raise NotImplementedError("trace level 3 is not available together with a user-supplied barcode prefilter")
    panel = compiled_panel(specimens, parameters, args, None, want_starts=True)
    bases, offsets, seqs = concat_records(seq_records)
    windows, lens = panel.pack_windows(bases, offsets)
    ops, extra, counts, hits, bdist = panel.run(windows, lens, want_hits=True)
    S, kidx = parameters.search_len, int(parameters.max_dist_index)
    primers = panel.primers
    # ---- stage 1: all optimal locations of every matched (read, primer, end)
    ends, stage1 = [], []
    for i, seq_str in enumerate(seqs):
        if ops["rtype"][i] == _lib.R_FILTERED:
            continue
        L, rs_str = len(seq_str), None
        for pi, primer in enumerate(primers):
            for X in (0, 1):
                if hits[i][pi * 2 + X]["pdist"] < 0:
                    continue
                if X == 0 and rs_str is None:
                    rs_str = reverse_complement(seq_str)
                q = rs_str if X == 0 else seq_str
                s0 = L - S
                t = q[(0 if s0 == -1 else s0):L]
                if t:
                    stage1.append((primer.primer_rc, t, int(parameters.max_dist_primers[primer.primer]), AlignMode.INFIX))
                ends.append((i, pi * 2 + X, q, primer))
    cache = AlignCache()
    cache.fill(stage1)
    # ---- stage 2: the barcode alignments the user's prefilter lets through (match_one_end, demultiplex.py:778-815)
    plan, stage2 = [], []
    with cache:
        for i, h, q, primer in ends:
            L = len(q)
            pm = align_seq(primer.primer_rc, q, int(parameters.max_dist_primers[primer.primer]), L - S, L)
            todo = []
            if pm.matched():
                for bi, b in enumerate(primer.barcodes):
                    b_rc = reverse_complement(b)
                    for loc in pm.locations():
                        start = loc[1] + 1
                        if not prefilter.match(b_rc, q[start:]):
                            continue
                        todo.append((bi, b_rc, start))
                        t = q[(0 if start == -1 else start):L][:len(b_rc) + kidx]
                        if t:
                            stage2.append((b_rc, t, kidx, AlignMode.PREFIX))
            plan.append((i, h, q, todo))
    cache.fill(stage2)
    hits2 = hits.copy()
    bdist2 = np.full(bdist.shape, -1, dtype=np.int8)
    with cache:
        for i, h, q, todo in plan:
            L, tail = len(q), None
            best = {}   # barcode slot -> (distance, alignment): lowest distance, first location on ties
            for bi, b_rc, start in todo:
                bm = align_seq(b_rc, q, kidx, start, L, AlignMode.PREFIX)
                if bm.matched() and (bi not in best or bm.distance() < best[bi][0]):
                    best[bi] = (bm.distance(), bm)
            for bi, (d, bm) in best.items():
                bdist2[i, h, bi] = d
                e_last = bm.locations()[-1][1]
                tail = e_last if tail is None or e_last > tail else tail
            hits2[i, h]["tail_end"] = -1 if tail is None else tail
    # ---- selection / dereplication / resolution: the reference's control flow replayed over the tables
    replayer = _trace.BatchReplayer(panel, parameters, specimens, args, False)
    tl = trace_logger if trace_logger is not None else _trace.NullTrace()
    write_ops: List[WriteOperation] = []
    matched = 0
    for i, record in enumerate(seq_records):
        sid = trace_logger.get_sequence_id(record, record_offset + i) if trace_logger is not None else None
        col = []
        replayer.replay(tl, record, seqs[i], sid, hits2[i], bdist2[i], None, collect=col)
        qual = getattr(record, "quality_string", None)
        if qual is None:
            ann = getattr(record, "letter_annotations", {}) or {}
            phred = ann.get("phred_quality")
            qual = "".join(chr(q + 33) for q in phred) if phred is not None else "I" * len(seqs[i])
        rc = None
        full = False
        for m, sample, rt, a, b, empty, asked in col:
            full = full or asked in (ResolutionType.FULL_MATCH, ResolutionType.DEREPLICATED_FULL)
            if m.o:
                if rc is None:
                    rc = (reverse_complement(seqs[i]), qual[::-1])
                s_, q_ = rc
            else:
                s_, q_ = seqs[i], qual
            s_, q_ = s_[a:b], q_[a:b]
            code = ",".join(str(d) if d >= 0 else "X" for d in (m.p1d, m.b1d(), m.b2d(), m.p2d))
            write_ops.append(WriteOperation(
                sample_id=sample, seq_id=record.id, distance_code=code, sequence=s_, quality_sequence=q_,
                quality_scores=[ord(c) - 33 for c in q_], p1_location=None, p2_location=None, b1_location=None, b2_location=None,
                primer_pool="unknown" if empty else (m.pool or "unknown"),
                p1_name="unknown" if (empty or not m.p1) else m.p1.name, p2_name="unknown" if (empty or not m.p2) else m.p2.name,
                resolution_type=rt, trace_sequence_id=sid))
        matched += 1 if full else 0
    return write_ops, len(seq_records), matched


def process_sequences(seq_records, parameters, specimens, args, prefilter, trace_logger=None,
                      record_offset: int = 0) -> Tuple[List[WriteOperation], int, int]:
    """Demultiplex one batch of reads on the GPU; see the module docstring.  With a trace_logger the kernel also
    returns the hit tables it scored from and trace.py replays the reference's events from them."""
    seq_records = list(seq_records)
    if _is_user_prefilter(prefilter):
        if not callable(getattr(prefilter, "match", None)):
            raise TypeError(f"prefilter {type(prefilter).__name__} has no match(barcode, sequence) method (databases.py:311-316)")
        return _process_with_user_prefilter(seq_records, parameters, specimens, args, prefilter,
                                            trace_logger if (trace_logger is not None and getattr(trace_logger, "enabled", True)) else None,
                                            record_offset)
    tracing = trace_logger is not None and getattr(trace_logger, "enabled", True)
    coloring = bool(getattr(args, "color", False)) and not getattr(args, "output_to_files", False)
    panel = compiled_panel(specimens, parameters, args, prefilter, want_starts=tracing or coloring)
    bases, offsets, seqs = concat_records(seq_records)
    windows, lens = panel.pack_windows(bases, offsets)
    trace_ids = {}
    if tracing:
        from . import trace as _trace
        ops, extra, counts, hits, bdist = panel.run(windows, lens, want_hits=True)
        replayer = panel.__dict__.get("_replayer")
        if replayer is None:
            replayer = panel.__dict__["_replayer"] = _trace.BatchReplayer(panel, parameters, specimens, args,
                                                                          _prefilter_enabled(prefilter))
        trace_ids = _trace.replay_batch(trace_logger, replayer, seq_records, seqs, ops, extra, hits, bdist, op_names,
                                        record_offset)
    elif coloring:
        ops, extra, counts, hits, bdist = panel.run(windows, lens, want_hits=True)
    else:
        ops, extra, counts = panel.run(windows, lens)
    locator = _Locator(panel, parameters, _prefilter_enabled(prefilter)) if coloring else None
    color_cache = None
    if coloring:
        # --color asks the device aligner for the best barcode's location of every painted end: all of them in two
        # launches (primer windows, then barcode targets), found again by align_seq through the cache
        from .alignment import AlignCache
        color_cache = AlignCache()
        locator.prefetch(color_cache, seqs, ops, extra, hits, bdist)
        color_cache.__enter__()
    shifts = {}   # (read, candidate) -> accumulated trim_locations shift (Q8)
    write_ops: List[WriteOperation] = []
    rc_cache = {}
    for i, rec in order_ops(ops, extra):
        record = seq_records[i]
        sample, pool, p1, p2, code, rtype = op_names(panel, rec)
        qual = getattr(record, "quality_string", None)
        if qual is None:
            ann = getattr(record, "letter_annotations", {}) or {}
            phred = ann.get("phred_quality")
            qual = "".join(chr(q + 33) for q in phred) if phred is not None else "I" * len(seqs[i])
        if rec["flags"] & _lib.OPF_REVERSE:
            if i not in rc_cache:
                rc_cache[i] = (reverse_complement(seqs[i]), qual[::-1])
            s, q = rc_cache[i]
        else:
            s, q = seqs[i], qual
        a, b = int(rec["trim_start"]), int(rec["trim_end"])
        locs = (None, None, None, None)
        if locator is not None and not (rec["flags"] & _lib.OPF_TRIM_EMPTY) and (rec["p1"] >= 0 or rec["p2"] >= 0):
            # the record does not name its candidate; (matched primers, orientation, pool) identifies it except when
            # two candidates of one read share all of those (then their Q8 shifts are pooled: cosmetic, --color only)
            key = (i, int(rec["p1"]), int(rec["p2"]), int(rec["flags"]) & _lib.OPF_REVERSE, int(rec["pool"]))
            if getattr(args, "trim", TrimMode.BARCODES) != TrimMode.NONE:
                shifts[key] = shifts.get(key, 0) + a     # create_write_operation shifts before it reads the locations
            if i not in rc_cache:
                rc_cache[i] = (reverse_complement(seqs[i]), qual[::-1])
            locs = locator.locate(seqs[i], rc_cache[i][0], rec, hits[i], bdist[i], shifts.get(key, 0))
        s, q = s[a:b], q[a:b]
        if rec["flags"] & _lib.OPF_NO_SPECIMEN:
            logging.warning(f"No Specimens for combo: read {record.id} ({code}, {p1}, {p2})")
        write_ops.append(WriteOperation(
            sample_id=sample, seq_id=record.id, distance_code=code, sequence=s, quality_sequence=q,
            quality_scores=[ord(c) - 33 for c in q], p1_location=locs[0], p2_location=locs[1], b1_location=locs[2],
            b2_location=locs[3], primer_pool=pool, p1_name=p1, p2_name=p2, resolution_type=rtype,
            trace_sequence_id=trace_ids.get(i)))
    if color_cache is not None:
        color_cache.__exit__(None, None, None)
    return write_ops, len(seq_records), int(counts[_lib.CNT_MATCHED])
