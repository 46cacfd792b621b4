"""Trace TSV emission (`-d 1|2|3`) for the GPU path.

Reference: src/specimux/trace.py (TraceLogger: file naming, header, event rows, verbosity rules) and the
`trace_logger.log_*` call sites in demultiplex.py / io_utils.py.  The file format, the event vocabulary and
the field order are the reference's; `specimux-stats` / `specimux-visualize` read these files unchanged.

Where the events come from here: the batch kernel runs in its dump mode and returns, per read, the hit table
it scored from (include/smx.h: `smx_hit` per (primer, end) + the best distance of every barcode at that end).
`replay_batch` walks that table in the reference's control flow (find_candidate_matches -> select_best_matches
-> dereplicate_* -> resolve_specimen -> create_write_operation) and logs what the reference would have logged.
Nothing is aligned on the host: levels 1 and 2 need only the table; level 3 (every barcode search attempt)
asks the device primitive `align_seq` (alignment.py -> smx_align) for the per-location results the table does
not keep.  The records that are written still come from the kernel; the replay cross-checks its own
conclusion against them and raises on any disagreement.

Canonical forms (as everywhere on this path): barcodes are enumerated in first-appearance order where the
reference iterates a `set`; timestamps are wall-clock and not comparable between runs."""
import csv
import logging
import os
from collections import OrderedDict
from datetime import datetime
from pathlib import Path
from typing import List, Optional

from . import _lib
from .constants import AlignMode, Primer, ResolutionType, SampleId, TrimMode
from .models import reverse_complement


class TraceLogger:
    """Same constructor, file layout and logging methods as the reference's TraceLogger (trace.py:24-334)."""

    def __init__(self, enabled: bool, verbosity: int, output_dir: str, worker_id: str, start_timestamp: str,
                 buffer_size: int = 1000):
        self.enabled = enabled
        self.verbosity = verbosity
        self.worker_id = worker_id
        self.event_counter = 0
        self.buffer = []
        self.buffer_size = buffer_size
        self.file_handle = None
        self.sequence_record_counter = 0
        if self.enabled:
            trace_dir = Path(output_dir) / "trace"
            trace_dir.mkdir(parents=True, exist_ok=True)
            self.filepath = trace_dir / f"specimux_trace_{start_timestamp}_{worker_id}.tsv"
            self.file_handle = open(self.filepath, "w", newline="")
            self.writer = csv.writer(self.file_handle, delimiter="\t")
            self.writer.writerow(["timestamp", "worker_id", "event_seq", "sequence_id", "event_type"])

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()

    def close(self):
        if self.enabled and self.file_handle:
            self._flush_buffer()
            self.file_handle.flush()
            self.file_handle.close()
            self.file_handle = None

    def _flush_buffer(self):
        if self.file_handle and self.buffer:
            self.writer.writerows(self.buffer)
            self.file_handle.flush()
            self.buffer = []

    def _log_event(self, sequence_id: str, event_type: str, *fields):
        if not self.enabled:
            return
        self.event_counter += 1
        self.buffer.append([datetime.now().isoformat(), self.worker_id, self.event_counter, sequence_id, event_type]
                           + list(fields))
        if len(self.buffer) >= self.buffer_size:
            self._flush_buffer()

    def get_sequence_id(self, seq_record, record_num: Optional[int] = None) -> str:
        if record_num is None:
            self.sequence_record_counter += 1
            record_num = self.sequence_record_counter
        return f"{seq_record.id}#{record_num:08d}#{self.worker_id}"

    # ---- level 1
    def log_sequence_received(self, sequence_id, sequence_length, sequence_name):
        self._log_event(sequence_id, "SEQUENCE_RECEIVED", sequence_length, sequence_name)

    def log_sequence_filtered(self, sequence_id, sequence_length, filter_reason):
        self._log_event(sequence_id, "SEQUENCE_FILTERED", sequence_length, filter_reason)

    def log_orientation_detected(self, sequence_id, orientation, forward_score, reverse_score, confidence):
        self._log_event(sequence_id, "ORIENTATION_DETECTED", orientation, forward_score, reverse_score, f"{confidence:.3f}")

    def log_primer_matched(self, sequence_id, match, pool, orientation_used):
        cid, p1, p2, _b1, _b2, _pres, _tot, p1d, p2d, _b1d, _b2d = match.info()
        match_type = "both" if match.p1 and match.p2 else "forward_only" if match.p1 else "reverse_only"
        self._log_event(sequence_id, "PRIMER_MATCHED", cid, match_type, p1, p2, p1d, p2d, pool, orientation_used)

    def log_barcode_matched(self, sequence_id, match):
        cid, p1, p2, b1, b2, pres, _tot, _p1d, _p2d, b1d, b2d = match.info()
        self._log_event(sequence_id, "BARCODE_MATCHED", cid, pres, b1, b2, b1d, b2d, p1, p2)

    def log_match_scored(self, sequence_id, match, score: float):
        cid, p1, p2, b1, b2, pres, tot = match.info()[:7]
        self._log_event(sequence_id, "MATCH_SCORED", cid, p1, p2, b1, b2, tot, pres, f"{score:.3f}")

    def log_match_selected(self, sequence_id, selection_strategy, forward_primer, reverse_primer, forward_barcode,
                           reverse_barcode, pool, is_unique):
        self._log_event(sequence_id, "MATCH_SELECTED", selection_strategy, forward_primer, reverse_primer,
                        forward_barcode, reverse_barcode, pool, str(is_unique).lower())

    def log_specimen_resolved(self, sequence_id, match, specimen_id, resolution_type, pool):
        _cid, p1, p2, b1, b2 = match.info()[:5]
        self._log_event(sequence_id, "SPECIMEN_RESOLVED", specimen_id, resolution_type, pool, p1, p2, b1, b2)

    def log_sequence_output(self, sequence_id, specimen_id, pool, primer_pair, file_path):
        self._log_event(sequence_id, "SEQUENCE_OUTPUT", specimen_id, pool, primer_pair, file_path)

    def log_sequence_trim_empty(self, sequence_id, trim_mode, trim_start, trim_end, seq_length, p1_name, p2_name):
        self._log_event(sequence_id, "SEQUENCE_TRIM_EMPTY", trim_mode, trim_start, trim_end, seq_length, p1_name, p2_name)

    def log_no_match_found(self, sequence_id, stage_failed, reason):
        self._log_event(sequence_id, "NO_MATCH_FOUND", stage_failed, reason)

    def log_match_discarded(self, sequence_id, match, score: float, discard_reason):
        cid, p1, p2, b1, b2 = match.info()[:5]
        self._log_event(sequence_id, "MATCH_DISCARDED", cid, p1, p2, b1, b2, score, discard_reason)

    def log_dereplicate_expanded(self, sequence_id, match_count, expanded_count):
        self._log_event(sequence_id, "DEREPLICATE_EXPANDED", match_count, expanded_count)

    def log_dereplicate_selected(self, sequence_id, specimen_id, alternatives_count, scores):
        barcode_dist, primer_dist, file_idx = scores
        self._log_event(sequence_id, "DEREPLICATE_SELECTED", specimen_id, alternatives_count, barcode_dist, primer_dist, file_idx)

    def log_dereplicate_partial_selected(self, sequence_id, direction, barcode, alternatives_count, scores):
        barcode_dist, neg_primer_count, primer_dist, file_idx = scores
        self._log_event(sequence_id, "DEREPLICATE_PARTIAL_SELECTED", direction, barcode, alternatives_count, barcode_dist,
                        -neg_primer_count, primer_dist, file_idx)

    def log_dereplicate_unknown_selected(self, sequence_id, alternatives_count, primer_count, primer_dist, file_idx):
        self._log_event(sequence_id, "DEREPLICATE_UNKNOWN_SELECTED", alternatives_count, primer_count, primer_dist, file_idx)

    # ---- level 2+
    def log_primer_search(self, sequence_id, primer_name, primer_direction, search_start, search_end, found,
                          edit_distance, match_position):
        if self.verbosity >= 2 and (self.verbosity >= 3 or found):
            self._log_event(sequence_id, "PRIMER_SEARCH", primer_name, primer_direction, search_start, search_end,
                            str(found).lower(), edit_distance, match_position)

    def log_barcode_search(self, sequence_id, barcode_name, barcode_type, primer_adjacent, search_start, search_end,
                           found, edit_distance, match_position):
        if self.verbosity >= 3:
            self._log_event(sequence_id, "BARCODE_SEARCH", barcode_name, barcode_type, primer_adjacent, search_start,
                            search_end, str(found).lower(), edit_distance, match_position)


# ------------------------------------------------------------------------------------------------
# Replay of one batch from the kernel's hit tables
class _Cand:
    """The state of one (primer pair, orientation) hypothesis as the reference's CandidateMatch would hold it
    (models.py:72-328), rebuilt from two smx_hit records and their per-barcode distances."""

    __slots__ = ("cid", "o", "L", "p1", "p2", "h1", "h2", "p1d", "p2d", "b1", "b2", "pool", "pair", "cum")

    def __init__(self, cid, o, L):
        self.cid, self.o, self.L = cid, o, L
        self.p1 = self.p2 = None          # PrimerInfo when matched
        self.h1 = self.h2 = None          # smx_hit rows
        self.p1d = self.p2d = -1
        self.b1, self.b2 = [], []         # [(barcode, dist)] stable-sorted by dist (models.py:97-108)
        self.pool = None
        self.pair = 0
        self.cum = 0                      # Q8: locations already shifted by earlier emissions of this candidate

    # -- the slice of CandidateMatch's interface that the reference's own TraceLogger reads (trace.py:132-212),
    #    so that either logger class can be handed to process_sequences
    class _Dist:
        def __init__(self, d):
            self._d = d

        def distance(self):
            return self._d

    @property
    def candidate_match_id(self):
        return self.cid

    @property
    def p1_match(self):
        return _Cand._Dist(self.p1d) if self.p1 else None

    @property
    def p2_match(self):
        return _Cand._Dist(self.p2d) if self.p2 else None

    def get_p1(self):
        return self.p1

    def get_p2(self):
        return self.p2

    def has_b1_match(self):
        return bool(self.b1)

    def has_b2_match(self):
        return bool(self.b2)

    def b1_distance(self):
        return self.b1d()

    def b2_distance(self):
        return self.b2d()

    def b1d(self):
        return self.b1[0][1] if self.b1 else -1

    def b2d(self):
        return self.b2[0][1] if self.b2 else -1

    def best_b1(self):
        return [b for b, d in self.b1 if d == self.b1[0][1]] if self.b1 else []

    def best_b2(self):
        return [b for b, d in self.b2 if d == self.b2[0][1]] if self.b2 else []

    def full(self):
        return bool(self.p1 and self.p2 and self.b1 and self.b2)

    def score(self):   # demultiplex.py:226-236
        p1, p2, b1, b2 = bool(self.p1), bool(self.p2), bool(self.b1), bool(self.b2)
        if p1 and p2 and b1 and b2:
            return 5
        if p1 and p2 and (b1 or b2):
            return 4
        if (p1 or p2) and (b1 or b2):
            return 3
        if p1 and p2:
            return 2
        return 1 if (p1 or p2) else 0

    def info(self):   # trace.py:169-212
        p1 = self.p1.name if self.p1 else "none"
        p2 = self.p2.name if self.p2 else "none"
        b1 = self.best_b1()[0] if self.b1 else "none"
        b2 = self.best_b2()[0] if self.b2 else "none"
        pres = "both" if self.b1 and self.b2 else "forward_only" if self.b1 else "reverse_only" if self.b2 else "none"
        total = sum(d for d in (self.p1d, self.p2d, self.b1d(), self.b2d()) if d >= 0)
        return self.cid or "unknown", p1, p2, b1, b2, pres, total, self.p1d, self.p2d, self.b1d(), self.b2d()

    def extent(self, trim, b_len):
        """models.py:278-319 on the stored first locations (in this candidate's orientation), after `cum`."""
        L = self.L
        s, e = 0, L
        if trim == TrimMode.BARCODES:
            if self.p1:
                s = (L - int(self.h1["first_end"]) - 1) - self.cum
            if self.p2:
                e = (int(self.h2["first_end"]) + 1) - self.cum
        elif trim in (TrimMode.PRIMERS, TrimMode.TAILS):
            ps, pe = 0, L
            if self.p1:
                ps = (L - int(self.h1["first_start"]) - 1) + 1 - self.cum
            if self.p2:
                pe = int(self.h2["first_start"]) - self.cum
            if trim == TrimMode.PRIMERS:
                s, e = ps, pe
            else:
                s = (L - int(self.h1["tail_end"]) - 1) - self.cum if self.b1 else max(0, ps - b_len)
                e = (int(self.h2["tail_end"]) + 1) - self.cum if self.b2 else min(L, pe + b_len)
        return s, e


_ORIENT = {1: "forward", 2: "reverse", 3: "unknown"}


class BatchReplayer:
    """Everything `replay` needs that does not change within a run."""

    def __init__(self, panel, parameters, specimens, args, prefilter_on: bool):
        self.panel, self.par, self.specimens, self.args = panel, parameters, specimens, args
        self.prefilter_on = prefilter_on
        self.primers = panel.primers
        self.pidx = {id(p): i for i, p in enumerate(self.primers)}
        self.fwd = specimens.get_primers(Primer.FWD)
        self.rev = specimens.get_primers(Primer.REV)
        self.trim = getattr(args, "trim", TrimMode.BARCODES)
        self.derep_best = getattr(args, "dereplicate", "best") == "best"
        self.b_len = specimens.b_length()
        self.pf_min = (len(panel.barcodes[0]) - parameters.max_dist_index) if prefilter_on else 0
        self._collect = None

    # ---- find_candidate_matches (demultiplex.py:668-746) + match_one_end (:748-820)
    def _end_events(self, tl, sid, L, seq_str, primer, which, hit, bd):
        S = self.par.search_len
        wdir = "forward" if which == 1 else "reverse"
        tl.log_primer_search(sid, primer.name, wdir, L - S, L, False, -1, -1)
        if hit["pdist"] < 0:
            tl.log_primer_search(sid, primer.name, wdir, L - S, L, False, -1, -1)
            return
        tl.log_primer_search(sid, primer.name, wdir, L - S, L, True, int(hit["pdist"]), int(hit["first_start"]))
        if tl.verbosity < 3:
            return
        # level 3: every (barcode, optimal primer location) attempt; per-location results from the device aligner
        from .alignment import align_seq
        pm = align_seq(primer.primer_rc, seq_str, self.par.max_dist_primers[primer.primer], L - S, L)
        for b in primer.barcodes:
            b_rc = reverse_complement(b)
            for loc in pm.locations():
                start = loc[1] + 1
                tl.log_barcode_search(sid, b, wdir, primer.name, start, L, False, -1, -1)
                if self.prefilter_on:
                    # BloomPrefilter.match as its exact set: the target must start with L - k plain bases
                    x = seq_str[start:][:self.pf_min]
                    if len(x) < self.pf_min or any(ch not in "ACGT" for ch in x):
                        continue
                bm = align_seq(b_rc, seq_str, self.par.max_dist_index, start, L, AlignMode.PREFIX)
                if bm.matched():
                    tl.log_barcode_search(sid, b, wdir, primer.name, start, L, True, bm.distance(),
                                          bm.location()[0] if bm.locations() else -1)

    def _barcodes(self, primer, bd_row):
        out = [(b, int(bd_row[i])) for i, b in enumerate(primer.barcodes) if bd_row[i] >= 0]
        out.sort(key=lambda x: x[1])   # stable, like the reference's re-sort on every insertion
        return out

    def replay(self, tl: TraceLogger, record, seq_str, sid, hits, bdist, rec_ops, collect=None):
        """Log the events of one read; returns nothing.  hits: smx_hit[2*NP], bdist: int8[2*NP][maxB],
        rec_ops: the kernel's records for this read in emission order (cross-check; None: no cross-check).
        collect (list): receives one (candidate, sample_id, ResolutionType, trim_start, trim_end, trim_was_empty,
        ResolutionType asked for) per emitted record -- the host-evaluated prefilter path builds its write operations from these."""
        self._collect = collect
        L = len(seq_str)
        tl.log_sequence_received(sid, L, record.id)
        a = self.args
        if getattr(a, "min_length", -1) != -1 and L < a.min_length:
            tl.log_sequence_filtered(sid, L, "too_short")
            return
        if getattr(a, "max_length", -1) != -1 and L > a.max_length:
            tl.log_sequence_filtered(sid, L, "too_long")
            return
        rs_str = reverse_complement(seq_str) if tl.verbosity >= 3 else None
        # orientation votes: fwd primer in A / rev primer in B vote "forward" 
        ori = 3
        if self.par.preorient:
            f = r = 0
            for p in self.fwd + self.rev:
                i = self.pidx[id(p)]
                va, vb = int(hits[2 * i]["flags"]) & 1, int(hits[2 * i + 1]["flags"]) & 1
                if p.direction == Primer.FWD:
                    f += va; r += vb
                else:
                    f += vb; r += va
            ori = 1 if (f > 0 and r == 0) else 2 if (r > 0 and f == 0) else 3
            conf = abs(f - r) / (f + r) if f + r > 0 else 0.0
            tl.log_orientation_detected(sid, _ORIENT[ori], f, r, conf)
        else:
            tl.log_orientation_detected(sid, "unknown", 0, 0, 0.0)
        cands: List[_Cand] = []
        pair_no = 0
        for fp in self.fwd:
            for rp in self.specimens.get_paired_primers(fp.primer):
                fi, ri = self.pidx[id(fp)], self.pidx[id(rp)]
                for o in (0, 1):
                    if (o == 0 and ori == 2) or (o == 1 and ori == 1):
                        continue
                    c = _Cand(f"{sid}_match_{len(cands)}", o, L)
                    c.pair = pair_no
                    h1i, h2i = fi * 2 + (0 if o == 0 else 1), ri * 2 + (1 if o == 0 else 0)
                    # as read: fwd primer searched in rs, rev primer in s; reverse complement: the other way round
                    self._end_events(tl, sid, L, (rs_str if o == 0 else seq_str), fp, 1, hits[h1i], bdist[h1i])
                    if hits[h1i]["pdist"] >= 0:
                        c.p1, c.h1, c.p1d = fp, hits[h1i], int(hits[h1i]["pdist"])
                        c.b1 = self._barcodes(fp, bdist[h1i])
                    self._end_events(tl, sid, L, (seq_str if o == 0 else rs_str), rp, 2, hits[h2i], bdist[h2i])
                    if hits[h2i]["pdist"] >= 0:
                        c.p2, c.h2, c.p2d = rp, hits[h2i], int(hits[h2i]["pdist"])
                        c.b2 = self._barcodes(rp, bdist[h2i])
                    if c.p1 or c.p2:
                        common = set(fp.pools) & set(rp.pools)   # get_pool_from_primers (:640-665) of the ATTEMPTED pair
                        c.pool = sorted(common)[0] if common else None
                        tl.log_primer_matched(sid, c, c.pool or "none", "as_is" if o == 0 else "reverse_complement")
                        tl.log_barcode_matched(sid, c)
                        cands.append(c)
                pair_no += 1
        emitted = []   # (sample_id, ResolutionType) in emission order
        if not cands:
            tl.log_no_match_found(sid, "primer_search", "No primer matches found")
            # the reference still builds the record from an empty CandidateMatch (demultiplex.py:202-210): an empty
            # read trims to nothing there too
            emitted.append(self._emit(tl, sid, _Cand(None, 0, L), SampleId.UNKNOWN, ResolutionType.UNKNOWN))
        else:
            best = self._select_best(tl, sid, cands)
            if self.derep_best:
                for m, spec in self._dereplicate(tl, sid, best):
                    if spec is not None:
                        m.pool = self.specimens.get_specimen_pool(spec)
                        emitted.append(self._emit(tl, sid, m, spec, ResolutionType.DEREPLICATED_FULL))
                    else:
                        fid, rt = self._resolve(tl, sid, m)
                        emitted.append(self._emit(tl, sid, m, fid, rt))
            else:
                for m in best:
                    fid, rt = self._resolve(tl, sid, m)
                    emitted.append(self._emit(tl, sid, m, fid, rt))
        got = [(s_, rt_) for s_, rt_ in rec_ops] if rec_ops is not None else emitted
        if got != emitted:
            raise RuntimeError(f"trace replay disagrees with the kernel for read {record.id}: kernel {got}, replay {emitted}")

    # ---- select_best_matches (demultiplex.py:216-259)
    def _select_best(self, tl, sid, cands):
        best = max(c.score() for c in cands)
        for c in cands:
            tl.log_match_scored(sid, c, float(c.score()))
        for c in sorted(cands, key=lambda c: c.score(), reverse=True):
            if c.score() < best:
                tl.log_match_discarded(sid, c, float(c.score()), "lower_score")
        return [c for c in cands if c.score() == best]

    # ---- dereplicate_* (demultiplex.py:262-538)
    @staticmethod
    def _fidx(p, missing):
        return p.file_index if p else missing

    def _derep_partial(self, tl, sid, ms):
        groups = OrderedDict()
        for m in ms:
            if m.b1 and not m.b2:
                d, bcs = "forward", m.best_b1()
            elif m.b2 and not m.b1:
                d, bcs = "reverse", m.best_b2()
            else:
                continue
            for b in bcs:
                groups.setdefault((d, b), []).append(m)
        out = []
        for (d, b), g in groups.items():
            def key(m):
                cnt = (1 if m.p1 else 0) + (1 if m.p2 else 0)
                pd = (m.p1d if m.p1 else 0) + (m.p2d if m.p2 else 0)
                return (m.b1d() if d == "forward" else m.b2d(), -cnt, pd, self._fidx(m.p1, 0) + self._fidx(m.p2, 0))
            win = sorted(g, key=key)[0]
            out.append(win)
            tl.log_dereplicate_partial_selected(sid, d, b, len(g), key(win))
        return out

    def _derep_unknown(self, tl, sid, ms):
        def key(m):
            cnt = (1 if m.p1 else 0) + (1 if m.p2 else 0)
            pd = (m.p1d if m.p1 else 0) + (m.p2d if m.p2 else 0)
            return (-cnt, pd, self._fidx(m.p1, 999) + self._fidx(m.p2, 999))
        win = sorted(ms, key=key)[0]
        if len(ms) > 1:
            k = key(win)
            tl.log_dereplicate_unknown_selected(sid, len(ms), -k[0], k[1], k[2])
        return [win]

    def _dereplicate(self, tl, sid, ms):
        expanded = []
        for m in ms:
            if not m.full():
                expanded.append((m, None, 999, 999))
                continue
            found = False
            for b1 in m.best_b1():
                for b2 in m.best_b2():
                    spec = self.specimens.specimen_for_exact_match(b1, b2, m.p1, m.p2)
                    if spec:
                        expanded.append((m, spec, m.b1d(), m.b2d()))
                        found = True
            if not found:
                expanded.append((m, None, 999, 999))
        tl.log_dereplicate_expanded(sid, len(ms), len(expanded))
        groups = OrderedDict()
        for e in expanded:
            groups.setdefault(e[1], []).append(e)
        res = []
        for spec, g in groups.items():
            if spec is None:
                one = [e[0] for e in g if bool(e[0].b1) != bool(e[0].b2)]
                none = [e[0] for e in g if not e[0].b1 and not e[0].b2]
                both = [e[0] for e in g if e[0].b1 and e[0].b2]
                res += [(m, None) for m in self._derep_partial(tl, sid, one)] if one else []
                res += [(m, None) for m in self._derep_unknown(tl, sid, none)] if none else []
                res += [(m, None) for m in both]
                continue

            def key(e):
                return (e[2] + e[3], e[0].p1d + e[0].p2d, self._fidx(e[0].p1, 999) + self._fidx(e[0].p2, 999))
            g = sorted(g, key=key)
            res.append((g[0][0], spec))
            tl.log_dereplicate_selected(sid, spec, len(g), key(g[0]))
        return res

    # ---- resolve_specimen (demultiplex.py:541-598)
    def _resolve(self, tl, sid, m):
        spec, rt = SampleId.UNKNOWN, ResolutionType.UNKNOWN
        if m.full():
            ids = self.specimens.specimens_for_barcodes_and_primers(m.best_b1(), m.best_b2(), m.p1, m.p2)
            if len(ids) > 1:
                m.pool = self.specimens.get_specimen_pool(ids[0])
                spec, rt = ids[0], ResolutionType.MULTIPLE_SPECIMENS
            elif len(ids) == 1:
                m.pool = self.specimens.get_specimen_pool(ids[0])
                spec, rt = ids[0], ResolutionType.FULL_MATCH
        else:
            b1s, b2s = m.best_b1(), m.best_b2()
            if m.b1 and not m.b2 and len(b1s) == 1:
                spec, rt = SampleId.PREFIX_FWD_MATCH + b1s[0], ResolutionType.PARTIAL_FORWARD
            elif m.b2 and not m.b1 and len(b2s) == 1:
                spec, rt = SampleId.PREFIX_REV_MATCH + b2s[0], ResolutionType.PARTIAL_REVERSE
        tl.log_specimen_resolved(sid, m, spec, rt.to_string(), m.pool or "none")
        return spec, rt

    # ---- create_write_operation's trace side (demultiplex.py:30-103): trim-to-empty fallback, Q8 shift
    def _emit(self, tl, sid, m, sample_id, rt):
        s, e = 0, m.L
        if self.trim != TrimMode.NONE:
            s, e = m.extent(self.trim, self.b_len)
            if s >= e:
                tl.log_sequence_trim_empty(sid, self.trim, s, e, m.L, m.p1.name if m.p1 else "unknown",
                                           m.p2.name if m.p2 else "unknown")
                if getattr(self, "_collect", None) is not None:
                    self._collect.append((m, SampleId.UNKNOWN, ResolutionType.UNKNOWN, 0, m.L, True, rt))
                return SampleId.UNKNOWN, ResolutionType.UNKNOWN
            m.cum += s
        if getattr(self, "_collect", None) is not None:
            self._collect.append((m, sample_id, rt, s, e, False, rt))
        return sample_id, rt


def _level3_requests(replayer, seqs, ops, hits, cache):
    """The alignments `_end_events` will ask for at verbosity 3, for the whole batch: first every matched (read, primer,
    end) over the search window, then -- from those locations -- every (barcode, location) attempt the exact-set prefilter
    lets through.  Two launches instead of one per alignment."""
    from .alignment import AlignMode
    par, S = replayer.par, replayer.par.search_len
    stage1 = []
    ends = []   # (read index, string searched, primer, request)
    for i, seq_str in enumerate(seqs):
        if ops["rtype"][i] == _lib.R_FILTERED:
            continue
        L = len(seq_str)
        rs_str = None
        for pi, primer in enumerate(replayer.primers):
            for X in (0, 1):
                if hits[i][pi * 2 + X]["pdist"] < 0:
                    continue
                if X == 0 and rs_str is None:
                    rs_str = reverse_complement(seq_str)
                q = rs_str if X == 0 else seq_str
                s0 = L - S
                s_ = 0 if s0 == -1 else s0
                t = q[s_:L]
                if not t:
                    continue
                req = (primer.primer_rc, t, int(par.max_dist_primers[primer.primer]), AlignMode.INFIX)
                stage1.append(req)
                ends.append((q, s_, primer, req))
    cache.fill(stage1)
    stage2 = []
    for q, s_, primer, req in ends:
        dist, locs = cache.table[req]
        if dist < 0:
            continue
        L = len(q)
        for b in primer.barcodes:
            b_rc = reverse_complement(b)
            for _a, e in locs:
                start = e + s_ + 1
                if replayer.prefilter_on:
                    x = q[start:][:replayer.pf_min]
                    if len(x) < replayer.pf_min or any(ch not in "ACGT" for ch in x):
                        continue
                s2 = 0 if start == -1 else start
                t = q[s2:L][:len(b_rc) + int(par.max_dist_index)]   # what align_seq keeps of a PREFIX target
                if t:
                    stage2.append((b_rc, t, int(par.max_dist_index), AlignMode.PREFIX))
    cache.fill(stage2)


def replay_batch(tl: TraceLogger, replayer: BatchReplayer, seq_records, seqs, ops, extra, hits, bdist, op_names,
                 record_offset: int):
    """Log every read of a batch; returns {read index: sequence_id} for the write operations."""
    from .alignment import AlignCache
    by_read = {}
    for j in range(len(extra)):
        by_read.setdefault(int(extra["read"][j]), []).append(extra[j])
    ids = {}
    cache = AlignCache()
    if tl.verbosity >= 3:
        _level3_requests(replayer, seqs, ops, hits, cache)
    with cache:
        for i, record in enumerate(seq_records):
            sid = tl.get_sequence_id(record, record_offset + i)
            ids[i] = sid
            if ops["rtype"][i] == _lib.R_FILTERED:
                recs = []
            else:
                recs = [ops[i]] + by_read.get(i, [])
            rec_ops = []
            for rec in recs:
                sample, _pool, _p1, _p2, _code, rtype = op_names(replayer.panel, rec)
                rec_ops.append((sample, rtype))
            replayer.replay(tl, record, seqs[i], sid, hits[i], bdist[i], rec_ops)
    return ids


class NullTrace:
    """A trace logger that logs nothing (verbosity 0): lets BatchReplayer.replay run as a plain scorer."""
    verbosity = 0

    def __getattr__(self, name):
        if name.startswith("log_"):
            return lambda *a, **k: None
        raise AttributeError(name)
