"""oracle/specimux_oracle.py -- TEST INFRASTRUCTURE ONLY (parity oracle, never shipped).

CPU restatement, in reference loop order, of the specimux hot path and the thin
host shell around it.  Every function cites the reference lines it follows
(paths relative to /root/reference/).  The third-party aligner is restated in
oracle/edlib_semantics.py.  PINNED against the reference's own golden suite
(tests/data/integration_test_suite/expected_output, 40 reads -> 46 records) by
tests/test_oracle_golden.py, comparing FULL records (header, sequence, quality).

Deliberate, documented choices where the reference is hash-order dependent:
  * Q4: ``PrimerInfo.barcodes`` is a Python ``set`` of str in the
    reference (models.py:28); iteration order there depends on PYTHONHASHSEED.  Here the
    canonical order is first appearance in specimens.txt.
  * Q7: the Bloom prefilter (bloom_filter.py:176-186) is restated as its exact set
    (no hash false positives): key ``target[:Lb-k]`` must be the truncation of some
    string within k edits of the barcode over {A,C,G,T}.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import csv
import gzip
import itertools
import math
import os
from collections import OrderedDict

from . import edlib_semantics as E

FWD, REV = "forward", "reverse"
UNKNOWN = "unknown"
# ResolutionType (constants.py:54-60)
R_FULL, R_PFWD, R_PREV, R_MULTI, R_UNKNOWN, R_DEREP = 1, 2, 3, 4, 5, 6

_COMP = str.maketrans("ACGTMRWSYKVHDBXNUacgtmrwsykvhdbxnu", "TGCAKYWSRMBDHVXNAtgcakywsrmbdhvxna")


def revcomp(s):
    """Bio.Seq.reverse_complement (ambiguous DNA table, case preserved, U->A)."""
    return s.translate(_COMP)[::-1]


# ------------------------------------------------------------------ panel (L1 data model)
class Primer:
    """models.py:20-31"""

    def __init__(self, name, seq, direction, pools, file_index):
        self.name = name
        self.primer = seq.upper()
        self.primer_rc = revcomp(self.primer)
        self.direction = direction
        self.pools = list(pools)
        self.file_index = file_index
        self.barcodes = []      # canonical order = first appearance (Q4)
        self.specimens = set()

    def __repr__(self):
        return f"Primer({self.name})"


class Panel:
    """PrimerDatabase + Specimens (databases.py:17-275) flattened into one object."""

    def __init__(self):
        self.by_name = OrderedDict()          # registry: name -> Primer (file order)
        self.pool_primers = OrderedDict()     # pool -> {FWD: [...], REV: [...]}
        self.specimens = []                   # (id, pool, b1, [p1], b2, [p2]) file order
        self.primers = OrderedDict()          # Specimens._primers: SEQUENCE -> Primer (Q5)
        self.b_length = 0
        self.active_pools = set()
        self._pairs = {}
        self._ids = set()

    # ---- io_utils.py:270-322
    @classmethod
    def read_primers(cls, path):
        pan = cls()
        for idx, (title, seq) in enumerate(read_fasta(path)):
            fields = title.split()
            name = fields[0]
            pools, pos = [], None
            for f in fields:
                if f.startswith("pool="):
                    pools = [p.strip() for p in f[5:].replace(";", ",").split(",")]
                elif f.startswith("position="):
                    pos = f[9:]
            if not pools:
                raise ValueError(f"Missing pool specification for primer {name}")
            if pos not in (FWD, REV):
                raise ValueError(f"Invalid primer position '{pos}' for {name}")
            if name in pan.by_name:
                raise ValueError(f"Duplicate primer name: {name}")
            pr = Primer(name, seq, pos, pools, idx)
            pan.by_name[name] = pr
            for pool in pools:
                pan.pool_primers.setdefault(pool, {FWD: [], REV: []})[pos].append(pr)
        for pool, d in pan.pool_primers.items():  # databases.py:91-103
            if not d[FWD]:
                raise ValueError(f"Pool {pool} has no forward primers")
            if not d[REV]:
                raise ValueError(f"Pool {pool} has no reverse primers")
        return pan

    # ---- databases.py:195-217
    def _resolve(self, name, pool, direction):
        if name in ("-", "*"):
            d = self.pool_primers.get(pool, {FWD: [], REV: []})
            got = [p for p in d[FWD] + d[REV] if p.direction == direction]
            if not got:
                raise ValueError(f"No primers found in pool {pool}")
            return got
        pr = self.by_name.get(name)
        if pr is None:
            raise ValueError(f"Primer not found: {name}")
        if pr.direction != direction:
            raise ValueError(f"Primer {name} has the wrong direction")
        if pool not in self.pool_primers or pr not in (self.pool_primers[pool][FWD] + self.pool_primers[pool][REV]):
            raise ValueError(f"Primer {name} is not in pool {pool}")
        return [pr]

    # ---- databases.py:135-167
    def add_specimen(self, sid, pool, b1, p1, b2, p2):
        if sid in self._ids:
            raise ValueError(f"Duplicate specimen id in index file: {sid}")
        self._ids.add(sid)
        self.active_pools.add(pool)
        self.b_length = max(self.b_length, len(b1), len(b2))
        p1s = self._resolve(p1, pool, FWD)
        p2s = self._resolve(p2, pool, REV)
        for plist, bc in ((p1s, b1), (p2s, b2)):
            for info in plist:
                reg = self.primers.setdefault(info.primer, info)  # keyed by SEQUENCE (Q5)
                if bc not in reg.barcodes:
                    reg.barcodes.append(bc)
                reg.specimens.add(sid)
        self.specimens.append((sid, pool, b1, p1s, b2, p2s))

    # ---- io_utils.py:324-377
    def read_specimens(self, path):
        need = {"SampleID", "PrimerPool", "FwIndex", "FwPrimer", "RvIndex", "RvPrimer"}
        with open(path, newline="") as fh:
            rd = csv.DictReader(fh, delimiter="\t")
            if need - set(rd.fieldnames or []):
                raise ValueError("Missing required columns in specimen file")
            empties = []
            for n, row in enumerate(rd, start=1):
                b1, b2 = row["FwIndex"].upper(), row["RvIndex"].upper()
                if not b1.strip() or not b2.strip():
                    empties.append(n)
                    continue
                self.add_specimen(row["SampleID"], row["PrimerPool"], b1, row["FwPrimer"], b2, row["RvPrimer"])
            if empties:
                raise ValueError(f"Empty barcodes found in {len(empties)} specimen(s)")
        if not self.specimens:
            raise ValueError("No valid data found in the specimen file")
        return self

    # ---- databases.py:169-193 (validate -> prune_unused_pools)
    def validate(self):
        unused = set(self.pool_primers) - self.active_pools
        if unused:
            for pr in self.primers.values():
                pr.pools = [p for p in pr.pools if p in self.active_pools]
            for pool in unused:
                del self.pool_primers[pool]
        return self

    def get_primers(self, direction):  # databases.py:247-249
        return [p for p in self.primers.values() if p.direction == direction]

    def get_paired(self, primer_seq):  # databases.py:251-264
        if primer_seq not in self._pairs:
            me = self.primers[primer_seq]
            self._pairs[primer_seq] = [p for p in self.primers.values()
                                       if p.direction != me.direction and p.specimens & me.specimens]
        return self._pairs[primer_seq]

    def specimen_for_exact(self, b1, b2, p1, p2):  # databases.py:232-245
        for sid, _pool, sb1, p1s, sb2, p2s in self.specimens:
            if p1 in p1s and p2 in p2s and sb1.upper() == b1.upper() and sb2.upper() == b2.upper():
                return sid
        return None

    def specimens_for(self, b1s, b2s, p1, p2):  # databases.py:219-230
        return [sid for sid, _pool, b1, p1s, b2, p2s in self.specimens
                if p1 in p1s and p2 in p2s and b1.upper() in b1s and b2.upper() in b2s]

    def specimen_pool(self, sid):  # databases.py:266-271
        for s in self.specimens:
            if s[0] == sid:
                return s[1]
        return None


def load_panel(primer_file, specimen_file):
    """orchestration.py:154-156"""
    return Panel.read_primers(primer_file).read_specimens(specimen_file).validate()


# ------------------------------------------------------------------ parameters
class Params:
    """MatchParameters (models.py:331-338) + the args flags the hot path reads."""

    def __init__(self, max_dist_primers, max_dist_index, search_len=80, preorient=True,
                 prefilter=True, trim="barcodes", dereplicate="best", min_length=-1, max_length=-1):
        self.max_dist_primers = max_dist_primers  # keyed by primer SEQUENCE (orchestration.py:605-614)
        self.max_dist_index = max_dist_index
        self.search_len = search_len
        self.preorient = preorient
        self.prefilter = prefilter
        self.trim = trim
        self.dereplicate = dereplicate
        self.min_length = min_length
        self.max_length = max_length


def bp_adjusted_length(primer):  # orchestration.py:564-570
    score = 0
    for b in primer:
        if b in "ACGT":
            score += 3
        elif b in "KMRSWY":
            score += 2
        elif b in "BDHV":
            score += 1
    return score / 3.0


def setup_params(panel, search_len=80, index_edit_distance=-1, primer_edit_distance=-1, **kw):
    """orchestration.py:548-628: k_idx = ceil(min NW(all b1 + rc(all b2)) / 2); k_p per primer."""
    b1s, b2s = [], []
    for p in panel.get_primers(FWD):
        b1s += [b for b in p.barcodes if b not in b1s]
    for p in panel.get_primers(REV):
        b2s += [b for b in p.barcodes if b not in b2s]
    combined = b1s + [revcomp(b) for b in b2s]
    if len(combined) <= 1:
        raise TypeError("need at least two barcodes to derive the index threshold (Q14)")
    min_bc = min(E.align(a, b, E.NW, -1, iupac=False)["editDistance"]
                 for a, b in itertools.combinations(combined, 2))
    k_idx = math.ceil(min_bc / 2.0) if index_edit_distance == -1 else index_edit_distance
    thr = {}
    for p in panel.get_primers(FWD) + panel.get_primers(REV):
        thr[p.primer] = primer_edit_distance if primer_edit_distance != -1 else int(bp_adjusted_length(p.primer) / 3)
    return Params(thr, k_idx, search_len, **kw)


# ------------------------------------------------------------------ alignment wrapper
class Aln:
    """AlignmentResult (models.py:34-69) without the dict."""
    __slots__ = ("dist", "locs")

    def __init__(self, dist, locs):
        self.dist, self.locs = dist, list(locs)

    def matched(self):
        return self.dist > -1

    def reversed(self, L):  # models.py:52-63
        if self.dist == -1:
            return Aln(self.dist, self.locs)
        return Aln(self.dist, [(L - b - 1, L - a - 1) for a, b in self.locs])

    def shift(self, s):  # models.py:65-69 (in place)
        if self.dist != -1:
            self.locs = [(a + s, b + s) for a, b in self.locs]


def align_seq(query, target, max_distance, start, end, mode=E.HW):
    """alignment.py:21-50, including Python slice semantics for negative starts (Q1)."""
    s = 0 if start == -1 else start
    e = len(target) if end == -1 else min(end, len(target))
    r = E.align(query, target[s:e], mode, max_distance, iupac=True)
    d = r["editDistance"]
    if d != -1 and d > max_distance:  # alignment.py:44-46 (empty target: A.4)
        d = -1
    a = Aln(d, r["locations"])
    a.shift(s)
    return a


# ------------------------------------------------------------------ prefilter (Q7)
class ExactPrefilter:
    """BloomPrefilter.match (bloom_filter.py:176-186) restated as its exact set.

    The filter holds ``barcode + v[:L-k]`` for every string v within k edits of the
    barcode over {A,C,G,T} (bloom_filter.py:57-66,70-101); every v has len >= L-k.  So
    ``barcode + x`` (x = target[:L-k]) is a member iff len(x) == L-k, x is over ACGT and
    some completion x+w is within k edits of the barcode.  The cheapest completion of a
    prefix alignment is free (w := the unaligned barcode tail), hence
        member(x)  <=>  min_j NW(x, barcode[:j]) <= k  ==  SHW(query=x, target=barcode) <= k.
    No hash, so none of the reference's 5 % Bloom false positives (they are resolved by
    the aligner anyway and only matter for non-ACGT targets: Q7)."""

    def __init__(self, barcodes_rc, k):
        self.barcodes = list(dict.fromkeys(barcodes_rc))
        self.k = k
        self.min_length = len(self.barcodes[0]) - k  # bloom_filter.py:41-44

    def match(self, barcode, sequence):
        if barcode not in self.barcodes:
            return True
        x = sequence[:self.min_length]
        if len(x) < self.min_length or any(c not in "ACGT" for c in x):
            return False
        return E.align(x, barcode, E.SHW, -1, iupac=False)["editDistance"] <= self.k


# ------------------------------------------------------------------ candidate state
class Cand:
    """CandidateMatch (models.py:72-328)."""

    def __init__(self, seq, oriented_reverse, b_len, cid=None):
        self.cid = cid                  # candidate_match_id (trace only)
        self.seq = seq                  # (id, bases, quals) in this candidate's orientation
        self.rev = oriented_reverse
        self.L = len(seq[1])
        self.p1 = self.p2 = None        # Primer
        self.p1m = self.p2m = None      # Aln
        self.b1 = []                    # [(barcode, Aln, dist)] stable-sorted by dist
        self.b2 = []
        self.pool = None
        self.b_len = b_len

    def add_barcode(self, aln, bc, reverse, which):  # models.py:97-108
        m = aln.reversed(self.L) if reverse else aln
        lst = self.b1 if which == 1 else self.b2
        lst.append((bc, m, m.dist))
        lst.sort(key=lambda x: x[2])

    def set_primer(self, aln, primer, reverse, which):  # models.py:128-137
        m = aln.reversed(self.L) if reverse else aln
        if which == 1:
            self.p1m, self.p1 = m, primer
        else:
            self.p2m, self.p2 = m, primer

    def b1d(self):
        return self.b1[0][2] if self.b1 else -1

    def b2d(self):
        return self.b2[0][2] if self.b2 else -1

    def best_b1(self):  # models.py:116-120 (tolerance 1.0 on ints == equality)
        return [b for b, _, d in self.b1 if d == self.b1d()] if self.b1 else []

    def best_b2(self):
        return [b for b, _, d in self.b2 if d == self.b2d()] if self.b2 else []

    def p1d(self):
        return self.p1m.dist if self.p1m else -1

    def p2d(self):
        return self.p2m.dist if self.p2m else -1

    def full(self):
        return bool(self.p1m and self.p2m and self.b1 and self.b2)

    def code(self):  # models.py:206-218
        return ",".join(str(d) if d >= 0 else "X" for d in (self.p1d(), self.b1d(), self.b2d(), self.p2d()))

    def extent(self, mode):  # models.py:278-319
        s, e = 0, self.L
        if mode == "primers":
            if self.p1m:
                s = self.p1m.locs[0][1] + 1
            if self.p2m:
                e = self.p2m.locs[0][0]
        elif mode == "barcodes":
            if self.p1m:
                s = self.p1m.locs[0][0]
            if self.p2m:
                e = self.p2m.locs[0][1] + 1
        elif mode == "tails":
            ps, pe = self.extent("primers")
            s = e = -1
            for _, m, _d in self.b1:
                for l in m.locs:
                    s = l[0] if s == -1 else min(s, l[0])
            for _, m, _d in self.b2:
                for l in m.locs:
                    e = l[1] + 1 if e == -1 else max(e, l[1] + 1)
            if s == -1:
                s = max(0, ps - self.b_len)
            if e == -1:
                e = min(self.L, pe + self.b_len)
        return s, e

    def trim_locations(self, start):  # models.py:321-328 (in place: Q8)
        for _, m, _d in self.b1 + self.b2:
            m.shift(-start)
        if self.p1m:
            self.p1m.shift(-start)
        if self.p2m:
            self.p2m.shift(-start)


# ------------------------------------------------------------------ trace events (trace.py:24-334)
class Tracer:
    """TraceLogger restated as an in-memory event list: rows are the TSV columns after the timestamp,
    i.e. [worker_id, event_seq, sequence_id, event_type, *fields], every field already str()-ed the way
    csv.writer would write it.  Verbosity rules: trace.py:318-334."""

    def __init__(self, level=1, worker_id="main"):
        self.level, self.worker_id, self.rows = level, worker_id, []

    def seq_id(self, rec_id, num):  # trace.py:109-116
        return f"{rec_id}#{num:08d}#{self.worker_id}"

    def log(self, sid, etype, *fields):  # trace.py:96-107
        self.rows.append([self.worker_id, str(len(self.rows) + 1), sid, etype] + [str(f) for f in fields])

    @staticmethod
    def info(c):  # trace.py:169-212 (_extract_match_info)
        p1 = c.p1.name if c.p1 else "none"
        p2 = c.p2.name if c.p2 else "none"
        b1 = c.best_b1()[0] if c.best_b1() else "none"
        b2 = c.best_b2()[0] if c.best_b2() else "none"
        presence = "both" if c.b1 and c.b2 else "forward_only" if c.b1 else "reverse_only" if c.b2 else "none"
        total = sum(d for d in (c.p1d(), c.p2d(), c.b1d(), c.b2d()) if d >= 0)
        return c.cid or "unknown", p1, p2, b1, b2, presence, total, c.p1d(), c.p2d(), c.b1d(), c.b2d()

    def primer_search(self, sid, name, direction, a, b, found, dist, pos):  # trace.py:318-325
        if self.level >= 2 and (self.level >= 3 or found):
            self.log(sid, "PRIMER_SEARCH", name, direction, a, b, str(found).lower(), dist, pos)

    def barcode_search(self, sid, bc, btype, primer, a, b, found, dist, pos):  # trace.py:327-334
        if self.level >= 3:
            self.log(sid, "BARCODE_SEARCH", bc, btype, primer, a, b, str(found).lower(), dist, pos)


# ------------------------------------------------------------------ the hot path
def match_one_end(prefilter, cand, par, sequence, reversed_sequence, primer, which, hits=None, tr=None, sid=None):
    """demultiplex.py:748-820"""
    L = len(sequence)
    wdir = "forward" if which == 1 else "reverse"   # Primer.to_string / Barcode.to_string (constants.py:91-118)
    if tr:
        tr.primer_search(sid, primer.name, wdir, L - par.search_len, L, False, -1, -1)
    pm = align_seq(primer.primer_rc, sequence, par.max_dist_primers[primer.primer], L - par.search_len, L)
    if hits is not None:
        hits.append(("P", primer.name, "A" if reversed_sequence ^ cand.rev else "B", pm.dist, list(pm.locs)))
    if not pm.matched():
        if tr:
            tr.primer_search(sid, primer.name, wdir, L - par.search_len, L, False, -1, -1)
        return
    pos = pm.locs[0][0] if pm.locs else -1   # read before set_primer: the reference logs the unreversed object
    cand.set_primer(pm, primer, reversed_sequence, which)
    if tr:
        tr.primer_search(sid, primer.name, wdir, L - par.search_len, L, True, pm.dist, pos)
    for b in primer.barcodes:
        b_rc = revcomp(b)
        best = None
        for loc in pm.locs:
            start = loc[1] + 1
            if tr:
                tr.barcode_search(sid, b, wdir, primer.name, start, L, False, -1, -1)
            if prefilter and not prefilter.match(b_rc, sequence[start:]):
                continue
            bm = align_seq(b_rc, sequence, par.max_dist_index, start, L, E.SHW)
            if bm.matched():
                if tr:
                    tr.barcode_search(sid, b, wdir, primer.name, start, L, True, bm.dist, bm.locs[0][0] if bm.locs else -1)
                if best is None or bm.dist < best.dist:
                    best = bm
        if best is not None:
            cand.add_barcode(best, b, reversed_sequence, which)


def determine_orientation(par, s, rs, fwd, rev, counts=False):
    """demultiplex.py:602-638 -> 'F', 'R' or 'U' (with counts=True: (orientation, fwd_score, rev_score))"""
    f = r = 0
    for p in fwd:
        k = par.max_dist_primers[p.primer]
        f += align_seq(p.primer, s, k, 0, par.search_len).matched()
        r += align_seq(p.primer, rs, k, 0, par.search_len).matched()
    for p in rev:
        k = par.max_dist_primers[p.primer]
        f += align_seq(p.primer, rs, k, 0, par.search_len).matched()
        r += align_seq(p.primer, s, k, 0, par.search_len).matched()
    o = "F" if f > 0 and r == 0 else "R" if r > 0 and f == 0 else "U"
    return (o, f, r) if counts else o


def pool_from_primers(p1, p2):  # demultiplex.py:640-665
    if p1 and p2:
        common = set(p1.pools) & set(p2.pools)
        return sorted(common)[0] if common else None
    if p1:
        return sorted(p1.pools)[0] if p1.pools else None
    if p2:
        return sorted(p2.pools)[0] if p2.pools else None
    return None


_ORI_NAME = {"F": "forward", "R": "reverse", "U": "unknown"}   # Orientation.to_string (constants.py:121-129)


def find_candidates(prefilter, par, panel, seq, rseq, tr=None, sid=None):
    """demultiplex.py:668-746"""
    s, rs = seq[1], rseq[1]
    if par.preorient:
        ori, fs, rv = determine_orientation(par, s, rs, panel.get_primers(FWD), panel.get_primers(REV), counts=True)
        if tr:
            conf = abs(fs - rv) / (fs + rv) if fs + rv > 0 else 0.0
            tr.log(sid, "ORIENTATION_DETECTED", _ORI_NAME[ori], fs, rv, f"{conf:.3f}")
    else:
        ori = "U"
        if tr:
            tr.log(sid, "ORIENTATION_DETECTED", "unknown", 0, 0, f"{0.0:.3f}")
    out = []

    def logged(c, pool, used):   # trace.py:132-167
        if not tr:
            return
        cid, p1, p2, b1, b2, _pres, _tot, p1d, p2d, b1d, b2d = Tracer.info(c)
        mtype = "both" if c.p1m and c.p2m else "forward_only" if c.p1m else "reverse_only"
        tr.log(sid, "PRIMER_MATCHED", cid, mtype, p1, p2, p1d, p2d, pool or "none", used)
        btype = "both" if c.b1 and c.b2 else "forward_only" if c.b1 else "reverse_only" if c.b2 else "none"
        tr.log(sid, "BARCODE_MATCHED", cid, btype, b1, b2, b1d, b2d, p1, p2)

    for fp in panel.get_primers(FWD):
        for rp in panel.get_paired(fp.primer):
            if ori in "FU":
                c = Cand(seq, False, panel.b_length, f"{sid}_match_{len(out)}")
                match_one_end(prefilter, c, par, rs, True, fp, 1, tr=tr, sid=sid)
                match_one_end(prefilter, c, par, s, False, rp, 2, tr=tr, sid=sid)
                if c.p1m or c.p2m:
                    c.pool = pool_from_primers(fp, rp)
                    logged(c, c.pool, "as_is")
                    out.append(c)
            if ori in "RU":
                c = Cand(rseq, True, panel.b_length, f"{sid}_match_{len(out)}")
                match_one_end(prefilter, c, par, s, True, fp, 1, tr=tr, sid=sid)
                match_one_end(prefilter, c, par, rs, False, rp, 2, tr=tr, sid=sid)
                if c.p1m or c.p2m:
                    c.pool = pool_from_primers(fp, rp)
                    logged(c, c.pool, "reverse_complement")
                    out.append(c)
    return out


def score(c):  # demultiplex.py:226-236
    p1, p2, b1, b2 = bool(c.p1m), bool(c.p2m), bool(c.b1), bool(c.b2)
    if p1 and p2 and b1 and b2:
        return 5
    if p1 and p2 and (b1 or b2):
        return 4
    if (p1 or p2) and (b1 or b2):
        return 3
    if p1 and p2:
        return 2
    if p1 or p2:
        return 1
    return 0


def select_best(cands, tr=None, sid=None):  # demultiplex.py:216-259 (stable)
    best = max(score(c) for c in cands)
    if tr:
        for c in cands:   # trace.py:214-222
            cid, p1, p2, b1, b2, pres, tot = Tracer.info(c)[:7]
            tr.log(sid, "MATCH_SCORED", cid, p1, p2, b1, b2, tot, pres, f"{float(score(c)):.3f}")
        for c in sorted(cands, key=score, reverse=True):   # stable: the reference sorts before discarding
            if score(c) < best:
                cid, p1, p2, b1, b2 = Tracer.info(c)[:5]
                tr.log(sid, "MATCH_DISCARDED", cid, p1, p2, b1, b2, float(score(c)), "lower_score")
    return [c for c in cands if score(c) == best]


def _fidx(p, missing):
    return p.file_index if p else missing


def derep_partial(ms, tr=None, sid=None):  # demultiplex.py:396-477
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
            pd = (m.p1d() if m.p1 else 0) + (m.p2d() if m.p2 else 0)
            fi = _fidx(m.p1, 0) + _fidx(m.p2, 0)
            return (m.b1d() if d == "forward" else m.b2d(), -cnt, pd, fi)
        win = sorted(g, key=key)[0]
        out.append(win)
        if tr:   # trace.py:283-297
            k = key(win)
            tr.log(sid, "DEREPLICATE_PARTIAL_SELECTED", d, b, len(g), k[0], -k[1], k[2], k[3])
    return out


def derep_unknown(ms, tr=None, sid=None):  # demultiplex.py:480-538
    if not ms:
        return []

    def key(m):
        cnt = (1 if m.p1 else 0) + (1 if m.p2 else 0)
        pd = (m.p1d() if m.p1 else 0) + (m.p2d() if m.p2 else 0)
        return (-cnt, pd, _fidx(m.p1, 999) + _fidx(m.p2, 999))
    win = sorted(ms, key=key)[0]
    if tr and len(ms) > 1:   # demultiplex.py:533-536
        k = key(win)
        tr.log(sid, "DEREPLICATE_UNKNOWN_SELECTED", len(ms), -k[0], k[1], k[2])
    return [win]


def dereplicate(ms, panel, tr=None, sid=None):  # demultiplex.py:262-393
    expanded = []
    for m in ms:
        if not m.full():
            expanded.append((m, None, None, None, 999, 999))
            continue
        found = False
        for b1 in m.best_b1():
            for b2 in m.best_b2():
                spec = panel.specimen_for_exact(b1, b2, m.p1, m.p2)
                if spec:
                    expanded.append((m, spec, b1, b2, m.b1d(), m.b2d()))
                    found = True
        if not found:
            expanded.append((m, None, None, None, 999, 999))
    if tr:
        tr.log(sid, "DEREPLICATE_EXPANDED", len(ms), len(expanded))
    groups = OrderedDict()
    for e in expanded:
        groups.setdefault(e[1], []).append(e)
    res = []
    for spec, g in groups.items():
        if spec is None:
            one = [e[0] for e in g if bool(e[0].b1) != bool(e[0].b2)]
            none = [e[0] for e in g if not e[0].b1 and not e[0].b2]
            both = [e[0] for e in g if e[0].b1 and e[0].b2]
            res += [(m, None, None, None) for m in derep_partial(one, tr, sid)] if one else []
            res += [(m, None, None, None) for m in derep_unknown(none, tr, sid)] if none else []
            res += [(m, None, None, None) for m in both]
            continue

        def key(e):
            return (e[4] + e[5], e[0].p1d() + e[0].p2d(), _fidx(e[0].p1, 999) + _fidx(e[0].p2, 999))
        g = sorted(g, key=key)
        res.append((g[0][0], spec, g[0][2], g[0][3]))
        if tr:   # demultiplex.py:385-391
            k = key(g[0])
            tr.log(sid, "DEREPLICATE_SELECTED", spec, len(g), k[0], k[1], k[2])
    return res


_RTYPE_NAME = {R_FULL: "full_match", R_PFWD: "partial_forward", R_PREV: "partial_reverse", R_MULTI: "multiple_specimens",
               R_UNKNOWN: "unknown", R_DEREP: "dereplicated_full"}   # ResolutionType.to_string (constants.py:62-75)


def resolve_specimen(m, panel, tr=None, sid=None):  # demultiplex.py:541-598
    spec, rt = UNKNOWN, R_UNKNOWN
    if m.full():
        ids = panel.specimens_for(m.best_b1(), m.best_b2(), m.p1, m.p2)
        if len(ids) > 1:
            m.pool = panel.specimen_pool(ids[0])
            spec, rt = ids[0], R_MULTI
        elif len(ids) == 1:
            m.pool = panel.specimen_pool(ids[0])
            spec, rt = ids[0], R_FULL
    else:
        b1s, b2s = m.best_b1(), m.best_b2()
        if m.b1 and not m.b2 and len(b1s) == 1:
            spec, rt = "barcode_fwd_" + b1s[0], R_PFWD
        elif m.b2 and not m.b1 and len(b2s) == 1:
            spec, rt = "barcode_rev_" + b2s[0], R_PREV
    if tr:   # demultiplex.py:592-594, trace.py:233-239
        _cid, p1, p2, b1, b2 = Tracer.info(m)[:5]
        tr.log(sid, "SPECIMEN_RESOLVED", spec, _RTYPE_NAME[rt], m.pool or "none", p1, p2, b1, b2)
    return spec, rt


class Op:
    """WriteOperation (models.py:341-357), only the fields that reach a file/stdout."""
    __slots__ = ("sample_id", "seq_id", "code", "sequence", "quality", "pool", "p1", "p2", "rtype",
                 "p1_loc", "p2_loc", "b1_loc", "b2_loc", "trim", "reverse", "trace_id")

    def key(self):
        return (self.seq_id, self.sample_id, self.code, self.pool, self.p1, self.p2, self.rtype,
                self.sequence, self.quality)


def make_op(sample_id, par, m, rtype, tr=None, sid=None):  # demultiplex.py:30-103
    rid, bases, quals = m.seq
    op = Op()
    op.seq_id, op.code, op.trace_id = rid, m.code(), sid
    op.trim, op.reverse = (0, len(bases)), m.rev     # extent actually cut from the oriented read (tests only)
    fallback = False
    if par.trim != "none":
        s, e = m.extent(par.trim)
        if s >= e:  # Q12: would trim to nothing -> untrimmed record to unknown/unknown/unknown-unknown
            fallback = True
            if tr:   # demultiplex.py:50-54
                tr.log(sid, "SEQUENCE_TRIM_EMPTY", par.trim, s, e, len(bases), m.p1.name if m.p1 else "unknown",
                       m.p2.name if m.p2 else "unknown")
        else:
            bases, quals = bases[s:e], quals[s:e]
            op.trim = (s, e)
            m.trim_locations(s)  # Q8: mutates the candidate for any later emission
    op.p1_loc = m.p1m.locs[0] if m.p1m else None
    op.p2_loc = m.p2m.locs[0] if m.p2m else None
    op.b1_loc = m.b1[0][1].locs[0] if m.b1 else None
    op.b2_loc = m.b2[0][1].locs[0] if m.b2 else None
    op.sequence, op.quality = bases, quals
    if fallback:
        op.sample_id, op.pool, op.p1, op.p2, op.rtype = UNKNOWN, "unknown", "unknown", "unknown", R_UNKNOWN
        return op
    op.sample_id = sample_id
    op.p1 = m.p1.name if m.p1 else "unknown"
    op.p2 = m.p2.name if m.p2 else "unknown"
    op.pool = m.pool if m.pool else "unknown"
    op.rtype = rtype
    return op


def process_sequences(records, par, panel, prefilter="auto", tr=None, record_offset=0):
    """demultiplex.py:108-212.  records: iterable of (id, bases, quality_string).
    Returns (ops, total, matched).  tr: optional Tracer collecting the reference's trace events."""
    if prefilter == "auto":
        prefilter = make_prefilter(panel, par) if par.prefilter else None
    ops, total, matched = [], 0, 0
    for idx, rec in enumerate(records):
        total += 1
        L = len(rec[1])
        sid = None
        if tr:
            sid = tr.seq_id(rec[0], record_offset + idx)
            tr.log(sid, "SEQUENCE_RECEIVED", L, rec[0])
        if par.min_length != -1 and L < par.min_length:
            if tr:
                tr.log(sid, "SEQUENCE_FILTERED", L, "too_short")
            continue
        if par.max_length != -1 and L > par.max_length:
            if tr:
                tr.log(sid, "SEQUENCE_FILTERED", L, "too_long")
            continue
        rrec = (rec[0], revcomp(rec[1]), rec[2][::-1])
        cands = find_candidates(prefilter, par, panel, rec, rrec, tr, sid)
        if not cands:
            if tr:
                tr.log(sid, "NO_MATCH_FOUND", "primer_search", "No primer matches found")
            ops.append(make_op(UNKNOWN, par, Cand(rec, False, panel.b_length), R_UNKNOWN, tr, sid))
            continue
        best = select_best(cands, tr, sid)
        full = False
        if par.dereplicate == "best":
            for m, spec, _b1, _b2 in dereplicate(best, panel, tr, sid):
                if spec is not None:
                    m.pool = panel.specimen_pool(spec)
                    ops.append(make_op(spec, par, m, R_DEREP, tr, sid))
                    full = True
                else:
                    fid, rt = resolve_specimen(m, panel, tr, sid)
                    ops.append(make_op(fid, par, m, rt, tr, sid))
                    full = full or rt in (R_FULL, R_DEREP)
        else:
            for m in best:
                fid, rt = resolve_specimen(m, panel, tr, sid)
                ops.append(make_op(fid, par, m, rt, tr, sid))
                full = full or rt in (R_FULL, R_DEREP)
        matched += 1 if full else 0
    return ops, total, matched


def trace_outputs(tr, ops, prefix="", fastq=True):
    """The SEQUENCE_OUTPUT events written when a batch's operations reach the OutputManager
    (io_utils.py:221-233): relative path of the primary file, primer pair "p1-p2"."""
    for op in ops:
        tr.log(op.trace_id, "SEQUENCE_OUTPUT", op.sample_id, op.pool, f"{op.p1}-{op.p2}", op_path(op, prefix, fastq)[0])


def make_prefilter(panel, par):
    """bloom_filter.py:200-209 barcode list + exact-set prefilter."""
    b1s, b2s = [], []
    for p in panel.get_primers(FWD):
        b1s += [b for b in p.barcodes if b not in b1s]
    for p in panel.get_primers(REV):
        b2s += [b for b in p.barcodes if b not in b2s]
    return ExactPrefilter([revcomp(b) for b in b1s + b2s], par.max_dist_index)


# ------------------------------------------------------------------ file formats (host shell)
def _open_text(path):
    return gzip.open(path, "rt") if path.endswith((".gz", ".gzip")) else open(path, "rt")


def read_fasta(path):
    """Bio.SeqIO 'fasta': yields (title, sequence)."""
    title, chunks = None, []
    with _open_text(path) as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if title is not None:
                    yield title, "".join(chunks)
                title, chunks = line[1:], []
            elif title is not None:
                chunks.append(line.strip())
    if title is not None:
        yield title, "".join(chunks)


def read_fastq(path):
    """Bio.SeqIO 'fastq' (FastqGeneralIterator semantics, multi-line tolerant):
    yields (id, bases, quality_string); id = first whitespace token of the title."""
    with _open_text(path) as fh:
        line = fh.readline()
        while line:
            if not line.strip():
                line = fh.readline()
                continue
            if not line.startswith("@"):
                raise ValueError("Records in Fastq files should start with '@' character")
            title = line[1:].rstrip("\r\n")
            seq = []
            line = fh.readline()
            while line and not line.startswith("+"):
                seq.append(line.strip())
                line = fh.readline()
            bases = "".join(seq)
            line = fh.readline()
            qual = line.strip()          # at least one quality line, always
            line = fh.readline()
            while line and not (line.startswith("@") and len(qual) >= len(bases)):
                qual += line.strip()
                line = fh.readline()
            if len(qual) != len(bases):
                raise ValueError("Lengths of sequence and quality values differs")
            yield (title.split(None, 1)[0] if title.split() else "", bases, qual)


def read_sequences(path):
    base = os.path.basename(path)
    for ext in (".gz", ".gzip"):
        if base.endswith(ext):
            base = base[:-len(ext)]
    if base.lower().endswith((".fastq", ".fq")):
        return list(read_fastq(path)), True
    return [(t.split(None, 1)[0], s, None) for t, s in read_fasta(path)], False


def op_path(op, prefix="", fastq=True):
    """OutputManager._make_filename (io_utils.py:197-219); returns [primary, (pool-level)]"""
    ext = ".fastq" if fastq else ".fasta"
    safe = "".join(c if c.isalnum() or c in "._-$#" else "_" for c in (op.sample_id or UNKNOWN))
    top = "unknown" if op.rtype == R_UNKNOWN else ("partial" if op.rtype in (R_PFWD, R_PREV) else "full")
    paths = [os.path.join(top, op.pool, f"{op.p1}-{op.p2}", f"{prefix}{safe}{ext}")]
    if op.rtype in (R_FULL, R_DEREP):  # io_utils.py:256-268
        paths.append(os.path.join("full", op.pool, f"{prefix}{safe}{ext}"))
    return paths


def op_record(op, fastq=True):
    """io_utils.py:239-253"""
    hdr = f"{op.seq_id} {op.code} pool={op.pool} primers={op.p1}+{op.p2} {op.sample_id}"
    if fastq:
        return f"@{hdr}\n{op.sequence}\n+\n{op.quality}\n"
    return f">{hdr}\n{op.sequence}\n"


def run_files(primer_file, specimen_file, sequence_file, num_seqs=-1, **kw):
    """End-to-end in memory: returns ({relative_path: [record_text, ...]}, total, matched)."""
    panel = load_panel(primer_file, specimen_file)
    par = setup_params(panel, **kw)
    recs, is_fastq = read_sequences(sequence_file)
    if num_seqs >= 0:
        recs = recs[:num_seqs]
    if not is_fastq:
        recs = [(i, s, "I" * len(s)) for i, s, _ in recs]
    ops, total, matched = process_sequences(recs, par, panel)
    tree = {}
    for op in ops:
        for p in op_path(op, fastq=is_fastq):
            tree.setdefault(p, []).append(op_record(op, is_fastq))
    return tree, total, matched


# ------------------------------------------------------------------ parity helper (tests only)
def hit_table(par, panel, rec, prefilter="auto"):
    """Search results of match_one_end for EVERY (primer, end) of one read, in align_seq coordinates
    (no AlignmentResult.reversed()): {(primer_name, 'A'|'B'): {'pdist', 'locs', 'barcodes': {bc: (dist, locs)}}}.
    End 'A' searches the reverse complement, end 'B' the read itself."""
    if prefilter == "auto":
        prefilter = make_prefilter(panel, par) if par.prefilter else None
    s = rec[1]
    rs = revcomp(s)
    out = {}
    for primer in panel.primers.values():
        for end, q in (("A", rs), ("B", s)):
            c = Cand((rec[0], q, None), False, panel.b_length)
            seen = []
            match_one_end(prefilter, c, par, q, False, primer, 1, hits=seen)
            _tag, _name, _e, pdist, locs = seen[0]
            out[(primer.name, end)] = {"pdist": pdist, "locs": locs,
                                       "barcodes": {bc: (d, list(a.locs)) for bc, a, d in c.b1}}
    return out


def orientation_of(par, panel, rec):
    """determine_orientation result 'F' / 'R' / 'U' ('U' when pre-orientation is disabled)."""
    if not par.preorient:
        return "U"
    return determine_orientation(par, rec[1], revcomp(rec[1]), panel.get_primers(FWD), panel.get_primers(REV))
