"""Deterministic synthetic workloads for the configs of BASELINE.json / SURVEY.md 8(d).

The reference ships no large dataset (its development set, ONT037, is not in the repo), so the bench and
the parity tests use generated panels and ONT-style reads:

  read = [tail 0-30][b1][P1][insert][rc P2][rc b2][tail 0-30], 50 % reverse-complemented,
  per-base error (sub:ins:del = 4:3:3), category mix 78 % intact / 8 % forward end truncated inside the
  barcode / 8 % reverse end truncated / 3 % primers only (random barcodes) / 2 % random sequence /
  1 % short (20-79 nt prefix of a construct).  ACGT only (no N in the end windows).

Everything is numpy-vectorised: the two read ends are generated as independent segments (errors are i.i.d.
per base), so the hot path's input -- the two `search_len` end windows plus the read length -- can be
produced for millions of reads without materialising the inserts (`windows_only=True`)."""
import os

import numpy as np

from .orchestration import edit_distance

ITS1F, ITS4 = "CTTGGTCATTTAGAGGAAGTAA", "TCCTCCGCTTATTGATATGC"
POOLS_C3 = [  # (pool, fwd name, fwd seq, rev name, rev seq)   README.md:99-102,170-173 of the reference
    ("ITS", "ITS1F", ITS1F, "ITS4", ITS4),
    ("RPB2", "fRPB2-5F", "GAYGAYMGWGATCAYTTYGG", "RPB2-7.1R", "CCCATRGCYTGYTTMCCCATDGC"),
    ("LSU", "LR0R", "ACCCGCTGAACTTAAGC", "LR5", "TCCTGAGGGAAACTTCG"),
    ("TEF1", "EF1-983F", "GCYCCYGGHCAYGGTGAYTTYAT", "EF1-1567R", "ACHGTRCCRATACCACCRATCTT"),
]
_IUPAC = {"A": "A", "C": "C", "G": "G", "T": "T", "R": "AG", "Y": "CT", "K": "GT", "M": "AC", "S": "CG", "W": "AT",
          "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT"}
_COMP = bytes.maketrans(b"ACGTMRWSYKVHDBN", b"TGCAKYWSRMBDHVN")
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def revcomp(s: str) -> str:
    return s.encode().translate(_COMP)[::-1].decode()


def make_barcodes(n_fwd, n_rev, length=13, min_dist=6, seed=0):
    """Greedy draw: pairwise NW distance >= min_dist over fwd + rc(rev)  =>  k_idx = ceil(min_dist/2)."""
    rng = np.random.default_rng(seed)
    chosen = []   # in the space where distances are measured: fwd barcodes and rc(rev barcodes)
    while len(chosen) < n_fwd + n_rev:
        cand = "".join("ACGT"[i] for i in rng.integers(0, 4, length))
        if all(edit_distance(cand, c) >= min_dist for c in chosen):
            chosen.append(cand)
    return chosen[:n_fwd], [revcomp(c) for c in chosen[n_fwd:]]


class Panel:
    """A synthetic panel: pools of (fwd primer, rev primer) and a specimen grid fwd x rev per pool."""

    def __init__(self, pools, fwd_barcodes, rev_barcodes, shared_rev=None):
        self.pools, self.fwd, self.rev = pools, fwd_barcodes, rev_barcodes
        self.shared_rev = shared_rev or {}   # primer name -> extra pools it is listed in

    def write(self, directory):
        os.makedirs(directory, exist_ok=True)
        primers, spec = os.path.join(directory, "primers.fasta"), os.path.join(directory, "specimens.txt")
        seen = set()
        with open(primers, "w") as fh:
            for pool, fn, fs, rn, rs in self.pools:
                for name, seq, pos in ((fn, fs, "forward"), (rn, rs, "reverse")):
                    if name in seen:
                        continue
                    seen.add(name)
                    plist = [p for p, a, _, b, _ in self.pools if name in (a, b)] + self.shared_rev.get(name, [])
                    fh.write(f">{name} pool={','.join(dict.fromkeys(plist))} position={pos}\n{seq}\n")
        with open(spec, "w") as fh:
            fh.write("SampleID\tPrimerPool\tFwIndex\tFwPrimer\tRvIndex\tRvPrimer\n")
            for pool, fn, _fs, rn, _rs in self.pools:
                for i, b1 in enumerate(self.fwd):
                    for j, b2 in enumerate(self.rev):
                        fh.write(f"{pool}_F{i:02d}_R{j:02d}\t{pool}\t{b1}\t{fn}\t{b2}\t{rn}\n")
        return primers, spec


def panel_c1(seed=1001):
    f, r = make_barcodes(2, 2, seed=seed)
    return Panel([("ITS", "ITS1F", ITS1F, "ITS4", ITS4)], f, r)


def panel_c2(seed=2002):
    """768 specimens = 32 fwd x 24 rev 13-nt barcodes, ITS1F / ITS4 (ONT037-style layout)."""
    f, r = make_barcodes(32, 24, seed=seed)
    return Panel([("ITS", "ITS1F", ITS1F, "ITS4", ITS4)], f, r)


def panel_c3(seed=2002):
    """3072 specimens over 4 pools, same 32 x 24 index grid per pool, ITS4 also listed in pool LSU."""
    f, r = make_barcodes(32, 24, seed=seed)
    return Panel(POOLS_C3, f, r, shared_rev={"ITS4": ["LSU"]})


# ------------------------------------------------------------------------------------------- reads
def _encode(s):
    return np.frombuffer(s.encode(), dtype=np.uint8)


def _instantiate(rng, seq, n):
    """n random ACGT instances of an IUPAC primer -> uint8 [n, len]"""
    out = np.empty((n, len(seq)), dtype=np.uint8)
    for i, ch in enumerate(seq):
        opts = _encode(_IUPAC[ch])
        out[:, i] = opts[rng.integers(0, len(opts), n)]
    return out


_BASE_IDX = np.zeros(256, dtype=np.uint8)
_BASE_IDX[[67, 71, 84]] = (1, 2, 3)


def _apply_errors(rng, seg, seglen, rate):
    """seg uint8 [n, W] (row i valid for seglen[i]); i.i.d. per-base sub/ins/del = 4:3:3 of `rate`.
    Returns (out [n, 2W], outlen)."""
    n, W = seg.shape
    u = rng.integers(0, 65536, (n, W), dtype=np.uint16)
    t1, t2, t3 = int(0.4 * rate * 65536), int(0.7 * rate * 65536), int(rate * 65536)
    valid = np.arange(W)[None, :] < seglen[:, None]
    sub = (u < t1) & valid
    ins = (u >= t1) & (u < t2) & valid
    keep = valid & ~((u >= t2) & (u < t3))
    base = seg.copy()
    nsub = int(sub.sum())
    base[sub] = _ACGT[(_BASE_IDX[base[sub]] + rng.integers(1, 4, nsub, dtype=np.uint8)) % 4]
    emit = keep.astype(np.int16) + ins.astype(np.int16)   # bases written per template position
    start = np.cumsum(emit, axis=1, dtype=np.int16) - emit
    outlen = (start[:, -1] + emit[:, -1]).astype(np.int32)
    out = np.zeros((n, 2 * W), dtype=np.uint8)
    flat = out.reshape(-1)
    rowbase = (np.arange(n, dtype=np.int64) * (2 * W))[:, None]
    pos = rowbase + start
    flat[pos[ins]] = _ACGT[rng.integers(0, 4, int(ins.sum()))]
    flat[(pos + ins)[keep]] = base[keep]
    return out, outlen


def _rc_rows(arr, lens):
    """reverse-complement each row's valid prefix, left aligned."""
    n, W = arr.shape
    j = np.arange(W)[None, :]
    src = np.clip(lens[:, None] - 1 - j, 0, W - 1)
    comp = np.frombuffer(bytes.maketrans(b"ACGTN", b"TGCAN"), dtype=np.uint8)
    out = comp[np.take_along_axis(arr, src, axis=1)]
    out[j >= lens[:, None]] = 0
    return out


def rebuild_read(head, tail, L, S):
    """A read with the given end windows and length: head + filler + tail.  The hot path reads nothing but the
    two windows and the length, so any consumer of full reads (the oracle) sees the same problem."""
    Sp = min(S, L)
    h = head[:Sp].tobytes().decode()
    t = tail[:Sp].tobytes().decode()
    if L <= S:
        return h
    if L < 2 * S:
        return h + t[2 * S - L:]
    return h + "A" * (L - 2 * S) + t


class ReadSet:
    """Generated reads.  Always holds the end windows; holds full reads only when windows_only=False."""

    def __init__(self):
        self.lens = None        # int32 [n]
        self.head = None        # uint8 [n, S]  first min(S, len) bases, zero padded
        self.tail = None        # uint8 [n, S]  last min(S, len) bases, zero padded (left aligned)
        self.reads = None       # list[str] or None
        self.quals = None
        self.truth = None       # dict of arrays: category, pool, fwd index, rev index, flipped

    def windows(self, stride):
        n, S = self.head.shape
        w = np.zeros((n, stride), dtype=np.uint8)
        w[:, :S] = self.head
        w[:, S:2 * S] = self.tail
        return w

    def packed_windows(self, pstride):
        """The same windows as 4-bit codes (include/smx.h smx_pack_windows4): [n, pstride] uint8."""
        n, S = self.head.shape
        lut = np.full(256, 15, dtype=np.uint8)
        lut[np.frombuffer(b"ACGTNRYKMSWBDHV", dtype=np.uint8)] = np.arange(15, dtype=np.uint8)
        hb = (S + 1) // 2
        out = np.full((n, pstride), 0xFF, dtype=np.uint8)
        for e, arr in enumerate((self.head, self.tail)):
            codes = np.full((n, 2 * hb), 15, dtype=np.uint8)
            codes[:, :S] = lut[arr]
            out[:, e * hb:(e + 1) * hb] = codes[:, 0::2] | (codes[:, 1::2] << 4)
        return out

    def write_fastq(self, path):
        with open(path, "w") as fh:
            for i, (s, q) in enumerate(zip(self.reads, self.quals)):
                fh.write(f"@read{i:07d} synthetic\n{s}\n+\n{q}\n")

    def write_fastq_rebuilt(self, path, S, seed=0):
        """FASTQ of a windows-only set: read i = rebuild_read(head, tail, len) (filler in the middle), quality = a slice of
        one random Phred block.  Byte-level loop, ~10^6 reads in seconds: the end-to-end bench's input file."""
        n = len(self.lens)
        hb, tb = np.ascontiguousarray(self.head).tobytes(), np.ascontiguousarray(self.tail).tobytes()
        W = self.head.shape[1]
        lens = self.lens.tolist()
        maxL = max(lens) if lens else 0
        qblock = (np.random.default_rng(seed).integers(3, 41, maxL + 4096) + 33).astype(np.uint8).tobytes()
        filler = b"A" * maxL
        with open(path, "wb") as fh:
            for lo in range(0, n, 8192):
                parts = []
                for i in range(lo, min(n, lo + 8192)):
                    L = lens[i]
                    Sp = S if L > S else L
                    h = hb[i * W:i * W + Sp]
                    if L <= S:
                        s = h
                    elif L < 2 * S:
                        s = h + tb[i * W + 2 * S - L:i * W + Sp]
                    else:
                        s = h + filler[:L - 2 * S] + tb[i * W:i * W + Sp]
                    off = i & 4095
                    parts.append(b"@read%07d synthetic\n%b\n+\n%b\n" % (i, s, qblock[off:off + L]))
                fh.write(b"".join(parts))


CHUNK = 65536   # part of the generator's definition: chunk c of a read set uses default_rng([seed, c])


def make_reads(panel: Panel, n, seed, workers=0, **kw):
    """n reads; generated in independent chunks of CHUNK reads so that memory stays bounded and the
    result does not depend on how many reads are requested after a given chunk (nor on `workers`: chunk c always
    uses default_rng([seed, c]); workers > 1 only runs chunks on a thread pool -- numpy releases the GIL)."""
    jobs = [(min(CHUNK, n - lo), c) for c, lo in enumerate(range(0, n, CHUNK))]
    if workers > 1 and len(jobs) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as ex:
            parts = list(ex.map(lambda j: _make_chunk(panel, j[0], np.random.default_rng([seed, j[1]]), **kw), jobs))
    else:
        parts = [_make_chunk(panel, m, np.random.default_rng([seed, c]), **kw) for m, c in jobs]
    if len(parts) == 1:
        return parts[0]
    rs = ReadSet()
    rs.lens = np.concatenate([p.lens for p in parts])
    rs.head = np.concatenate([p.head for p in parts])
    rs.tail = np.concatenate([p.tail for p in parts])
    rs.truth = {k: np.concatenate([p.truth[k] for p in parts]) for k in parts[0].truth}
    if parts[0].reads is not None:
        rs.reads = [r for p in parts for r in p.reads]
        rs.quals = [q for p in parts for q in p.quals]
    return rs


def _make_chunk(panel: Panel, n, rng, search_len=80, error_rate=0.06, insert_mean=650, insert_sd=120,
                insert_min=200, insert_max=1500, windows_only=True, n_frac=0.0):
    S = search_len
    K = S + 20   # insert bases generated on each side (enough to fill the window after deletions)
    npool = len(panel.pools)
    pool = rng.integers(0, npool, n)
    fi = rng.integers(0, len(panel.fwd), n)
    ri = rng.integers(0, len(panel.rev), n)
    cat = np.searchsorted(np.cumsum([0.78, 0.08, 0.08, 0.03, 0.02, 0.01]), rng.random(n), side="right")
    cat = np.minimum(cat, 5)   # 0 intact 1 fwd-trunc 2 rev-trunc 3 primers-only 4 random 5 short
    flipped = rng.random(n) < 0.5
    ta, tb = rng.integers(0, 31, n), rng.integers(0, 31, n)
    Lb = len(panel.fwd[0])
    fwd_bc = np.stack([_encode(b) for b in panel.fwd])
    rev_bc_rc = np.stack([_encode(revcomp(b)) for b in panel.rev])
    pmax = max(max(len(p[2]), len(p[4])) for p in panel.pools)
    W = 30 + Lb + pmax + K
    rand = lambda shape: _ACGT[rng.integers(0, 4, shape, dtype=np.uint8)]  # noqa: E731
    # ---- head template: [tailA][b1][P1][insert head K], left aligned
    head = np.zeros((n, W), dtype=np.uint8)
    headlen = np.zeros(n, dtype=np.int32)
    tail = np.zeros((n, W), dtype=np.uint8)
    taillen = np.zeros(n, dtype=np.int32)
    b1 = fwd_bc[fi].copy()
    b2 = rev_bc_rc[ri].copy()
    po = cat == 3
    b1[po] = rand((int(po.sum()), Lb))
    b2[po] = rand((int(po.sum()), Lb))
    tailA, tailB = rand((n, 30)), rand((n, 30))
    insH, insT = rand((n, K)), rand((n, K))
    for p, (_pool, _fn, fs, _rn, rs) in enumerate(panel.pools):
        sel = np.nonzero(pool == p)[0]
        if len(sel) == 0:
            continue
        P1 = _instantiate(rng, fs, len(sel))
        P2rc = _instantiate(rng, revcomp(rs), len(sel))
        m1, m2 = len(fs), len(rs)
        # head rows: tail (right aligned in 30) + b1 + P1 + insert
        block = np.concatenate([tailA[sel], b1[sel], P1, insH[sel]], axis=1)
        cut = (30 - ta[sel]).astype(np.int64)                       # drop unused tail positions
        c1 = cat[sel] == 1                                          # forward end truncated inside the barcode
        cut[c1] = 30 + rng.integers(1, Lb, int(c1.sum()))
        width = block.shape[1]
        idx = np.clip(np.arange(W)[None, :] + cut[:, None], 0, width - 1)
        rows = np.take_along_axis(block, idx[:, :W], axis=1)
        hl = (width - cut).astype(np.int32)
        rows[np.arange(W)[None, :] >= hl[:, None]] = 0
        head[sel], headlen[sel] = rows, hl
        # tail rows: insert + rcP2 + rc b2 + tail (left aligned), truncated on the right
        block = np.concatenate([insT[sel], P2rc, b2[sel], tailB[sel]], axis=1)
        tl = (K + m2 + Lb + tb[sel]).astype(np.int32)
        c2 = cat[sel] == 2
        tl[c2] = K + m2 + rng.integers(0, Lb - 1, int(c2.sum()))
        rows = np.zeros((len(sel), W), dtype=np.uint8)
        rows[:, :block.shape[1]] = block
        rows[np.arange(W)[None, :] >= tl[:, None]] = 0
        tail[sel], taillen[sel] = rows, tl
        del m1
    rnd = cat == 4
    head[rnd] = rand((int(rnd.sum()), W)); headlen[rnd] = W - 30
    tail[rnd] = rand((int(rnd.sum()), W)); taillen[rnd] = W - 30
    head[np.arange(W)[None, :] >= headlen[:, None]] = 0
    tail[np.arange(W)[None, :] >= taillen[:, None]] = 0
    # ---- errors
    hout, hlen = _apply_errors(rng, head, headlen, error_rate)
    tout, tlen = _apply_errors(rng, tail, taillen, error_rate)
    if n_frac > 0:   # a share of the reads carries ambiguity codes ('N') inside the end windows
        for out_, len_ in ((hout, hlen), (tout, tlen)):
            sel = np.nonzero(rng.random(n) < n_frac)[0]
            for k in range(3):
                pos = rng.integers(0, np.maximum(len_[sel], 1))
                ok = pos < len_[sel]
                out_[sel[ok], pos[ok]] = 78
    ins = np.clip(rng.normal(insert_mean, insert_sd, n), max(insert_min, 2 * K), insert_max).astype(np.int32)
    middle = ins - 2 * K
    short = cat == 5
    shortlen = rng.integers(20, 80, n).astype(np.int32)
    shortlen = np.minimum(shortlen, hlen)
    # ---- assemble windows (before the strand flip)
    L = np.where(short, shortlen, hlen + middle + tlen).astype(np.int32)
    Sp = np.minimum(L, S)
    j = np.arange(S)[None, :]
    hw = hout[:, :S].copy() if hout.shape[1] >= S else np.pad(hout, ((0, 0), (0, S - hout.shape[1])))
    # normal reads have hlen >= S (K=100 insert bases guarantee it); short reads: whole read
    hw[j >= Sp[:, None]] = 0
    tsrc = np.clip(tlen[:, None] - Sp[:, None] + j, 0, tout.shape[1] - 1)
    tw = np.take_along_axis(tout, tsrc, axis=1)
    tw[short] = hw[short]
    tw[j >= Sp[:, None]] = 0
    rs = ReadSet()
    rs.lens = L
    rs.truth = {"category": cat, "pool": pool, "fwd": fi, "rev": ri, "flipped": flipped}
    fl = flipped
    rs.head = np.where(fl[:, None], _rc_rows(tw, Sp), hw)
    rs.tail = np.where(fl[:, None], _rc_rows(hw, Sp), tw)
    if not windows_only:
        reads, quals = [], []
        for i in range(n):
            if short[i]:
                s = hout[i, :L[i]].tobytes().decode()
            else:
                mid = rand(int(middle[i])).tobytes().decode()
                s = hout[i, :hlen[i]].tobytes().decode() + mid + tout[i, :tlen[i]].tobytes().decode()
            if fl[i]:
                s = revcomp(s)
            reads.append(s)
            quals.append((rng.integers(3, 41, len(s)) + 33).astype(np.uint8).tobytes().decode())
        rs.reads, rs.quals = reads, quals
    return rs
