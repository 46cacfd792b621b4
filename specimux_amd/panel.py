"""Panel compiler: Specimens + MatchParameters + flags  ->  smx_panel_desc (include/smx.h) -> smx_panel.

Flattens exactly the state the reference's hot path reads from its Python objects:
  * primer order  = Specimens._primers registration order (databases.py:151-165)
  * barcode lists = PrimerInfo.barcodes in canonical (first appearance) order (Q4)
  * pair list     = find_candidate_matches' nested loops (demultiplex.py:699-700) with
                    get_paired_primers (databases.py:251-264) and get_pool_from_primers (:640-665)
  * specimen rows = file order with p1/p2 membership masks (wildcards already expanded)
"""
import ctypes as C

import numpy as np

from . import _lib
from .constants import Primer
from .models import reverse_complement


class CompiledPanel:
    """Owns the smx_panel handle plus the index -> name tables needed to print results."""

    def __init__(self, specimens, parameters, trim="barcodes", dereplicate="best", prefilter=True,
                 min_length=-1, max_length=-1, want_starts=False):
        lib = _lib.load()
        self._lib = lib
        self.handle = None
        primers = list(specimens._primers.values())
        if not primers:
            raise ValueError("no primers registered")
        self.primers = primers
        self.primer_names = [p.name for p in primers]
        pidx = {id(p): i for i, p in enumerate(primers)}
        # global barcode list, first appearance over primers in registration order
        self.barcodes = []
        bidx = {}
        for p in primers:
            for b in p.barcodes:
                if b not in bidx:
                    bidx[b] = len(self.barcodes)
                    self.barcodes.append(b)
        # pools: index table over every pool name that can be printed
        self.pools = []
        pool_idx = {}

        def pool_id(name):
            if name is None:
                return -1
            if name not in pool_idx:
                pool_idx[name] = len(self.pools)
                self.pools.append(name)
            return pool_idx[name]

        fwd = [p for p in primers if p.direction == Primer.FWD]
        pairs = []
        for fp in fwd:
            for rp in specimens.get_paired_primers(fp.primer):
                common = set(fp.pools) & set(rp.pools)
                pairs.append((pidx[id(fp)], pidx[id(rp)], pool_id(sorted(common)[0] if common else None)))
        if not pairs:
            raise ValueError("no primer pairs share a specimen")
        self.specimen_ids = [s[0] for s in specimens._specimens]
        self.specimen_pools = [s[1] for s in specimens._specimens]

        def mask(infos):
            m = 0
            for info in infos:       # membership is by object identity in the reference (`p in p1s`)
                reg = specimens._primers.get(info.primer)
                if reg is info:
                    m |= 1 << pidx[id(info)]
            return m

        rc_primers = [p.primer_rc for p in primers]
        rc_barcodes = [reverse_complement(b) for b in self.barcodes]
        bc_lens = {len(b) for b in self.barcodes}
        if prefilter:
            # BloomPrefilter.min_length = len(barcodes[0]) - k (bloom_filter.py:41-44); only defined for
            # uniform barcode lengths (with mixed lengths the reference depends on set order: Q7)
            if len(bc_lens) != 1:
                raise ValueError("barcode prefilter needs barcodes of one length; use --disable-prefilter")
            pf_min = bc_lens.pop() - parameters.max_dist_index
        else:
            pf_min = 0

        def offsets(strings):
            off = np.zeros(len(strings) + 1, dtype=np.uint32)
            off[1:] = np.cumsum([len(s) for s in strings])
            return off

        keep = self._keep = {}
        keep["primer_rc"] = "".join(rc_primers).encode("ascii")
        keep["primer_rc_off"] = offsets(rc_primers)
        keep["primer_dir"] = np.array([0 if p.direction == Primer.FWD else 1 for p in primers], dtype=np.uint8)
        keep["primer_k"] = np.array([parameters.max_dist_primers[p.primer] for p in primers], dtype=np.int32)
        keep["primer_file_index"] = np.array([p.file_index for p in primers], dtype=np.int32)
        keep["primer_bc_off"] = offsets([p.barcodes for p in primers])
        keep["primer_bc"] = np.array([bidx[b] for p in primers for b in p.barcodes] or [0], dtype=np.uint32)
        keep["barcode_rc"] = "".join(rc_barcodes).encode("ascii")
        keep["barcode_rc_off"] = offsets(rc_barcodes)
        keep["pair_fwd"] = np.array([a for a, _, _ in pairs], dtype=np.uint32)
        keep["pair_rev"] = np.array([b for _, b, _ in pairs], dtype=np.uint32)
        keep["pair_pool"] = np.array([c for _, _, c in pairs], dtype=np.int32)
        keep["spec_b1"] = np.array([bidx[s[2].upper()] for s in specimens._specimens], dtype=np.uint32)
        keep["spec_b2"] = np.array([bidx[s[4].upper()] for s in specimens._specimens], dtype=np.uint32)
        keep["spec_p1mask"] = np.array([mask(s[3]) for s in specimens._specimens], dtype=np.uint64)
        keep["spec_p2mask"] = np.array([mask(s[5]) for s in specimens._specimens], dtype=np.uint64)
        keep["spec_pool"] = np.array([pool_id(s[1]) for s in specimens._specimens], dtype=np.int32)

        d = _lib.PanelDesc()
        d.abi_version = _lib.ABI_VERSION
        d.n_primers, d.n_barcodes = len(primers), len(self.barcodes)
        d.n_specimens, d.n_pools, d.n_pairs = len(self.specimen_ids), len(self.pools), len(pairs)
        d.primer_rc, d.barcode_rc = keep["primer_rc"], keep["barcode_rc"]
        for name in ("primer_rc_off", "primer_dir", "primer_k", "primer_file_index", "primer_bc_off", "primer_bc",
                     "barcode_rc_off", "pair_fwd", "pair_rev", "pair_pool", "spec_b1", "spec_b2", "spec_p1mask",
                     "spec_p2mask", "spec_pool"):
            setattr(d, name, keep[name].ctypes.data)
        d.k_index = int(parameters.max_dist_index)
        d.search_len = int(parameters.search_len)
        d.barcode_len_max = int(specimens.b_length())
        d.prefilter_min_len = int(pf_min)
        d.preorient = 1 if parameters.preorient else 0
        d.trim = _lib.TRIM[trim]
        d.dereplicate = _lib.DEREP[dereplicate]
        d.min_length, d.max_length = int(min_length), int(max_length)
        d.want_starts = 1 if want_starts else 0
        self.desc = d
        self.pairs = pairs
        self.search_len = int(parameters.search_len)
        self.trim, self.dereplicate = trim, dereplicate
        handle = C.c_void_p()
        _lib.check(lib.smx_panel_create(C.byref(d), C.byref(handle)))
        self.handle = handle
        self.counts_len = lib.smx_counts_len(handle)
        self.window_stride = lib.smx_window_stride(handle)
        self.hits_per_read = lib.smx_hits_per_read(handle)
        self.bdist_per_read = lib.smx_bdist_per_read(handle)
        self.max_barcodes = self.bdist_per_read // self.hits_per_read

    def set_streams(self, n: int):
        """The caller keeps n batches in flight on n streams (smx_panel_set_streams): each demux launch then takes 1/n of
        the CUs' workgroup slots and kernels of different batches run side by side."""
        _lib.check(self._lib.smx_panel_set_streams(self.handle, int(n)))

    def close(self):
        if self.handle is not None:
            self._lib.smx_panel_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ host helpers
    def pack_windows(self, bases: np.ndarray, offsets: np.ndarray):
        """bases: uint8 concatenated reads, offsets: uint64 n+1  ->  (windows uint8 [n, stride], lens int32 [n])"""
        n = len(offsets) - 1
        windows = np.empty((n, self.window_stride), dtype=np.uint8)
        lens = np.empty(n, dtype=np.int32)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if len(bases) == 0:
            bases = np.zeros(1, dtype=np.uint8)
        _lib.check(self._lib.smx_pack_windows(_lib.ptr(bases), _lib.ptr(offsets), n, self.search_len,
                                              _lib.ptr(windows), _lib.ptr(lens)))
        return windows, lens

    def run(self, windows: np.ndarray, lens: np.ndarray, counts=None, want_hits=False):
        """Host-buffer convenience path (smx_batch_run): returns (ops, extra, counts[, hits, bdist]).
        want_hits=True asks for the hit table AND the per-barcode distances (per-barcode "slots" kernel);
        want_hits="lean" asks for the hit table only, from the kernel the flags select (bdist is None)."""
        n = len(lens)
        ops = np.zeros(n, dtype=_lib.OP_DTYPE)
        cap = max(64, n // 4)
        if counts is None:
            counts = np.zeros(self.counts_len, dtype=np.uint64)
        hits = np.zeros((n, self.hits_per_read), dtype=_lib.HIT_DTYPE) if want_hits else None
        bdist = np.zeros((n, self.hits_per_read, self.max_barcodes), dtype=np.int8) if want_hits is True else None
        while True:
            extra = np.zeros(cap, dtype=_lib.OP_DTYPE)
            n_extra = C.c_uint32(0)
            before = counts.copy()
            rc = self._lib.smx_batch_run(self.handle, _lib.ptr(windows), _lib.ptr(lens), n, _lib.ptr(ops),
                                         _lib.ptr(extra), cap, C.byref(n_extra), _lib.ptr(counts),
                                         _lib.ptr(hits), _lib.ptr(bdist))
            if rc == _lib.ERR_OVERFLOW and n_extra.value > cap:
                cap = int(n_extra.value)      # rerun the batch with a large enough extra buffer
                counts[:] = before
                continue
            _lib.check(rc)
            break
        extra = extra[:n_extra.value]
        if want_hits:
            return ops, extra, counts, hits, bdist
        return ops, extra, counts
