"""Python face of the native streaming helpers in libsmx.so (include/smx.h, "Host streaming helpers"):
FASTQ/FASTA reader -> batches, window packer, output writer.  SURVEY.md section 8(f) rows 1-2."""
import ctypes as C

import numpy as np

from . import _lib


class Batch:
    """One parsed batch; owns its memory on the C side."""

    def __init__(self):
        self._lib = _lib.load()
        self.handle = C.c_void_p(self._lib.smx_batch_new())

    def __len__(self):
        return int(self._lib.smx_batch_size(self.handle))

    def record(self, i):
        """(id, sequence, quality or None) as str -- for tests and the compatibility path."""
        pid, pseq, pqual = C.c_char_p(), C.c_void_p(), C.c_void_p()
        nid, nseq = C.c_uint32(), C.c_uint32()
        _lib.check(self._lib.smx_batch_record(self.handle, i, C.byref(pid), C.byref(nid), C.byref(pseq),
                                              C.byref(pqual), C.byref(nseq)))
        rid = C.string_at(pid, nid.value).decode("latin-1")
        seq = C.string_at(pseq.value, nseq.value).decode("latin-1") if nseq.value else ""
        qual = None
        if pqual.value:
            qual = C.string_at(pqual.value, nseq.value).decode("latin-1") if nseq.value else ""
        return rid, seq, qual

    def pack_windows(self, search_len, stride):
        n = len(self)
        windows = np.empty((n, stride), dtype=np.uint8)
        lens = np.empty(n, dtype=np.int32)
        _lib.check(self._lib.smx_pack_windows_batch(self.handle, search_len, _lib.ptr(windows), _lib.ptr(lens)))
        return windows, lens

    def pack_windows_into(self, search_len, windows, lens):
        """Cut the end windows straight into caller-owned buffers (a Lane's pinned staging): no allocation, no copy."""
        _lib.check(self._lib.smx_pack_windows_batch(self.handle, search_len, _lib.ptr(windows), _lib.ptr(lens)))

    def pack_windows4_into(self, search_len, packed, lens):
        """Cut the end windows as 4-bit codes (include/smx.h smx_pack_windows4_batch) into caller-owned buffers.  Returns
        the number of reads that cannot travel in this format (a 'U' inside a window): 0 in practice."""
        n_ascii = C.c_uint32()
        _lib.check(self._lib.smx_pack_windows4_batch(self.handle, search_len, _lib.ptr(packed), _lib.ptr(lens), C.byref(n_ascii)))
        return int(n_ascii.value)

    def close(self):
        if self.handle:
            self._lib.smx_batch_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Lane:
    """One asynchronous batch slot of a panel (include/smx.h "Lanes"): page-locked staging + device buffers + a HIP stream.
    `windows` / `lens` are numpy views of the pinned staging; submit() returns at once, wait() blocks and returns views
    of the pinned result records (valid until the next submit on this lane)."""

    def __init__(self, panel, max_reads):
        self._lib = _lib.load()
        self.panel = panel
        self.max_reads = int(max_reads)
        self.handle = C.c_void_p()
        _lib.check(self._lib.smx_lane_create(panel.handle, self.max_reads, C.byref(self.handle)))
        wp = C.cast(self._lib.smx_lane_windows(self.handle), C.POINTER(C.c_uint8))
        lp = C.cast(self._lib.smx_lane_lens(self.handle), C.POINTER(C.c_int32))
        self.windows = np.ctypeslib.as_array(wp, shape=(self.max_reads, panel.window_stride))
        self.lens = np.ctypeslib.as_array(lp, shape=(self.max_reads,))
        # the same staging seen as 4-bit windows (submit_packed): half the bytes per read
        self.packed_stride = int(self._lib.smx_packed_stride(panel.handle))
        self.packed = self.windows.reshape(-1)[:self.max_reads * self.packed_stride].reshape(self.max_reads, self.packed_stride)
        self.n = 0

    def submit(self, n):
        self.n = int(n)
        _lib.check(self._lib.smx_lane_submit(self.handle, self.n))

    def submit_packed(self, n):
        """The staging holds 4-bit windows (Batch.pack_windows4_into(..., lane.packed, lane.lens))."""
        self.n = int(n)
        _lib.check(self._lib.smx_lane_submit_packed(self.handle, self.n))

    def wait(self, counts):
        """-> (ops, extra) numpy views; `counts` (uint64, panel.counts_len) is accumulated into."""
        ops_p, extra_p, n_extra = C.c_void_p(), C.c_void_p(), C.c_uint32()
        _lib.check(self._lib.smx_lane_wait(self.handle, C.byref(ops_p), C.byref(extra_p), C.byref(n_extra), _lib.ptr(counts)))
        if self.n == 0:
            return np.zeros(0, dtype=_lib.OP_DTYPE), np.zeros(0, dtype=_lib.OP_DTYPE)
        ops = np.ctypeslib.as_array(C.cast(ops_p, C.POINTER(C.c_uint8)), shape=(self.n * 32,)).view(_lib.OP_DTYPE)
        if n_extra.value:
            extra = np.ctypeslib.as_array(C.cast(extra_p, C.POINTER(C.c_uint8)), shape=(n_extra.value * 32,)).view(_lib.OP_DTYPE)
        else:
            extra = np.zeros(0, dtype=_lib.OP_DTYPE)
        return ops, extra

    def close(self):
        if self.handle:
            self._lib.smx_lane_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Reader:
    def __init__(self, path, byte_range=None):
        """byte_range = (lo, hi): only the records that start inside those bytes (uncompressed 4-line FASTQ)."""
        self._lib = _lib.load()
        self.handle = C.c_void_p()
        fq = C.c_int()
        if byte_range is None:
            _lib.check(self._lib.smx_reader_open(path.encode(), C.byref(self.handle), C.byref(fq)))
        else:
            _lib.check(self._lib.smx_reader_open_range(path.encode(), int(byte_range[0]), int(byte_range[1]),
                                                       C.byref(self.handle), C.byref(fq)))
        self.is_fastq = bool(fq.value)

    def next_batch(self, max_reads, max_bytes=0, into=None):
        """Fill (or create) a Batch with up to max_reads records; returns None at end of file."""
        batch = into or Batch()
        n = C.c_uint32()
        _lib.check(self._lib.smx_reader_next(self.handle, max_reads, max_bytes, batch.handle, C.byref(n)))
        return batch if n.value else None

    def close(self):
        if self.handle:
            self._lib.smx_reader_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _table(strings):
    blob = "".join(strings).encode("utf-8")
    off = np.zeros(len(strings) + 1, dtype=np.uint32)
    if strings:
        off[1:] = np.cumsum([len(s.encode("utf-8")) for s in strings])
    return blob, off


class Writer:
    """Appends formatted records to the output tree; names come from a CompiledPanel."""

    def __init__(self, output_dir, prefix, is_fastq, panel):
        self._lib = _lib.load()
        self._keep = [_table(panel.specimen_ids), _table(panel.pools), _table(panel.primer_names), _table(panel.barcodes)]
        nm = _lib.Names()
        (nm.specimens, so), (nm.pools, po), (nm.primers, pr), (nm.barcodes, bo) = self._keep
        nm.specimen_off, nm.pool_off, nm.primer_off, nm.barcode_off = (a.ctypes.data for a in (so, po, pr, bo))
        nm.n_specimens, nm.n_pools = len(panel.specimen_ids), len(panel.pools)
        nm.n_primers, nm.n_barcodes = len(panel.primer_names), len(panel.barcodes)
        self.handle = C.c_void_p()
        _lib.check(self._lib.smx_writer_open(output_dir.encode(), (prefix or "").encode(), 1 if is_fastq else 0,
                                             C.byref(nm), C.byref(self.handle)))

    def write(self, batch, ops, extra):
        ops = np.ascontiguousarray(ops)
        extra = np.ascontiguousarray(extra)
        _lib.check(self._lib.smx_writer_write(self.handle, batch.handle, _lib.ptr(ops), len(ops),
                                              _lib.ptr(extra) if len(extra) else None, len(extra)))

    def close(self):
        if self.handle:
            h, self.handle = self.handle, None
            _lib.check(self._lib.smx_writer_close(h))
