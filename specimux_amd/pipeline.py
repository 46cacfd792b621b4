"""Streaming file -> output-tree pipeline around the GPU hot path (the `-F` production mode).

Replaces the reference's parent-parses / pool-of-workers structure (orchestration.py:153-237: serial
SeqIO.parse, pickled 1000-read batches, per-worker lockf+fsync appends) with three overlapped stages in one
process: native reader + window packer (thread), GPU batch run (main thread), native writer (thread).
A batch owns its memory, so stage i+1 of batch k overlaps stage i of batch k+1."""
import queue
import threading

import numpy as np

from . import _lib
from .native_io import Reader, Writer

BATCH_READS = 131072          # reads per kernel launch
BATCH_BYTES = 256 << 20       # ... or this many bytes of input, whichever comes first


def run_streaming(sequence_file, panel, output_dir, prefix, start_seq=1, num_seqs=-1, on_batch=None):
    """Returns (total_reads, matched_reads, counts vector, is_fastq)."""
    reader = Reader(sequence_file)
    writer = Writer(output_dir, prefix, reader.is_fastq, panel)
    counts = np.zeros(panel.counts_len, dtype=np.uint64)
    q_in, q_out = queue.Queue(maxsize=2), queue.Queue(maxsize=2)
    errors = []

    def produce():
        try:
            to_skip = max(0, start_seq - 1)
            left = num_seqs if num_seqs >= 0 else None
            while to_skip > 0:   # -n start,num: discard the first start-1 records
                b = reader.next_batch(min(to_skip, BATCH_READS), BATCH_BYTES)
                if b is None:
                    break
                to_skip -= len(b)
                b.close()
            while left is None or left > 0:
                want = BATCH_READS if left is None else min(BATCH_READS, left)
                b = reader.next_batch(want, BATCH_BYTES)
                if b is None:
                    break
                if left is not None:
                    left -= len(b)
                windows, lens = b.pack_windows(panel.search_len, panel.window_stride)
                q_in.put((b, windows, lens))
        except BaseException as e:   # surfaced in the main thread
            errors.append(e)
        finally:
            q_in.put(None)

    def consume():
        try:
            while True:
                item = q_out.get()
                if item is None:
                    return
                b, ops, extra = item
                if not errors:
                    writer.write(b, ops, extra)
                b.close()
        except BaseException as e:
            errors.append(e)
            while q_out.get() is not None:   # keep draining so the main thread never blocks
                pass

    tp = threading.Thread(target=produce, name="smx-reader", daemon=True)
    tc = threading.Thread(target=consume, name="smx-writer", daemon=True)
    tp.start()
    tc.start()
    try:
        while True:
            item = q_in.get()
            if item is None or errors:
                break
            b, windows, lens = item
            ops, extra, _ = panel.run(windows, lens, counts=counts)
            q_out.put((b, ops, extra))
            if on_batch:
                on_batch(int(counts[_lib.CNT_TOTAL]), int(counts[_lib.CNT_MATCHED]))
    finally:
        q_out.put(None)
        tc.join()
        while tp.is_alive():   # unblock a producer stuck on a full queue after an error
            try:
                q_in.get_nowait()
            except queue.Empty:
                pass
            tp.join(timeout=0.05)
        reader.close()
    if errors:
        try:
            writer.close()
        except Exception:
            pass
        raise errors[0]
    writer.close()
    return int(counts[_lib.CNT_TOTAL]), int(counts[_lib.CNT_MATCHED]), counts, reader.is_fastq
