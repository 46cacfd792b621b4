"""Streaming file -> output-tree pipeline around the GPU hot path (the `-F` production mode).

Replaces the reference's parent-parses / pool-of-workers structure (orchestration.py:153-237: serial
SeqIO.parse, pickled 1000-read batches, per-worker lockf+fsync appends) with three overlapped stages in one
process: native reader + window packer (thread), GPU batch run (main thread), native writer (thread).
A batch owns its memory, so stage i+1 of batch k overlaps stage i of batch k+1."""
import os
import queue
import sys
import threading
import time

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
    timing = {"read": 0.0, "pack": 0.0, "gpu": 0.0, "write": 0.0} if os.environ.get("SMX_PIPELINE_TIMING") else None
    t_start = time.perf_counter()

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
                t0 = time.perf_counter()
                b = reader.next_batch(want, BATCH_BYTES)
                if b is None:
                    break
                if left is not None:
                    left -= len(b)
                t1 = time.perf_counter()
                windows, lens = b.pack_windows(panel.search_len, panel.window_stride)
                if timing:
                    timing["read"] += t1 - t0
                    timing["pack"] += time.perf_counter() - t1
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
                t0 = time.perf_counter()
                if not errors:
                    writer.write(b, ops, extra)
                b.close()
                if timing:
                    timing["write"] += time.perf_counter() - t0
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
            t0 = time.perf_counter()
            ops, extra, _ = panel.run(windows, lens, counts=counts)
            if timing:
                timing["gpu"] += time.perf_counter() - t0
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
    t0 = time.perf_counter()
    writer.close()
    if timing:
        print("[smx pipeline] wall %.3f s: reader %.3f + pack %.3f (thread 1) | gpu %.3f (main) | writer %.3f + close %.3f "
              "(thread 2)" % (time.perf_counter() - t_start, timing["read"], timing["pack"], timing["gpu"], timing["write"],
                              time.perf_counter() - t0), file=sys.stderr)
    return int(counts[_lib.CNT_TOTAL]), int(counts[_lib.CNT_MATCHED]), counts, reader.is_fastq
