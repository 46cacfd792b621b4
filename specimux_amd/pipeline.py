"""Streaming file -> output-tree pipeline around the GPU hot path (the `-F` production mode).

Replaces the reference's parent-parses / pool-of-workers structure (orchestration.py:153-237: serial
SeqIO.parse, pickled 1000-read batches, per-worker lockf+fsync appends) with overlapped stages in one process:

  reader thread   native reader -> parsed batch -> end windows cut STRAIGHT INTO a lane's page-locked staging
                  (no per-batch allocation) -> lane.submit(): asynchronous H2D copy, prescan + demux kernels, D2H copy
                  on the lane's own HIP stream
  main thread     lane.wait() in submission order; counts accumulate
  writer thread   native writer formats the records (views of the lane's pinned result buffers) and appends to the
                  tree, then hands the lane back

Three lanes are in flight: while lane A's kernels run, lane B's windows cross PCIe and lane C's records are being
written (reference steps replaced: io_utils.py:429-450, orchestration.py:447-456).
`byte_range` restricts the run to the records that START inside [lo, hi) of an uncompressed FASTQ (multi-GPU file
sharding, specimux_amd/distributed.py)."""
import os
import queue
import sys
import threading
import time

import numpy as np

from . import _lib
from .native_io import Lane, Reader, Writer

BATCH_READS = int(os.environ.get("SMX_BATCH_READS", "131072"))   # reads per kernel launch (tuning hook: tools/e2e_sweep.py)
BATCH_BYTES = 256 << 20       # ... or this many bytes of input, whichever comes first
FIRST_BATCH_READS = 32768     # the first batch is small: the writer (the longest stage) starts that much earlier
N_LANES = 3


def run_streaming(sequence_file, panel, output_dir, prefix, start_seq=1, num_seqs=-1, on_batch=None, byte_range=None,
                  stats=None, stride=None):
    """Returns (total_reads, matched_reads, counts vector, is_fastq).  `stats` (dict), if given, receives the stage
    seconds: read, pack, submit, gpu_wait (main thread blocked on a lane), write, close, wall."""
    reader = Reader(sequence_file, byte_range=byte_range)
    keep_batch = [0]   # stride = (rank, world): batch i belongs to rank i % world (inputs that cannot be cut by byte range)
    writer = Writer(output_dir, prefix, reader.is_fastq, panel)
    counts = np.zeros(panel.counts_len, dtype=np.uint64)
    lanes = []
    free_lanes, q_gpu, q_out = queue.Queue(), queue.Queue(maxsize=N_LANES), queue.Queue(maxsize=N_LANES)
    errors = []
    ascii_lanes = bool(os.environ.get("SMX_LANES_ASCII"))   # A/B and test hook: ship 8-bit windows
    timing = {"read": 0.0, "pack": 0.0, "submit": 0.0, "gpu_wait": 0.0, "write": 0.0, "close": 0.0}
    n_delivered = [0]
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
            first = True
            while (left is None or left > 0) and not errors:
                want = min(BATCH_READS, FIRST_BATCH_READS) if first else BATCH_READS
                want = want if left is None else min(want, left)
                first = False
                t0 = time.perf_counter()
                b = reader.next_batch(want, BATCH_BYTES)
                if b is None:
                    break
                if left is not None:
                    left -= len(b)
                if stride is not None:
                    mine = keep_batch[0] % stride[1] == stride[0]
                    keep_batch[0] += 1
                    if not mine:
                        b.close()
                        continue
                n_delivered[0] += len(b)
                t1 = time.perf_counter()
                lane = free_lanes.get()
                if lane is None:      # shutdown after an error elsewhere
                    b.close()
                    break
                t2 = time.perf_counter()
                # 4-bit windows across PCIe (84 instead of 164 bytes per read at -l 80), unpacked on the device; a batch
                # with a 'U' inside a window (the one letter the 4-bit alphabet cannot carry) travels as ASCII
                if ascii_lanes or b.pack_windows4_into(panel.search_len, lane.packed, lane.lens) > 0:
                    b.pack_windows_into(panel.search_len, lane.windows, lane.lens)
                    t3 = time.perf_counter()
                    lane.submit(len(b))
                else:
                    t3 = time.perf_counter()
                    lane.submit_packed(len(b))
                timing["read"] += t1 - t0
                timing["pack"] += t3 - t2
                timing["submit"] += time.perf_counter() - t3
                q_gpu.put((b, lane))
        except BaseException as e:   # surfaced in the main thread
            errors.append(e)
        finally:
            q_gpu.put(None)

    def consume():
        try:
            while True:
                item = q_out.get()
                if item is None:
                    return
                b, lane, ops, extra = item
                t0 = time.perf_counter()
                if not errors:
                    writer.write(b, ops, extra)
                free_lanes.put(lane)
                q_retire.put(b)
                timing["write"] += time.perf_counter() - t0
        except BaseException as e:
            errors.append(e)
            free_lanes.put(None)
            while q_out.get() is not None:   # keep draining so the main thread never blocks
                pass

    q_retire = queue.SimpleQueue()

    def retire():
        # freeing a batch drops the page-table entries of its part of the mapped input (smx_io.cpp smx_batch): off the
        # writer's thread, which is the longest stage; the reader's mapping goes last
        while True:
            b = q_retire.get()
            if b is None:
                break
            b.close()
        reader.close()

    tp = threading.Thread(target=produce, name="smx-reader", daemon=True)
    tc = threading.Thread(target=consume, name="smx-writer", daemon=True)
    tr = threading.Thread(target=retire, name="smx-retire", daemon=True)
    tp.start()
    tc.start()
    tr.start()
    try:
        # the lanes (pinned staging, device buffers, a stream each) are created while the reader parses its first batch
        for _ in range(N_LANES):
            lanes.append(Lane(panel, BATCH_READS))
            free_lanes.put(lanes[-1])
        while True:
            item = q_gpu.get()
            if item is None:
                break
            b, lane = item
            if errors:
                b.close()
                continue
            t0 = time.perf_counter()
            try:
                ops, extra = lane.wait(counts)
            except _lib.SmxError as e:
                if e.code != _lib.ERR_OVERFLOW or "extra buffer" not in str(e):
                    raise
                # more extra records than a lane holds (pathological tie storms): this batch again, synchronously,
                # with a buffer of the size the kernel asked for
                windows, lens = b.pack_windows(panel.search_len, panel.window_stride)
                ops, extra, _ = panel.run(windows, lens, counts=counts)
            timing["gpu_wait"] += time.perf_counter() - t0
            q_out.put((b, lane, ops, extra))
            if on_batch:
                on_batch(int(counts[_lib.CNT_TOTAL]), int(counts[_lib.CNT_MATCHED]))
    except BaseException as e:
        errors.append(e)
    finally:
        if errors:
            free_lanes.put(None)
        q_out.put(None)
        tc.join()
        while tp.is_alive():   # unblock a producer stuck on a full queue after an error
            try:
                q_gpu.get_nowait()
            except queue.Empty:
                pass
            free_lanes.put(None)
            tp.join(timeout=0.05)
        q_retire.put(None)
    if errors:
        tr.join()
        try:
            writer.close()
        except Exception:
            pass
        for ln in lanes:
            try:
                ln.close()
            except Exception:
                pass
        raise errors[0]
    t0 = time.perf_counter()
    writer.close()
    timing["close"] = time.perf_counter() - t0
    for ln in lanes:
        ln.close()
    tr.join()
    timing["wall"] = time.perf_counter() - t_start
    if int(counts[_lib.CNT_TOTAL]) != n_delivered[0]:
        raise RuntimeError(f"pipeline accounting: the reader delivered {n_delivered[0]} reads, the kernels counted "
                           f"{int(counts[_lib.CNT_TOTAL])}")
    if stats is not None:
        stats.update(timing)
    if os.environ.get("SMX_PIPELINE_STATS_JSON"):
        import json
        with open(os.environ["SMX_PIPELINE_STATS_JSON"], "w") as fh:
            json.dump(timing, fh)
    if os.environ.get("SMX_PIPELINE_TIMING"):
        print("[smx pipeline] wall %.3f s: reader %.3f + pack %.3f + submit %.3f (thread 1) | waiting for the GPU %.3f (main) | "
              "writer %.3f + close %.3f (thread 2)" % (timing["wall"], timing["read"], timing["pack"], timing["submit"],
                                                        timing["gpu_wait"], timing["write"], timing["close"]), file=sys.stderr)
    return int(counts[_lib.CNT_TOTAL]), int(counts[_lib.CNT_MATCHED]), counts, reader.is_fastq
