#!/usr/bin/env python3
"""Host side of the file pipeline without a GPU: native reader -> window packer -> (fabricated records) -> native writer,
the same calls and batch sizes as specimux_amd/pipeline.py, stage by stage and overlapped.  For tuning smx_io.cpp on a box
without a GPU:  python tools/io_bench.py [--reads 300000]"""
import argparse
import os
import shutil
import sys
import tempfile
import threading
import time
import queue

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fake_ops(n, lens, rng, n_spec):
    from specimux_amd import _lib
    ops = np.zeros(n, dtype=_lib.OP_DTYPE)
    u = rng.random(n)
    full = u < 0.8
    part = (u >= 0.8) & (u < 0.9)
    ops["rtype"] = np.where(full, _lib.R_DEREP_FULL, np.where(part, _lib.R_PARTIAL_FWD, _lib.R_UNKNOWN))
    ops["sample"] = np.where(full, rng.integers(0, n_spec, n), -1)
    ops["pool"] = 0
    ops["p1"] = 0
    ops["p2"] = np.where(full, 1, -1)
    ops["barcode"] = np.where(part, rng.integers(0, 32, n), -1)
    ops["dist"] = rng.integers(0, 4, (n, 4))
    ops["trim_start"] = np.minimum(35, lens // 4)
    ops["trim_end"] = np.maximum(lens - 35, ops["trim_start"] + 1)
    ops["flags"] = rng.integers(0, 2, n)
    ops["n_ops"] = 1
    ops["read"] = np.arange(n)
    return ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=300000)
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    from specimux_amd import _lib, synth
    from specimux_amd.native_io import Reader, Writer
    from specimux_amd.pipeline import BATCH_BYTES, BATCH_READS
    _lib.load()
    pan = synth.panel_c2(2002)
    d = tempfile.mkdtemp(prefix="smx_io_", dir="/dev/shm")
    pf, sf = pan.write(d)
    rs = synth.make_reads(pan, a.reads, 2002, workers=8)
    fq = os.path.join(d, "reads.fastq")
    rs.write_fastq_rebuilt(fq, 80, 1)
    size = os.path.getsize(fq)

    class FakePanel:   # what Writer needs of a CompiledPanel
        specimen_ids = [f"ITS_F{i:02d}_R{j:02d}" for i in range(32) for j in range(24)]
        pools = ["ITS"]
        primer_names = ["ITS1F", "ITS4"]
        barcodes = pan.fwd + pan.rev
    rng = np.random.default_rng(3)
    for rep in range(3):
        out = os.path.join(d, f"out{rep}")
        t = dict(read=0.0, pack=0.0, write=0.0, close=0.0)
        t0 = time.perf_counter()
        reader = Reader(fq)
        writer = Writer(out, "", True, FakePanel)
        q = queue.Queue(maxsize=3)

        def consume():
            while True:
                item = q.get()
                if item is None:
                    return
                b, ops = item
                t1 = time.perf_counter()
                writer.write(b, ops, np.zeros(0, dtype=_lib.OP_DTYPE))
                b.close()
                t["write"] += time.perf_counter() - t1
        th = threading.Thread(target=consume)
        th.start()
        packed = np.zeros((BATCH_READS, 80), dtype=np.uint8)
        lens = np.zeros(BATCH_READS, dtype=np.int32)
        n_total = 0
        while True:
            t1 = time.perf_counter()
            b = reader.next_batch(BATCH_READS, BATCH_BYTES)
            if b is None:
                break
            t2 = time.perf_counter()
            b.pack_windows4_into(80, packed, lens)
            t3 = time.perf_counter()
            ops = fake_ops(len(b), lens[:len(b)], rng, 768)
            t["read"] += t2 - t1
            t["pack"] += t3 - t2
            n_total += len(b)
            q.put((b, ops))
        q.put(None)
        th.join()
        t1 = time.perf_counter()
        writer.close()
        t["close"] = time.perf_counter() - t1
        reader.close()
        wall = time.perf_counter() - t0
        nbytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _s, fs in os.walk(out) for f in fs)
        print(f"rep {rep}: {n_total} reads, wall {wall:.3f} s = {n_total / wall:.3e} reads/s | in {size / 1e9:.2f} GB, out {nbytes / 1e9:.2f} GB "
              f"| read {t['read']:.3f} pack {t['pack']:.3f} write {t['write']:.3f} close {t['close']:.3f}", flush=True)
        shutil.rmtree(out, ignore_errors=True)
    if not a.keep:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
