#!/usr/bin/env python3
"""End-to-end throughput: FASTQ file -> output tree through the CLI entry point (native reader, GPU, native writer).
Run on the GPU box: python tools/e2e_bench.py [n_reads]"""
import os
import shutil
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from specimux_amd import cli, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
tmp = tempfile.mkdtemp(prefix="smx_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    pan = synth.panel_c2()
    pf, sf = pan.write(tmp)
    t0 = time.time()
    rs = synth.make_reads(pan, n, 2002, windows_only=False)
    fq = os.path.join(tmp, "reads.fastq")
    rs.write_fastq(fq)
    size = os.path.getsize(fq)
    print(f"generated {n} reads, {size / 1e6:.0f} MB FASTQ in {time.time() - t0:.1f} s", flush=True)
    for rep in range(2):
        out = os.path.join(tmp, f"out{rep}")
        t0 = time.time()
        cli.main(["specimux", pf, sf, fq, "-F", "-O", out])
        dt = time.time() - t0
        print(f"run {rep}: {dt:.2f} s  {n / dt:,.0f} reads/s  {size / dt / 1e6:,.0f} MB/s input", flush=True)
        shutil.rmtree(out)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
