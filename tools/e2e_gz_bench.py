#!/usr/bin/env python3
"""End-to-end throughput on gzip input (the usual way ONT reads are stored): FASTQ.gz -> output tree through the CLI.
Run on the GPU box: python tools/e2e_gz_bench.py [n_reads]"""
import gzip
import os
import shutil
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from specimux_amd import cli, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
tmp = tempfile.mkdtemp(prefix="smx_e2egz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    pan = synth.panel_c2()
    pf, sf = pan.write(tmp)
    rs = synth.make_reads(pan, n, 2002, windows_only=False)
    fq = os.path.join(tmp, "reads.fastq")
    rs.write_fastq(fq)
    gz = fq + ".gz"
    t0 = time.time()
    with open(fq, "rb") as src, gzip.open(gz, "wb", compresslevel=4) as dst:
        shutil.copyfileobj(src, dst, 1 << 24)
    raw, comp = os.path.getsize(fq), os.path.getsize(gz)
    os.unlink(fq)
    print(f"{n} reads, {raw / 1e6:.0f} MB FASTQ -> {comp / 1e6:.0f} MB gz ({time.time() - t0:.1f} s to compress)", flush=True)
    for rep in range(2):
        out = os.path.join(tmp, f"out{rep}")
        t0 = time.time()
        cli.main(["specimux", pf, sf, gz, "-F", "-O", out])
        dt = time.time() - t0
        print(f"run {rep}: {dt:.2f} s  {n / dt:,.0f} reads/s  {raw / dt / 1e6:,.0f} MB/s of inflated FASTQ", flush=True)
        shutil.rmtree(out)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
