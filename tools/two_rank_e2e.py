#!/usr/bin/env python3
"""File -> tree with one process and with two ranks on the ONE GPU of the box (gloo for the counts): what the per-rank trees
and their merge cost beside the demultiplexing itself.  python tools/two_rank_e2e.py [--reads 765000]"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=765000)
    a = ap.parse_args()
    from specimux_amd import synth
    d = tempfile.mkdtemp(prefix="smx_2rank_", dir="/dev/shm")
    pan = synth.panel_c2(2002)
    pf, sf = pan.write(d)
    rs = synth.make_reads(pan, a.reads, 2002, workers=16)
    fq = os.path.join(d, "reads.fastq")
    rs.write_fastq_rebuilt(fq, 80, 2002)
    env = dict(os.environ, PYTHONPATH=REPO, SMX_DIST_BACKEND="gloo")
    for world in (1, 2, 1, 2):
        out = os.path.join(d, f"out{world}")
        shutil.rmtree(out, ignore_errors=True)
        cli = ["-m", "specimux_amd.cli", pf, sf, fq, "-F", "-O", out]
        cmd = [sys.executable] + cli if world == 1 else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
             "--master-port", "29611"] + cli
        t0 = time.perf_counter()
        p = subprocess.run(cmd, env=env, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        el = [ln for ln in (p.stderr + p.stdout).splitlines() if "Elapsed time" in ln or "merge" in ln.lower()]
        nfiles = sum(len(fs) for _d, _s, fs in os.walk(out))
        print(f"world {world}: process wall {dt:.2f} s (incl. interpreter / torch start-up), rc {p.returncode}, {nfiles} files; {el[-2:]}", flush=True)
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
