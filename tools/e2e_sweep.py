#!/usr/bin/env python3
"""File -> tree through the CLI on the GPU box under different host settings (one subprocess per setting: the I/O thread
count is fixed per process).  python tools/e2e_sweep.py [--reads 765000]  -> one line per setting."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

CHILD = r'''
import json, os, sys, time, shutil
sys.path.insert(0, %r)
from specimux_amd import cli
pf, sf, fq, out = sys.argv[1:5]
best = None
for rep in range(3):
    shutil.rmtree(out, ignore_errors=True)
    t0 = time.perf_counter()
    cli.main(["specimux", pf, sf, fq, "-F", "-O", out])
    dt = time.perf_counter() - t0
    st = json.load(open(os.environ["SMX_PIPELINE_STATS_JSON"]))
    if rep and (best is None or dt < best[0]):
        best = (dt, st)
shutil.rmtree(out, ignore_errors=True)
print("RESULT " + json.dumps({"seconds": best[0], "stages": {k: round(v, 3) for k, v in best[1].items()}}))
''' % REPO


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=765000)
    ap.add_argument("--settings", default="")
    a = ap.parse_args()
    from specimux_amd import synth
    d = tempfile.mkdtemp(prefix="smx_sweep_", dir="/dev/shm")
    pan = synth.panel_c2(2002)
    pf, sf = pan.write(d)
    rs = synth.make_reads(pan, a.reads, 2002, workers=16)
    fq = os.path.join(d, "reads.fastq")
    rs.write_fastq_rebuilt(fq, 80, 2002)
    settings = [dict()] + [dict(kv.split("=") for kv in s.split(",")) for s in a.settings.split(";") if s]
    for env in settings:
        e = dict(os.environ, SMX_PIPELINE_STATS_JSON=os.path.join(d, "stats.json"), PYTHONPATH=REPO, **env)
        p = subprocess.run([sys.executable, "-c", CHILD, pf, sf, fq, os.path.join(d, "out")], env=e, capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print(env, "FAILED", p.stderr[-400:], flush=True)
            continue
        r = json.loads(line[0][7:])
        print(json.dumps({"env": env, "reads_per_s": a.reads / r["seconds"], **r}), flush=True)
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
