#!/usr/bin/env python3
"""CONTEXT ONLY (BASELINE.md section 2, B-native): build and run tools/cpu_native/cpu_scan.c -- Myers bit-vector scans,
OpenMP over reads -- on the configs[1] workload of bench.py and print one JSON line.  Nothing in the package, the tests
or bench.py imports this; it exists so that gpu_over_cpu (measured against the Python-loop oracle) can be read next
to a native CPU figure for the alignment work.
    python tools/cpu_native/run.py [n_reads]     (any host; uses all cores it may use)"""
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from specimux_amd import synth  # noqa: E402  (workload generator only)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 765000
tmp = tempfile.mkdtemp(prefix="smx_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
exe = os.path.join(tempfile.mkdtemp(prefix="smx_cpu_exe_"), "cpu_scan")   # /dev/shm may be mounted noexec
subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", os.path.join(REPO, "tools", "cpu_native", "cpu_scan.c"), "-o", exe])
pan = synth.panel_c2(2002)
rs = synth.make_reads(pan, n, 2002, workers=8)
keep = rs.lens >= 80                       # the comparator handles full-length ACGT windows only
win = rs.windows(160)[keep]
path = os.path.join(tmp, "windows.bin")
win.tofile(path)
fwd_rc = [synth.revcomp(b) for b in pan.fwd]          # barcodes as searched: reverse complements
rev_rc = [synth.revcomp(b) for b in pan.rev]
args = [exe, path, str(len(win)), "80", "3", synth.revcomp(synth.ITS1F), "7", synth.revcomp(synth.ITS4), "6",
        str(len(fwd_rc))] + fwd_rc + [str(len(rev_rc))] + rev_rc
out = json.loads(subprocess.check_output(args).decode())
out["note"] = ("alignment work only (4 primer HW scans per read + barcode SHW scans of matched ends), Myers 64-bit, "
               "gcc -O3 -march=native -fopenmp; an upper bound on a native CPU demultiplexer of this design")
print(json.dumps(out))
