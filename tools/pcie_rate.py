#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: host windows in -> records out through smx_batch_run (H2D copy, kernel, D2H copy)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import specimux_amd as sa
from specimux_amd import synth, cli
from specimux_amd.bloom_filter import BloomPrefilter, barcodes_for_bloom_prefilter
from specimux_amd.demultiplex import compiled_panel
pan = synth.panel_c2(); d = tempfile.mkdtemp(); pf, sf = pan.write(d)
args = cli.parse_args(["specimux", pf, sf, "x.fastq"])
reg = sa.read_primers_file(pf); sp = sa.read_specimen_file(sf, reg); sp.validate()
par = sa.setup_match_parameters(args, sp)
cp = compiled_panel(sp, par, args, BloomPrefilter(barcodes_for_bloom_prefilter(sp), par.max_dist_index))
n = 765000
rs = synth.make_reads(pan, n, 2002)
w = rs.windows(cp.window_stride)
cp.run(w, rs.lens)
ts = []
for _ in range(5):
    t = time.perf_counter(); cp.run(w, rs.lens); ts.append(time.perf_counter() - t)
print(f"smx_batch_run host->host, {n} reads: median {sorted(ts)[2]*1e3:.1f} ms = {n/sorted(ts)[2]/1e6:.1f} M reads/s")
