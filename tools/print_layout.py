#!/usr/bin/env python3
"""Tile geometry of the demux kernel for the bench panels (SMX_DEBUG output of libsmx): LDS bytes per tile and the
workgroups per CU the launch glue counts on.  Run on a GPU box:  python tools/print_layout.py"""
import os
import sys
import tempfile

os.environ["SMX_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from specimux_amd import synth  # noqa: E402

d = tempfile.mkdtemp(prefix="smx_layout_")
for config, pan in (("c2", synth.panel_c2(2002)), ("c3", synth.panel_c3()), ("c5", synth.panel_c3())):
    sub = os.path.join(d, config)
    os.makedirs(sub)
    pf, sf = pan.write(sub)
    for trim in (None, "tails", "primers"):
        cp, par = bench.build_compiled_panel(pf, sf, config, trim)
        S = cp.search_len
        rs = synth.make_reads(pan, 4096, 1, search_len=S)
        print(f"== {config} trim={trim or 'barcodes'} S={S}", file=sys.stderr, flush=True)
        cp.run(rs.windows(cp.window_stride), rs.lens)
