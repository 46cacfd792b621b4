import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from specimux_amd import synth, cli
import specimux_amd as sa
from specimux_amd.bloom_filter import BloomPrefilter, barcodes_for_bloom_prefilter
from specimux_amd.demultiplex import compiled_panel
import tempfile
pan = synth.panel_c2(); d = tempfile.mkdtemp(); pf, sf = pan.write(d)
args = cli.parse_args(["specimux", pf, sf, "x.fastq"])
reg = sa.read_primers_file(pf); sp = sa.read_specimen_file(sf, reg); sp.validate()
par = sa.setup_match_parameters(args, sp)
cp = compiled_panel(sp, par, args, BloomPrefilter(barcodes_for_bloom_prefilter(sp), par.max_dist_index))
rs = synth.make_reads(pan, 200000, 2002)
ops, extra, counts = cp.run(rs.windows(cp.window_stride), rs.lens)
print("multi-op reads", counts[6], "extra", len(extra))
