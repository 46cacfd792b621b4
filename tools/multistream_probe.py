#!/usr/bin/env python3
"""Probe: kernel-resident throughput with the batches spread over 1 / 2 / 3 HIP streams (what the product's lanes do), for a
given number of resident demux workgroups per CU (SMX_BLOCKS_PER_CU): fewer of them leave room for the next batch's
prescan kernels to run beside the demux kernel.   python tools/multistream_probe.py c2|c3|c5"""
import os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from specimux_amd import _lib, synth

def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n = 765000 if cfg == "c2" else 1000000
    pan = synth.panel_c2(2002) if cfg == "c2" else synth.panel_c3(2002)
    tmp = tempfile.mkdtemp()
    pf, sf = pan.write(tmp)
    kw = dict(search_len=160, error_rate=0.15) if cfg == "c5" else {}
    sets = [synth.make_reads(pan, n, 2002 + 1000 * b, workers=16, **kw) for b in range(3)]
    lib = _lib.load()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    for bpc in (None, "3", "2"):
        if bpc is None:
            os.environ.pop("SMX_BLOCKS_PER_CU", None)
        else:
            os.environ["SMX_BLOCKS_PER_CU"] = bpc
        for ns in (1, 2, 3):
            cp, _ = bench.build_compiled_panel(pf, sf, cfg)
            streams = [torch.cuda.Stream() for _ in range(ns)]
            dbs = [bench.DeviceBatches(lib, cp, sets, dev, s) for s in streams]
            for i in range(6):
                dbs[i % ns].step(i)
            torch.cuda.synchronize()
            K = 30
            t0 = time.perf_counter()
            for i in range(K):
                dbs[i % ns].step(i // ns)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(cfg, "blocks/CU", bpc or "default", ns, "streams: %.4f ms per batch, %.3e reads/s" % (dt / K * 1e3, K * n / dt), flush=True)
            del dbs
            torch.cuda.empty_cache()

main()
