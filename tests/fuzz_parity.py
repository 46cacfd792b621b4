#!/usr/bin/env python3
"""Manual, longer-running parity fuzz (not collected by pytest): GPU records vs the oracle over many seeds, panel
shapes, flag sets and search lengths.  Run on a GPU box:  python tests/fuzz_parity.py [--seeds 8] [--reads 1200]"""
import argparse
import itertools
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from parity_utils import Both  # noqa: E402


def reads_of(rs, n, S):
    from parity_utils import reads_from_set
    return reads_from_set(rs, range(n), S)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--reads", type=int, default=1200)
    ap.add_argument("--trim", default=None, help="force this trim mode onto every flag set (e.g. tails)")
    ap.add_argument("--index-k", type=int, default=None, help="force this index edit distance onto every flag set")
    ap.add_argument("--panel", default=None, choices=["c1", "c2", "c3"], help="this panel only (default: all three in rotation)")
    a = ap.parse_args()
    from specimux_amd import synth
    from parity_utils import FUZZ_FLAG_SETS
    flag_sets = [dict(f) for f in FUZZ_FLAG_SETS]
    if a.trim:
        flag_sets = [dict(f, trim=a.trim) for f in flag_sets if "trim" not in f]
    if a.index_k is not None:
        flag_sets = [dict(f, index_edit_distance=a.index_k) for f in flag_sets if "index_edit_distance" not in f]
    tmp = tempfile.mkdtemp(prefix="smx_fuzz_")
    panels = {"c2": synth.panel_c2(), "c3": synth.panel_c3(), "c1": synth.panel_c1()}
    files = {}
    for name, pan in panels.items():
        d = os.path.join(tmp, name)
        os.makedirs(d)
        files[name] = pan.write(d)
    t0 = time.time()
    checked = 0
    for seed, (fi, flags) in itertools.product(range(a.seeds), enumerate(flag_sets)):
        name = a.panel or ("c2", "c3", "c1")[(seed + fi) % 3]
        pan, (pf, sf) = panels[name], files[name]
        flags = dict(flags)
        gen = {k: flags.pop(k) for k in ("error_rate", "n_frac") if k in flags}   # generator-only knobs
        S = flags.get("search_len", 80)
        rs = synth.make_reads(pan, a.reads, 9000 + 131 * seed + fi, search_len=S, windows_only=False, **gen)
        reads = reads_of(rs, a.reads, S)
        both = Both(pf, sf, **flags)
        both.assert_hits_equal(reads[:100], f"{name} seed {seed} {flags}")
        both.assert_ops_equal(reads, f"{name} seed {seed} {flags}")
        checked += len(reads)
        print(f"ok {name} seed {seed} {flags}  ({checked} reads, {time.time() - t0:.0f} s)", flush=True)
    print(f"fuzz parity OK: {checked} reads")


if __name__ == "__main__":
    main()
