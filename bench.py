#!/usr/bin/env python3
"""bench.py -- reads/sec demultiplexed, 768-specimen ITS panel, 765k ONT-style reads (BASELINE.json).

One "step" = one pass of the hot path (libsmx: primer prescan kernels + demux kernel = primer scan, barcode scan,
scorer) over one whole 765 000-read batch of configs[1], end windows already resident in HBM.  The timed loop ROTATES
over three distinct resident batches (3 x 122 MB of windows: more than the 256 MiB Infinity Cache), so no step finds
its input in the last-level cache; the rate on one batch relaunched in place is reported beside it (`single_buffer`).
`--gpus N` runs one process per GPU: under torch.distributed.run (RANK / WORLD_SIZE set) this process is one rank;
started plainly with --gpus N > 1 it launches its own N ranks (python -m torch.distributed.run, before anything here
touches a GPU), relays rank 0's line and exits with the ranks' status.  Every rank owns its own 765k-read shards
(seed + rank, weak scaling, no data-path collective) and the per-specimen counts are summed once at the end with RCCL
(smx_counts_allreduce, C ABI).  Rank 0 prints ONE JSON line.

`value` is the kernel-resident rate the contract asks for.  The same line carries the wider scopes, so that no
ratio has to mix scopes (N = 1 only, after the timed region):
  step_kernels    device time of every kernel of a step (HIP events on the launch stream, smx_debug_kernel_times)
  roofline        vs the 8 TB/s HBM3E peak, ALGORITHMIC bytes = 196 B/read (2*search_len window bytes + 4 length +
                  32 result record) over ALL kernels of a step -- the same scope as `value`;
                  `dominant_kernel` = the same bytes over the demux kernel alone.  `traffic` = HBM bytes per step from
                  the committed rocprofv3 PMC passes of this workload (profiles/, replayed: not measured in this run).
  other_configs   the configs[2]- and configs[4]-shaped workloads (3072 specimens / 8 primers; + -l 160, 15 % errors),
                  1 M reads per step, timed here after the headline region: value, per-kernel ms, path roofline
  valu_roofline   the bound that binds (integer VALU issue): wave64 VALU instructions per step (committed PMC summary,
                  replayed) over the measured kernel time, against the guide's peak (one wave64 VALU per SIMD-32 every
                  2 cycles) and against the measured-mix bound of tools/ubench/issue_rate.hip (2.75 cycles).
  pcie_inclusive  host windows in, records out through the pinned, asynchronous lanes (smx_lane_*), three lanes in flight
  end_to_end      FASTQ file in /dev/shm -> output tree through the CLI pipeline (native reader, lanes, native writer):
                  reads/s, input GB/s, stage seconds, share of the wall time the main thread was not waiting for the GPU
  cpu_baseline    the oracle (reference-shaped Python loop + C DP aligner) on a bounded sample of the same reads, all host
                  cores: `value` over in-memory reads (same scope as `value`), `end_to_end` over FASTQ files -> trees
                  (same scope as end_to_end).  `gpu_over_cpu` lists the two like-for-like ratios.
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_READS = 765_000
SEED = 2002
SEARCH_LEN = 80
BYTES_PER_READ = 2 * SEARCH_LEN + 4 + 32    # algorithmic bytes per read = 196
HBM_PEAK_GBPS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GUIDE = 256 * 4 * 2.4e9 / 2.0     # wave64 VALU instr/s: 1024 SIMD-32, one per 2 cycles (MI355X_MICROARCH.md)
VALU_PEAK_MEASURED_MIX = 256 * 4 * 2.4e9 / 2.75   # tools/ubench/issue_rate.hip: v_and / v_bitop3 mix, 4 waves per SIMD


# ------------------------------------------------------------------ cpu baseline (oracle, test infrastructure)
_cpu_state = {}


def _cpu_init(pf, sf):
    from oracle import specimux_oracle as O
    panel = O.load_panel(pf, sf)
    _cpu_state["O"] = O
    _cpu_state["panel"] = panel
    _cpu_state["par"] = O.setup_params(panel)
    _cpu_state["files"] = (pf, sf)


def _cpu_work(reads):
    O = _cpu_state["O"]
    ops, total, matched = O.process_sequences(reads, _cpu_state["par"], _cpu_state["panel"])
    return total, matched


def _cpu_file_work(path):
    """One worker's share of the end-to-end CPU leg: FASTQ file -> parsed records -> oracle -> record texts written to a
    tree (the oracle's run_files + plain appends: the reference's worker + writer, without its locking)."""
    O = _cpu_state["O"]
    pf, sf = _cpu_state["files"]
    tree, total, matched = O.run_files(pf, sf, path)
    out = path + ".out"
    for rel, recs in tree.items():
        full = os.path.join(out, rel)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, "a") as fh:
            fh.write("".join(recs))
    return total, matched


def host_cores():
    """CPU share of this process: cgroup quota if one is set, else the affinity mask; capped at 64 workers."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 64)


def cpu_baseline(pf, sf, rs, e2e_dir, budget_reads_per_core=32000, file_reads_per_core=8000):
    """Reference-shaped CPU path (the oracle) on bounded samples, all host cores, 1000-read batches like the
    reference's worker pool (orchestration.py:165).  Two scopes: in-memory reads (what `value` covers) and FASTQ files
    -> output trees (what `end_to_end` covers)."""
    from specimux_amd.synth import rebuild_read
    cores = host_cores()
    n = min(len(rs.lens), budget_reads_per_core * cores)
    reads = [(f"r{i}", rebuild_read(rs.head[i], rs.tail[i], int(rs.lens[i]), SEARCH_LEN), None) for i in range(n)]
    reads = [(i, s, "I" * len(s)) for i, s, _ in reads]
    batches = [reads[i:i + 1000] for i in range(0, n, 1000)]
    nf = min(n, file_reads_per_core * cores)
    paths = []
    for w in range(cores):   # one FASTQ file per worker: the end-to-end sample
        lo, hi = nf * w // cores, nf * (w + 1) // cores
        path = os.path.join(e2e_dir, f"cpu_{w}.fastq")
        with open(path, "w") as fh:
            for rid, s, q in reads[lo:hi]:
                fh.write(f"@{rid} synthetic\n{s}\n+\n{q}\n")
        paths.append(path)
    ctx = mp.get_context("fork")    # forked BEFORE this process touches the GPU
    with ctx.Pool(cores, initializer=_cpu_init, initargs=(pf, sf)) as pool:
        pool.map(_cpu_work, batches[:cores])         # warm: imports, panel, C library
        t0 = time.perf_counter()
        res = pool.map(_cpu_work, batches)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        fres = pool.map(_cpu_file_work, paths)
        fdt = time.perf_counter() - t0
    total = sum(t for t, _ in res)
    return {"value": total / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {n} of the {len(rs.lens)} reads, {dt:.1f} s wall, oracle/specimux_oracle.py "
                      f"(reference loop order, one C DP alignment per call), multiprocessing fork x{cores}",
            "matched_fraction": sum(m for _, m in res) / max(total, 1),
            "end_to_end": {"value": sum(t for t, _ in fres) / fdt, "unit": "reads/s",
                           "sample": f"first {nf} reads as {cores} FASTQ files -> {cores} output trees, {fdt:.1f} s wall "
                                     f"(oracle run_files: parse, demultiplex, format, append)"}}


# ------------------------------------------------------------------ wider scopes (N = 1, after the timed region)
def pcie_inclusive(cp, rs, rounds=6):
    """Host windows in, records out: three pinned lanes in flight (H2D of one batch overlaps the kernels of the previous
    and the D2H of the one before).  The windows cross the link as 4-bit codes (smx_lane_submit_packed: what the streaming
    pipeline ships) and are unpacked on the device; `ascii_windows` = the same with 8-bit windows.  The packer's pass over
    the bases is not part of it (staging filled once)."""
    from specimux_amd.native_io import Lane
    n = len(rs.lens)
    lanes = [Lane(cp, n) for _ in range(3)]
    counts = np.zeros(cp.counts_len, dtype=np.uint64)

    def run(packed):
        for ln in lanes:               # warm
            (ln.submit_packed if packed else ln.submit)(n)
        for ln in lanes:
            ln.wait(counts)
        t0 = time.perf_counter()
        inflight = []
        for i in range(rounds):
            ln = lanes[i % 3]
            if len(inflight) == 3:
                inflight.pop(0).wait(counts)
            (ln.submit_packed if packed else ln.submit)(n)
            inflight.append(ln)
        for ln in inflight:
            ln.wait(counts)
        dt = time.perf_counter() - t0
        h2d = n * ((lanes[0].packed_stride if packed else cp.window_stride) + 4)
        return {"value": rounds * n / dt, "unit": "reads/s", "ms_per_batch": dt / rounds * 1e3, "lanes": 3,
                "h2d_bytes_per_batch": h2d, "d2h_bytes_per_batch": n * 32, "h2d_gbps": rounds * h2d / dt / 1e9}
    try:
        windows = rs.windows(cp.window_stride)
        for ln in lanes:
            ln.windows[:n] = windows
            ln.lens[:n] = rs.lens
        asc = run(False)
        pk = rs.packed_windows(lanes[0].packed_stride)
        for ln in lanes:
            ln.packed[:n] = pk
        out = run(True)
    finally:
        for ln in lanes:
            ln.close()
    out["ascii_windows"] = asc
    out["note"] = ("pinned staging -> hipMemcpyAsync H2D of 4-bit windows -> unpack kernel -> prescan + demux kernels -> D2H of the "
                   "32-byte records, own stream per lane")
    return out


def tmpfs_write_ceiling(directory, nbytes, threads):
    """A plain `threads`-way write of `nbytes` into `directory` (the file system the end-to-end leg writes its tree to):
    what the output side of the pipeline could reach if formatting cost nothing.  Returns GB/s."""
    import threading
    d = os.path.join(directory, "ceiling")
    os.makedirs(d, exist_ok=True)
    block = bytes(8 << 20)
    per = max(1, nbytes // threads)

    def work(k):
        fd = os.open(os.path.join(d, f"f{k}"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        left = per
        while left > 0:
            left -= os.write(fd, block[:min(left, len(block))])
        os.close(fd)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    shutil.rmtree(d, ignore_errors=True)
    return per * threads / dt / 1e9


def end_to_end(pan, pf, sf, n_reads, e2e_dir):
    """FASTQ in /dev/shm -> output tree through the product pipeline (what `specimux -F` runs), at the metric's size."""
    from specimux_amd import synth
    rs = synth.make_reads(pan, n_reads, SEED, workers=host_cores())
    fq = os.path.join(e2e_dir, "reads.fastq")
    rs.write_fastq_rebuilt(fq, SEARCH_LEN, SEED)
    size = os.path.getsize(fq)
    # The CLI runs in a process of its own, as a user's does (this one holds gigabytes of synthetic batches, pinned buffers
    # and torch: tearing mappings down in it costs several times what it costs a fresh process); the clock is around
    # cli.main() inside that process, the way the CPU baseline is timed around its own call.
    child = r"""
import json, os, shutil, sys, time
sys.path.insert(0, sys.argv[1])
from specimux_amd import cli
pf, sf, fq, e2e_dir = sys.argv[2:6]
best = None
for rep in range(3):   # first run warms the page cache / allocations; the best of the next two is reported
    out = os.path.join(e2e_dir, f"out{rep}")
    t0 = time.perf_counter()
    cli.main(["specimux", pf, sf, fq, "-F", "-O", out])
    dt = time.perf_counter() - t0
    with open(os.environ["SMX_PIPELINE_STATS_JSON"]) as fh:
        stats = json.load(fh)
    nbytes = sum(os.path.getsize(os.path.join(d, f)) for d, _s, fs in os.walk(out) for f in fs)
    shutil.rmtree(out, ignore_errors=True)
    if rep and (best is None or dt < best["seconds"]):
        best = {"seconds": dt, "output_bytes": nbytes, "stats": stats}
print("RESULT " + json.dumps(best))
"""
    env = dict(os.environ, SMX_PIPELINE_STATS_JSON=os.path.join(e2e_dir, "stats.json"))
    p = subprocess.run([sys.executable, "-c", child, REPO, pf, sf, fq, e2e_dir], env=env, capture_output=True, text=True)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    if p.returncode or not line:
        raise RuntimeError("end_to_end: the CLI process failed:\n" + (p.stderr or p.stdout)[-2000:])
    r = json.loads(line[0][7:])
    dt, stats = r["seconds"], r["stats"]
    best = {"value": n_reads / dt, "unit": "reads/s", "seconds": dt, "reads": n_reads, "input_gbps": size / dt / 1e9,
            "input_bytes": size, "output_bytes": r["output_bytes"], "output_gbps": r["output_bytes"] / dt / 1e9,
            "stage_seconds": {k: round(v, 4) for k, v in stats.items()},
            "main_thread_not_waiting_for_gpu_pct": (100.0 * (1.0 - stats["gpu_wait"] / stats["wall"]) if stats.get("wall") else None),
            "note": "CLI entry point (cli.main in a process of its own, clock around the call) incl. panel compilation and "
                    "log/primers side files; reader, lanes and writer overlapped (specimux_amd/pipeline.py)"}
    os.remove(fq)
    if best:
        threads = host_cores()
        gbps = tmpfs_write_ceiling(e2e_dir, best["output_bytes"], threads)
        best["write_ceiling"] = {"gbps": gbps, "threads": threads, "seconds_for_output": best["output_bytes"] / gbps / 1e9,
                                 "reads_per_s_if_write_bound": n_reads / (best["output_bytes"] / gbps / 1e9),
                                 "note": "plain multi-threaded write of the same number of bytes into the same tmpfs"}
    return best


def committed_profile(name):
    path = os.path.join(REPO, "profiles", name)
    if os.path.exists(path):
        with open(path) as fh:
            return json.load(fh)
    return None


# ------------------------------------------------------------------ one configuration on the device
N_ROTATE = 3    # distinct resident input batches the timed loop cycles through (3 x 122 MB > the 256 MiB Infinity Cache)


def build_compiled_panel(pf, sf, config, trim=None, cli_args=""):
    import specimux_amd as sa
    from specimux_amd.bloom_filter import BloomPrefilter, barcodes_for_bloom_prefilter
    from specimux_amd.cli import parse_args
    from specimux_amd.demultiplex import compiled_panel
    args = parse_args(["specimux", pf, sf, "reads.fastq"] + (["-l", "160"] if config == "c5" else []) +
                      (["--trim", trim] if trim else []) + cli_args.split())   # default flags
    reg = sa.read_primers_file(pf)
    specimens = sa.read_specimen_file(sf, reg)
    specimens.validate()
    parameters = sa.setup_match_parameters(args, specimens)
    prefilter = BloomPrefilter(barcodes_for_bloom_prefilter(specimens), parameters.max_dist_index)
    return compiled_panel(specimens, parameters, args, prefilter), parameters


class DeviceBatches:
    """N resident input batches of one shape, and one set of output buffers per HIP stream; step(i) runs the hot path on
    batch i % N, on stream i % n_streams (round robin: what a caller with several batches in flight does -- the product's
    lanes are that caller)."""

    def __init__(self, lib, cp, read_sets, dev, streams):
        import torch
        from specimux_amd import _lib
        self.lib, self.cp, self._lib = lib, cp, _lib
        self.streams = list(streams) if isinstance(streams, (list, tuple)) else [streams]
        self.stream = self.streams[0]
        self.n = len(read_sets[0].lens)
        assert all(len(r.lens) == self.n for r in read_sets)
        self.windows = [torch.from_numpy(r.windows(cp.window_stride)).to(dev) for r in read_sets]
        self.lens = [torch.from_numpy(r.lens).to(dev) for r in read_sets]
        self.extra_cap = self.n
        # one counts vector for all streams (the kernels add to it with global atomics); records per stream
        self.counts = torch.zeros(cp.counts_len, dtype=torch.int64, device=dev)
        self.out = [dict(ops=torch.empty(self.n * 32, dtype=torch.uint8, device=dev),
                         extra=torch.empty(self.extra_cap * 32, dtype=torch.uint8, device=dev),
                         nextra=torch.zeros(4, dtype=torch.int32, device=dev)) for _ in self.streams]
        self.nextra = self.out[0]["nextra"]
        torch.cuda.synchronize()

    def step(self, i=0, n_streams=None):
        b = i % len(self.windows)
        k = i % (n_streams or len(self.streams))
        o, st = self.out[k], self.streams[k]
        self._lib.check(self.lib.smx_batch_run_device(
            self.cp.handle, C.c_void_p(st.cuda_stream), C.c_void_p(self.windows[b].data_ptr()),
            C.c_void_p(self.lens[b].data_ptr()), self.n, C.c_void_p(o["ops"].data_ptr()), C.c_void_p(o["extra"].data_ptr()),
            self.extra_cap, C.c_void_p(o["nextra"].data_ptr()), C.c_void_p(self.counts.data_ptr()), None, None))

    def zero_counts(self):
        self.counts.zero_()

    def max_extra(self):
        return max(int(o["nextra"][0].item()) for o in self.out)

    def kernel_times(self, reps=6):
        """Mean device time of the three kernels of a step (HIP events on the launch stream), one batch at a time on one
        stream with the launches at their full size, rotating over the resident batches like the timed loop."""
        import torch
        torch.cuda.synchronize()
        self.cp.set_streams(1)
        kms = np.zeros((reps, 3), dtype=np.float32)
        self._lib.check(self.lib.smx_debug_kernel_times(self.cp.handle, 1, None))
        for i in range(reps):
            self.step(i, 1)
            self._lib.check(self.lib.smx_debug_kernel_times(self.cp.handle, 1, kms[i].ctypes.data_as(C.POINTER(C.c_float))))
        self._lib.check(self.lib.smx_debug_kernel_times(self.cp.handle, 0, None))
        self.cp.set_streams(len(self.streams))
        return tuple(float(x) for x in kms.mean(axis=0))

    def event_timed(self, steps, rotate=True, n_streams=1):
        """ms per step over `steps` launches (after the headline region): wall clock between two device synchronisations."""
        import torch
        self.cp.set_streams(n_streams)
        for i in range(2):
            self.step(i if rotate else 0, n_streams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(i if rotate else 0, n_streams)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.cp.set_streams(len(self.streams))
        return dt / steps * 1e3


def step_kernel_list(k_t, k_d, k_b, demux_name):
    return [{"kernel": "smx::prescan_transpose_kernel", "ms": k_t}, {"kernel": "smx::prescan_dp_kernel", "ms": k_d},
            {"kernel": demux_name, "ms": k_b}]


def side_config(lib, config, dev, stream, tmp, n_reads=1_000_000, steps=10):
    """configs[2]- / configs[4]-shaped workload on this GPU, timed after the headline region (N = 1 only)."""
    import torch
    from specimux_amd import _lib, synth
    pan = synth.panel_c3(SEED)
    d = os.path.join(tmp, config)
    pf, sf = pan.write(d)
    cp, _par = build_compiled_panel(pf, sf, config)
    gen_kw = dict(search_len=160, error_rate=0.15) if config == "c5" else {}
    sets = [synth.make_reads(pan, n_reads, SEED + 7000 * (b + 1), workers=host_cores(), **gen_kw) for b in range(2)]
    db = DeviceBatches(lib, cp, sets, dev, stream)
    for i in range(3):
        db.step(i)
    ms = db.event_timed(steps)
    counts = db.counts.cpu().numpy().astype(np.uint64)
    assert counts[_lib.CNT_TOTAL] == n_reads * (steps + 5) and counts[_lib.CNT_OVERFLOW] == 0
    k_t, k_d, k_b = db.kernel_times(4)
    bpr = (2 * 160 + 4 + 32) if config == "c5" else BYTES_PER_READ
    kernels_ms = k_t + k_d + k_b
    ach = bpr * n_reads / (kernels_ms * 1e-3) / 1e9
    out = {"workload": ("configs[2]-style: 3072 specimens over 4 pools (ITS/RPB2/LSU/TEF1, ITS4 shared, degenerate primers), "
                        "default flags" if config == "c3" else
                        "configs[4]-style: the 3072-specimen panel, 15 % error reads, -l 160"),
           "reads_per_step": n_reads, "steps": steps, "resident_batches": 2, "value": n_reads / (ms * 1e-3), "unit": "reads/s",
           "ms_per_step": ms, "matched_fraction": float(counts[_lib.CNT_MATCHED] / counts[_lib.CNT_TOTAL]),
           "step_kernels": step_kernel_list(k_t, k_d, k_b, "smx::demux_kernel<unsigned int, 256, 1, 1, ...> (compact + redo launch)"),
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                        "algorithmic_bytes_per_read": bpr, "kernels_ms": kernels_ms}}
    prof = committed_profile(f"r03_{config}_pmc_summary.json")
    if prof and prof.get("reads_per_launch") == n_reads:
        out["roofline"]["traffic"] = prof.get("hbm_bytes_per_step")
        out["roofline"]["traffic_source"] = (f"profiles/r03_{config}_pmc_summary.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE over the "
                                             "kernels of a step; replayed, not measured in this run)")
        if prof.get("valu_instr_per_step"):
            out["valu_roofline"] = {"instr_per_step": prof["valu_instr_per_step"],
                                    "frac": prof["valu_instr_per_step"] / (kernels_ms * 1e-3) / VALU_PEAK_GUIDE,
                                    "source": "instruction counts replayed from the same committed profile"}
    del db
    torch.cuda.empty_cache()
    return out


def self_launch(a, argv):
    """`python3 bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU,
    torch.distributed.run), before this process has touched a GPU; relay rank 0's JSON line; exit with their status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line)
    sys.exit(proc.returncode if proc.returncode != 0 or line else 1)


# ------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=N_READS, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true", help=argparse.SUPPRESS)   # skip other_configs / pcie_inclusive / end_to_end (profiling runs)
    ap.add_argument("--e2e-reads", type=int, default=N_READS, help=argparse.SUPPRESS)
    ap.add_argument("--rotate", type=int, default=N_ROTATE, help=argparse.SUPPRESS)   # resident input batches (1: relaunch in place)
    ap.add_argument("--streams", type=int, default=None, help=argparse.SUPPRESS)      # batches in flight (default: 3 on configs[1], else 1)
    ap.add_argument("--trim", default=None, help=argparse.SUPPRESS)   # e.g. tails: not the headline flags, DESIGN.md side numbers
    ap.add_argument("--cli-args", default="", help=argparse.SUPPRESS)  # extra specimux flags for side measurements, e.g. "-e 4"
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c5"], help=argparse.SUPPRESS)   # c3: 3072 specimens / 4 pools; c5: + 160-nt windows, 15 % errors
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    a.gpus = world

    from specimux_amd import synth
    pan = synth.panel_c2(SEED) if a.config == "c2" else synth.panel_c3(SEED)
    tmp = tempfile.mkdtemp(prefix="smx_bench_")
    e2e_dir = tempfile.mkdtemp(prefix="smx_bench_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    pf, sf = pan.write(tmp)
    from specimux_amd.distributed import shard_seed
    gen_kw = dict(search_len=160, error_rate=0.15) if a.config == "c5" else {}
    gen_workers = max(1, host_cores() // world)
    # this rank's shards (weak scaling): batch b of rank r is generated from seed SEED + 1000 b + r
    sets = [synth.make_reads(pan, a.reads, shard_seed(SEED + 1000 * b, rank), workers=gen_workers, **gen_kw)
            for b in range(max(1, a.rotate))]
    rs = sets[0]

    solo = rank == 0 and a.gpus == 1
    cpu = None
    if solo and not a.no_cpu_baseline and a.config == "c2":
        cpu = cpu_baseline(pf, sf, rs, e2e_dir)

    import torch
    import torch.distributed as dist
    from specimux_amd import _lib

    rehearsal = bool(os.environ.get("SMX_BENCH_SAME_DEVICE"))   # 1-GPU box only: every rank on GPU 0, gloo (RCCL refuses two ranks per device)
    if rehearsal:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        sys.exit(f"--gpus {world}: only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    lib = _lib.load()
    cp, parameters = build_compiled_panel(pf, sf, a.config, a.trim, a.cli_args)
    assert (parameters.max_dist_index == 3 or a.cli_args) and len(cp.specimen_ids) == (768 if a.config == "c2" else 3072)

    n = a.reads
    stream = torch.cuda.current_stream()
    # Batches in flight: the timed loop hands the steps round robin to n_streams HIP streams (smx_batch_run_device takes the
    # stream; smx_panel_set_streams sizes each persistent demux launch to its share of the CUs), so that the next batch's
    # memory-bound transpose and VALU-bound DP kernels run beside this batch's latency-bound demux kernel.  Measured on
    # configs[1], two passes each: one stream 0.326-0.333 ms per step, two 0.315, three 0.307-0.309, four 0.44 (a quarter of
    # the slots per launch is too few); on the 8-primer panel nothing (its DP and compact demux kernels already fill the CUs).
    n_streams = a.streams if a.streams else (3 if a.config == "c2" else 1)
    if n_streams > 1:
        # none of them the legacy default stream: a launch there waits for, and holds up, the work of every other stream
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        stream = streams[0]
        torch.cuda.set_stream(stream)      # torch's own (tiny) operations on the counts go there too
    else:
        streams = [stream]
    db = DeviceBatches(lib, cp, sets, dev, streams)
    cp.set_streams(n_streams)
    d_counts = db.counts

    from specimux_amd.distributed import CountsReducer
    # RCCL communicator of the C ABI (its 128-byte id travels over torch.distributed)
    reducer = CountsReducer(world, rank, "torch" if rehearsal else "rccl")

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(a.warmup):
        db.step(i)
    if world > 1:   # warm-up of the exchange too: the first collective on a fresh communicator sets up its rings
        reducer.allreduce_(torch.zeros_like(d_counts), stream.cuda_stream)
    torch.cuda.synchronize()
    db.zero_counts()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        db.step(i)
    if n_streams > 1:
        torch.cuda.synchronize()                       # every stream's batches are done: the counts vector is complete
    reducer.allreduce_(d_counts, stream.cuda_stream)   # the one exchange of the path: counts, once per job
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    counts = d_counts.cpu().numpy().astype(np.uint64)
    total_reads = world * n * a.steps
    assert counts[_lib.CNT_TOTAL] == total_reads, (counts[:8], total_reads)
    assert counts[_lib.CNT_OVERFLOW] == 0 and db.max_extra() <= db.extra_cap
    n_rccl_ranks = world if (world > 1 and reducer.backend == "rccl") else (1 if world == 1 else 0)
    reducer.close()
    if world > 1:
        dist.destroy_process_group()
    if rank != 0:
        return

    # ---- after the timed region (rank 0): the same launches on ONE batch relaunched in place (Infinity-Cache assisted),
    # per-kernel device times (HIP events on the launch stream)
    one_stream_ms = db.event_timed(a.steps, rotate=True, n_streams=1)
    single_ms = db.event_timed(a.steps, rotate=False, n_streams=1) if len(sets) > 1 else None
    k_t, k_d, k_b = db.kernel_times()
    bsv = 1 if parameters.max_dist_index < 4 and (a.trim or "") != "tails" else (3 if parameters.max_dist_index < 4 else 2)
    # (template arguments after the scan variant -- compact / redo mode, default-flags specialisation -- are chosen by the
    # launch glue from the panel; panels with compact tiles launch the kernel twice per step, both are in this time)
    demux_name = f"smx::demux_kernel<unsigned int, 256, {bsv}, ...>"
    kernels_ms = k_t + k_d + k_b

    bytes_per_read = (2 * 160 + 4 + 32) if a.config == "c5" else BYTES_PER_READ
    step_ms_value = elapsed / a.steps * 1e3 / 1.0                       # per step of THIS rank: the scope of `value`
    achieved = bytes_per_read * n / (step_ms_value * 1e-3) / 1e9        # algorithmic bytes over the measured time per step
    dom_achieved = bytes_per_read * n / (k_b * 1e-3) / 1e9              # the demux kernel alone, launched by itself
    traffic = None
    valu = None
    prof_name = "r03_pmc_summary.json" if os.path.exists(os.path.join(REPO, "profiles", "r03_pmc_summary.json")) else "r02_pmc_summary.json"
    prof = committed_profile(prof_name) if a.config == "c2" else None
    if prof and prof.get("reads_per_launch") == n:
        traffic = prof.get("hbm_bytes_per_step")
        insts = prof.get("valu_instr_per_step")
        if insts:
            ach = insts / (step_ms_value * 1e-3)
            valu = {"bound": "valu-issue", "achieved": ach, "unit": "wave64 VALU instr/s", "instr_per_step": insts,
                    "instr_by_kernel": prof.get("valu_instr_by_kernel"),
                    "peak": VALU_PEAK_GUIDE, "frac": ach / VALU_PEAK_GUIDE,
                    "peak_note": "MI355X_MICROARCH.md: one wave64 VALU instruction per SIMD-32 every 2 cycles, 1024 SIMDs, 2.4 GHz",
                    "measured_mix_peak": VALU_PEAK_MEASURED_MIX, "frac_of_measured_mix": ach / VALU_PEAK_MEASURED_MIX,
                    "measured_mix_note": "tools/ubench/issue_rate.hip: 2.55-2.97 cycles per v_and/v_bitop3 at 4 waves per SIMD",
                    "source": f"instruction counts replayed from profiles/{prof_name} (rocprofv3 SQ_INSTS_VALU of this "
                              "workload), not measured in this run; time = this run's measured time per step"}
    matched = counts[_lib.CNT_MATCHED] / total_reads
    out = {
        "metric": "reads/sec demultiplexed, 768-specimen ITS panel on 765k ONT-style reads",
        "value": total_reads / elapsed, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": ("configs[1]: 768-specimen ONT037-style ITS panel (32x24 13-nt barcodes, ITS1F/ITS4, "
                                "k_idx=3, k_p=7/6), 765k reads per GPU per step, search_len 80, prefilter+preorient on, "
                                "trim=barcodes, dereplicate=best") if a.config == "c2" else
                               ("configs[2]-style: 3072 specimens over 4 pools (ITS/RPB2/LSU/TEF1, ITS4 shared), default flags"
                                if a.config == "c3" else "configs[4]-style: the 3072-specimen panel, 15 % error reads, -l 160"),
                   "reads_per_gpu_per_step": n, "seed": SEED, "parallelism": f"read-sharded x{world}",
                   "rccl_ranks": n_rccl_ranks, "matched_fraction": float(matched),
                   "resident_batches": len(sets), "streams": n_streams,
                   "scope": f"kernel-resident: end windows already in HBM, records left in HBM; the timed loop rotates over "
                            f"{len(sets)} distinct resident batches ({len(sets) * n * cp.window_stride / 1e6:.0f} MB of windows) and hands "
                            f"the steps round robin to {n_streams} HIP stream(s)"},
        "step_kernels": step_kernel_list(k_t, k_d, k_b, demux_name),
        "step_kernels_note": "each kernel launched by itself at full size on one stream (HIP events on that stream)",
        "one_stream": {"value": n / (one_stream_ms * 1e-3), "unit": "reads/s", "ms_per_step": one_stream_ms,
                       "note": "the same rotation with one batch in flight at a time"},
        "single_buffer": ({"value": n / (single_ms * 1e-3), "unit": "reads/s", "ms_per_step": single_ms,
                           "note": "one stream, one resident batch relaunched in place (inputs + intermediates fit the 256 MiB "
                                   "Infinity Cache): NOT the headline"} if single_ms else None),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_source": (f"profiles/{prof_name} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE over the three kernels "
                                        "of a step; replayed, not measured in this run)") if traffic else None,
                     "kernels": "all three kernels of a step, batches overlapped as in the timed loop (the scope of `value`)",
                     "ms_per_step_this_rank": step_ms_value, "kernels_ms_one_at_a_time": kernels_ms,
                     "algorithmic_bytes_per_read": bytes_per_read,
                     "dominant_kernel": {"kernel": demux_name, "kernel_ms_avg": k_b, "achieved": dom_achieved,
                                         "frac": dom_achieved / HBM_PEAK_GBPS},
                     "note": "integer-VALU / latency bound, not HBM bound: see DESIGN.md sections 4-5"},
        "cpu_baseline": cpu,
    }
    if valu:
        out["valu_roofline"] = valu
    cp.set_streams(1)
    if solo and not a.no_extras and a.config == "c2":
        del db
        torch.cuda.empty_cache()
        out["other_configs"] = {c: side_config(lib, c, dev, stream, tmp) for c in ("c3", "c5")}
        out["pcie_inclusive"] = pcie_inclusive(cp, rs)
        out["end_to_end"] = end_to_end(pan, pf, sf, a.e2e_reads, e2e_dir)
    if cpu:
        out["gpu_over_cpu"] = out["value"] / cpu["value"]   # kernel-resident rate over the oracle's in-memory rate, same reads
        out["gpu_over_cpu_scopes"] = {"kernel_resident_over_cpu_in_memory": out["gpu_over_cpu"],
                                      "note": "like-for-like scopes only; both CPU legs are the Python-loop oracle, not a tuned CPU code"}
        if out.get("end_to_end"):
            out["gpu_over_cpu_scopes"]["end_to_end_over_cpu_end_to_end"] = out["end_to_end"]["value"] / cpu["end_to_end"]["value"]
    shutil.rmtree(e2e_dir, ignore_errors=True)
    shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
