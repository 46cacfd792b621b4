#!/usr/bin/env python3
"""bench.py -- reads/sec demultiplexed, 768-specimen ITS panel, 765k ONT-style reads (BASELINE.json).

One "step" = one pass of the hot path (libsmx demux kernel: primer scan + barcode scan + scorer) over the
whole 765 000-read batch of configs[1], end windows already resident in HBM.  `--gpus N` runs one process
per GPU (launched by torch.distributed.run); every rank owns its own 765k-read shard (seed + rank, weak
scaling, no data-path collective) and the per-specimen counts are summed once at the end with RCCL
(smx_counts_allreduce, C ABI).  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel (demux_kernel) vs the 8 TB/s HBM3E peak, ALGORITHMIC bytes = 196 B/read
                (2*search_len window bytes + 4 length + 32 result record), duration from HIP
                events on the launch stream.  `traffic` = HBM bytes per launch from rocprofv3 PMC passes
                (profiles/traffic_r01.json, written by tools/collect_traffic.py), null if absent.
  alu           the bound that actually applies (integer VALU): Myers column-steps per launch, measured
                by the oracle-free closed form in DESIGN.md, vs 256 CU x 128 lanes x 2.4 GHz.
  cpu_baseline  the oracle (reference-shaped Python loop + C DP aligner, one alignment per call) on a bounded
                sample of the same reads, multiprocessing over the host cores; N=1, rank 0 only.
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
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
ALU_PEAK_LANE_OPS = 256 * 128 * 2.4e9       # 256 CU x 4 SIMD-32 x 2.4 GHz int32 lane-ops/s


# ------------------------------------------------------------------ cpu baseline (oracle, test infrastructure)
_cpu_state = {}


def _cpu_init(pf, sf):
    from oracle import specimux_oracle as O
    panel = O.load_panel(pf, sf)
    _cpu_state["O"] = O
    _cpu_state["panel"] = panel
    _cpu_state["par"] = O.setup_params(panel)


def _cpu_work(reads):
    O = _cpu_state["O"]
    ops, total, matched = O.process_sequences(reads, _cpu_state["par"], _cpu_state["panel"])
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


def cpu_baseline(pf, sf, rs, budget_reads_per_core=12000):
    """Reference-shaped CPU path (the oracle) on a bounded sample, all host cores, 1000-read batches like the
    reference's worker pool (orchestration.py:165)."""
    from specimux_amd.synth import rebuild_read
    cores = host_cores()
    n = min(len(rs.lens), budget_reads_per_core * cores)
    reads = [(f"r{i}", rebuild_read(rs.head[i], rs.tail[i], int(rs.lens[i]), SEARCH_LEN), None) for i in range(n)]
    reads = [(i, s, "I" * len(s)) for i, s, _ in reads]
    batches = [reads[i:i + 1000] for i in range(0, n, 1000)]
    ctx = mp.get_context("fork")    # forked BEFORE this process touches the GPU
    with ctx.Pool(cores, initializer=_cpu_init, initargs=(pf, sf)) as pool:
        pool.map(_cpu_work, batches[:cores])         # warm: imports, panel, C library
        t0 = time.perf_counter()
        res = pool.map(_cpu_work, batches)
        dt = time.perf_counter() - t0
    total = sum(t for t, _ in res)
    return {"value": total / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {n} of the {len(rs.lens)} reads, {dt:.1f} s wall, oracle/specimux_oracle.py "
                      f"(reference loop order, one C DP alignment per call), multiprocessing fork x{cores}",
            "matched_fraction": sum(m for _, m in res) / max(total, 1)}


# ------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=N_READS, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--trim", default=None, help=argparse.SUPPRESS)   # e.g. tails: not the headline flags, DESIGN.md side numbers
    ap.add_argument("--cli-args", default="", help=argparse.SUPPRESS)  # extra specimux flags for side measurements, e.g. "-e 4"
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c5"], help=argparse.SUPPRESS)   # c3: 3072 specimens / 4 pools; c5: + 160-nt windows, 15 % errors
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit(f"--gpus {a.gpus} needs one process per GPU: launch with "
                     f"python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...")
        a.gpus = world

    from specimux_amd import synth
    pan = synth.panel_c2(SEED) if a.config == "c2" else synth.panel_c3(SEED)
    tmp = tempfile.mkdtemp(prefix="smx_bench_")
    pf, sf = pan.write(tmp)
    from specimux_amd.distributed import shard_seed
    gen_kw = dict(search_len=160, error_rate=0.15) if a.config == "c5" else {}
    rs = synth.make_reads(pan, a.reads, shard_seed(SEED, rank), **gen_kw)     # this rank's shard (weak scaling)

    cpu = None
    if rank == 0 and a.gpus == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(pf, sf, rs)

    import torch
    import torch.distributed as dist
    from specimux_amd import _lib
    import specimux_amd as sa
    from specimux_amd.bloom_filter import BloomPrefilter, barcodes_for_bloom_prefilter
    from specimux_amd.demultiplex import compiled_panel
    from specimux_amd.cli import parse_args

    if os.environ.get("SMX_BENCH_SAME_DEVICE"):   # rehearsal on a 1-GPU box only: every rank on GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rehearsal = bool(os.environ.get("SMX_BENCH_SAME_DEVICE"))   # gloo + shared GPU 0: RCCL refuses two ranks per device
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    lib = _lib.load()
    args = parse_args(["specimux", pf, sf, "reads.fastq"] + (["-l", "160"] if a.config == "c5" else []) +
                      (["--trim", a.trim] if a.trim else []) + a.cli_args.split())   # default flags
    reg = sa.read_primers_file(pf)
    specimens = sa.read_specimen_file(sf, reg)
    specimens.validate()
    parameters = sa.setup_match_parameters(args, specimens)
    prefilter = BloomPrefilter(barcodes_for_bloom_prefilter(specimens), parameters.max_dist_index)
    cp = compiled_panel(specimens, parameters, args, prefilter)
    assert (parameters.max_dist_index == 3 or a.cli_args) and len(cp.specimen_ids) == (768 if a.config == "c2" else 3072)

    n = a.reads
    d_windows = torch.from_numpy(rs.windows(cp.window_stride)).to(dev)
    d_lens = torch.from_numpy(rs.lens).to(dev)
    d_ops = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    extra_cap = n
    d_extra = torch.empty(extra_cap * 32, dtype=torch.uint8, device=dev)
    d_nextra = torch.zeros(4, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(cp.counts_len, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()

    from specimux_amd.distributed import CountsReducer
    # RCCL communicator of the C ABI (its 128-byte id travels over torch.distributed)
    reducer = CountsReducer(world, rank, "torch" if rehearsal else "rccl")

    def step():
        _lib.check(lib.smx_batch_run_device(cp.handle, C.c_void_p(stream.cuda_stream), C.c_void_p(d_windows.data_ptr()),
                                            C.c_void_p(d_lens.data_ptr()), n, C.c_void_p(d_ops.data_ptr()),
                                            C.c_void_p(d_extra.data_ptr()), extra_cap, C.c_void_p(d_nextra.data_ptr()),
                                            C.c_void_p(d_counts.data_ptr()), None, None))

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    if world > 1:   # warm-up of the exchange too: the first collective on a fresh communicator sets up its rings
        reducer.allreduce_(torch.zeros_like(d_counts), stream.cuda_stream)
    torch.cuda.synchronize()
    d_counts.zero_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s0, s1 in ev:
        s0.record(stream)
        step()
        s1.record(stream)
    reducer.allreduce_(d_counts, stream.cuda_stream)   # the one exchange of the path: counts, once per job
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kernel_ms = [s0.elapsed_time(s1) for s0, s1 in ev]

    counts = d_counts.cpu().numpy().astype(np.uint64)
    total_reads = world * n * a.steps
    assert counts[_lib.CNT_TOTAL] == total_reads, (counts[:8], total_reads)
    assert counts[_lib.CNT_OVERFLOW] == 0 and int(d_nextra[0].item()) <= extra_cap
    reducer.close()
    if world > 1:
        dist.destroy_process_group()
    if rank != 0:
        return

    avg_ms = float(np.mean(kernel_ms))
    bytes_per_read = (2 * 160 + 4 + 32) if a.config == "c5" else BYTES_PER_READ
    achieved = bytes_per_read * n / (avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(REPO, "profiles", "traffic_r01.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            tj = json.load(fh)
        if tj.get("reads_per_launch") == n:
            traffic = tj.get("hbm_bytes_per_launch")
    # secondary, informational: the integer-VALU issue roofline (the bound that binds).  The wave
    # instructions per launch come from the committed rocprofv3 PMC summary (SQ_INSTS_VALU) of this same workload.
    valu = None
    ppath = os.path.join(REPO, "profiles", "r01_l_pmc_summary.json")
    if a.config == "c2" and os.path.exists(ppath):
        with open(ppath) as fh:
            pj = json.load(fh)
        insts = pj.get("counters_per_launch", {}).get("SQ_INSTS_VALU")
        if insts and pj.get("hbm", {}).get("reads_per_launch") == n:
            peak = 256 * 4 * 2.4e9 / 4.0          # wave64 instructions/s: 1024 SIMDs, one per 4 cycles, 2.4 GHz
            ach = insts / (avg_ms * 1e-3)
            valu = {"bound": "valu-issue", "achieved": ach, "peak": peak, "unit": "wave64 VALU instr/s", "frac": ach / peak,
                    "instr_per_launch": insts, "source": "profiles/r01_l_pmc_summary.json (SQ_INSTS_VALU)"}
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
                   "matched_fraction": float(matched)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel": "smx::demux_kernel<unsigned int, 256, 1>",
                     "kernel_ms_avg": avg_ms, "kernel_ms_min": float(np.min(kernel_ms)),
                     "algorithmic_bytes_per_read": bytes_per_read,
                     "note": "integer-VALU bound, not HBM bound: see DESIGN.md section 5"},
        "cpu_baseline": cpu,
    }
    if valu:
        out["valu_roofline"] = valu
    if cpu:
        out["gpu_over_cpu"] = out["value"] / cpu["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
