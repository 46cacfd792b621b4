#!/usr/bin/env python3
"""bench.py -- reads/sec demultiplexed, 768-specimen ITS panel, 765k ONT-style reads (BASELINE.json).

One "step" = one pass of the hot path (libsmx: primer prescan kernels + demux kernel = primer scan, barcode scan,
scorer) over the whole 765 000-read batch of configs[1], end windows already resident in HBM.  `--gpus N` runs one
process per GPU (launched by torch.distributed.run); every rank owns its own 765k-read shard (seed + rank, weak
scaling, no data-path collective) and the per-specimen counts are summed once at the end with RCCL
(smx_counts_allreduce, C ABI).  Rank 0 prints ONE JSON line.

`value` is the kernel-resident rate the contract asks for.  The same line carries the wider scopes, so that no
ratio has to mix scopes (N = 1 only, after the timed region):
  step_kernels    device time of every kernel of a step (HIP events on the launch stream, smx_debug_kernel_times)
  roofline        dominant kernel (demux_kernel) vs the 8 TB/s HBM3E peak, ALGORITHMIC bytes = 196 B/read
                  (2*search_len window bytes + 4 length + 32 result record); `path` = the same bytes
                  over all kernels of the step.  `traffic` = HBM bytes per launch from the committed rocprofv3 PMC
                  passes of this workload (profiles/, replayed: not measured in this run), null if absent.
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
def pcie_inclusive(cp, windows, lens, rounds=6):
    """Host windows in, records out: three pinned lanes in flight (H2D of one batch overlaps the kernels of the previous
    and the D2H of the one before).  The packer's memcpy into the staging is not part of it (staging filled once)."""
    from specimux_amd.native_io import Lane
    n = len(lens)
    lanes = [Lane(cp, n) for _ in range(3)]
    counts = np.zeros(cp.counts_len, dtype=np.uint64)
    try:
        for ln in lanes:
            ln.windows[:n] = windows
            ln.lens[:n] = lens
        for ln in lanes:               # warm
            ln.submit(n)
        for ln in lanes:
            ln.wait(counts)
        t0 = time.perf_counter()
        inflight = []
        for i in range(rounds):
            ln = lanes[i % 3]
            if len(inflight) == 3:
                inflight.pop(0).wait(counts)
            ln.submit(n)
            inflight.append(ln)
        for ln in inflight:
            ln.wait(counts)
        dt = time.perf_counter() - t0
    finally:
        for ln in lanes:
            ln.close()
    h2d = n * (cp.window_stride + 4)
    d2h = n * 32
    return {"value": rounds * n / dt, "unit": "reads/s", "ms_per_batch": dt / rounds * 1e3, "lanes": 3,
            "h2d_bytes_per_batch": h2d, "d2h_bytes_per_batch": d2h, "h2d_gbps": rounds * h2d / dt / 1e9,
            "note": "pinned staging -> hipMemcpyAsync H2D -> kernels -> D2H of the 32-byte records, own stream per lane"}


def end_to_end(pan, pf, sf, n_reads, e2e_dir):
    """FASTQ in /dev/shm -> output tree through the product pipeline (what `specimux -F` runs)."""
    from specimux_amd import cli, synth
    rs = synth.make_reads(pan, n_reads, SEED, windows_only=False)
    fq = os.path.join(e2e_dir, "reads.fastq")
    rs.write_fastq(fq)
    size = os.path.getsize(fq)
    best = None
    for rep in range(3):   # first run warms the page cache / allocations; the best of the next two is reported
        out = os.path.join(e2e_dir, f"out{rep}")
        stats = {}
        os.environ["SMX_PIPELINE_STATS_JSON"] = os.path.join(e2e_dir, "stats.json")
        t0 = time.perf_counter()
        cli.main(["specimux", pf, sf, fq, "-F", "-O", out])
        dt = time.perf_counter() - t0
        try:
            with open(os.environ["SMX_PIPELINE_STATS_JSON"]) as fh:
                stats = json.load(fh)
        except OSError:
            pass
        nbytes = sum(os.path.getsize(os.path.join(d, f)) for d, _s, fs in os.walk(out) for f in fs)
        shutil.rmtree(out, ignore_errors=True)
        if rep and (best is None or dt < best["seconds"]):
            best = {"value": n_reads / dt, "unit": "reads/s", "seconds": dt, "reads": n_reads, "input_gbps": size / dt / 1e9,
                    "output_bytes": nbytes, "stage_seconds": {k: round(v, 4) for k, v in stats.items()},
                    "main_thread_not_waiting_for_gpu_pct": (100.0 * (1.0 - stats["gpu_wait"] / stats["wall"])
                                                            if stats.get("wall") else None),
                    "note": "CLI entry point incl. panel compilation and log/primers side files; reader, lanes and "
                            "writer overlapped (specimux_amd/pipeline.py)"}
    os.environ.pop("SMX_PIPELINE_STATS_JSON", None)
    return best


def committed_profile(name):
    path = os.path.join(REPO, "profiles", name)
    if os.path.exists(path):
        with open(path) as fh:
            return json.load(fh)
    return None


# ------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=N_READS, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true", help=argparse.SUPPRESS)   # skip pcie_inclusive / end_to_end (profiling runs)
    ap.add_argument("--e2e-reads", type=int, default=200_000, help=argparse.SUPPRESS)
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
    e2e_dir = tempfile.mkdtemp(prefix="smx_bench_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    pf, sf = pan.write(tmp)
    from specimux_amd.distributed import shard_seed
    gen_kw = dict(search_len=160, error_rate=0.15) if a.config == "c5" else {}
    rs = synth.make_reads(pan, a.reads, shard_seed(SEED, rank), **gen_kw)     # this rank's shard (weak scaling)

    solo = rank == 0 and a.gpus == 1
    cpu = None
    if solo and not a.no_cpu_baseline and a.config == "c2":
        cpu = cpu_baseline(pf, sf, rs, e2e_dir)

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
    h_windows = rs.windows(cp.window_stride)
    d_windows = torch.from_numpy(h_windows).to(dev)
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
    step_ms = [s0.elapsed_time(s1) for s0, s1 in ev]

    counts = d_counts.cpu().numpy().astype(np.uint64)
    total_reads = world * n * a.steps
    assert counts[_lib.CNT_TOTAL] == total_reads, (counts[:8], total_reads)
    assert counts[_lib.CNT_OVERFLOW] == 0 and int(d_nextra[0].item()) <= extra_cap
    n_rccl_ranks = world if (world > 1 and reducer.backend == "rccl") else (1 if world == 1 else 0)
    reducer.close()
    if world > 1:
        dist.destroy_process_group()
    if rank != 0:
        return

    # ---- per-kernel device times (HIP events on the launch stream), outside the timed region
    kms = np.zeros((5, 3), dtype=np.float32)
    _lib.check(lib.smx_debug_kernel_times(cp.handle, 1, None))
    for i in range(len(kms)):
        step()
        _lib.check(lib.smx_debug_kernel_times(cp.handle, 1, kms[i].ctypes.data_as(C.POINTER(C.c_float))))
    _lib.check(lib.smx_debug_kernel_times(cp.handle, 0, None))
    k_t, k_d, k_b = (float(x) for x in kms.mean(axis=0))
    bsv = 1 if parameters.max_dist_index < 4 and (a.trim or "") != "tails" else (3 if parameters.max_dist_index < 4 else 2)
    # (template arguments after the scan variant -- compact / redo mode, default-flags specialisation -- are chosen by the
    # launch glue from the panel; panels with compact tiles launch the kernel twice per step, both are in this time)
    demux_name = f"smx::demux_kernel<unsigned int, 256, {bsv}, ...>"
    step_kernels = [{"kernel": "smx::prescan_transpose_kernel", "ms": k_t}, {"kernel": "smx::prescan_dp_kernel", "ms": k_d},
                    {"kernel": demux_name, "ms": k_b}]
    kernels_ms = k_t + k_d + k_b

    avg_ms = float(np.mean(step_ms))
    bytes_per_read = (2 * 160 + 4 + 32) if a.config == "c5" else BYTES_PER_READ
    achieved = bytes_per_read * n / (k_b * 1e-3) / 1e9
    path_achieved = bytes_per_read * n / (kernels_ms * 1e-3) / 1e9
    traffic = None
    valu = None
    prof = committed_profile("r02_pmc_summary.json") if a.config == "c2" else None
    if prof and prof.get("reads_per_launch") == n:
        traffic = prof.get("hbm_bytes_per_step")
        insts = prof.get("valu_instr_per_step")
        if insts:
            ach = insts / (kernels_ms * 1e-3)
            valu = {"bound": "valu-issue", "achieved": ach, "unit": "wave64 VALU instr/s", "instr_per_step": insts,
                    "instr_by_kernel": prof.get("valu_instr_by_kernel"),
                    "peak": VALU_PEAK_GUIDE, "frac": ach / VALU_PEAK_GUIDE,
                    "peak_note": "MI355X_MICROARCH.md: one wave64 VALU instruction per SIMD-32 every 2 cycles, 1024 SIMDs, 2.4 GHz",
                    "measured_mix_peak": VALU_PEAK_MEASURED_MIX, "frac_of_measured_mix": ach / VALU_PEAK_MEASURED_MIX,
                    "measured_mix_note": "tools/ubench/issue_rate.hip: 2.55-2.97 cycles per v_and/v_bitop3 at 4 waves per SIMD",
                    "source": "instruction counts replayed from profiles/r02_pmc_summary.json (rocprofv3 SQ_INSTS_VALU of this "
                              "workload), not measured in this run; time = this run's HIP events"}
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
                   "scope": "kernel-resident: end windows already in HBM, records left in HBM"},
        "step_kernels": step_kernels, "step_ms_events": {"avg": avg_ms, "min": float(np.min(step_ms))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_source": ("profiles/r02_pmc_summary.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE over the three kernels "
                                        "of a step; replayed, not measured in this run)") if traffic else None,
                     "kernel": demux_name, "kernel_ms_avg": k_b, "algorithmic_bytes_per_read": bytes_per_read,
                     "path": {"achieved": path_achieved, "frac": path_achieved / HBM_PEAK_GBPS, "kernels_ms": kernels_ms,
                              "note": "same algorithmic bytes over the three kernels of a step"},
                     "note": "integer-VALU / latency bound, not HBM bound: see DESIGN.md sections 4-5"},
        "cpu_baseline": cpu,
    }
    if valu:
        out["valu_roofline"] = valu
    if solo and not a.no_extras and a.config == "c2":
        out["pcie_inclusive"] = pcie_inclusive(cp, h_windows, rs.lens)
        out["end_to_end"] = end_to_end(pan, pf, sf, a.e2e_reads, e2e_dir)
    if cpu:
        out["gpu_over_cpu"] = {"kernel_resident_over_cpu_in_memory": out["value"] / cpu["value"],
                               "note": "like-for-like scopes only; both CPU legs are the Python-loop oracle, not a tuned CPU code"}
        if out.get("end_to_end"):
            out["gpu_over_cpu"]["end_to_end_over_cpu_end_to_end"] = out["end_to_end"]["value"] / cpu["end_to_end"]["value"]
    shutil.rmtree(e2e_dir, ignore_errors=True)
    shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
