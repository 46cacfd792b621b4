"""Run-level orchestration of the drop-in interface (reference: src/specimux/orchestration.py).

setup_match_parameters (:548-641) derives the thresholds exactly as the reference; specimux_mp /
specimux (:153, :458) stream the read file in large batches through process-local GPU batches instead
of a multiprocessing.Pool of 1000-read batches.  Same log lines (`Processed N sequences, match rate: X%`,
`Elapsed time`), same output tree."""
import itertools
from datetime import datetime
import logging
import math
import os
import timeit

from .bloom_filter import BloomPrefilter, barcodes_for_bloom_prefilter
from .constants import Primer
from .io_utils import (OutputManager, cleanup_empty_directories, cleanup_locks, open_sequence_file,
                       output_write_operation, read_primers_file, read_specimen_file)
from .models import MatchParameters, reverse_complement

GPU_BATCH_READS = 65536   # reads per kernel launch on the streaming path


def edit_distance(a: str, b: str) -> int:
    """Plain global (NW) edit distance, exact character equality (edlib.align(a, b, task='distance'),
    orchestration.py:552) -- bit-parallel on Python integers; init-time only."""
    if not a or not b:
        return max(len(a), len(b))
    peq = {}
    for i, ch in enumerate(a):
        peq[ch] = peq.get(ch, 0) | (1 << i)
    m = len(a)
    mask, top = (1 << m) - 1, 1 << (m - 1)
    pv, mv, score = mask, 0, m
    for ch in b:
        eq = peq.get(ch, 0)
        xv = eq | mv
        xh = (((eq & pv) + pv) ^ pv) | eq
        ph = mv | (~(xh | pv) & mask)
        mh = pv & xh
        if ph & top:
            score += 1
        elif mh & top:
            score -= 1
        ph = ((ph << 1) | 1) & mask
        mh = (mh << 1) & mask
        pv = mh | (~(xv | ph) & mask)
        mv = ph & xv
    return score


def _native_min_pairwise(seqs) -> int:
    """min over all pairs of edit_distance(x, y), in libsmx (smx_min_pairwise_distance): a 224-barcode panel has 25,000
    pairs, which the Python loop above takes a second over."""
    import ctypes as C
    import numpy as np
    from . import _lib
    enc = [s.encode("utf-8") for s in seqs]
    if any(len(e) != len(s) for e, s in zip(enc, seqs)):   # non-ASCII text: compare code points, not bytes
        return min(edit_distance(x, y) for x, y in itertools.combinations(seqs, 2))
    off = np.zeros(len(enc) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(e) for e in enc])
    out = C.c_int32()
    _lib.check(_lib.load().smx_min_pairwise_distance(b"".join(enc), _lib.ptr(off), len(enc), C.byref(out)))
    return int(out.value)


def _bp_adjusted_length(primer: str) -> float:
    weight = {**dict.fromkeys("ACGT", 3), **dict.fromkeys("KMRSWY", 2), **dict.fromkeys("BDHV", 1)}
    return sum(weight.get(ch, 0) for ch in primer) / 3.0


def _min_pairwise(seqs, label, args):
    if len(seqs) <= 1:
        return None
    best = _native_min_pairwise(seqs)
    if getattr(args, "diagnostics", None):
        logging.info(f"Minimum edit distance is {best} for {label}")
    return best


def setup_match_parameters(args, specimens) -> MatchParameters:
    fwd_bcs, rev_bcs = [], []
    for primer in specimens.get_primers(Primer.FWD):
        fwd_bcs.extend(b for b in primer.barcodes if b not in fwd_bcs)
    for primer in specimens.get_primers(Primer.REV):
        rev_bcs.extend(b for b in primer.barcodes if b not in rev_bcs)
    _min_pairwise(fwd_bcs, "Forward Barcodes", args)
    _min_pairwise(rev_bcs, "Reverse Barcodes", args)
    min_bc = _min_pairwise(fwd_bcs + [reverse_complement(b) for b in rev_bcs],
                           "Forward Barcodes + Reverse Complement of Reverse Barcodes", args)
    all_primers = specimens.get_primers(Primer.FWD) + specimens.get_primers(Primer.REV)
    _min_pairwise(sorted({s for p in all_primers for s in (p.primer, p.primer_rc)}),
                  "All Primers and Reverse Complements", args)

    if args.index_edit_distance != -1:
        max_dist_index = args.index_edit_distance
    else:
        if min_bc is None:   # the reference dies with a TypeError here (None / 2.0): SURVEY Q14
            raise ValueError("cannot derive the barcode edit distance from fewer than two barcodes; use -e")
        max_dist_index = math.ceil(min_bc / 2.0)
    logging.info(f"Using Edit Distance Thresholds {max_dist_index} for barcode indexes")

    thresholds = {}
    for primer in all_primers:
        thresholds[primer.primer] = (args.primer_edit_distance if args.primer_edit_distance != -1
                                     else int(_bp_adjusted_length(primer.primer) / 3))
    for seq, k in thresholds.items():
        logging.info(f"Using Edit Distance Threshold {k} for primer {seq}")
    logging.info(f"Using dereplication strategy: {args.dereplicate}")
    preorient = not args.disable_preorient
    if not preorient:
        logging.info("Sequence pre-orientation disabled, may run slower")
    parameters = MatchParameters(thresholds, max_dist_index, args.search_len, preorient)
    if not args.disable_prefilter:
        if specimens.b_length() > 13:
            logging.warning("Barcode prefilter not tested for barcodes longer than 13 nt.  You may need to use --disable-prefilter")
        if max_dist_index > 3:
            logging.warning("Barcode prefilter not tested for edit distance greater than 3.  You may need to use --disable-prefilter")
        logging.info("Using Bloom Filter optimization for barcode matching")
    else:
        logging.info("Barcode prefiltering disabled, may run slower")
    return parameters


def _write_primer_files(directory, fwd_primers, rev_primers):
    rows = [(p, "forward") for p in fwd_primers] + [(p, "reverse") for p in rev_primers]
    with open(os.path.join(directory, "primers.fasta"), "w") as fa, open(os.path.join(directory, "primers.txt"), "w") as tx:
        for p, pos in rows:
            fa.write(f">{p.name} position={pos} pool={','.join(p.pools)}\n{p.primer}\n")
            tx.write(f">{p.name}\n{p.primer}\n")


def write_primers_fasta(output_dir, fwd_primer, rev_primer):
    _write_primer_files(output_dir, [fwd_primer], [rev_primer])


def write_all_primers_fasta(output_dir, fwd_primers, rev_primers):
    _write_primer_files(output_dir, fwd_primers, rev_primers)


def create_output_files(args, specimens):
    """Directory skeleton + primers.fasta / primers.txt (orchestration.py:314-372)."""
    if not args.output_to_files:
        return
    out, reg = args.output_dir, specimens._primer_registry
    kinds = ("full", "partial", "unknown")
    for kind in kinds:
        os.makedirs(os.path.join(out, kind), exist_ok=True)
    for kind in ("partial", "unknown"):
        os.makedirs(os.path.join(out, kind, "unknown", "unknown-unknown"), exist_ok=True)
    for pool in reg.get_pools():
        fwd, rev = reg.get_pool_primers(pool, Primer.FWD), reg.get_pool_primers(pool, Primer.REV)
        for kind in kinds:
            os.makedirs(os.path.join(out, kind, pool), exist_ok=True)
        _write_primer_files(os.path.join(out, "full", pool), fwd, rev)
        for f in fwd:
            for r in rev:
                for kind in kinds:
                    os.makedirs(os.path.join(out, kind, pool, f"{f.name}-{r.name}"), exist_ok=True)
                _write_primer_files(os.path.join(out, "full", pool, f"{f.name}-{r.name}"), [f], [r])
            for kind in ("partial", "unknown"):
                os.makedirs(os.path.join(out, kind, pool, f"{f.name}-unknown"), exist_ok=True)
        for r in rev:
            for kind in ("partial", "unknown"):
                os.makedirs(os.path.join(out, kind, pool, f"unknown-{r.name}"), exist_ok=True)


def subsample_top_quality(output_dir: str, top_n: int):
    """--sample-topq N: for every FASTQ under full/, write subsample/<same path> with the N records of highest mean
    Phred quality (stable order on ties) and copy the primer side files (reference: orchestration.py:374-445)."""
    import shutil
    from .io_utils import parse_fastq
    full_dir = os.path.join(output_dir, "full")
    if not os.path.exists(full_dir):
        logging.warning(f"Full directory not found: {full_dir}")
        return
    logging.info(f"Creating subsamples with top {top_n} sequences by quality...")
    done = failed = 0
    for root, _dirs, files in os.walk(full_dir):
        for fname in files:
            if not fname.endswith(".fastq"):
                continue
            dest_dir = os.path.join(output_dir, "subsample", os.path.relpath(root, full_dir))
            os.makedirs(dest_dir, exist_ok=True)
            for side in ("primers.fasta", "primers.txt"):
                if os.path.exists(os.path.join(root, side)):
                    shutil.copy2(os.path.join(root, side), os.path.join(dest_dir, side))
            try:
                with open(os.path.join(root, fname)) as fh:
                    records = list(parse_fastq(fh))
                if not records:
                    continue

                def mean_q(rec):
                    q = rec.quality_string
                    return (sum(map(ord, q)) / len(q) - 33) if q else 0
                records.sort(key=mean_q, reverse=True)
                with open(os.path.join(dest_dir, fname), "w") as out:
                    for rec in records[:top_n]:
                        out.write(f"@{rec.description}\n{rec.seq}\n+\n{rec.quality_string}\n")
                done += 1
            except Exception as e:
                logging.warning(f"Failed to subsample {os.path.join(root, fname)}: {e}")
                failed += 1
    logging.info(f"Subsampling complete: {done} files processed, {failed} failed")


def iter_batches(seq_records, batch_size: int, max_seqs: int, all_seqs: bool):
    done = 0
    while all_seqs or done < max_seqs:
        want = batch_size if all_seqs else min(batch_size, max_seqs - done)
        batch = list(itertools.islice(seq_records, want))
        if not batch:
            return
        yield batch
        done += len(batch)


def _load(args):
    registry = read_primers_file(args.primer_file)
    specimens = read_specimen_file(args.specimen_file, registry)
    specimens.validate()
    parameters = setup_match_parameters(args, specimens)
    prefilter = None
    if not args.disable_prefilter:
        prefilter = BloomPrefilter(barcodes_for_bloom_prefilter(specimens), parameters.max_dist_index)
    return specimens, parameters, prefilter


def _finish(total, matched, start):
    if total > 0:
        logging.info(f"Processed {total:,} sequences, match rate: {matched / total:.1%}")
    logging.info(f"Elapsed time: {timeit.default_timer() - start:.2f} seconds")


def _run_native(args):
    """`-F`: native reader -> GPU -> native writer, overlapped (specimux_amd/pipeline.py)."""
    from .demultiplex import compiled_panel   # needs libsmx.so: import late so --help works without it
    from .io_utils import detect_file_format
    from .pipeline import run_streaming
    specimens, parameters, prefilter = _load(args)
    args.isfastq = detect_file_format(args.sequence_file) == "fastq"
    from .distributed import env_rank
    rank, _local, world = env_rank()
    if rank == 0:
        create_output_files(args, specimens)
    start = timeit.default_timer()
    panel = compiled_panel(specimens, parameters, args, prefilter)
    if world > 1:
        # one process per GPU (python -m torch.distributed.run ... -m specimux_amd.cli ...): the input file is cut
        # into byte ranges at record boundaries, every rank appends its records to the one output tree (or writes its own
        # tree, merged afterwards: SMX_RANK_MERGE=1), one RCCL all-reduce sums the counts
        # (specimux_amd/distributed.py; reference: the worker pool of orchestration.py:181-207).
        # -n start,num (cli.py:54-68, orchestration.py:170-172) counts records from the start of the file: every rank then
        # reads the whole file, applies the window, and keeps batch i iff i mod world == rank (batch striding).
        from .distributed import run_sharded
        window = args.start_seq > 1 or args.num_seqs >= 0

        def shard_runner(seqfile, out_dir, byte_range, stride):
            t, m, c, _fq = run_streaming(seqfile, panel, out_dir, args.output_file_prefix, byte_range=byte_range, stride=stride,
                                         start_seq=args.start_seq if window else 1, num_seqs=args.num_seqs if window else -1)
            return t, m, c

        total, matched, _counts, _w = run_sharded(args.sequence_file, args.output_dir, args.output_file_prefix,
                                                  panel.counts_len, shard_runner, window=window)
        if rank == 0:
            logging.info(f"Demultiplexed on {world} GPUs (read-sharded by " + ("batch striding inside the -n window" if window
                         else "byte range") + ", counts summed by all-reduce)")
            _finish(total, matched, start)
            cleanup_empty_directories(args.output_dir)
            if getattr(args, "sample_topq", 0) > 0:
                subsample_top_quality(args.output_dir, args.sample_topq)
            cleanup_locks(args.output_dir)
        return
    total, matched, _counts, _fq = run_streaming(args.sequence_file, panel, args.output_dir, args.output_file_prefix,
                                                 start_seq=args.start_seq, num_seqs=args.num_seqs)
    _finish(total, matched, start)
    cleanup_empty_directories(args.output_dir)
    if getattr(args, "sample_topq", 0) > 0:
        subsample_top_quality(args.output_dir, args.sample_topq)
    cleanup_locks(args.output_dir)


def _run_records(args, to_files: bool):
    """Record-object path (stdout mode, --color): Python parser + process_sequences + OutputManager."""
    from .demultiplex import process_sequences
    specimens, parameters, prefilter = _load(args)
    if to_files:   # `-F -d`: trace events need record objects, but the input still comes through the native reader
        from .io_utils import native_sequence_records
        seq_records = native_sequence_records(args.sequence_file, args)
    else:
        seq_records = open_sequence_file(args.sequence_file, args)
    create_output_files(args, specimens)
    start = timeit.default_timer()
    if args.start_seq > 1:
        for _ in itertools.islice(seq_records, args.start_seq - 1):
            pass
    total = matched = 0
    manager = OutputManager(args.output_dir, args.output_file_prefix, args.isfastq) if to_files else None
    try:
        if manager:
            manager.__enter__()
        for batch in iter_batches(seq_records, GPU_BATCH_READS, args.num_seqs, args.num_seqs < 0):
            # one trace file per batch, like the reference's single-process loop (orchestration.py:501-521);
            # with -F the reference's files carry the worker's name: the GPU path has one "worker"
            trace_logger = None
            if getattr(args, "diagnostics", None):
                from .trace import TraceLogger
                trace_logger = TraceLogger(enabled=True, verbosity=args.diagnostics, output_dir=args.output_dir,
                                           worker_id="worker_1" if to_files else "main",
                                           start_timestamp=datetime.now().strftime("%Y%m%d_%H%M%S"))
            try:
                ops, n, m = process_sequences(batch, parameters, specimens, args, prefilter, trace_logger, total)
                for op in ops:
                    output_write_operation(op, manager, args, trace_logger)
            finally:
                if trace_logger:
                    trace_logger.close()
            total += n
            matched += m
    finally:
        if manager:
            manager.close()
    _finish(total, matched, start)
    if to_files:
        cleanup_empty_directories(args.output_dir)
        cleanup_locks(args.output_dir)


def _only_rank0_runs(what: str) -> bool:
    """The record path (trace logging, stdout mode) is one stream of records through one process: under a
    one-process-per-GPU launch rank 0 runs it alone and the other ranks leave at once -- otherwise every rank would push
    the whole input through, append to the same files and write trace files under the same worker id.
    Returns True when this process should return without doing anything."""
    from .distributed import env_rank
    rank, _local, world = env_rank()
    if world > 1 and rank == 0:
        logging.warning(f"{what} runs on one GPU: ranks 1..{world - 1} of this launch stay idle")
    return world > 1 and rank > 0


def specimux_mp(args):
    """File-output entry (`-F`).  The reference forks a worker pool here; the GPU path needs no host
    parallelism for the matching itself."""
    if getattr(args, "diagnostics", None):
        # trace events are attached to record objects: the record path (Python parser) carries them
        if _only_rank0_runs("-F with -d (trace logging)"):
            return
        _run_records(args, to_files=True)
        return
    _run_native(args)


def specimux(args):
    """stdout entry (no `-F`)."""
    if _only_rank0_runs("stdout mode"):
        return
    _run_records(args, to_files=False)
