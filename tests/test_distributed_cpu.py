"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): reads sharded by rank, per-rank counts
vectors summed once; the sum must equal the single-process result.  The per-shard worker here is the
oracle (no GPU in this tier); the sharding and reduction code is the product's (specimux_amd.distributed)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, REPO


def _counts_for(reads):
    from oracle import specimux_oracle as O
    panel = O.load_panel(f"{GOLDEN}/primers.fasta", f"{GOLDEN}/specimens.txt")
    par = O.setup_params(panel)
    ops, total, matched = O.process_sequences(reads, par, panel)
    ids = [s[0] for s in panel.specimens]
    vec = np.zeros(8 + len(ids), dtype=np.int64)
    vec[0], vec[1] = total, matched
    for op in ops:
        cls = 5 if op.rtype == O.R_UNKNOWN else (4 if op.rtype in (O.R_PFWD, O.R_PREV) else 3)
        vec[cls] += 1
        if cls == 3:
            vec[8 + ids.index(op.sample_id)] += 1
    return vec


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from oracle import specimux_oracle as O
    from specimux_amd.distributed import CountsReducer, env_rank, shard_range
    assert env_rank() == (rank, rank, world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    reads, _ = O.read_sequences(f"{GOLDEN}/sequences.fastq")
    lo, hi = shard_range(len(reads), rank, world)
    counts = torch.from_numpy(_counts_for(reads[lo:hi]))
    red = CountsReducer(world, rank, "torch")
    red.allreduce_(counts)
    np.save(os.path.join(out_dir, f"counts_{rank}.npy"), counts.numpy())
    dist.destroy_process_group()


def test_shard_range_partitions():
    from specimux_amd.distributed import shard_range, shard_seed
    for n in (0, 1, 7, 40, 765000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert [shard_seed(2002, r) for r in range(3)] == [2002, 2003, 2004]


@pytest.mark.timeout(300)
def test_two_rank_counts_reduce_equals_single_process(tmp_path):
    from oracle import specimux_oracle as O
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, os.fspath(tmp_path)), nprocs=2, join=True)
    reads, _ = O.read_sequences(f"{GOLDEN}/sequences.fastq")
    whole = _counts_for(reads)
    for rank in range(2):
        assert np.array_equal(np.load(tmp_path / f"counts_{rank}.npy"), whole)
    assert whole[0] == 40 and whole[1] == 6 and whole[8:].tolist() == [2, 3, 1]


# ------------------------------------------------------------------ the FILE path: shard by byte range, per-rank trees, merge
def _file_worker(rank, world, port, tmp, seqfile, pf, sf):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SMX_DIST_BACKEND="gloo")
    from oracle import specimux_oracle as O
    from parity_utils import Both
    from specimux_amd import _lib
    from specimux_amd.demultiplex import compiled_panel
    from specimux_amd.distributed import run_sharded, stride_batches
    from specimux_amd.native_io import Reader, Writer
    from test_native_io_cpu import _ops_from_oracle
    both = Both(pf, sf)
    cp = compiled_panel(both.specimens, both.parameters, both.args, both.prefilter)   # host only: names + sharding metadata

    def shard_runner(path, out_dir, byte_range, stride):
        """The oracle stands in for the kernels (no GPU in this tier); reader, byte ranges, writer are the product's."""
        reader = Reader(path, byte_range=byte_range)
        writer = Writer(out_dir, "", reader.is_fastq, cp)
        counts = np.zeros(cp.counts_len, dtype=np.uint64)
        batches = stride_batches(reader, stride[0], stride[1], 300) if stride else iter(lambda: reader.next_batch(300), None)
        for b in batches:
            recs = [b.record(i) for i in range(len(b))]
            ops, total, matched = O.process_sequences(recs, both.opar, both.opanel)
            sops, extra = _ops_from_oracle(cp, ops, len(recs), {r[0]: i for i, r in enumerate(recs)})
            writer.write(b, sops, extra)
            counts[_lib.CNT_TOTAL] += total
            counts[_lib.CNT_MATCHED] += matched
            for op in ops:
                if op.rtype in (O.R_FULL, O.R_DEREP) and op.sample_id in cp.specimen_ids:
                    counts[_lib.CNT_SPECIMEN0 + cp.specimen_ids.index(op.sample_id)] += 1
            b.close()
        writer.close()
        return int(counts[_lib.CNT_TOTAL]), int(counts[_lib.CNT_MATCHED]), counts

    total, matched, gcounts, w = run_sharded(seqfile, os.path.join(tmp, "out"), "", cp.counts_len, shard_runner)
    np.save(os.path.join(tmp, f"gcounts_{rank}.npy"), gcounts)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind,world,mode", [("fastq", 2, "append"), ("fastq", 2, "merge"), ("gz", 2, "append"), ("gz", 2, "merge"),
                                             ("fastq", 8, "append"), ("fastq", 8, "merge")])
def test_two_rank_file_path_equals_single_process(tmp_path, kind, world, mode, monkeypatch):
    """`specimux -F` under a 2-process launch (and an 8-process one: north_star's node size, rehearsed on gloo), in both
    output modes of run_sharded: ranks appending to the one tree (default; same records per file, any order, no record
    torn), and per-rank trees merged (SMX_RANK_MERGE=1; every file in input order). byte-range shards (stride fallback for gzip), per-rank trees, one
    all-reduce of the counts, merge by rank 0.  The merged tree equals the single-process tree FILE BY FILE (records in
    input order), and each specimen's counter equals the records in full/<pool>/<specimen>.fastq."""
    import gzip
    import shutil
    from conftest import read_expected_tree
    from oracle import specimux_oracle as O
    from specimux_amd import _lib, synth
    pan = synth.panel_c1()
    pf, sf = pan.write(os.fspath(tmp_path / "panel"))
    rs = synth.make_reads(pan, 1500, 77, windows_only=False)
    seqfile = os.fspath(tmp_path / "reads.fastq")
    rs.write_fastq(seqfile)
    if kind == "gz":
        with open(seqfile, "rb") as a, gzip.open(seqfile + ".gz", "wb") as b:
            shutil.copyfileobj(a, b)
        seqfile += ".gz"
    port = 29500 + ((os.getpid() + 7) % 2000)
    if mode == "merge":
        monkeypatch.setenv("SMX_RANK_MERGE", "1")
    else:
        monkeypatch.delenv("SMX_RANK_MERGE", raising=False)
    mp.spawn(_file_worker, args=(world, port, os.fspath(tmp_path), seqfile, pf, sf), nprocs=world, join=True)
    exp_tree, total, matched = O.run_files(pf, sf, seqfile)
    got = {}
    out = tmp_path / "out"
    assert not any(p.name.startswith(".smx_rank_") for p in out.iterdir())   # rank trees merged and removed
    for dirpath, _d, files in os.walk(out):
        for fn in files:
            full = os.path.join(dirpath, fn)
            got[os.path.relpath(full, out)] = open(full).read()
    if kind == "fastq" and mode == "merge":   # byte ranges + merge keep the input order inside every file
        assert got == {k: "".join(v) for k, v in exp_tree.items()}
    else:                 # appending ranks / stride sharding interleave batches: same records, file by file, none torn
        assert {k: sorted(v.split("@read")) for k, v in got.items()} == {k: sorted("".join(v).split("@read")) for k, v in exp_tree.items()}
    g0 = np.load(tmp_path / "gcounts_0.npy")
    for r in range(1, world):
        assert np.array_equal(g0, np.load(tmp_path / f"gcounts_{r}.npy"))
    assert int(g0[_lib.CNT_TOTAL]) == total == 1500 and int(g0[_lib.CNT_MATCHED]) == matched
    panel = O.load_panel(pf, sf)
    for i, spec in enumerate(panel.specimens):
        path = out / "full" / spec[1] / f"{spec[0]}.fastq"
        n_rec = path.read_text().count("\n") // 4 if path.exists() else 0
        assert int(g0[_lib.CNT_SPECIMEN0 + i]) == n_rec, spec[0]
    assert int(g0[_lib.CNT_SPECIMEN0:].sum()) > 500


def test_parallel_merge_equals_rank_order_concatenation(tmp_path):
    """merge_rank_trees split over three mergers (what the three ranks do after the barrier): every output file is the
    concatenation of the rank trees' files in rank order, whoever built it; a file that existed before is appended to;
    every file is built by exactly one merger."""
    from specimux_amd.distributed import merge_rank_trees, rank_dir, remove_rank_tree
    out = tmp_path / "out"
    world = 3
    expect = {}
    rng = np.random.default_rng(5)
    names = [f"full/P/spec{i}.fastq" for i in range(40)] + [f"partial/P/a-b/x{i}.fastq" for i in range(25)] + ["unknown/u.fastq"]
    (out / "full" / "P").mkdir(parents=True)
    (out / "full" / "P" / "spec3.fastq").write_text("@old\nAC\n+\nII\n")      # left by an earlier run: appended to
    expect["full/P/spec3.fastq"] = "@old\nAC\n+\nII\n"
    for k in range(world):
        for nm in names:
            if rng.random() < 0.6 or nm == "unknown/u.fastq":
                path = os.path.join(rank_dir(os.fspath(out), k), nm)
                os.makedirs(os.path.dirname(path), exist_ok=True)
                text = "".join(f"@r{k}_{j}\nACGT\n+\nIIII\n" for j in range(int(rng.integers(1, 50))))
                with open(path, "w") as fh:
                    fh.write(text)
                expect[nm] = expect.get(nm, "") + text
    built = [merge_rank_trees(os.fspath(out), world, r, world) for r in range(world)]
    for r in range(world):
        remove_rank_tree(os.fspath(out), r)
    got = {}
    for dirpath, _d, files in os.walk(out):
        for fn in files:
            full = os.path.join(dirpath, fn)
            got[os.path.relpath(full, out)] = open(full).read()
    assert got == expect
    assert sum(built) == len(expect) and min(built) > 0
    assert not any(p.name.startswith(".smx_rank_") for p in out.iterdir())


def _failing_worker(rank, world, port, tmp, seqfile):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SMX_DIST_BACKEND="gloo")
    from specimux_amd.distributed import run_sharded

    def shard_runner(path, out_dir, byte_range, stride):
        if rank == 1:
            raise ValueError("rank 1 cannot read its shard")
        with open(os.path.join(out_dir, "x.fastq"), "w") as fh:
            fh.write("@r\nA\n+\nI\n")
        return 1, 0, np.zeros(9, dtype=np.uint64)

    try:
        run_sharded(seqfile, os.path.join(tmp, "out"), "", 9, shard_runner)
        outcome = "returned"
    except ValueError as e:
        outcome = f"ValueError: {e}"
    except RuntimeError as e:
        outcome = f"RuntimeError: {e}"
    with open(os.path.join(tmp, f"outcome_{rank}.txt"), "w") as fh:
        fh.write(outcome)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["append", "merge"])
def test_two_rank_failure_is_raised_on_every_rank(tmp_path, mode, monkeypatch):
    """An exception in one rank's shard must not leave the other rank waiting in the counts all-reduce: the ranks agree on a
    status word first, the failing rank re-raises its exception, the other raises too, and a stale rank tree left by an
    earlier (killed) run is gone.  With per-rank trees (SMX_RANK_MERGE=1) no tree is merged; ranks that append to the one
    tree leave what they wrote before the failure, as a failing single-process run does."""
    if mode == "merge":
        monkeypatch.setenv("SMX_RANK_MERGE", "1")
    else:
        monkeypatch.delenv("SMX_RANK_MERGE", raising=False)
    from specimux_amd.distributed import rank_dir
    seqfile = tmp_path / "reads.fastq"
    seqfile.write_text("@a\nACGT\n+\nIIII\n" * 50)
    stale = rank_dir(os.fspath(tmp_path / "out"), 5)
    os.makedirs(stale)
    open(os.path.join(stale, "old.fastq"), "w").write("@old\nA\n+\nI\n")
    port = 29500 + ((os.getpid() + 13) % 2000)
    mp.spawn(_failing_worker, args=(2, port, os.fspath(tmp_path), os.fspath(seqfile)), nprocs=2, join=True)
    o0, o1 = (tmp_path / "outcome_0.txt").read_text(), (tmp_path / "outcome_1.txt").read_text()
    assert o1 == "ValueError: rank 1 cannot read its shard"
    assert o0.startswith("RuntimeError: another rank failed")
    assert not os.path.exists(stale)
    if mode == "merge":
        assert not (tmp_path / "out" / "x.fastq").exists()
