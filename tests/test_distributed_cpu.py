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
