"""Multi-GPU plumbing: one process per GPU, reads sharded by rank, no collective on the data path.

The only exchange of the whole job is the sum of the per-rank counts vectors
(`[total, matched, filtered, ..., per-specimen...]`, include/smx.h SMX_CNT_*) -- the analogue of the
reference's parent process adding up `(batch_total, batch_matched)` tuples (orchestration.py:203-207).
On GPUs it runs through the C ABI (`smx_counts_allreduce`, RCCL over xGMI); the rendezvous for the RCCL
unique id and the CPU rehearsal (gloo) use torch.distributed."""
import ctypes as C
import os


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) of `n_items` owned by `rank` (strong scaling over one input file)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seed(seed: int, rank: int) -> int:
    """Weak scaling over synthetic data: every rank generates its own shard."""
    return seed + rank


def stride_batches(reader, rank, world, batch_reads, batch_bytes=0):
    """Fallback sharding for inputs that cannot be cut by byte range (gzip, FASTA, wrapped FASTQ): every rank parses the
    whole file and keeps batch i iff i % world == rank.  Yields the kept batches."""
    i = 0
    while True:
        b = reader.next_batch(batch_reads, batch_bytes)
        if b is None:
            return
        if i % world == rank:
            yield b
        else:
            b.close()
        i += 1


def merge_rank_trees(output_dir, world, keep=()):
    """Rank 0, after the barrier: append every file of <output_dir>/.smx_rank_<k>/ (k ascending = file order of the
    input, so each merged file holds its records in input order exactly like a single-process run) to the same relative
    path under output_dir, then remove the rank trees.  The analogue of the reference's workers appending to shared
    files under a lock (io_utils.py:108-121), done once instead of per record."""
    import shutil
    n_files = 0
    for k in range(world):
        root = rank_dir(output_dir, k)
        if not os.path.isdir(root):
            continue
        for dirpath, _dirs, files in sorted(os.walk(root)):
            for fn in sorted(files):
                if fn in keep:
                    continue
                src = os.path.join(dirpath, fn)
                dst = os.path.join(output_dir, os.path.relpath(src, root))
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                with open(src, "rb") as a, open(dst, "ab") as b:
                    shutil.copyfileobj(a, b, 8 << 20)
                n_files += 1
        shutil.rmtree(root, ignore_errors=True)
    return n_files


def rank_dir(output_dir, rank):
    return os.path.join(output_dir, f".smx_rank_{rank}")


def run_sharded(sequence_file, output_dir, prefix, counts_len, shard_runner, backend=None):
    """One input file over WORLD_SIZE processes (one per GPU, launched by torch.distributed.run before anything touches
    a GPU).  Rank k demultiplexes the records that start inside its byte range into its own tree
    (`shard_runner(sequence_file, rank_output_dir, byte_range, stride) -> (total, matched, counts uint64[counts_len])`;
    `stride` = (rank, world) instead of a byte range when the file cannot be cut), the counts vectors are summed over
    the ranks with one all-reduce (RCCL through the C ABI on GPUs), rank 0 merges the trees.
    Returns (global total, global matched, global counts, world) on every rank."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    on_gpu = torch.cuda.is_available() and torch.cuda.device_count() > 0
    backend = backend or os.environ.get("SMX_DIST_BACKEND") or ("nccl" if on_gpu else "gloo")
    own_group = not dist.is_initialized()
    if own_group:
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    try:
        if on_gpu:
            from . import _lib
            _lib.check(_lib.load().smx_device_init(local_rank if backend == "nccl" else 0, None))
        size = os.path.getsize(sequence_file)
        byte_range, stride = shard_range(size, rank, world), None
        out = rank_dir(output_dir, rank)
        os.makedirs(out, exist_ok=True)
        try:
            total, matched, counts = shard_runner(sequence_file, out, byte_range, None)
            cut_ok = 1
        except Exception as e:   # not an uncompressed 4-line FASTQ: every rank must switch, or none
            if "byte range" not in str(e):
                raise
            cut_ok = 0
        flag = torch.tensor([cut_ok], dtype=torch.int32, device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            import shutil
            shutil.rmtree(out, ignore_errors=True)
            os.makedirs(out, exist_ok=True)
            stride = (rank, world)
            total, matched, counts = shard_runner(sequence_file, out, None, stride)
        assert len(counts) == counts_len
        dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
        t = torch.from_numpy(np.ascontiguousarray(counts).astype(np.int64)).to(dev)
        reducer = CountsReducer(world, rank, "rccl" if backend == "nccl" else "torch")
        reducer.allreduce_(t)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        reducer.close()
        gcounts = t.cpu().numpy().astype(np.uint64)
        dist.barrier()          # every rank's tree is complete and closed
        if rank == 0:
            merge_rank_trees(output_dir, world)
        dist.barrier()
        return int(gcounts[0]), int(gcounts[1]), gcounts, world
    finally:
        if own_group:
            dist.destroy_process_group()


class CountsReducer:
    """Sums a counts vector across ranks once per job.

    backend 'rccl': device tensor, C-ABI communicator (smx_comm_* / smx_counts_allreduce);
    backend 'torch': whatever torch.distributed backend is initialised (gloo on CPU in the tests)."""

    def __init__(self, world: int, rank: int, backend: str):
        self.world, self.rank, self.backend = world, rank, backend
        self.comm = C.c_void_p()
        if world > 1 and backend == "rccl":
            import sys
            import torch
            import torch.distributed as dist
            # Every step below is collective-safe: rank 0 always broadcasts (the id or None), and the ranks agree by an
            # all-reduce on whether everybody can go on -- a failure on one rank must not leave the others waiting.
            lib = None
            payload = None
            try:
                from . import _lib
                lib = _lib.load()
                if rank == 0:
                    uid = (C.c_uint8 * 128)()
                    _lib.check(lib.smx_comm_unique_id(uid))
                    payload = bytes(uid)
            except Exception as e:
                print(f"[specimux_amd] rank {rank}: C-ABI RCCL communicator unavailable ({e})", file=sys.stderr)
                lib = None
            box = [payload]
            dist.broadcast_object_list(box, src=0)
            ready = torch.tensor([1 if (lib is not None and box[0] is not None) else 0], dtype=torch.int32,
                                 device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(ready, op=dist.ReduceOp.MIN)
            if int(ready.item()) == 1:
                from . import _lib
                uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
                _lib.check(lib.smx_comm_init(uid, world, rank, C.byref(self.comm)))
            else:   # the same collective through torch.distributed's RCCL communicator instead
                if rank == 0:
                    print("[specimux_amd] using torch.distributed all_reduce for the counts", file=sys.stderr)
                self.comm = C.c_void_p()
                self.backend = "torch"

    def allreduce_(self, counts, stream_ptr=None):
        """In-place sum of an int64/uint64 tensor over all ranks."""
        if self.world == 1:
            return counts
        if self.backend == "rccl":
            from . import _lib
            _lib.check(_lib.load().smx_counts_allreduce(C.c_void_p(counts.data_ptr()), counts.numel(), self.comm,
                                                        C.c_void_p(stream_ptr) if stream_ptr else None))
        else:
            import torch.distributed as dist
            if stream_ptr and counts.is_cuda:
                # counts were produced on `stream_ptr`: the collective runs on torch's current stream, which must wait
                # for the PRODUCING stream (not for itself)
                import torch
                ev = torch.cuda.Event()
                ev.record(torch.cuda.ExternalStream(stream_ptr))
                torch.cuda.current_stream().wait_event(ev)
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        return counts

    def close(self):
        if self.comm:
            from . import _lib
            _lib.load().smx_comm_destroy(self.comm)
            self.comm = C.c_void_p()
