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
