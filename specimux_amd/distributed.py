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


def _append_file(src, dst_fd):
    """Append the whole of `src` to the open descriptor `dst_fd` inside the kernel when the platform allows it."""
    with open(src, "rb") as fh:
        left = os.fstat(fh.fileno()).st_size
        try:
            while left > 0:
                n = os.sendfile(dst_fd, fh.fileno(), None, min(left, 1 << 30))
                if n == 0:
                    break
                left -= n
        except OSError:   # sendfile not available for this pair of files: plain copy of what is left
            while True:
                buf = fh.read(8 << 20)
                if not buf:
                    break
                os.write(dst_fd, buf)


def merge_rank_trees(output_dir, world, rank=0, n_mergers=1, keep=()):
    """After the barrier that says every rank's tree is complete and closed, EVERY rank calls this with its own rank and
    n_mergers = world: merger r owns the output files whose relative path hashes to r and builds each of them from the rank
    trees <output_dir>/.smx_rank_<k>/ in ascending k (= file order of the input under byte-range sharding, so a merged
    file holds its records in input order exactly like a single-process run).  The first piece of a file that does not
    exist yet is renamed into place (same file system: no bytes move), the others are appended with sendfile.  The
    analogue of the reference's workers appending to shared files under a per-file lock (io_utils.py:108-121), done once
    per file instead of once per record, and in parallel over the files.  Returns the number of files this caller built.
    The rank trees themselves are removed by their owners afterwards (remove_rank_tree)."""
    import zlib
    roots = [rank_dir(output_dir, k) for k in range(world)]
    rels = set()
    for root in roots:
        if not os.path.isdir(root):
            continue
        for dirpath, _dirs, files in os.walk(root):
            for fn in files:
                if fn not in keep:
                    rels.add(os.path.relpath(os.path.join(dirpath, fn), root))
    n_files = 0
    for rel in sorted(rels):
        if zlib.crc32(rel.encode()) % n_mergers != rank:
            continue
        dst = os.path.join(output_dir, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        srcs = [os.path.join(root, rel) for root in roots if os.path.exists(os.path.join(root, rel))]
        if not os.path.exists(dst):
            os.rename(srcs[0], dst)
            srcs = srcs[1:]
        if srcs:
            fd = os.open(dst, os.O_WRONLY | os.O_APPEND)
            try:
                for s in srcs:
                    _append_file(s, fd)
            finally:
                os.close(fd)
        n_files += 1
    return n_files


def remove_rank_tree(output_dir, rank):
    import shutil
    shutil.rmtree(rank_dir(output_dir, rank), ignore_errors=True)


def rank_dir(output_dir, rank):
    return os.path.join(output_dir, f".smx_rank_{rank}")


ST_OK, ST_CANNOT_CUT, ST_ERROR = 0, 1, 2


def _validate_shard(sequence_file, byte_range):
    """Parse the records of a byte range without keeping them: ST_OK, ST_CANNOT_CUT (compressed input, FASTA, wrapped FASTQ:
    only a reader of the whole file can deliver the records) or ST_ERROR."""
    from . import _lib
    from .native_io import Reader
    try:
        reader = Reader(sequence_file, byte_range=byte_range)
        try:
            while True:
                b = reader.next_batch(262144, 256 << 20)
                if b is None:
                    return ST_OK
                b.close()
        finally:
            reader.close()
    except _lib.SmxError as e:
        return ST_CANNOT_CUT if e.code == _lib.ERR_UNSUPPORTED else ST_ERROR
    except Exception:   # noqa: BLE001 -- agreed on by all ranks, raised by the caller
        return ST_ERROR


def run_sharded(sequence_file, output_dir, prefix, counts_len, shard_runner, backend=None, window=None):
    """One input file over WORLD_SIZE processes (one per GPU, launched by torch.distributed.run before anything touches
    a GPU).  Rank k demultiplexes the records that start inside its byte range into its own tree
    (`shard_runner(sequence_file, rank_output_dir, byte_range, stride) -> (total, matched, counts uint64[counts_len])`;
    `stride` = (rank, world) instead of a byte range when the file cannot be cut -- gzip, FASTA, wrapped FASTQ -- or when
    the run is restricted to a record window, `window` = True: -n start,num counts records from the start of the file,
    which only a reader of the whole file can do), the counts vectors are summed over the ranks with one all-reduce
    (RCCL through the C ABI on GPUs).

    Where the records go.  Default: every rank APPENDS to the files of the one output tree, as the reference's worker
    processes do under their per-file lock (io_utils.py:108-121): the native writer hands a file's pending records to the
    kernel as one O_APPEND writev per flush, which a local file system (tmpfs, ext4, xfs) serialises per inode, so records
    of different ranks interleave in whole flushes and never inside a record.  Like the reference's, the order of records
    inside a file is then not the input order.  Because nothing written can be taken back, a byte-range shard is parsed
    once WITHOUT writing first (a few per cent of the run) so that the ranks agree on byte ranges or batch striding before
    the first record is out.  SMX_RANK_MERGE=1: every rank writes its own tree <out>/.smx_rank_k and all ranks merge the
    trees in parallel (merge_rank_trees): every file then holds its records in input order, exactly like a single-process
    run, at the price of copying (world - 1) / world of the output once more (two ranks sharing one GPU and its 16 host
    cores, 765 000 reads: 0.73-0.76 s against 0.52-0.55 s; `tools/two_rank_e2e.py`).

    A failure on one rank is agreed on by all ranks (one MAX all-reduce of a status word) before anybody raises, so no
    rank is left waiting in a collective.  Returns (global total, global matched, global counts, world) on every rank."""
    import glob
    import shutil
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
    dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    def agree(code):
        """The worst status over all ranks."""
        flag = torch.tensor([code], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return int(flag.item())

    def attempt(byte_range, stride):
        """-> (status, result or exception): every exception is caught so that the ranks can agree on what to do next."""
        from . import _lib
        try:
            return ST_OK, shard_runner(sequence_file, out, byte_range, stride)
        except _lib.SmxError as e:
            return (ST_CANNOT_CUT if (byte_range is not None and e.code == _lib.ERR_UNSUPPORTED) else ST_ERROR), e
        except Exception as e:   # noqa: BLE001 -- re-raised after the collective
            return ST_ERROR, e

    try:
        if on_gpu:
            from . import _lib
            _lib.check(_lib.load().smx_device_init(local_rank if backend == "nccl" else 0, None))
        # stale rank trees of a killed earlier run must not be merged into this one (the writers append)
        if rank == 0:
            for stale in glob.glob(os.path.join(output_dir, ".smx_rank_*")):
                shutil.rmtree(stale, ignore_errors=True)
        dist.barrier()
        merge_mode = bool(os.environ.get("SMX_RANK_MERGE"))
        size = os.path.getsize(sequence_file)
        if merge_mode:
            out = rank_dir(output_dir, rank)
            os.makedirs(out, exist_ok=True)
            status, res = (ST_CANNOT_CUT, None) if window else attempt(shard_range(size, rank, world), None)
            worst = agree(status)
            if worst == ST_CANNOT_CUT:   # every rank switches to batch striding, or none
                shutil.rmtree(out, ignore_errors=True)
                os.makedirs(out, exist_ok=True)
                status, res = attempt(None, (rank, world))
                worst = agree(status)
        else:
            out = output_dir
            os.makedirs(out, exist_ok=True)
            cut = ST_CANNOT_CUT if window else _validate_shard(sequence_file, shard_range(size, rank, world))
            cut = agree(cut)
            if cut == ST_ERROR:
                raise RuntimeError(f"{sequence_file}: the input could not be parsed (this rank or another; see the messages)")
            status, res = attempt(None, (rank, world)) if cut == ST_CANNOT_CUT else attempt(shard_range(size, rank, world), None)
            if status == ST_CANNOT_CUT:   # cannot happen after the validation pass; never silently
                status = ST_ERROR
            worst = agree(status)
        if worst != ST_OK:
            if merge_mode:
                shutil.rmtree(out, ignore_errors=True)
            if status != ST_OK:
                raise res
            raise RuntimeError("another rank failed while demultiplexing its shard (see its message)")
        total, matched, counts = res
        assert len(counts) == counts_len
        t = torch.from_numpy(np.ascontiguousarray(counts).astype(np.int64)).to(dev)
        reducer = CountsReducer(world, rank, "rccl" if backend == "nccl" else "torch")
        reducer.allreduce_(t)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        reducer.close()
        gcounts = t.cpu().numpy().astype(np.uint64)
        dist.barrier()          # every rank's records are written and its files closed
        if merge_mode:
            merge_rank_trees(output_dir, world, rank, world)
            dist.barrier()          # every output file is complete
            remove_rank_tree(output_dir, rank)
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
