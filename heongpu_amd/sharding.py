"""Multi-GPU plumbing: one process per GPU, batches of independent ciphertexts
sharded across ranks, evaluation keys broadcast once from rank 0 over
RCCL/xGMI (torch.distributed backend "nccl"; "gloo" in the CPU tests).
There is no per-operation collective: ciphertexts never cross ranks
(SURVEY.md 8e -- the reference itself is single-GPU)."""
import os


def init_distributed(backend=None, device_index=None):
    """Initialise torch.distributed from the torchrun environment.  Returns
    (rank, world, local_rank); world == 1 means single process, no group.
    device_index: the GPU this rank uses (default: LOCAL_RANK)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            idx = local_rank if device_index is None else device_index
            torch.cuda.set_device(idx)
            kwargs["device_id"] = torch.device("cuda", idx)
        dist.init_process_group(backend, **kwargs)
    return rank, world, local_rank


def shard_range(total, world, rank):
    """contiguous slice [start, start+count) of `total` units owned by `rank`;
    the first total % world ranks get one extra unit."""
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def broadcast_eval_key(key, src=0, chunk_elems=1 << 27):
    """Broadcast an evaluation key tensor (int64 bit patterns of the uint64
    residues, layout [digit][2][Q'][N]) from `src` to every rank, in chunks of
    at most `chunk_elems` elements (1 GiB) so that a 272 MiB relin key is one
    message and multi-GiB Galois key sets stay below per-message limits."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return key
    flat = key.view(-1)
    for off in range(0, flat.numel(), chunk_elems):
        dist.broadcast(flat[off:off + chunk_elems], src=src)
    return key
