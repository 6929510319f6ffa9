'''Work sharding over the GPUs of one node: one process per GPU, joined by
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  The scan is embarrassingly parallel - every (window, region)
pair is independent and the selection rule folds only over one window's own
regions - so work is split by contiguous window ranges (row bands of the window
grid) and the only communication is an all-gather of the small per-window
results.  Without an initialised process group everything degenerates to a
single shard.
'''
import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except ImportError:           # pragma: no cover
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def world():
    '''(rank, world_size) of the current process group, (0, 1) without one.'''
    dist = _dist()
    if dist is None:
        return 0, 1
    return dist.get_rank(), dist.get_world_size()


def shard_range(count, rank=None, size=None):
    '''[begin, end) of the contiguous, balanced share of `count` items that
    belongs to `rank` (the first count % size ranks get one extra item).'''
    if rank is None or size is None:
        rank, size = world()
    base, extra = divmod(int(count), int(size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def _device_for_backend(dist):
    import torch
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def all_gather_concat(local, total):
    '''All-gathers 1-D or N-D per-rank arrays that were sharded along axis 0 with
    shard_range(total) and returns the concatenation (a numpy array on every
    rank).  Shares may differ by one row, so they are padded to a common length
    for the collective and trimmed afterwards.'''
    dist = _dist()
    local = np.ascontiguousarray(local)
    if dist is None or dist.get_world_size() == 1:
        return local
    import torch
    size = dist.get_world_size()
    counts = [shard_range(total, r, size) for r in range(size)]
    longest = max(e - b for b, e in counts)
    padded = np.zeros((longest,) + local.shape[1:], dtype=local.dtype)
    padded[:len(local)] = local
    device = _device_for_backend(dist)
    send = torch.from_numpy(padded).to(device)
    recv = torch.empty((size * longest,) + tuple(send.shape[1:]), dtype=send.dtype, device=device)
    dist.all_gather_into_tensor(recv, send)
    recv = recv.cpu().numpy().reshape((size, longest) + local.shape[1:])
    return np.concatenate([recv[r, :e - b] for r, (b, e) in enumerate(counts)], axis=0)
