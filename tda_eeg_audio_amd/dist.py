"""
dist.py -- one process per GPU; recordings are sharded, results are all-gathered once.

Replaces the reference's only parallel/merge machinery: joblib workers over recordings
(scripts/tda_eeg_classification_v2.py:569-572) and the file-based partial merge
(BATCH_START/BATCH_END + features/partials/*.npz, v2:55-60,608-638) by
  * shard_recordings(): whole recordings per rank (keeps the per-recording mean/std v2:429-436
    and the per-recording-band tau cmp:83 rank-local), dealt by descending window count so the
    per-rank work is balanced (slow recordings have ~80 windows, fast ~49), and
  * all_gather_rows(): ONE torch.distributed all_gather (RCCL over xGMI on the GPU box, gloo in
    the CPU tests) of the padded per-rank result block, then the inverse permutation restores
    the reference's row order (features/filenames.txt).
No other collective exists on this path: windows are independent.
"""
import numpy as np


def shard_recordings(n_windows, world_size):
    """n_windows: per-recording window counts.  Returns a list (per rank) of recording indices."""
    n_windows = np.asarray(n_windows)
    order = np.argsort(-n_windows, kind="stable")
    shards = [[] for _ in range(world_size)]
    load = np.zeros(world_size, dtype=np.int64)
    for k, rec in enumerate(order):
        # snake deal: 0..W-1, W-1..0 -- balanced totals without a priority queue
        rnd, pos = divmod(k, world_size)
        r = pos if rnd % 2 == 0 else world_size - 1 - pos
        shards[r].append(int(rec))
        load[r] += n_windows[rec]
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


_IDX_CACHE = {}


def _index_tensor(rows, device):
    """Row-index tensor on `device`, uploaded once per distinct index list (a pageable H2D copy per
    step would put a host synchronisation into the timed loop)."""
    import torch
    key = (device, np.asarray(rows, np.int64).tobytes())
    t = _IDX_CACHE.get(key)
    if t is None:
        t = torch.as_tensor(np.asarray(rows, np.int64), device=device)
        _IDX_CACHE[key] = t
    return t


def all_gather_rows(local_rows, my_recs, shards, n_total, group=None):
    """local_rows: (len(my_recs), k) tensor of this rank's results, rows in the order of
    shards[rank].  Returns the (n_total, k) tensor in original recording order on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    k = local_rows.shape[1]
    if world == 1:
        out = torch.empty((n_total, k), dtype=local_rows.dtype, device=local_rows.device)
        out[_index_tensor(my_recs, local_rows.device)] = local_rows
        return out
    pad = max(len(s) for s in shards)
    send = torch.zeros((pad, k), dtype=local_rows.dtype, device=local_rows.device)
    send[: local_rows.shape[0]] = local_rows
    recv = torch.empty((world * pad, k), dtype=local_rows.dtype, device=local_rows.device)
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        # rehearsal on a shared GPU: gloo collectives are staged through host memory
        r_h = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r_h, send.cpu(), group=group)
        recv.copy_(r_h)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    out = torch.empty((n_total, k), dtype=local_rows.dtype, device=local_rows.device)
    for r, s in enumerate(shards):
        if len(s):
            out[_index_tensor(s, local_rows.device)] = recv[r * pad: r * pad + len(s)]
    return out


def init_from_env(backend=None):
    """torch.distributed init from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun contract)."""
    import os
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
