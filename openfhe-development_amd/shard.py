"""Multi-GPU driver: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU tests).

The DCRTPoly hot path shards naturally — every polynomial / ciphertext of a batch is independent (the reference
has no cross-ciphertext dependence in EvalMult, key switching or bootstrapping) — so:
  * the batch is partitioned into contiguous slices, one per rank (`shard_range`);
  * every rank holds a full replica of the context tables (built locally from (N, q_i, psi_i));
  * evaluation keys are produced once (rank 0) and replicated with ONE broadcast over xGMI (`broadcast_key`);
  * there is no collective on the data path; results stay on the rank that computed them.
"""
import numpy as np


def shard_range(total, rank, world):
    """contiguous slice [lo, hi) of `total` independent units for `rank`; sizes differ by at most one"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_key(plan, keyB_host, keyA_host, device, src=0):
    """Replicates an evaluation key (b and a vectors, uint64[numPartQ][sizeQ+sizeP][N]) from rank `src` to every
    rank with torch.distributed.broadcast and hands the rank-local copy to the key-switch plan.
    `keyB_host`/`keyA_host` are only read on rank `src` (other ranks may pass None).
    Returns the tensors (keep them alive as long as the plan uses the key)."""
    import torch
    import torch.distributed as dist

    words = plan.key_words()
    tens = []
    for host in (keyB_host, keyA_host):
        t = torch.empty(words, dtype=torch.int64, device=device)
        if dist.get_rank() == src:
            h = np.ascontiguousarray(host, dtype=np.uint64).reshape(-1)
            assert h.size == words
            t.copy_(torch.from_numpy(h.view(np.int64)))
        dist.broadcast(t, src=src)  # the only collective: eval keys over xGMI, once, at setup
        tens.append(t)
    plan.wrap_key(tens[0].data_ptr(), tens[1].data_ptr())
    return tens


def broadcast_rotation_keys(plan, host_keys, device, src=0):
    """Replicates a SET of evaluation keys (the rotation keys of a linear transform / of bootstrapping) with ONE
    broadcast: rank `src` packs them as int64[nKeys][2][words]; every rank wraps the slices of its copy as key handles.
    host_keys: list of (keyB, keyA) on rank `src` (only its length matters on the other ranks).
    Returns (handles, tensor); keep the tensor alive as long as the handles are used, destroy the handles with
    fhe_ks_key_destroy."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    words, n = plan.key_words(), len(host_keys)
    t = torch.empty((n, 2, words), dtype=torch.int64, device=device)
    if dist.get_rank() == src:
        for i, (kb, ka) in enumerate(host_keys):
            for h, arr in enumerate((kb, ka)):
                flat = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
                assert flat.size == words
                t[i, h].copy_(torch.from_numpy(flat.view(np.int64)))
    dist.broadcast(t, src=src)
    L, handles = plan.ctx.lib.L, []
    for i in range(n):
        k = C.c_void_p()
        plan.ctx.lib.check(L.fhe_ks_key_wrap(plan.h, C.c_void_p(t[i, 0].data_ptr()), C.c_void_p(t[i, 1].data_ptr()), C.byref(k)))
        handles.append(k)
    return handles, t


def broadcast_rows(host_rows, shape, device, src=0):
    """Replicates read-only tables of residues (e.g. the encoded diagonals of a linear transform, uint64 `shape`) from
    rank `src` with one broadcast; returns the rank-local int64 tensor (its data_ptr() + offsets are device pointers)."""
    import torch
    import torch.distributed as dist

    t = torch.empty(shape, dtype=torch.int64, device=device)
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(host_rows, dtype=np.uint64).view(np.int64).reshape(shape)))
    dist.broadcast(t, src=src)
    return t


def allgather_words(host_words, shape, device, src=0):
    """Replicates a large read-only table of 64-bit words (a bootstrapping rotation-key set: ~13 GB at N = 2^17, SURVEY.md
    8e) from rank `src` to every rank as scatter + all-gather: `src` sends each rank ONE 1/world slice, then every rank
    gathers the other slices from its peers — every xGMI link carries 1/world of the table per step, where a ring
    broadcast pushes the whole table through every link of the ring.  `host_words`: uint64 array of `shape` on rank
    `src` (or an int64 tensor already on `device`; ignored elsewhere).  Returns the rank-local int64 tensor of `shape`
    (its data_ptr() + offsets are device pointers; keep it alive while they are in use)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    total = int(np.prod(shape))
    per = -(-total // world)  # slice length, the last slice is padded
    full = torch.empty(per * world, dtype=torch.int64, device=device)
    mine = torch.empty(per, dtype=torch.int64, device=device)
    chunks = None
    if rank == src:
        if isinstance(host_words, torch.Tensor):
            flat = host_words.reshape(-1)
        else:
            flat = torch.from_numpy(np.ascontiguousarray(host_words, dtype=np.uint64).reshape(-1).view(np.int64))
        if flat.numel() == per * world and flat.device == torch.device(device):
            staged = flat
        else:
            staged = torch.zeros(per * world, dtype=torch.int64, device=device)
            staged[:total].copy_(flat)
        chunks = list(staged.split(per))
    dist.scatter(mine, chunks, src=src)
    try:
        dist.all_gather_into_tensor(full, mine)
    except Exception:  # backends without the flat form
        parts = [torch.empty(per, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(parts, mine)
        full = torch.cat(parts)
    return full[:total].view(*shape)


def allgather_rotation_keys(plan, host_keys, device, src=0):
    """`broadcast_rotation_keys` with the scatter + all-gather replication of `allgather_words`: host_keys = list of
    (keyB, keyA) on rank `src` (only its length matters elsewhere).  Returns (handles, tensor)."""
    import ctypes as C

    import torch.distributed as dist

    words, n = plan.key_words(), len(host_keys)
    packed = None
    if dist.get_rank() == src:
        packed = np.empty((n, 2, words), np.uint64)
        for i, (kb, ka) in enumerate(host_keys):
            packed[i, 0] = np.ascontiguousarray(kb, dtype=np.uint64).reshape(-1)
            packed[i, 1] = np.ascontiguousarray(ka, dtype=np.uint64).reshape(-1)
    t = allgather_words(packed, (n, 2, words), device, src)
    L, handles = plan.ctx.lib.L, []
    for i in range(n):
        k = C.c_void_p()
        plan.ctx.lib.check(L.fhe_ks_key_wrap(plan.h, C.c_void_p(t[i, 0].data_ptr()), C.c_void_p(t[i, 1].data_ptr()), C.byref(k)))
        handles.append(k)
    return handles, t
