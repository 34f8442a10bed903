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
