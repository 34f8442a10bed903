"""BASELINE configs[3] as north_star states it: a batch of CKKS ciphertexts bootstrapped through the reference's own API
(cc->EvalBootstrap) on the HIP backend of DCRTPoly, the batch sharded over one process per GPU, the evaluation keys generated on
rank 0 and replicated over xGMI with scatter + all-gather (shard.allgather_words).

The native side is openfhe-development_amd/hal/bootstrap_batch.cpp (a C ABI over the reference's CryptoContext, built into
hal/_build/libfhe_boot_batch_hip.so; the same source against the stock libraries is the TEST-ONLY byte-for-byte reference).  This
module is the ctypes binding and the per-rank sequence; bench.py and tests/test_multi_gpu_gloo.py call `run_rank`."""
import ctypes as C
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Two builds of the same backend: the reference's sources patched at source level (integration/with_hip.patch, plain compiler flags) and
# the unmodified sources with the hooks bound on object files (objcopy --weaken-symbol, hal/Makefile).  The patched build is the one
# used when it is present; FHE_HAL_BUILD=objcopy selects the other.
OBJCOPY_SO = os.path.join(ROOT, "openfhe-development_amd", "hal", "_build", "libfhe_boot_batch_hip.so")
PATCHED_SO = os.path.join(ROOT, "integration", "_build", "lib", "libfhe_boot_batch_hip.so")


def _newest_backend_source():
    """modification time of the newest source either build is made of (the backend's sources and headers, the C ABI): a patched build
    older than that is STALE (round-5 advisor: after `make -C hal` alone the benchmark kept running an old patched backend)"""
    import glob
    hal = os.path.join(ROOT, "openfhe-development_amd", "hal")
    files = [os.path.join(ROOT, "include", "fhe_hip.h"), os.path.join(ROOT, "integration", "with_hip.patch")]
    for pat in ("*.cpp", "*.h", "lattice/**/*.h", "math/**/*.h"):
        files += glob.glob(os.path.join(hal, pat), recursive=True)
    return max((os.path.getmtime(f) for f in files if os.path.exists(f)), default=0.0)


def _choose_build():
    if os.environ.get("FHE_HAL_BUILD", "") == "objcopy" or not os.path.exists(PATCHED_SO):
        return OBJCOPY_SO, "unmodified sources, hooks bound with objcopy (hal/Makefile)"
    # (on the GPU box the snapshot's files all carry the copy's time: only a build that is older than the sources AND older than the
    # other build is refused)
    stale = os.path.getmtime(PATCHED_SO) < _newest_backend_source() and os.path.exists(OBJCOPY_SO) and \
        os.path.getmtime(PATCHED_SO) < os.path.getmtime(OBJCOPY_SO)
    if stale:
        import sys
        print("boot_batch: integration/_build is older than the backend's sources and than hal/_build: using the objcopy build "
              "(run integration/build_patched.sh)", file=sys.stderr)
        return OBJCOPY_SO, "unmodified sources, hooks bound with objcopy (hal/Makefile) — the patched build was stale"
    return PATCHED_SO, "patched sources (integration/with_hip.patch)"


HIP_SO, HAL_BUILD = _choose_build()
u32, u64, vp = C.c_uint32, C.c_uint64, C.c_void_p


class BootBatch:
    def __init__(self, so, logN, slots, budget=(4, 4), levels_after=5, prng=None, device=0, omp_threads=None):
        if not os.path.exists(so):
            raise RuntimeError(f"{so} not built (./build.sh hal needs the reference sources)")
        L = self.L = C.CDLL(so)
        L.fbb_set_omp_threads.argtypes = [C.c_int]
        if omp_threads:  # (before the context exists: key-pair generation already runs pke's OpenMP regions)
            L.fbb_set_omp_threads(omp_threads)
        L.fbb_create.restype, L.fbb_create.argtypes = vp, [u32, u32, u32, u32, u32, C.c_char_p, C.c_int]
        L.fbb_error.restype, L.fbb_error.argtypes = C.c_char_p, [vp]
        L.fbb_destroy.argtypes = [vp]
        L.fbb_set_omp_threads.argtypes = [C.c_int]
        L.fbb_set_active_levels.argtypes = [C.c_int]
        L.fbb_shape.argtypes = [vp, C.POINTER(u32)]
        L.fbb_encrypt.argtypes = [vp, u32, u32, u32]
        L.fbb_keygen.argtypes = [vp]
        L.fbb_key_count.restype, L.fbb_key_count.argtypes = u32, [vp, C.POINTER(u32), u32]
        L.fbb_key_layout.argtypes = [vp, C.POINTER(u64), C.POINTER(u32)]
        L.fbb_make_key_shells.argtypes = [vp, C.POINTER(u32), u32]
        L.fbb_export_keys.argtypes = [vp, vp]
        L.fbb_adopt_keys.argtypes = [vp, vp]
        L.fbb_bootstrap_all.restype, L.fbb_bootstrap_all.argtypes = C.c_double, [vp, C.c_int, C.c_int, C.c_int]
        if hasattr(L, "fbb_bootstrap_wide"):
            L.fbb_bootstrap_wide.restype, L.fbb_bootstrap_wide.argtypes = C.c_double, [vp, u32, C.c_int]
            L.fbb_bootstrap_wide_mt.restype, L.fbb_bootstrap_wide_mt.argtypes = C.c_double, [vp, u32, C.c_int, C.c_int]
        L.fbb_check.restype, L.fbb_check.argtypes = C.c_double, [vp, u32, C.POINTER(C.c_double)]
        L.fbb_counters.argtypes = [C.POINTER(u64)]
        L.fbb_member_stats.restype, L.fbb_member_stats.argtypes = C.c_size_t, [C.c_char_p, C.c_size_t]
        if hasattr(L, "fbb_alloc_stats"):
            L.fbb_alloc_stats.argtypes = [C.POINTER(u64)]
            L.fbb_reserve.restype, L.fbb_reserve.argtypes = C.c_int, [u64]
        L.fbb_save_outputs.argtypes = [vp]
        L.fbb_compare_saved.restype, L.fbb_compare_saved.argtypes = C.c_long, [vp]
        L.fbb_dump.argtypes = [vp, C.c_char_p, u32, u32]
        self.h = L.fbb_create(logN, slots, budget[0], budget[1], levels_after, prng.encode() if prng else None, device)
        self._ok(0)
        sh = (u32 * 5)()
        L.fbb_shape(self.h, sh)
        self.N, self.sizeQ, self.sizeP, self.dnum, self.depth = (int(v) for v in sh)

    def _ok(self, rc):
        err = self.L.fbb_error(self.h).decode()
        if rc != 0 or err:
            raise RuntimeError("bootstrap batch: " + (err or f"status {rc}"))

    def encrypt(self, total, first, count):
        self._ok(self.L.fbb_encrypt(self.h, total, first, count))

    def keygen(self):
        self._ok(self.L.fbb_keygen(self.h))

    def key_indices(self):
        n = self.L.fbb_key_count(self.h, None, 0)
        idx = (u32 * n)()
        self.L.fbb_key_count(self.h, idx, n)
        return np.array(list(idx), np.uint32)

    def key_layout(self):
        w, per = u64(), u32()
        self.L.fbb_key_layout(self.h, C.byref(w), C.byref(per))
        return int(w.value), int(per.value)

    def make_key_shells(self, indices):
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        self._ok(self.L.fbb_make_key_shells(self.h, idx.ctypes.data_as(C.POINTER(u32)), len(idx)))

    def export_keys(self, dev_ptr):
        self._ok(self.L.fbb_export_keys(self.h, vp(dev_ptr)))

    def adopt_keys(self, dev_ptr):
        self._ok(self.L.fbb_adopt_keys(self.h, vp(dev_ptr)))

    def single_thread_latency(self, reps=1):
        """seconds per bootstrap with every OpenMP region of the process confined to its calling thread: ONE host thread, ONE stream"""
        self.L.fbb_set_active_levels(0)
        try:
            s = self.bootstrap_all(1, reps, 0)
        finally:
            self.L.fbb_set_active_levels(1)
        return s

    def bootstrap_all(self, threads, reps, warmup=1):
        s = self.L.fbb_bootstrap_all(self.h, threads, reps, warmup)
        self._ok(0 if s >= 0 else 1)
        return s

    def bootstrap_wide(self, group, reps, threads=1):
        """the rank's ciphertexts in lockstep, `group` per wide evaluation (0 = all): one cc->EvalBootstrap on a ciphertext whose towers hold
        `group` towers each (hal/bootstrap_batch.cpp fbb_bootstrap_wide); one narrow pass must have run before.  threads > 1: the groups
        spread over that many host threads (streams), so that one group's kernels run while another's thread is in pke's host code.
        Seconds per pass."""
        s = self.L.fbb_bootstrap_wide_mt(self.h, group, reps, threads)
        self._ok(0 if s >= 0 else 1)
        return s

    def counters(self):
        """{operand bytes read / written by the device operations so far (every tower or key an operation touches, once per operation),
        kernel launches, PCIe bytes host->device / device->host} of the backend in this process"""
        c = (u64 * 5)()
        self.L.fbb_counters(c)
        return dict(zip(("operand_read_bytes", "operand_write_bytes", "launches", "h2d_bytes", "d2h_bytes"), (int(v) for v in c)))

    def member_bytes(self):
        """{member (the outermost pke / DCRTPoly scope): operand bytes of its device operations so far}"""
        n = self.L.fbb_member_stats(None, 0)
        buf = C.create_string_buffer(n)
        self.L.fbb_member_stats(buf, n)
        out = {}
        for ln in buf.value.decode().split("\n"):
            f = ln.split()
            if len(f) >= 5:  # "<name (may hold blanks)> deviceOps hostOps hostReads operandBytes"
                out[" ".join(f[:-4])] = int(f[-1])
        return out

    def alloc_stats(self):
        """the backend's buffer caches (fhe_hal_alloc_stats / _stats2)"""
        if not hasattr(self.L, "fbb_alloc_stats"):
            return None
        c = (u64 * 9)()
        self.L.fbb_alloc_stats(c)
        names = ("cached_bytes", "served_from_another_threads_cache", "requests_that_reached_the_device", "cache_releases", "device_free_bytes",
                 "device_total_bytes", "held_from_the_device_bytes", "held_high_water_bytes", "served_behind_the_buffers_own_mark")
        return dict(zip(names, (int(v) for v in c)))

    def reserve(self, nbytes):
        """pre-sizes the calling thread's cache with one buffer of nbytes (fhe_hal_reserve)"""
        return int(self.L.fbb_reserve(nbytes)) if hasattr(self.L, "fbb_reserve") else 1

    def member_stats(self):
        """{member: (device operations, host-mirror executions, host reads after a device->host copy, operand bytes)} so far"""
        n = self.L.fbb_member_stats(None, 0)
        buf = C.create_string_buffer(n)
        self.L.fbb_member_stats(buf, n)
        out = {}
        for ln in buf.value.decode().split("\n"):
            f = ln.split()
            if len(f) >= 5:
                out[" ".join(f[:-4])] = tuple(int(v) for v in f[-4:])
        return out

    def save_outputs(self):
        """keeps the current outputs for compare_saved (the next pass produces new objects)"""
        self._ok(self.L.fbb_save_outputs(self.h))

    def compare_saved(self):
        """number of current outputs that differ from the saved ones in any word (host comparison, limb by limb); -1: nothing saved"""
        n = self.L.fbb_compare_saved(self.h)
        self._ok(0)
        return int(n)

    def check(self, i):
        vals = (C.c_double * 8)()
        return self.L.fbb_check(self.h, i, vals), list(vals)

    def dump(self, path, lo, hi):
        self._ok(self.L.fbb_dump(self.h, path.encode(), lo, hi))

    def close(self):
        if self.h:
            self.L.fbb_destroy(self.h)
            self.h = None


def run_rank(logN, slots, total, threads, reps, device, prng, dist=None, torch_device="cpu", budget=(4, 4), levels_after=5, so=HIP_SO,
             dump_path=None, warmup=1, key_threads=None, keep=None, eval_threads=None, force_replication=False):
    """One rank of the sharded batch.  dist: torch.distributed (initialised) or None for a single process.  Returns a dict of
    timings; with dump_path the rank's bootstrapped ciphertexts are written there (tests).  key_threads: the OpenMP team during
    set-up, encryption and key generation (pke draws from thread-local PRNGs there: equal teams give equal keys); eval_threads: the
    team of pke's inner loops during the bootstraps (default: unchanged).  keep: bootstrap only the first `keep` ciphertexts of the
    rank's slice (all `total` are still encrypted, so the kept ones are the batch's — the byte-comparison reference of bench.py).
    force_replication: run the key replication (export -> scatter + all-gather -> adoption of windows of the gathered tensor) even
    in a world of one rank — the multi-GPU code path on one GPU's memory (tests, FHE_BENCH_FORCE_DIST)."""
    from . import shard
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    t0 = time.perf_counter()
    bb = BootBatch(so, logN, slots, budget, levels_after, prng, device, omp_threads=key_threads)
    lo, hi = shard.shard_range(total, rank, world)
    if keep is not None:
        hi = min(hi, lo + keep)
    bb.encrypt(total, lo, hi - lo)  # (before any key generation: every rank draws the same randomness for the same ciphertext)
    t_setup = time.perf_counter() - t0
    res = {"rank": rank, "world": world, "ciphertexts": hi - lo, "setup_s": round(t_setup, 2)}
    t0 = time.perf_counter()
    if rank == 0:
        bb.keygen()
    res["keygen_s"] = round(time.perf_counter() - t0, 2)
    if world > 1 or (force_replication and dist is not None):
        import torch
        # rank 0 tells the others which keys exist, then the packed key words travel: scatter + all-gather over xGMI
        n = torch.zeros(1, dtype=torch.int64, device=torch_device)
        idx = bb.key_indices() if rank == 0 else None
        if rank == 0:
            n[0] = len(idx)
        dist.broadcast(n, src=0)
        it = torch.zeros(int(n[0]), dtype=torch.int64, device=torch_device)
        if rank == 0:
            it.copy_(torch.from_numpy(idx.astype(np.int64)))
        dist.broadcast(it, src=0)
        if rank != 0:
            bb.make_key_shells(it.cpu().numpy().astype(np.uint32))
        words, per = bb.key_layout()
        shape = (int(n[0]), per, words)
        packed = None
        if rank == 0:
            packed = torch.empty(shape, dtype=torch.int64, device=torch_device)
            bb.export_keys(packed.data_ptr())
        dist.barrier()
        t0 = time.perf_counter()
        keys = shard.allgather_words(packed, shape, torch_device, src=0)
        if str(torch_device).startswith("cuda"):
            torch.cuda.synchronize()
        dist.barrier()
        t_rep = time.perf_counter() - t0
        bb.adopt_keys(keys.data_ptr())
        res["keys"] = keys  # (keep the tensor alive: the key towers are windows of it)
        res["key_set_GB"] = round(keys.numel() * 8 / 1e9, 3)
        res["key_replication_s"] = round(t_rep, 3)
        res["key_replication_GBps"] = round(keys.numel() * 8 / 1e9 / max(t_rep, 1e-9), 1)
        dist.barrier()
    if eval_threads:
        bb.L.fbb_set_omp_threads(eval_threads)
    sec = bb.bootstrap_all(threads, reps, warmup)
    res["seconds_per_pass"] = sec
    res["bootstraps_per_s"] = (hi - lo) / sec if sec > 0 else 0.0
    worst = 0.0
    for i in range(hi - lo):
        worst = max(worst, bb.check(i)[0])
    res["max_abs_error"] = worst
    if dump_path:
        bb.dump(dump_path, 0, hi - lo)
    res["handle"] = bb
    return res
