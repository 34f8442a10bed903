#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X DCRTPoly backend.

Metric (BASELINE.json): "NTT GB/s vs HBM roofline + CKKS EvalMultKeySwitch/sec, N=2^16 L=30, 1/2/4/8 GPU".
One step = one pass of the hot path over one batch of synthetic input already resident in HBM:
    forward NTT + inverse NTT of a batch of B = 1024 polynomials, N = 2^16, L = 30 60-bit RNS limbs
    (BASELINE config 2: [1024][30][65536] uint64 = 16.1 GB per GPU, ILDCRTParams(2N, 30, 60) moduli).
value = algorithmic GB/s = (4 * 8 * N * L * B bytes per step, i.e. each transform reads and writes the tower
once) * n_gpus / step time.  Multi-GPU: the batch of independent polynomials is sharded, B per GPU (weak
scaling), no collective on the data path; the barrier + max-over-ranks timing uses torch.distributed (RCCL).

Extra fields on the same JSON line: `roofline` (dominant kernel, live hipEvent timing), `cpu_baseline`
(reference compiled from its own sources, or the oracle port, timed on this host) and `evalmult` (CKKS
EvalMult + HYBRID key switch throughput at config 3's shape: N=2^16, l=21, k=7, dnum=3).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from openfhe_amd import fhe_hip as fh  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--limbs", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1024, help="polynomials per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-evalmult", action="store_true")
    ap.add_argument("--evalmult-batch", type=int, default=256, help="ciphertexts per GPU in the EvalMult leg (BASELINE configs[2]: 256)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-hadamard", action="store_true", help="skip the Hadamard-product leg")
    ap.add_argument("--no-bfv", action="store_true", help="skip the BFV EvalMult (BEHZ) leg (BASELINE configs[4] shape)")
    ap.add_argument("--bfv-batch", type=int, default=64)
    ap.add_argument("--no-lt", action="store_true", help="skip the BSGS linear-transform leg (bootstrapping's inner loop)")
    ap.add_argument("--no-bootstrap", action="store_true",
                    help="skip the EvalBootstrap leg (BASELINE configs[3] shape through the reference's CryptoContext on the HIP backend of DCRTPoly)")
    ap.add_argument("--bootstrap-logn", type=int, default=17)
    ap.add_argument("--bootstrap-batch", type=int, default=64, help="ciphertexts per GPU in the bootstrap leg (BASELINE configs[3]: 512 over 8 GPUs = 64 per GPU; "
                                                                     "the driver's 1-GPU run is one rank's share)")
    ap.add_argument("--bootstrap-wide-threads", type=int, default=2,
                    help="host threads (= streams) the lockstep groups of the bootstrap leg are spread over (round 5: 16 x 2 49.2-51.5 in every "
                         "position of a sweep, 8 x 4 50.1, 16 x 4 37-40, 32 x 2 52 with warm caches but 13-17 cold: profiles/r05_sweeps.md section 3)")
    ap.add_argument("--bootstrap-group", type=int, default=16,
                    help="ciphertexts per wide (lockstep) evaluation in the bootstrap leg; 0 = the rank's whole slice (one group of 64 needs "
                         "BSGS workspaces of more than 100 GiB)")
    ap.add_argument("--bootstrap-threads", type=int, default=8, help="host threads (= HIP streams) the rank's ciphertexts are spread over")
    ap.add_argument("--no-cc-evalmult", action="store_true",
                    help="skip the leg that runs BASELINE configs[2]'s EvalMult through the reference's CryptoContext on the HIP backend")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparisons of the legs' results")
    ap.add_argument("--no-power", action="store_true", help="skip the package-power reading of the headline leg")
    ap.add_argument("--evalmult-logn", type=int, default=16, help="ring of the EvalMult leg (config 3: 16)")
    ap.add_argument("--evalmult-limbs", type=int, default=21, help="Q limbs of the EvalMult leg (config 3: 21)")
    return ap.parse_args()


def source_sha():
    """identity of the kernel sources the resident library was built from: PMC records carry it, so that a record made with
    other kernels is never quoted as this run's traffic"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "openfhe-development_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_oracle():
    """the parity CHECKER (oracle/libfhe_oracle.so): used after the timed regions only, never inside them"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import libs
        return libs.load_oracle()
    except Exception:
        return None


def sample_towers(batch):
    return sorted({0, batch // 2, batch - 1})


def download_tower(lib, ctx, dev, tw, n_limbs):
    out = np.empty((n_limbs, ctx.N), np.uint64)
    lib.check(lib.L.fhe_memcpy_d2h(ctx.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(dev.value + tw * out.nbytes), out.nbytes, None))
    ctx.sync()
    return out


def device_checksums(lib, ctx, dev, rows):
    """{sum, position-weighted sum} mod 2^64 of EVERY limb-row of a resident batch (fhe_checksum: one read of the batch) -> uint64[rows][2]"""
    d = ctx.malloc(rows * 16)
    lib.check(lib.L.fhe_checksum(ctx.h, dev, rows, d, None))
    out = np.empty((rows, 2), np.uint64)
    lib.check(lib.L.fhe_memcpy_d2h(ctx.h, out.ctypes.data_as(C.c_void_p), d, out.nbytes, None))
    ctx.sync()
    ctx.free(d)
    return out


def host_checksums(towers):
    """the same two words per limb-row of host towers [T][L][N] -> uint64[T][L][2]"""
    w = (2 * np.arange(towers.shape[2], dtype=np.uint64) + np.uint64(1))  # (uint64 arithmetic wraps modulo 2^64, like the kernel's)
    return np.stack([towers.sum(axis=2, dtype=np.uint64), (towers * w).sum(axis=2, dtype=np.uint64)], axis=2)


def all_towers_match(lib, ctx, dev, batch, n_limbs, want):
    """every tower t of the resident batch [batch][n_limbs][N] against want[t % len(want)] (host towers), by checksums of every
    limb-row; returns the index of the first differing tower or -1"""
    got = device_checksums(lib, ctx, dev, batch * n_limbs).reshape(batch, n_limbs, 2)
    ref = host_checksums(want)
    exp = ref[np.arange(batch) % ref.shape[0]]
    bad = np.flatnonzero((got != exp).any(axis=(1, 2)))
    return int(bad[0]) if len(bad) else -1


def host_seed_towers(q, N, batch, seed, seed_polys):
    """the host image fill_random_tower() replicated: tower t of the device batch is seed tower t % seed_polys"""
    rng = np.random.default_rng(seed)
    seed_polys = min(seed_polys, batch)
    host = np.empty((seed_polys, len(q), N), np.uint64)
    for i, qi in enumerate(q):
        host[:, i, :] = rng.integers(0, int(qi), size=(seed_polys, N), dtype=np.uint64)
    return host


def fill_random_tower(ctx, q, batch, seed, seed_polys=8):
    """[batch][L][N] uniform residues in HBM: `seed_polys` towers generated on the host, replicated on device."""
    rng = np.random.default_rng(seed)
    L, N = len(q), ctx.N
    seed_polys = min(seed_polys, batch)
    host = np.empty((seed_polys, L, N), np.uint64)
    for i, qi in enumerate(q):
        host[:, i, :] = rng.integers(0, int(qi), size=(seed_polys, N), dtype=np.uint64)
    dev = ctx.malloc(batch * L * N * 8)
    lib = ctx.lib
    lib.check(lib.L.fhe_memcpy_h2d(ctx.h, dev, host.ctypes.data_as(C.c_void_p), host.nbytes, None))
    ctx.sync()
    done = seed_polys
    stride = L * N * 8
    while done < batch:
        n = min(done, batch - done)
        lib.check(lib.L.fhe_memcpy_d2d(ctx.h, C.c_void_p(dev.value + done * stride), dev, n * stride, None))
        done += n
    ctx.sync()
    return dev


def cpu_baseline(q, psi, logN, L, seconds):
    """fwd+inv NTT of DCRTPoly-shaped towers on the host cores: the reference itself (oracle/_ref, OpenMP over
    limbs as DCRTPolyImpl::SwitchFormat does) when its build travelled with the repo, else the oracle port."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import libs
    N = 1 << logN
    polys = 4
    rng = np.random.default_rng(7)
    x = np.empty((polys, L, N), np.uint64)
    for i, qi in enumerate(q):
        x[:, i, :] = rng.integers(0, int(qi), size=(polys, N), dtype=np.uint64)
    cores = os.cpu_count() or 1
    if libs.have_ref():
        r = libs.load_ref()
        h = r.ref_towers_create(N, L, q, psi, x, polys, 0)
        r.ref_towers_switch_format(h)  # warm the twiddle cache
        r.ref_towers_switch_format(h)
        # DCRTPolyImpl::SwitchFormat parallelises over the L limbs only (dcrtpoly-impl.h:1932-1940), so a team far
        # larger than L just adds fork/join cost: give the reference its best team size instead of all host threads
        gomp = C.CDLL("libgomp.so.1")
        best, best_t = None, None
        for nt in sorted({cores, 128, 64, 32, 16, 8}):
            if nt > cores:
                continue
            gomp.omp_set_num_threads(nt)
            r.ref_towers_switch_format(h)  # let the new team start up
            r.ref_towers_switch_format(h)
            dt1 = None
            for _ in range(3):  # best of three: a probe disturbed by another process must not pick a bad team
                t1 = time.time()
                r.ref_towers_switch_format(h)
                r.ref_towers_switch_format(h)
                d = time.time() - t1
                dt1 = d if dt1 is None else min(dt1, d)
            if best is None or dt1 < best_t:
                best, best_t = nt, dt1
        gomp.omp_set_num_threads(best)
        t0 = time.time()
        reps = 0
        while time.time() - t0 < seconds:
            r.ref_towers_switch_format(h)  # COEFF -> EVAL
            r.ref_towers_switch_format(h)  # EVAL -> COEFF
            reps += 1
        dt = time.time() - t0
        threads = best
        r.ref_towers_destroy(h)
        kind = "reference"
    else:
        o = libs.load_oracle()
        octx = o.orc_ctx_create(N, L, q, psi)
        o.orc_ntt_fwd_tower(octx, x, None, L, polys, 0)
        o.orc_ntt_inv_tower(octx, x, None, L, polys, 0)
        t0 = time.time()
        reps = 0
        while time.time() - t0 < seconds:
            o.orc_ntt_fwd_tower(octx, x, None, L, polys, 0)
            o.orc_ntt_inv_tower(octx, x, None, L, polys, 0)
            reps += 1
        dt = time.time() - t0
        threads = cores
        o.orc_ctx_destroy(octx)
        kind = "port"
    bytes_done = 4.0 * 8 * N * L * polys * reps
    return {"value": round(bytes_done / dt / 1e9, 3), "unit": "GB/s", "cores": int(threads), "kind": kind,
            "sample": f"{reps} x fwd+inv NTT of {polys} towers (N=2^{logN}, L={L}) in {dt:.1f} s; "
                      f"{dt / (reps * polys) * 1e3:.1f} ms per tower fwd+inv; host has {cores} logical cores"}


def timed_sequence(lib, ctx, st, call):
    """The composite legs are 40-60 kernel launches per step: record them once into a HIP graph (fhe_graph_begin/end) and
    replay it, so that the figure measures the GPU and not the launching thread.  Falls back to direct calls if capture
    is unavailable.  Returns (step function, description, graph handle or None)."""
    call()  # first call builds the per-level tables (not capturable)
    lib.check(lib.L.fhe_stream_sync(ctx.h, st))
    g = C.c_void_p()
    try:
        lib.check(lib.L.fhe_graph_begin(ctx.h, st))
        call()
        lib.check(lib.L.fhe_graph_end(ctx.h, st, C.byref(g)))
        return (lambda: lib.check(lib.L.fhe_graph_launch(ctx.h, g, st))), "one HIP graph launch per step", g
    except Exception as e:
        return call, f"direct calls ({e})", None


def evalmult_leg(lib, device, logN, batch, steps, warmup, sync, dist=None, sizeQ=21, parity=True, tdev=None):
    """CKKS EvalMult + HYBRID key switch at config 3's shape (depth 20: l=21 limbs, dnum=3 => alpha=7, k=7)."""
    dnum = 3
    q, psiQ = lib.ckks_like_chain(logN, sizeQ, 60, 59)
    p, psiP = lib.select_p(logN, q, dnum, 60)
    allq = np.concatenate([q, p])
    ctx = fh.Context(lib, logN, allq, np.concatenate([psiQ, psiP]), device=device)
    plan = fh.KeySwitchPlan(ctx, sizeQ, len(p), dnum)
    N = ctx.N
    rng = np.random.default_rng(3)

    def host_key():  # evaluation key half: uniform residues ([dnum][l+k][N]), generated per limb on the host
        host = np.empty((dnum, len(allq), N), np.uint64)
        for i, qi in enumerate(allq):
            host[:, i, :] = rng.integers(0, int(qi), size=(dnum, N), dtype=np.uint64)
        return host
    key_dist, keep = "local (single process)", None
    hostKeys = None  # (keyB, keyA) images when this rank generated the key itself (needed by the parity check)
    if dist is not None:
        # the evaluation key exists on rank 0 and reaches every GPU with ONE broadcast over RCCL/xGMI; it is
        # the only collective of the whole path
        try:
            import torch
            from openfhe_amd import shard
            kb = ka = None
            if dist.get_rank() == 0:
                kb, ka = host_key(), host_key()
            hostKeys = (kb, ka) if kb is not None else None
            keep = shard.broadcast_key(plan, kb, ka, tdev if tdev is not None else torch.device("cuda", device))
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            key_dist = (f"{'rccl' if dist.get_backend() == 'nccl' else dist.get_backend()} broadcast from rank 0 to "
                        f"{dist.get_world_size()} rank(s), {2 * plan.key_words() * 8 / 2**20:.1f} MiB")
        except Exception as e:  # keep the benchmark alive; the line says what happened
            key_dist, keep = f"local (broadcast failed: {type(e).__name__}: {e})", None
            plan.key = None
    if plan.key is None:
        lib.check(lib.L.fhe_ks_key_alloc(plan.h, C.byref(key := C.c_void_p())))
        plan.key = key
        assert lib.L.fhe_ks_key_words(key) == dnum * len(allq) * N
        hk = []
        for which in (0, 1):
            host = host_key()
            hk.append(host)
            lib.check(lib.L.fhe_memcpy_h2d(ctx.h, lib.L.fhe_ks_key_devptr(key, which), host.ctypes.data_as(C.c_void_p),
                                           host.nbytes, None))
            ctx.sync()
        hostKeys = tuple(hk)
    key = plan.key
    ops = [fh.Tower(ctx, fill_random_tower(ctx, q, batch, 100 + i, seed_polys=2), batch, sizeQ) for i in range(4)]
    c0, c1 = ops[0].like(), ops[0].like()
    ws, wsb = plan.workspace(sizeQ, batch)

    st = C.c_void_p()
    lib.check(lib.L.fhe_stream_create(ctx.h, C.byref(st)))

    def call():
        lib.check(lib.L.fhe_ckks_eval_mult(plan.h, key, ops[0].ptr, ops[1].ptr, ops[2].ptr, ops[3].ptr, sizeQ, batch,
                                           c0.ptr, c1.ptr, ws, wsb, st))
    step, mode, graph = timed_sequence(lib, ctx, st, call)
    for _ in range(warmup):
        step()
    lib.check(lib.L.fhe_stream_sync(ctx.h, st))
    psamp = PowerSampler(device) if dist is None or dist.get_rank() == 0 else None
    if psamp is not None:
        psamp.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    lib.check(lib.L.fhe_stream_sync(ctx.h, st))
    dt = (time.perf_counter() - t0) / steps
    em_power = psamp.stop() if psamp is not None else None
    # ---- parity of the TIMED launch's result: towers {0, B/2, B-1} of both output elements against the oracle on the same
    # inputs (tower t of every operand is seed tower t % 2; the key is this leg's host image)
    par = "skipped"
    o = load_oracle() if parity else None
    if parity and o is None:
        par = "oracle library not available"
    elif parity and hostKeys is None:
        par = "skipped on this rank (the key was generated on rank 0)"
    elif parity:
        seeds = [host_seed_towers(q, N, batch, 100 + i, 2) for i in range(4)]
        hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, len(p), p, psiP, dnum)
        nseed = seeds[0].shape[0]
        w0, w1 = np.empty((nseed, sizeQ, N), np.uint64), np.empty((nseed, sizeQ, N), np.uint64)
        for sp in range(nseed):
            o.orc_ckks_eval_mult_relin(hy, seeds[0][sp], seeds[1][sp], seeds[2][sp], seeds[3][sp], sizeQ, hostKeys[0], hostKeys[1], w0[sp], w1[sp])
        b0, b1 = all_towers_match(lib, ctx, c0.ptr, batch, sizeQ, w0), all_towers_match(lib, ctx, c1.ptr, batch, sizeQ, w1)
        par = (f"bit-exact vs oracle on ALL {batch} ciphertexts, both elements (orc_ckks_eval_mult_relin; checksums of every limb-row, "
               "ciphertext 0 word for word)")
        if b0 >= 0 or b1 >= 0 or not (np.array_equal(download_tower(lib, ctx, c0.ptr, 0, sizeQ), w0[0]) and
                                      np.array_equal(download_tower(lib, ctx, c1.ptr, 0, sizeQ), w1[0])):
            par = f"MISMATCH vs oracle at ciphertext {max(b0, b1, 0)}"
        o.orc_hybrid_destroy(hy)
    lib.L.fhe_graph_destroy(graph)
    lib.check(lib.L.fhe_stream_destroy(ctx.h, st))
    plan.close()
    ctx.close()
    del keep
    # algorithmic HBM bytes of one EvalMult + key switch: every limb-NTT reads and writes its limb once
    # (l INTT + beta*(l+k-alpha') forward in ModUp + 2k INTT + 2l forward in the two ModDowns), the tensor product moves 7l
    # limbs, a ModUp conversion reads a digit and writes its complement, the inner product reads beta*(l+k) digit limbs and
    # writes 2(l+k) (the key is shared by the batch), a ModDown conversion reads k and writes l limbs per element and its
    # epilogue reads the extended element and the accumulator and writes the result (3l per element)
    k_, l_ = len(p), sizeQ
    alpha = -(-l_ // dnum)
    beta = -(-l_ // alpha)
    compl = sum(l_ + k_ - min(alpha, l_ - alpha * j) for j in range(beta))
    ntt_limbs = l_ + compl + 2 * k_ + 2 * l_
    limb_moves = 2 * ntt_limbs + 7 * l_ + (l_ + compl) + (beta * (l_ + k_) + 2 * (l_ + k_)) + 2 * (k_ + l_) + 2 * 3 * l_
    alg = 8.0 * N * limb_moves
    ach = alg * batch / dt / 1e9
    # SURVEY.md 8(d)'s own per-unit figure counts the NTT traffic and the tensor product only (~0.25 GB per op at config 3)
    alg_survey = 8.0 * N * (2 * ntt_limbs + 9 * l_)
    ach_survey = alg_survey * batch / dt / 1e9
    return {"ops_per_s_per_gpu": round(batch / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "batch": batch,
            "shape": f"N=2^{logN}, l={sizeQ}, k={len(p)}, dnum={dnum}, workspace {wsb / 2**30:.1f} GiB",
            "eval_key": key_dist, "launch": mode, "parity": par, "power": em_power,
            # frac is quoted on SURVEY.md 8(d)'s per-unit figure (the contract's byte count); the itemised count of every stage's operands
            # (what the operation sequence has to move with each limb-NTT in two passes) is reported beside it as moved_*
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_op": alg_survey, "limb_ntts_per_op": ntt_limbs,
                         "achieved": round(ach_survey, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach_survey / HBM_PEAK_GBPS, 4),
                         "itemised_bytes_per_op": alg, "moved_achieved": round(ach, 1), "moved_frac": round(ach / HBM_PEAK_GBPS, 4),
                         "byte_counts": "algorithmic_bytes_per_op = SURVEY.md 8(d)'s figure (limb-NTT traffic + the tensor product's 9 limb moves); "
                                        "itemised_bytes_per_op counts every stage's operands once (DESIGN.md §7): moved_* is quoted on it",
                         "dominant_kernel": "ntt_static_kernel (57 % over its pass kernels; then switch_basis_kernel 16 %, "
                                            "ks_inner_multi_kernel 15 %, tensor_kernel 11 %): shares in "
                                            "profiles/r05_rocprof_kernel_stats_evalmult256.csv"}}


def linear_transform_leg(lib, device, batches, steps, warmup, with_cpu, parity=True):
    """FHECKKSRNS::EvalLinearTransform (BSGS with double hoisting, the linear-transform loop of CKKS bootstrapping, BASELINE
    configs[3]'s inner loop) at config 4's ring: N=2^17, l=21 limbs, dnum=3, 64 diagonals as 8 baby x 8 giant steps.
    One HIP graph replay per transform; the CPU figure is the reference's own EvalLinearTransform (oracle/_ref)."""
    logN, sizeQ, dnum, slots, bStep = 17, 21, 3, 64, 8
    gStep = slots // bStep
    q, psiQ = lib.ckks_like_chain(logN, sizeQ, 60, 59)
    p, psiP = lib.select_p(logN, q, dnum, 60)
    allq = np.concatenate([q, p])
    ctx = fh.Context(lib, logN, np.concatenate([q, p]), np.concatenate([psiQ, psiP]), device=device)
    plan = fh.KeySwitchPlan(ctx, sizeQ, len(p), dnum)
    N = ctx.N
    rng = np.random.default_rng(9)

    def rows(mods, lead):
        host = np.empty((lead, len(mods), N), np.uint64)
        for i, m in enumerate(mods):
            host[:, i, :] = rng.integers(0, int(m), size=(lead, N), dtype=np.uint64)
        return host
    # distinct device buffers for every key and diagonal (so that nothing is served from the Infinity Cache by accident);
    # their content is uniform random either way, two host images are enough
    kimg = [rows(allq, dnum) for _ in range(2)]
    keys, keyImg = {}, {}
    for n, index in enumerate(list(range(1, bStep)) + [bStep * j for j in range(1, gStep)]):
        keys[index] = (lib.find_automorphism_index(index, 2 * N), plan.make_key(kimg[n % 2], kimg[(n + 1) % 2]))
        keyImg[index] = (kimg[n % 2], kimg[(n + 1) % 2])
    dimg = [rows(allq, 1)[0] for _ in range(2)]
    A = [ctx.upload(dimg[i % 2]) for i in range(slots)]
    in_rot = [None] + [keys[i] for i in range(1, bStep)]
    out_rot = [None] + [keys[bStep * j] for j in range(1, gStep)]
    diag = [[A[bStep * j + i] for i in range(bStep)] for j in range(gStep)]
    res = {"shape": f"N=2^{logN}, l={sizeQ}, k={len(p)}, dnum={dnum}, {slots} diagonals = {bStep} baby x {gStep} giant steps, "
                    f"{bStep - 1}+{gStep - 1} rotation keys", "per_batch": {}}
    st = C.c_void_p()
    lib.check(lib.L.fhe_stream_create(ctx.h, C.byref(st)))
    for batch in batches:
        c0 = fh.Tower(ctx, fill_random_tower(ctx, q, batch, 400, seed_polys=2), batch, sizeQ)
        c1 = fh.Tower(ctx, fill_random_tower(ctx, q, batch, 401, seed_polys=2), batch, sizeQ)
        out = (c0.like(), c0.like())
        wsb = lib.L.fhe_ckks_bsgs_workspace_bytes(plan.h, sizeQ, batch, bStep, gStep)
        ws = (ctx.malloc(wsb), wsb)

        def call():
            plan.BsgsTransform(c0, c1, in_rot, out_rot, diag, ws=ws, out=out, stream=st)
        step, mode, graph = timed_sequence(lib, ctx, st, call)
        t0 = time.perf_counter()
        n = 0
        while n < warmup or time.perf_counter() - t0 < 0.4:  # a step is a few ms: warm up by time, so that the clocks have ramped
            step()
            lib.check(lib.L.fhe_stream_sync(ctx.h, st))
            n += 1
        reps = max(steps, int(0.5 / max((time.perf_counter() - t0) / n, 1e-4)))
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        lib.check(lib.L.fhe_stream_sync(ctx.h, st))
        dt = (time.perf_counter() - t0) / reps
        lib.L.fhe_graph_destroy(graph)
        res["per_batch"][str(batch)] = {"transforms_per_s_per_gpu": round(batch / dt, 2), "ms_per_batch": round(dt * 1e3, 3),
                                        "workspace_GiB": round(wsb / 2**30, 2), "launch": mode}
        if parity and batch == batches[0]:  # the timed replay's result against the oracle's sequential EvalLinearTransform
            o = load_oracle()
            if o is None:
                res["parity"] = "oracle library not available"
            else:
                import libs
                hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, len(p), p, psiP, dnum)
                s0, s1 = host_seed_towers(q, N, batch, 400, 2), host_seed_towers(q, N, batch, 401, 2)
                inI = [None] + list(range(1, bStep))
                outI = [None] + [bStep * j for j in range(1, gStep)]
                inK = np.array([0 if i is None else keys[i][0] for i in inI], np.uint32)
                outK = np.array([0 if i is None else keys[i][0] for i in outI], np.uint32)
                pa = libs.ptr_array
                flat = [dimg[(bStep * j + i) % 2] for j in range(gStep) for i in range(bStep)]
                w0, w1 = np.zeros((sizeQ, N), np.uint64), np.zeros((sizeQ, N), np.uint64)
                o.orc_ckks_bsgs_transform(hy, s0[0], s1[0], sizeQ, bStep, inK, pa([None if i is None else keyImg[i][0] for i in inI]),
                                          pa([None if i is None else keyImg[i][1] for i in inI]), gStep, outK,
                                          pa([None if i is None else keyImg[i][0] for i in outI]),
                                          pa([None if i is None else keyImg[i][1] for i in outI]), pa(flat), w0, w1)
                ok = (np.array_equal(download_tower(lib, ctx, out[0].ptr, 0, sizeQ), w0) and
                      np.array_equal(download_tower(lib, ctx, out[1].ptr, 0, sizeQ), w1))
                res["parity"] = ("bit-exact vs oracle (orc_ckks_bsgs_transform, both elements, batch %d)" % batch) if ok else "MISMATCH vs oracle"
                o.orc_hybrid_destroy(hy)
        for t in (c0, c1) + out:
            ctx.free(t.ptr)
        ctx.free(ws[0])
    lib.check(lib.L.fhe_stream_destroy(ctx.h, st))
    plan.close()
    ctx.close()
    res["transforms_per_s_per_gpu"] = max(v["transforms_per_s_per_gpu"] for v in res["per_batch"].values())
    res["cpu_baseline"] = None
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import libs
        if libs.have_ref():
            r = libs.load_ref()
            gomp = C.CDLL("libgomp.so.1")
            cores = os.cpu_count() or 1
            h = r.ref_ckks_create(1 << logN, sizeQ - 1, 59, 60, dnum, 0)
            idx = np.array(sorted(keys), np.int32)
            r.ref_ckks_rotate_keygen(h, idx, len(idx))
            vals, M = np.zeros((slots, 2)), np.zeros((slots, slots, 2))
            vals[:, 0], M[:, :, 0] = rng.uniform(-1, 1, slots), rng.uniform(-1, 1, (slots, slots))
            ct = r.ref_ckks_encrypt_slots(h, vals, 0, slots)
            lt = r.ref_ckks_lt_create(h, slots, bStep, M, 0)
            best, best_t = None, None
            for nt in sorted({cores, 64, 32, 16, 8}):  # give the reference its best OpenMP team size
                if nt > cores:
                    continue
                gomp.omp_set_num_threads(nt)
                sec1 = r.ref_ckks_time_linear_transform(h, lt, ct, 1)
                if best is None or sec1 < best_t:
                    best, best_t = nt, sec1
            gomp.omp_set_num_threads(best)
            reps = max(2, min(10, int(6.0 / best_t)))
            sec = r.ref_ckks_time_linear_transform(h, lt, ct, reps)
            res["cpu_baseline"] = {"value": round(1.0 / sec, 3), "unit": "transforms/s", "cores": int(best), "kind": "reference",
                                   "sample": f"{reps} x FHECKKSRNS::EvalLinearTransform, N=2^{logN}, {sizeQ} Q limbs, {slots} diagonals, "
                                             f"bStep={bStep}; {sec * 1e3:.0f} ms each; best OpenMP team of {{8..{cores}}}"}
            r.ref_ckks_lt_destroy(lt)
            r.ref_ckks_destroy(h)
    return res


def bfv_leg(lib, device, batch, steps, warmup, sync, with_cpu, parity=True):
    """BFV EvalMult (BEHZ, no relinearisation) at BASELINE configs[4]'s shape: N=2^15, Q = 7 x 60-bit limbs and
    Bsk = 8 limbs (log2(Q*Bsk) ~ 900).  Product: fhe_bfv_eval_mult_behz on `batch` ciphertext pairs; CPU: the reference's
    own cc->EvalMultNoRelin (oracle/_ref) on one pair when its build travelled with the repo."""
    logN, numQ, t = 15, 7, 65537
    q, psiQ = lib.ckks_like_chain(logN, numQ, 60, 60)
    bsk, psiB = lib.behz_bsk(logN, q, t)
    ctx = fh.Context(lib, logN, np.concatenate([q, bsk]), np.concatenate([psiQ, psiB]), device=device)
    plan = fh.Behz(ctx, np.arange(numQ), np.arange(numQ, numQ + len(bsk)), t)
    ops = [fh.Tower(ctx, fill_random_tower(ctx, q, batch, 200 + i, seed_polys=2), batch, numQ) for i in range(4)]
    d = [ops[0].like() for _ in range(3)]
    wsb = lib.L.fhe_bfv_eval_mult_behz_workspace_bytes(plan.h, batch)
    ws = ctx.malloc(wsb)

    st = C.c_void_p()
    lib.check(lib.L.fhe_stream_create(ctx.h, C.byref(st)))

    def call():
        lib.check(lib.L.fhe_bfv_eval_mult_behz(plan.h, ops[0].ptr, ops[1].ptr, ops[2].ptr, ops[3].ptr, d[0].ptr,
                                               d[1].ptr, d[2].ptr, 0, batch, ws, wsb, st))
    step, mode, graph = timed_sequence(lib, ctx, st, call)
    for _ in range(warmup):
        step()
    lib.check(lib.L.fhe_stream_sync(ctx.h, st))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    lib.check(lib.L.fhe_stream_sync(ctx.h, st))
    dt = (time.perf_counter() - t0) / steps
    par = "skipped"
    o = load_oracle() if parity else None
    hb = None
    if parity and o is None:
        par = "oracle library not available"
    elif parity:
        N = ctx.N
        hb = o.orc_behz_create(N, numQ, q, t)
        call = o.orc_ctx_create(N, numQ + len(bsk), np.concatenate([q, bsk]), np.concatenate([psiQ, psiB]))
        seeds = [host_seed_towers(q, N, batch, 200 + i, 2) for i in range(4)]
        par = "bit-exact vs oracle on towers " + str(sample_towers(batch)) + " of the three product elements (orc_bfv_eval_mult_behz)"
        wantNR = {}
        for tw in sample_towers(batch):
            sp = tw % 2
            if sp not in wantNR:
                w = [np.zeros((numQ, N), np.uint64) for _ in range(3)]
                o.orc_bfv_eval_mult_behz(hb, call, seeds[0][sp], seeds[1][sp], seeds[2][sp], seeds[3][sp], w[0], w[1], w[2])
                wantNR[sp] = w
            if not all(np.array_equal(download_tower(lib, ctx, d[k].ptr, tw, numQ), wantNR[sp][k]) for k in range(3)):
                par = f"MISMATCH vs oracle at tower {tw}"
                break
        o.orc_ctx_destroy(call)
    lib.L.fhe_graph_destroy(graph)
    lib.check(lib.L.fhe_stream_destroy(ctx.h, st))
    ctx.free(ws)
    plan.close()
    ctx.close()
    # ... and with relinearisation (cc->EvalMult): context = Q, P (HYBRID, dnum 3), Bsk; uniform random evaluation key
    dnum = 3
    p, psiP = lib.select_p(logN, q, dnum, 60)
    ctx = fh.Context(lib, logN, np.concatenate([q, p, bsk]), np.concatenate([psiQ, psiP, psiB]), device=device)
    ks = fh.KeySwitchPlan(ctx, numQ, len(p), dnum)
    rng = np.random.default_rng(5)
    allqp = np.concatenate([q, p])

    def host_key():
        host = np.empty((dnum, len(allqp), ctx.N), np.uint64)
        for i, qi in enumerate(allqp):
            host[:, i, :] = rng.integers(0, int(qi), size=(dnum, ctx.N), dtype=np.uint64)
        return host
    hkB, hkA = host_key(), host_key()
    ks.upload_key(hkB, hkA)
    plan = fh.Behz(ctx, np.arange(numQ), np.arange(numQ + len(p), numQ + len(p) + len(bsk)), t)
    ops = [fh.Tower(ctx, fill_random_tower(ctx, q, batch, 300 + i, seed_polys=2), batch, numQ) for i in range(4)]
    c0, c1 = ops[0].like(), ops[0].like()
    wsb = lib.L.fhe_bfv_eval_mult_relin_workspace_bytes(plan.h, ks.h, batch)
    ws = ctx.malloc(wsb)
    st = C.c_void_p()
    lib.check(lib.L.fhe_stream_create(ctx.h, C.byref(st)))

    def call2():
        lib.check(lib.L.fhe_bfv_eval_mult_relin_behz(plan.h, ks.h, ks.key, ops[0].ptr, ops[1].ptr, ops[2].ptr, ops[3].ptr,
                                                     c0.ptr, c1.ptr, batch, ws, wsb, st))
    step2, mode2, graph2 = timed_sequence(lib, ctx, st, call2)
    for _ in range(warmup):
        step2()
    lib.check(lib.L.fhe_stream_sync(ctx.h, st))
    t0 = time.perf_counter()
    for _ in range(steps):
        step2()
    lib.check(lib.L.fhe_stream_sync(ctx.h, st))
    dt2 = (time.perf_counter() - t0) / steps
    par2 = "skipped"
    if parity and o is not None:
        # cc->EvalMult = EvalMultNoRelin, SetFormat(EVALUATION), KeySwitchCore on the third element, adds (base-leveledshe.cpp:201-214)
        N = ctx.N
        hy = o.orc_hybrid_create(N, numQ, q, psiQ, len(p), p, psiP, dnum)
        call = o.orc_ctx_create(N, numQ + len(bsk), np.concatenate([q, bsk]), np.concatenate([psiQ, psiB]))
        octxQ = o.orc_ctx_create(N, numQ, q, psiQ)
        seeds = [host_seed_towers(q, N, batch, 300 + i, 2) for i in range(4)]
        par2 = "bit-exact vs oracle on towers " + str(sample_towers(batch)) + " of both elements (BEHZ product + orc_hybrid_key_switch)"
        wantR = {}
        for tw in sample_towers(batch):
            sp = tw % 2
            if sp not in wantR:
                w = [np.zeros((numQ, N), np.uint64) for _ in range(3)]
                o.orc_bfv_eval_mult_behz(hb, call, seeds[0][sp], seeds[1][sp], seeds[2][sp], seeds[3][sp], w[0], w[1], w[2])
                for k in range(3):
                    o.orc_ntt_fwd_tower(octxQ, w[k], None, numQ, 1, 0)
                k0, k1 = np.empty((numQ, N), np.uint64), np.empty((numQ, N), np.uint64)
                o.orc_hybrid_key_switch(hy, w[2], numQ, hkB, hkA, k0, k1)
                for l in range(numQ):
                    o.orc_vec_add(w[0][l], w[0][l], k0[l], N, q[l])
                    o.orc_vec_add(w[1][l], w[1][l], k1[l], N, q[l])
                wantR[sp] = w
            if not (np.array_equal(download_tower(lib, ctx, c0.ptr, tw, numQ), wantR[sp][0]) and
                    np.array_equal(download_tower(lib, ctx, c1.ptr, tw, numQ), wantR[sp][1])):
                par2 = f"MISMATCH vs oracle at tower {tw}"
                break
        for h in (call, octxQ):
            o.orc_ctx_destroy(h)
        o.orc_hybrid_destroy(hy)
    if hb is not None:
        o.orc_behz_destroy(hb)
    lib.L.fhe_graph_destroy(graph2)
    lib.check(lib.L.fhe_stream_destroy(ctx.h, st))
    ctx.free(ws)
    plan.close()
    ks.close()
    ctx.close()
    relin = {"ops_per_s_per_gpu": round(batch / dt2, 1), "ms_per_batch": round(dt2 * 1e3, 3),
             "shape": f"+ HYBRID relinearisation, dnum={dnum}, {len(p)} P limbs", "launch": mode2, "parity": par2,
             "cpu_baseline": None}
    cpu = None
    if with_cpu:  # after the GPU leg: the reference's OpenMP team keeps spinning for a while and slows kernel launches
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import libs
        if libs.have_ref():
            r = libs.load_ref()
            for depth in range(2, 12):  # the reference sizes Q from the depth: first context with >= numQ limbs
                h = r.ref_bfv_create(1 << logN, t, depth, 60, 0)
                info = np.zeros(3, np.uint32)
                r.ref_bfv_info(h, info)
                if int(info[1]) >= numQ:
                    break
                r.ref_bfv_destroy(h)
            r.ref_bfv_keygen(h)
            ca, cb = r.ref_bfv_encrypt(h, 1), r.ref_bfv_encrypt(h, 2)
            gomp = C.CDLL("libgomp.so.1")
            cores = os.cpu_count() or 1
            best, best_t = None, None
            for nt in sorted({cores, 128, 64, 32, 16, 8}):  # give the reference its best OpenMP team size
                if nt > cores:
                    continue
                gomp.omp_set_num_threads(nt)
                r.ref_bfv_time_eval_mult_no_relin(h, ca, cb, 1)
                sec1 = r.ref_bfv_time_eval_mult_no_relin(h, ca, cb, 2)
                if best is None or sec1 < best_t:
                    best, best_t = nt, sec1
            gomp.omp_set_num_threads(best)
            reps = max(3, min(50, int(3.0 / best_t)))
            sec = r.ref_bfv_time_eval_mult_no_relin(h, ca, cb, reps)
            cpu = {"value": round(1.0 / sec, 2), "unit": "EvalMult/s", "cores": int(best), "kind": "reference",
                   "sample": f"{reps} x cc->EvalMultNoRelin (BEHZ), N=2^{logN}, {int(info[1])} Q limbs (depth {depth}); "
                             f"{sec * 1e3:.1f} ms each; best OpenMP team of {{8..{cores}}}; host has {cores} logical cores"}
            r.ref_bfv_destroy(h)
            # the same with relinearisation: cc->EvalMult on a HYBRID context of the same depth
            h2 = r.ref_bfv_create_hybrid(1 << logN, t, depth, 60, dnum)
            ca, cb = r.ref_bfv_encrypt(h2, 1), r.ref_bfv_encrypt(h2, 2)
            r.ref_bfv_time_eval_mult(h2, ca, cb, 1)
            reps2 = max(3, min(30, int(3.0 / max(best_t, 1e-3))))
            sec2 = r.ref_bfv_time_eval_mult(h2, ca, cb, reps2)
            relin["cpu_baseline"] = {"value": round(1.0 / sec2, 2), "unit": "EvalMult/s", "cores": int(best), "kind": "reference",
                                     "sample": f"{reps2} x cc->EvalMult (BEHZ + HYBRID relinearisation), N=2^{logN}; {sec2 * 1e3:.1f} ms each"}
            r.ref_bfv_destroy(h2)
    return {"with_relinearisation": relin, "ops_per_s_per_gpu": round(batch / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "batch": batch,
            "shape": f"N=2^{logN}, {numQ} Q limbs + {len(bsk)} Bsk limbs, t={t}, no relinearisation", "launch": mode,
            "parity": par, "cpu_baseline": cpu}


class PowerSampler:
    """Package power and shader clock of the device while a leg runs (the headline leg runs at the package power cap: DESIGN.md 4.12).
    Reads the amdgpu hwmon files when the box exposes them (a read costs microseconds, sampled from a thread every 10 ms); otherwise
    `sample_rocm_smi()` is called once by the caller inside a longer window."""

    def __init__(self, device=0):
        import glob
        self.files = None
        # the hwmon directory of THIS HIP device (a host exposes every GPU of the node in sysfs): by PCI bus id
        hw = []
        try:
            hip = C.CDLL("libamdhip64.so")
            buf = C.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) == 0:
                hw = sorted(glob.glob(f"/sys/bus/pci/devices/{buf.value.decode().lower()}/hwmon/hwmon*"))
        except Exception:
            hw = []
        for h in hw:
            pw = next((os.path.join(h, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, n))), None)
            if pw:
                cap, fq = os.path.join(h, "power1_cap"), os.path.join(h, "freq1_input")
                self.files = (pw, cap if os.path.exists(cap) else None, fq if os.path.exists(fq) else None)
                break
        self.samples, self._stop, self._thr = [], False, None

    @staticmethod
    def _read(path):
        try:
            return float(open(path).read().strip())
        except Exception:
            return None

    def start(self):
        if not self.files:
            return
        import threading

        def loop():
            while not self._stop:
                w = self._read(self.files[0])
                f = self._read(self.files[2]) if self.files[2] else None
                if w is not None:
                    self.samples.append((w / 1e6, f / 1e6 if f else None))
                time.sleep(0.01)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        self._stop = True
        if self._thr is not None:
            self._thr.join()
        if not self.samples:
            return None
        ws = [w for w, _ in self.samples]
        fs = [f for _, f in self.samples if f]
        cap = self._read(self.files[1]) if self.files[1] else None
        return {"source": f"amdgpu hwmon of this device ({self.files[0]}), sampled every 10 ms over the timed steps", "samples": len(ws),
                "package_W_mean": round(sum(ws) / len(ws), 1), "package_W_max": round(max(ws), 1),
                "sclk_MHz_mean": round(sum(fs) / len(fs)) if fs else None, "package_cap_W": round(cap / 1e6, 1) if cap else None}

    @staticmethod
    def sample_rocm_smi():
        import re
        import subprocess
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=20).stdout
        except Exception as e:
            return {"error": f"{type(e).__name__}: {e}"}
        w = re.search(r"Current Socket Graphics Package Power \(W\): ([0-9.]+)", t)
        c = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", t)
        f = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", t)
        return {"source": "one rocm-smi reading in the middle of a window of extra (untimed) steps of the same leg",
                "package_W": float(w.group(1)) if w else None, "package_cap_W": float(c.group(1)) if c else None,
                "sclk_MHz": int(f.group(1)) if f else None}


def shutil_which(name):
    import shutil
    return shutil.which(name)


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


STOCK_BOOT_SO = os.path.join(ROOT, "tests", "hal", "_build", "libfhe_boot_batch_stock.so")  # TEST-ONLY: the same source on the stock backend


def bootstrap_batch_leg(logN, per_gpu, threads, rank, world, device, dist, tdev, with_cpu, libpath, group=0, wide_threads=1):
    """BASELINE configs[3] as north_star states it: a BATCH of ciphertexts bootstrapped through the reference's own API
    (cc->EvalBootstrap at N = 2^17, 2^16 slots, {4,4}, SPARSE_TERNARY, FLEXIBLEAUTO: benchmark/src/ckks-bootstrapping.cpp:70) on the HIP backend
    of DCRTPoly, `per_gpu` ciphertexts per rank spread over `threads` host threads (one HIP stream each); with several ranks the
    relinearisation + rotation key set is generated on rank 0 only and replicated by scatter + all-gather over RCCL/xGMI into key
    objects that never had words (openfhe-development_amd/boot_batch.py, hal/bootstrap_batch.cpp).  Parity, computed in this run: every
    output of the lockstep (wide) pass against the threaded (narrow) pass's word for word; at N = 1 the narrow pass's ciphertext 0
    against the same batch on the stock backend (same PRNG, same OpenMP team), byte for byte; every output decrypted and checked."""
    import subprocess
    import tempfile
    from openfhe_amd import boot_batch as bb
    prng = os.path.join(ROOT, "tests", "hal", "_build", "libdetprng.so")
    if not (os.path.exists(bb.HIP_SO) and os.path.exists(prng)):
        return {"skipped": "openfhe-development_amd/hal/_build/libfhe_boot_batch_hip.so not built (./build.sh hal needs the reference sources)"}
    os.environ["FHE_HIP_LIB"] = libpath
    os.environ["FHE_HAL_REQUIRE_DEVICE"] = "1"
    # host threads: a node's cores are shared by its ranks — OpenMP teams and stream threads of a rank stay within cores / world
    team_cap = max(1, (os.cpu_count() or 1) // max(1, world))
    threads = max(1, min(threads, team_cap))
    slots, total, key_threads = 1 << (logN - 1), per_gpu * world, max(1, min(8, team_cap))
    tmp = tempfile.mkdtemp(prefix="fhe_bootbatch_")
    forced = dist is not None and world == 1  # FHE_BENCH_FORCE_DIST: the replication path (RCCL, adopted key windows) on one GPU
    if os.environ.get("FHE_BENCH_TEST_STALL_RANK") == str(rank):  # (tests: a rank that never reaches the leg's first collective)
        time.sleep(1e6)
    # (FHE_BENCH_BOOT_SHAPE="budgetEnc,budgetDec,levelsAfter": the CPU rehearsal of the multi-rank path runs a ring of 2^8 on the lane emulator)
    shape = [int(v) for v in os.environ.get("FHE_BENCH_BOOT_SHAPE", "4,4,5").split(",")]
    r = bb.run_rank(logN, slots, total, threads, 2, device, prng, dist=dist if (world > 1 or forced) else None,
                    torch_device=tdev if tdev is not None else "cpu", dump_path=None, warmup=1, key_threads=key_threads,
                    force_replication=forced, budget=(shape[0], shape[1]), levels_after=shape[2])
    h = r.pop("handle")
    keep_keys = r.pop("keys", None)  # (the replicated key tensor: the key towers are windows of it until h.close())
    h.save_outputs()  # the narrow (threaded) pass's outputs: the lockstep pass is compared with them word for word below
    # ciphertexts the stock backend bootstraps for the byte comparison below (11.9 s each on 32 host cores: three at one rank, one
    # when other ranks wait for rank 0)
    nstock = min(3 if world == 1 else 1, per_gpu)
    ct0 = os.path.join(tmp, "hip_ct0.bin")
    h.dump(ct0, 0, nstock)
    # EVERY output of the rank's narrow pass as one digest: compared below with the committed digest of the stock backend's bootstraps of
    # the same 64 ciphertexts (tools/stock_boot_digest.py 64:64: 25 minutes of host time, done once; round 5 compared 3 of 64 with stock)
    all_digest = None
    if world == 1:
        import hashlib
        call = os.path.join(tmp, "hip_all.bin")
        h.dump(call, 0, per_gpu)
        hh = hashlib.sha256()
        with open(call, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                hh.update(chunk)
        all_digest = hh.hexdigest()
        os.remove(call)
    # latency of one bootstrap: ciphertexts of the slice on one thread, one stream (at most 8 of them: the figure is per bootstrap)
    c0 = h.counters()
    single = h.single_thread_latency() / max(1, r["ciphertexts"])
    c1 = h.counters()
    nct = max(1, r["ciphertexts"])

    def per_bootstrap(a, b, passes, rate_per_s):
        """roofline of one bootstrap from the backend's counters between two readings: operand bytes = every tower / key a device
        operation reads or writes, once per operation (the algorithmic bytes of the operation sequence pke issues)"""
        n = passes * nct
        gb = (b["operand_read_bytes"] - a["operand_read_bytes"] + b["operand_write_bytes"] - a["operand_write_bytes"]) / n / 1e9
        return {"bound": "hbm", "operand_GB_per_bootstrap": round(gb, 3), "achieved": round(gb * rate_per_s, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(gb * rate_per_s / HBM_PEAK_GBPS, 4),
                "launches_per_bootstrap": round((b["launches"] - a["launches"]) / n, 1),
                "h2d_MB_per_bootstrap": round((b["h2d_bytes"] - a["h2d_bytes"]) / n / 1e6, 2),
                "d2h_MB_per_bootstrap": round((b["d2h_bytes"] - a["d2h_bytes"]) / n / 1e6, 2), "traffic": None}
    narrow_roof = per_bootstrap(c0, c1, 1, 1.0 / single if single > 0 else 0.0)
    threaded_rate = r["bootstraps_per_s"]
    # the same ciphertexts in LOCKSTEP: one ciphertext whose towers hold all of the rank's towers, cc->EvalBootstrap runs once, every launch
    # works on K towers and every key is read once for all (wide towers; the passes above were the narrow first use of
    # every composite).  The rank's figure is the better of the two ways of running the batch.
    wide = None
    try:
        # (the first lockstep passes after the threaded pass run 5-10 % below the later ones — its buffer caches go back to the device on
        # demand, profiles/r04_sweeps.md sessions h / i — so two passes precede the readings)
        h.bootstrap_wide(group, 1, wide_threads)
        w0, m0, s0 = h.counters(), h.member_bytes(), h.member_stats()
        bsamp = PowerSampler(device) if rank == 0 else None
        if bsamp is not None:
            bsamp.start()
        wsec = h.bootstrap_wide(group, 3, wide_threads)
        boot_power = bsamp.stop() if bsamp is not None else None
        w1, m1, s1 = h.counters(), h.member_bytes(), h.member_stats()
        # members of DCRTPolyHipImpl that ran on its host mirror (the reference's own DCRTPolyImpl) between the two readings: the
        # lockstep passes the rate is quoted on.  A device figure with mirror executions in it is not a device figure.
        mirror = {k: v[1] - s0.get(k, (0, 0, 0, 0))[1] for k, v in s1.items() if v[1] - s0.get(k, (0, 0, 0, 0))[1] > 0}
        ndiff = h.compare_saved()
        wide = {"seconds_per_pass": round(wsec, 4), "bootstraps_per_s": round(r["ciphertexts"] / wsec, 2),
                "group": group if 0 < group < r["ciphertexts"] else r["ciphertexts"],
                "max_abs_error_vs_message": max(h.check(i)[0] for i in range(r["ciphertexts"])),
                "parity": (f"all {r['ciphertexts']} outputs identical word for word to the threaded (narrow) pass's outputs of the same ciphertexts "
                           "(compared in this run, every limb on the host)" if ndiff == 0 else
                           f"MISMATCH: {ndiff} of {r['ciphertexts']} outputs differ from the narrow pass's"),
                "host_threads": wide_threads, "power": boot_power,
                "mirror_executions": {"total": sum(mirror.values()), "by_member": mirror,
                                      "window": "the 1 untimed + 3 timed lockstep passes between the counter readings (fhe_hal_member_stats)"},
                "how": f"one cc->EvalBootstrap per group on a ciphertext of K-tower towers, the groups over {wide_threads} host thread(s) / stream(s)",
                "passes": "2 untimed, then 1 untimed + 3 timed",
                "roofline": per_bootstrap(w0, w1, 4, r["ciphertexts"] / wsec)}  # (1 untimed + 3 timed passes between the readings)
        # the operand bytes itemised by the pke / DCRTPoly scope that issued the operations (GB per bootstrap, largest first)
        per = {k: (m1[k] - m0.get(k, 0)) / (4.0 * nct) / 1e9 for k in m1}
        # (every scope above 5 MB per bootstrap: round 4 listed twelve and left 9.6 GB as "everything else")
        wide["roofline"]["operand_GB_by_member"] = {k: round(v, 3) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:48] if v > 0.005}
        wide["roofline"]["operand_GB_listed"] = round(sum(wide["roofline"]["operand_GB_by_member"].values()), 3)
        # HBM bytes per bootstrap from the committed counter passes of this setting (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the lockstep
        # passes, tools/boot_wide_profile.py pmc) — quoted only when recorded with the kernel sources this run executes
        for recname in ("r06_bootstrap_pmc.json", "r05_bootstrap_pmc.json"):
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", recname)))
            except Exception:
                continue
            if rec.get("kernel_source_sha") == source_sha():
                wide["roofline"]["traffic"] = round(rec["bytes_per_bootstrap"])
                wide["roofline"]["traffic_source"] = f"profiles/{recname} (FETCH_SIZE x2 + WRITE_SIZE over the lockstep passes, groups of 16 on 2 host threads, same kernel sources)"
                wide["roofline"]["wasted_traffic_ratio"] = round(rec["bytes_per_bootstrap"] / 1e9 / max(wide["roofline"]["operand_GB_per_bootstrap"], 1e-9), 3)
                break
            wide["roofline"]["traffic_source"] = f"profiles/{recname} was recorded with other kernel sources: not quoted"
        if mirror:
            wide["parity"] = f"HOST-MIRROR EXECUTIONS in the timed passes ({sum(mirror.values())}: {mirror}); " + wide["parity"]
        if ndiff != 0 or mirror:
            wide["bootstraps_per_s_unverified"] = wide.pop("bootstraps_per_s")  # a figure without parity is not reported as the rate
    except Exception as e:
        wide = {"error": f"{type(e).__name__}: {e}"}
    rate = max(threaded_rate, wide.get("bootstraps_per_s", 0.0))
    total_rate = rate
    if dist is not None and world > 1:
        import torch
        tt = torch.tensor([rate], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        total_rate = float(tt.item())
    res = {"workload": f"cc->EvalBootstrap, ring 2^{logN}, {slots} slots, level budget {{4,4}}, {h.sizeQ} Q + {h.sizeP} P limbs, dnum {h.dnum}; "
                       f"{per_gpu} ciphertexts per GPU over {threads} host threads, {world} rank(s)",
           "bootstraps_per_s_per_gpu": round(rate, 2), "bootstraps_per_s_total": round(total_rate, 2),
           "bootstraps_per_s_over_host_threads": round(threaded_rate, 2), "lockstep": wide,
           "seconds_per_bootstrap": round(single, 5), "one_stream_roofline": narrow_roof, "seconds_per_pass": round(r["seconds_per_pass"], 4),
           "max_abs_error_vs_message": r["max_abs_error"], "setup_s": r["setup_s"], "keygen_s_rank0": r["keygen_s"], "hal_build": bb.HAL_BUILD,
           "host_threads": {"cores": os.cpu_count() or 1, "ranks": world, "cap_per_rank": team_cap, "stream_threads": threads,
                            "openmp_team_during_setup": key_threads},
           "how": "reference pke (unmodified sources) on the HIP backend of DCRTPoly; 1 warm-up + 2 timed passes over the rank's ciphertexts; "
                  "seconds_per_bootstrap = the same ciphertexts on ONE host thread / stream",
           "parity": "every output decrypted and compared with its message; byte comparison with the stock backend not run", "cpu_baseline": None}
    # the leg's roofline = that of the way of running the batch whose rate is reported (kernel census of the lockstep pass:
    # profiles/r05_bootstrap_wide_kernels.txt; model and itemisation: DESIGN.md 7.2 / 7.3)
    if isinstance(wide, dict) and wide.get("bootstraps_per_s", 0.0) >= threaded_rate and "roofline" in wide:
        res["roofline"] = dict(wide["roofline"], path="lockstep groups")
    else:
        res["roofline"] = dict(narrow_roof, path="one stream, narrow towers")
    for k in ("key_set_GB", "key_replication_s", "key_replication_GBps"):
        if k in r:
            res[k] = r[k]
    try:  # device memory in use after the passes = live towers + keys + every cached released buffer: the leg's high-water mark
        mlib = fh.Lib(libpath)
        mq, mpsi = mlib.dcrt_chain(12, 1, 60)
        mctx = fh.Context(mlib, 12, mq, mpsi, device=device)
        fr, tot = C.c_size_t(), C.c_size_t()
        mlib.check(mlib.L.fhe_mem_info(mctx.h, C.byref(fr), C.byref(tot)))
        mctx.close()
        res["device_memory"] = {"in_use_GB_after_the_passes": round((tot.value - fr.value) / 1e9, 1), "total_GB": round(tot.value / 1e9, 1),
                                "what": "hipMemGetInfo at the end of the leg: live towers, the key set and every cached released buffer (the caches "
                                        "keep the evaluation's high-water mark)"}
    except Exception as e:
        res["device_memory"] = {"error": f"{type(e).__name__}: {e}"}
    try:  # the backend's own account of its buffers: what it holds from the device, the high-water mark, how requests were served
        res["allocator"] = h.alloc_stats()
    except Exception as e:
        res["allocator"] = {"error": f"{type(e).__name__}: {e}"}
    h.close()
    del keep_keys
    try:
        digests = json.load(open(os.path.join(ROOT, "tests", "golden", "stock_bootstrap_digests.json")))
    except Exception:
        digests = {}
    dkey = f"logN{logN}_slots{slots}_total{total}_team{key_threads}_first{nstock}"
    # one rank: live.  Several ranks: against the committed digest of the same stock run when there is one (the other ranks do not wait
    # for rank 0's host run), else live (about a minute on the GPU box's host: all `total` ciphertexts are encrypted there)
    live_stock = with_cpu and rank == 0 and os.path.exists(STOCK_BOOT_SO) and (world == 1 or dkey not in digests or
                                                                                 bool(os.environ.get("FHE_BENCH_STOCK_AT_SCALE")))
    if rank == 0 and world > 1 and not live_stock:
        # several ranks: the other ranks would wait minutes for rank 0's stock run (all `total` ciphertexts encrypted on the host, the
        # key set generated there): the byte comparison is against the COMMITTED digest of that run (tools/stock_boot_digest.py wrote
        # it with the same program on the stock backend), or live with FHE_BENCH_STOCK_AT_SCALE=1
        import hashlib
        digest = hashlib.sha256(open(ct0, "rb").read()).hexdigest()
        key, table = dkey, digests
        if key in table:
            res["parity"] = ((f"ciphertext 0 of rank 0's narrow pass identical to the stock backend's bootstrap of the same {total}-ciphertext batch's "
                              f"ciphertext 0: sha256 of the byte dump = the committed digest (tests/golden/stock_bootstrap_digests.json[{key}], "
                              f"{table[key].get('made_by', 'tools/stock_boot_digest.py')}); every rank's lockstep outputs identical word for word to its narrow "
                              "pass's; every output decrypted and compared with its message")
                             if table[key].get("sha256") == digest else
                             f"digest of rank 0's ciphertext 0 ({digest[:16]}…) differs from the committed stock digest for {key}: run with "
                             "FHE_BENCH_STOCK_AT_SCALE=1 for the live comparison")
        else:
            res["parity"] = (f"every output decrypted and compared with its message; lockstep == narrow on every rank; no committed stock digest for {key} "
                             f"(sha256 of rank 0's ciphertext 0 dump: {digest}); FHE_BENCH_STOCK_AT_SCALE=1 runs the stock backend live")
    if live_stock:
        # parity + CPU baseline: the SAME batch on the stock backend in a process of its own — same deterministic PRNG, same OpenMP
        # team during set-up, encryption and key generation (pke's samplers read thread-local PRNGs: equal teams give equal keys), all
        # `total` ciphertexts encrypted, ciphertext 0 bootstrapped (the reference's best team) and compared byte for byte with
        # ciphertext 0 of THIS run's narrow pass (which the lockstep pass was compared with above)
        cthreads = min(32, os.cpu_count() or 1)
        env = dict(os.environ, OMP_NUM_THREADS=str(cthreads))
        env.pop("FHE_HAL_REQUIRE_DEVICE", None)
        sdump = os.path.join(tmp, "stock_ct0.bin")
        code = (f"import sys; sys.path.insert(0, {ROOT!r}); from openfhe_amd import boot_batch as bb; "
                f"r = bb.run_rank({logN}, {slots}, {total}, 1, 1, 0, {prng!r}, so={STOCK_BOOT_SO!r}, dump_path={sdump!r}, warmup=0, "
                f"key_threads={key_threads}, keep={nstock}, eval_threads={cthreads}); print('seconds', r['seconds_per_pass'])")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=2400)
        m = [ln for ln in p.stdout.split("\n") if ln.startswith("seconds")]
        if p.returncode == 0 and m:
            csec = float(m[0].split()[1]) / nstock
            res["cpu_baseline"] = {"value": round(1.0 / csec, 4), "unit": "bootstraps/s", "seconds_per_bootstrap": round(csec, 3), "cores": cthreads,
                                   "kind": "reference",
                                   "sample": f"ciphertexts 0..{nstock - 1} of the batch, the same program on the stock backend (oracle/_ref), {nstock} bootstrap(s)"}
            res["speedup_vs_cpu"] = round(rate * csec, 1)
            same = open(ct0, "rb").read() == open(sdump, "rb").read()
            try:  # (cross-host check of the committed digests the multi-rank runs rely on: this run's stock dump against the table's entry)
                import hashlib
                ent = json.load(open(os.path.join(ROOT, "tests", "golden", "stock_bootstrap_digests.json"))).get(
                    f"logN{logN}_slots{slots}_total{total}_team{key_threads}_first{nstock}")
                if ent:
                    res["stock_digest_check"] = ("the stock run of THIS host reproduces the committed digest (" + ent.get("made_by", "") + ")"
                                                 if ent["sha256"] == hashlib.sha256(open(sdump, "rb").read()).hexdigest()
                                                 else "this host's stock run differs from the committed digest: digests are host-specific here")
            except Exception:
                pass
            res["parity"] = (f"ciphertexts 0..{nstock - 1} of rank 0's narrow pass identical byte for byte to the stock backend's bootstraps of the same "
                             f"batch's ciphertexts (same PRNG, same OpenMP team, all {total} ciphertexts of the {world}-rank batch encrypted on both "
                             "sides); the lockstep pass's outputs — ALL of the rank's — identical word for word to the narrow pass's (lockstep.parity); "
                             "every output of the batch decrypted and compared with its message"
                             if same else f"MISMATCH vs the stock backend (ciphertexts 0..{nstock - 1})")
        else:
            res["parity"] = "byte comparison failed to run: " + (p.stdout + p.stderr)[-300:]
    akey = f"logN{logN}_slots{slots}_total{total}_team{key_threads}_first{per_gpu}"
    if all_digest is not None and akey in digests:
        ok = digests[akey].get("sha256") == all_digest
        res["all_outputs_vs_stock"] = (f"ALL {per_gpu} outputs of the narrow pass identical to the stock backend's bootstraps of the same ciphertexts: sha256 of the "
                                       f"byte dump of all of them = the committed digest (tests/golden/stock_bootstrap_digests.json[{akey}], "
                                       f"{digests[akey].get('made_by', '')})" if ok else
                                       f"MISMATCH: sha256 of all {per_gpu} outputs ({all_digest[:16]}…) differs from the committed stock digest [{akey}]")
        if not ok:
            res["parity"] = res["all_outputs_vs_stock"] + "; " + res.get("parity", "")
        elif isinstance(res.get("parity"), str) and not res["parity"].startswith("MISMATCH"):
            res["parity"] = res["all_outputs_vs_stock"] + "; " + res["parity"]
    elif all_digest is not None:
        res["all_outputs_vs_stock"] = f"no committed stock digest for {akey} (sha256 of this run's {per_gpu} outputs: {all_digest})"
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return res


def cc_evalmult_leg(with_cpu, libpath):
    """BASELINE configs[2]'s operation through the reference's own API: cc->EvalMult (tensor + HYBRID key switch) on a batch of
    ciphertexts at N = 2^16, depth 20 (21 Q limbs, dnum 3), spread over host threads — the reference's pke sources on the HIP backend
    of DCRTPoly against the same program on the stock backend (tests/hal/shim_ckks.cpp multbatch); every product compared (a 128-bit
    digest of every word per product, first and last product byte for byte; 64 ciphertexts, same PRNG, same thread count)."""
    import re
    import shutil
    import subprocess
    import tempfile
    bdir = os.path.join(ROOT, "tests", "hal", "_build")
    hip, stock, prng = (os.path.join(bdir, n) for n in ("shim_ckks_hip", "shim_ckks_stock", "libdetprng.so"))
    # the program built from the PATCHED reference sources (integration/with_hip.patch: hooks bound at source level, plain compiler flags)
    # when it is present, else the one whose hooks are bound on object files (objcopy, hal/Makefile); FHE_HAL_BUILD=objcopy forces the latter
    patched = os.path.join(ROOT, "integration", "_build", "shim_ckks_hip_patched")
    which = "unmodified sources, hooks bound with objcopy (hal/Makefile)"
    if os.path.exists(patched) and os.environ.get("FHE_HAL_BUILD", "") != "objcopy":
        hip, which = patched, "patched sources (integration/with_hip.patch)"
    if not (os.path.exists(hip) and os.path.exists(prng)):
        return {"skipped": "tests/hal/_build/shim_ckks_hip not built (./build.sh hal needs the reference sources)"}
    tmp = tempfile.mkdtemp(prefix="fhe_ccmult_")

    def run(exe, out, batch, reps, threads, extra_env, group=0):
        env = dict(os.environ)
        env.update(extra_env)
        env["OMP_NUM_THREADS"] = str(threads)
        p = subprocess.run([exe, out, prng, "multbatch", "16", "20", str(batch), str(reps), str(group)], env=env, capture_output=True, text=True,
                           timeout=900)
        m = re.search(r"multbatch seconds per pass \S+ EvalMult per second ([0-9.eE+-]+)", p.stdout)
        # (lockstep runs also time the multiplications on operands that STAY wide, and compare those products with the packed pass's)
        mr = re.search(r"multbatch resident seconds per pass \S+ EvalMult per second ([0-9.eE+-]+)", p.stdout)
        md = re.search(r"resident products differing from the packed pass's: (\d+) of", p.stdout)
        run.resident = (float(mr.group(1)), int(md.group(1))) if p.returncode == 0 and mr and md else None
        # PCIe bytes of the evaluation phase (the program resets the backend's counters after key generation and encryption; the phase
        # also downloads the two dumped products and the one decrypted for the check)
        mp = re.search(r"hal: available 1 deviceOps \d+ hostOps (\d+) h2dBytes (\d+) d2hBytes (\d+)", p.stdout)
        run.pcie = (int(mp.group(2)), int(mp.group(3))) if p.returncode == 0 and mp else None
        # host-mirror executions of the evaluation phase, by member ("halmember <member> <device ops> <mirror executions> <reads> <bytes>")
        run.mirror = None
        if p.returncode == 0 and mp:
            by = {}
            for ln in p.stdout.splitlines():
                f = ln.split()
                if len(f) >= 6 and f[0] == "halmember" and int(f[-3]) > 0:
                    by[" ".join(f[1:-4])] = int(f[-3])
            run.mirror = {"total": int(mp.group(1)), "by_member": by,
                          "window": "the program's evaluation phase: warm-up + timed passes + the dump / decryption of the checked products "
                                    "(counters reset after key generation and encryption)"}
        return (float(m.group(1)) if p.returncode == 0 and m else None), (p.stdout + p.stderr)[-400:]

    hipenv = {"FHE_HIP_LIB": libpath}
    # (10 timed passes after the warm-up pass: ~0.5 s, so that the clocks of a device that idled through key generation have ramped up;
    # 5 passes of 64 ciphertexts — 60 ms — measured anything between 2.1 k and 5.1 k/s for one binary, profiles/r03_sweeps.md session i)
    rate, txt = run(hip, os.path.join(tmp, "h256.bin"), 256, 10, 8, hipenv)
    if rate is None:
        shutil.rmtree(tmp, ignore_errors=True)
        return {"error": txt}
    mirror_threaded = run.mirror
    pcie = None
    if run.pcie is not None:  # (1 warm-up + 10 timed passes of 256 products)
        pcie = {"h2d_MB_per_EvalMult": round(run.pcie[0] / (11 * 256) / 1e6, 4), "d2h_MB_per_EvalMult": round(run.pcie[1] / (11 * 256) / 1e6, 4),
                "note": "evaluation phase of the threaded run (counters reset after key generation and encryption); includes the download of the "
                        "three products the program dumps / decrypts"}
    # the same 256 ciphertexts in lockstep: 64 at a time as one ciphertext of 64-tower towers, cc->EvalMult once per group, one host thread
    # (the same OpenMP team as the threaded run: pke's key generation draws from thread-local PRNGs, equal teams give equal keys)
    wrate, wtxt = run(hip, os.path.join(tmp, "w256.bin"), 256, 10, 8, hipenv, group=64)
    threaded = rate
    if wrate is not None:
        same = open(os.path.join(tmp, "w256.bin"), "rb").read() == open(os.path.join(tmp, "h256.bin"), "rb").read()
        lock = {"ops_per_s": round(wrate, 1), "group": 64, "host_threads": 1,  # (one thread issues the group's launches)
                "mirror_executions": run.mirror,
                "how": "PackWide of the operands, cc->EvalMult and UnpackTower of the products INSIDE the timed region; towers of a lockstep group "
                       "are windows of one allocation (wide from birth: packing operands that are already consecutive windows and unpacking cost no "
                       "copy); the 10 timed passes are enqueued back to back and the device queue is drained once behind the last",
                "parity": ("all 256 products identical to the threaded run's (128-bit digest of every word of every product; first and last "
                           "product byte for byte)") if same else "MISMATCH vs the threaded run"}
        if run.resident is not None:  # the same multiplications on ciphertexts that stay wide (no PackWide / UnpackTower in the timed region)
            lock["resident_ops_per_s"] = round(run.resident[0], 1)
            lock["resident_parity"] = ("all 256 products identical word for word to the packed pass's (every limb on the host)" if run.resident[1] == 0
                                       else f"MISMATCH: {run.resident[1]} products differ from the packed pass's")
        if same:
            rate = max(rate, wrate)
    else:
        lock = {"error": wtxt}
    res = {"workload": "cc->EvalMult(ct, ct) with HYBRID relinearisation, N=2^16, 21 Q + 7 P limbs, dnum 3, 256 ciphertexts: over 8 host threads "
                       "(one tower per operation) and in lockstep (wide towers: 64 ciphertexts per cc->EvalMult call, one host thread)",
           "ops_per_s": round(rate, 1), "ops_per_s_over_host_threads": round(threaded, 1), "lockstep": lock,
           "mirror_executions": mirror_threaded, "pcie": pcie, "hal_build": which,
           "how": "reference pke (unmodified sources) on the HIP backend of DCRTPoly; ops_per_s = the better of the two ways of running the batch",
           "parity": "not checked", "cpu_baseline": None}
    nmirror = sum((m or {}).get("total", 0) for m in (mirror_threaded, lock.get("mirror_executions")))
    if nmirror:
        res["parity"] = f"HOST-MIRROR EXECUTIONS in the evaluation phase ({nmirror}); products not yet compared"
    if with_cpu and os.path.exists(stock):
        threads = min(32, os.cpu_count() or 1)
        crate, ctxt = run(stock, os.path.join(tmp, "s64.bin"), 64, 1, threads, {})
        hrate, _ = run(hip, os.path.join(tmp, "h64.bin"), 64, 1, threads, hipenv)
        lrate, _ = run(hip, os.path.join(tmp, "w64.bin"), 64, 1, threads, hipenv, group=64)  # the lockstep path on the stock run's batch
        if crate is not None:
            res["cpu_baseline"] = {"value": round(crate, 2), "unit": "EvalMult/s", "cores": threads, "kind": "reference",
                                   "sample": "the same program on the stock backend, 64 ciphertexts over the same threads, 1 timed pass"}
            res["speedup_vs_cpu"] = round(rate / crate, 1)
            try:
                sref = open(os.path.join(tmp, "s64.bin"), "rb").read()
                same = hrate is not None and open(os.path.join(tmp, "h64.bin"), "rb").read() == sref
                lsame = lrate is not None and open(os.path.join(tmp, "w64.bin"), "rb").read() == sref
                res["parity"] = ("ALL 64 products of the 64-ciphertext batch identical to the stock backend's, over host threads AND in lockstep "
                                 "(one group of 64): 128-bit digest of every word of every product, first and last product byte for byte"
                                 if same and lsame else f"MISMATCH vs the stock backend (threaded {same}, lockstep {lsame})")
                if nmirror:
                    res["parity"] = f"HOST-MIRROR EXECUTIONS in the evaluation phase ({nmirror}); " + res["parity"]
            except OSError as e:
                res["parity"] = f"dumps unreadable: {e}"
    shutil.rmtree(tmp, ignore_errors=True)
    return res


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` invoked like the N=1 run: start one rank per GPU ourselves (the driver's own
        # `python -m torch.distributed.run ... bench.py --gpus N` sets WORLD_SIZE and comes straight through)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    # FHE_BENCH_BACKEND=gloo: the CPU rehearsal of the multi-rank path (with FHE_HIP_LIB pointing at the test-only lane
    # emulator build and small --logn/--batch); the product path is "nccl" = RCCL over xGMI
    backend = os.environ.get("FHE_BENCH_BACKEND", "nccl")
    tdev = None
    if world > 1 or a.gpus > 1 or os.environ.get("FHE_BENCH_FORCE_DIST"):
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            tdev = torch.device("cuda", local)
            dist.init_process_group("nccl", device_id=tdev)
        else:
            tdev = torch.device("cpu")
            dist.init_process_group(backend)
    elif not os.environ.get("FHE_BENCH_NO_TORCH"):
        try:
            import torch
        except Exception:
            torch = None
    if world > 1 and "OMP_NUM_THREADS" not in os.environ:
        # one process per GPU on one node: every rank's OpenMP teams (the backend's own loops, pke's inside the bootstrap leg) stay within
        # its share of the host cores
        os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // world))
    lib = fh.Lib()  # the HIP library or nothing
    if lib.device_count() < 1:
        raise fh.FheError("bench.py needs a HIP device; there is no CPU fallback")
    device = local if (world > 1 and backend == "nccl") else 0

    logN, L, B = a.logn, a.limbs, a.batch
    N = 1 << logN
    q, psi = lib.dcrt_chain(logN, L, 60)  # ILDCRTParams(2N, L, 60): LastPrime then PreviousPrime chain
    ctx = fh.Context(lib, logN, q, psi, device=device)
    x = fill_random_tower(ctx, q, B, seed=2 + rank)

    def gpu_sync():
        ctx.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    def step():
        lib.check(lib.L.fhe_ntt_fwd(ctx.h, x, None, L, B, None))
        lib.check(lib.L.fhe_ntt_inv(ctx.h, x, None, L, B, None))

    for _ in range(a.warmup):
        step()
    barrier()
    gpu_sync()
    psamp = PowerSampler(device) if rank == 0 else None
    if psamp is not None:
        psamp.start()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    gpu_sync()
    barrier()
    dt = time.perf_counter() - t0
    power = psamp.stop() if psamp is not None else None
    if rank == 0 and world == 1 and power is None and not a.no_power and shutil_which("rocm-smi"):
        # no hwmon files: ~3 s of extra steps of the same leg (untimed) with one rocm-smi reading in the middle
        import threading
        box = {}
        thr = threading.Timer(1.2, lambda: box.update(PowerSampler.sample_rocm_smi()))
        thr.start()
        tw = time.perf_counter()
        while time.perf_counter() - tw < 3.0:
            step()
            ctx.sync()
        thr.join()
        power = box or None
    per_rank_ms = [round(dt / a.steps * 1e3, 4)]
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=tdev)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(gathered, tt)
        per_rank_ms = [round(float(g.item()) / a.steps * 1e3, 4) for g in gathered]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / a.steps * 1e3
    bytes_per_step = 4.0 * 8 * N * L * B  # fwd + inv, each: read once + write once
    value = bytes_per_step * world / (ms_per_step * 1e-3) / 1e9

    # ---- parity at the full workload.  Every timed step is NTT followed by INTT, so the resident batch must still be the
    # generated one (round trip); and ONE more forward transform of the resident 16 GB batch is compared word for word with
    # the oracle's forward transform of the same towers (first / middle / last: tower t is seed tower t % 8), so that a
    # wrong-but-invertible transform cannot pass.  The inverse transform that follows restores the batch.
    def full_size_parity():
        seed_polys = min(8, B)
        host = host_seed_towers(q, N, B, 2 + rank, 8)
        # EVERY tower of the resident batch (checksum of every limb-row, one read of the batch), and tower 0 word for word
        bad = all_towers_match(lib, ctx, x, B, L, host)
        if bad >= 0 or not np.array_equal(download_tower(lib, ctx, x, 0, L), host[0]):
            return f"MISMATCH (round trip) at tower {max(bad, 0)}"
        if a.no_parity:
            return f"fwd+inv round trip bit-exact on ALL {B} towers after all steps (checksums of every limb-row; oracle comparison skipped)"
        o = load_oracle()
        if o is None:
            return f"fwd+inv round trip bit-exact on ALL {B} towers; oracle library not available"
        octx = o.orc_ctx_create(N, L, q, psi)
        lib.check(lib.L.fhe_ntt_fwd(ctx.h, x, None, L, B, None))
        want = host.copy()
        for sp in range(want.shape[0]):
            o.orc_ntt_fwd_tower(octx, want[sp], None, L, 1, 0)
        bad = all_towers_match(lib, ctx, x, B, L, want)
        res = (f"forward NTT of ALL {B} towers of the resident batch == oracle (orc_ntt_fwd_tower; checksums of every limb-row, tower 0 word "
               "for word), and fwd+inv round trip bit-exact on all towers after all steps")
        if bad >= 0 or not np.array_equal(download_tower(lib, ctx, x, 0, L), want[0]):
            res = f"MISMATCH vs oracle (forward NTT) at tower {max(bad, 0)}"
        lib.check(lib.L.fhe_ntt_inv(ctx.h, x, None, L, B, None))
        ctx.sync()
        o.orc_ctx_destroy(octx)
        return res
    roundtrip = full_size_parity()

    # ---- Hadamard product a o b over the same batch shape: 3 streams, HBM-bound ----
    hadamard = None
    if not a.no_hadamard:
        y = fill_random_tower(ctx, q, B, seed=50 + rank)
        z = ctx.malloc(B * L * N * 8)

        def hstep():
            lib.check(lib.L.fhe_mul(ctx.h, z, x, y, None, L, B, None))
        for _ in range(2):
            hstep()
        gpu_sync()
        t1 = time.perf_counter()
        hs = max(5, a.steps)
        for _ in range(hs):
            hstep()
        gpu_sync()
        hdt = (time.perf_counter() - t1) / hs
        hbytes = 3.0 * 8 * N * L * B
        hadamard = {"GB_per_s_per_gpu": round(hbytes / hdt / 1e9, 1), "ms": round(hdt * 1e3, 3),
                    "hbm_roofline_frac": round(hbytes / hdt / 1e9 / HBM_PEAK_GBPS, 4),
                    "bytes": hbytes, "note": "out = a*b mod q_i per limb (generalized Barrett), read 2 + write 1 streams"}
        # negacyclic polynomial product c = a * b, a, b, c in COEFFICIENT form (SURVEY 8(d) "fused fwd o mul o inv"):
        # fhe_poly_mul (column passes + two fused row kernels) against the same product as four separate calls
        wsb = lib.L.fhe_poly_mul_workspace_bytes(ctx.h, L, B)
        wsp = ctx.malloc(wsb)
        wa, wb = wsp, C.c_void_p(wsp.value + wsb // 2)

        def pfused():
            lib.check(lib.L.fhe_poly_mul(ctx.h, x, y, z, None, L, B, wsp, wsb, None))

        def pplain():
            lib.check(lib.L.fhe_ntt_fwd_oop(ctx.h, x, wa, None, L, B, None))
            lib.check(lib.L.fhe_ntt_fwd_oop(ctx.h, y, wb, None, L, B, None))
            lib.check(lib.L.fhe_mul(ctx.h, z, wa, wb, None, L, B, None))
            lib.check(lib.L.fhe_ntt_inv(ctx.h, z, None, L, B, None))
        times = {}
        for nm, fn in (("fused", pfused), ("separate", pplain), ("fused", pfused)):
            fn()
            gpu_sync()
            t2 = time.perf_counter()
            ps = max(3, a.steps // 2)
            for _ in range(ps):
                fn()
            gpu_sync()
            times[nm] = min(times.get(nm, 1e9), (time.perf_counter() - t2) / ps)
        pbytes = 5.0 * 8 * N * L * B  # read a, b; write c; + one workspace round trip counted as algorithmic
        ppar = "skipped"
        if not a.no_parity:
            o = load_oracle()
            if o is None:
                ppar = "oracle library not available"
            else:
                pfused()
                gpu_sync()
                octx = o.orc_ctx_create(N, L, q, psi)
                hx, hy = host_seed_towers(q, N, B, 2 + rank, 8), host_seed_towers(q, N, B, 50 + rank, 8)
                ppar = "bit-exact vs oracle (INTT(NTT(a) o NTT(b))) on towers " + str(sample_towers(B))
                done = {}
                for tw in sample_towers(B):
                    sp = tw % hx.shape[0]
                    if sp not in done:
                        u, v = hx[sp].copy(), hy[sp].copy()
                        o.orc_ntt_fwd_tower(octx, u, None, L, 1, 0)
                        o.orc_ntt_fwd_tower(octx, v, None, L, 1, 0)
                        w = np.empty_like(u)
                        for l in range(L):
                            o.orc_vec_mul(w[l], u[l], v[l], N, q[l])
                        o.orc_ntt_inv_tower(octx, w, None, L, 1, 0)
                        done[sp] = w
                    if not np.array_equal(download_tower(lib, ctx, z, tw, L), done[sp]):
                        ppar = f"MISMATCH vs oracle at tower {tw}"
                        break
                o.orc_ctx_destroy(octx)
        hadamard["polymul_fused"] = {"ms_per_batch": round(times["fused"] * 1e3, 3), "ms_per_batch_four_separate_calls": round(times["separate"] * 1e3, 3),
                                     "GB_per_s_algorithmic": round(pbytes / times["fused"] / 1e9, 1), "algorithmic_bytes": pbytes,
                                     "k_limb_products_per_s": round(B * L / times["fused"] / 1e3, 1), "parity": ppar,
                                     "note": "fhe_poly_mul: 2 forward column passes, forward row pass of a, [forward row pass of b + "
                                             "Hadamard product + inverse row pass] in one kernel, inverse column pass"}
        ctx.free(wsp)
        ctx.free(y)
        ctx.free(z)

    # ---- roofline of the dominant kernel (row pass of the forward transform), live hipEvent timing ----
    roof = None
    per_kernel = {}
    if rank == 0:
        ms = C.c_float()
        names = {10: "fwd_column_pass", 11: "fwd_row_pass", 12: "inv_row_pass", 13: "inv_column_pass"}
        if logN > 12:
            for d, nm in names.items():
                lib.check(lib.L.fhe_time_ntt(ctx.h, x, None, L, B, d, 5, None, C.byref(ms)))
                per_kernel[nm + "_ms"] = round(ms.value, 4)
            dom = max(per_kernel, key=per_kernel.get)
            alg = 2.0 * 8 * N * L * B  # a pass kernel reads every word once and writes it once
            ach = alg / (per_kernel[dom] * 1e-3) / 1e9
            # HBM bytes per launch from the committed rocprofv3 PMC passes of this exact workload — quoted only when the record
            # was made with the kernels this run executes (same kernel-source identity), else null
            traffic, tsrc, wasted = None, None, None
            for rec in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", rec)))
                    if pmc.get("workload") == f"logN{logN}_L{L}_B{B}" and pmc.get("kernel_source_sha") == source_sha():
                        traffic = pmc["per_launch_bytes"][dom[:-3]]["total"]
                        tsrc = f"profiles/{rec} (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, same kernel sources)"
                        # a TRANSFORM is two launches (column pass + row pass), each moving the whole tower: bytes moved per fwd+inv
                        # step over the step's algorithmic bytes (one read + one write per transform)
                        wasted = round(sum(v["total"] for v in pmc["per_launch_bytes"].values()) / bytes_per_step, 3)
                        break
                    tsrc = f"profiles/{rec} was recorded with other kernel sources: not quoted"
                except Exception:
                    continue
            # the BINDING roofline of the dominant kernel is integer issue, not HBM: SQ counters of this workload (committed record)
            binding = None
            try:
                sqrec = next(r for r in ("r06_pmc_valu.json", "r05_pmc_valu.json", "r04_pmc_valu.json", "r03_pmc_valu.json") if os.path.exists(os.path.join(ROOT, "profiles", r)))
                sq = json.load(open(os.path.join(ROOT, "profiles", sqrec)))["legs"]["ntt"]
                ent = next(v for k, v in sq.items() if k.startswith(dom[:-3]))
                instr, act = ent["valu_instructions_per_wave"], ent["active_valu_over_wave_cycles"]
                simds = 256 * 4
                ns_per_instr = per_kernel[dom] * 1e6 / (instr * ent["waves"] / simds)  # wall time per VALU instruction issued on one SIMD
                # what binds the leg is the PACKAGE POWER CAP (profiles/r06_sweeps.md section 2): 1386-1393 W of 1400 W in every sample, shader clock
                # 1.93-1.96 GHz instead of 2.4; at that clock the row passes issue ~90 % of one VALU instruction per quad-cycle and SIMD
                clk = ((power or {}).get("sclk_MHz_mean") or (power or {}).get("sclk_MHz") or 1950) / 1e3
                peak_ns = 4.17 / clk  # the multiplier class issues once per 4.17 cycles per SIMD (tools/energybench.hip)
                binding = {"bound": "package power cap (time = energy / cap)", "instr_per_wave_tile": instr,
                           "valu_active_over_wave_cycles": act,
                           "ns_per_instr_per_simd": round(ns_per_instr, 3), "sclk_GHz_used": clk,
                           "frac_of_issue_peak_at_that_clock": round(peak_ns / ns_per_instr, 3),
                           "power": power,
                           "energy_model": "per step: floor 590 W x t + HBM 128.8 GB moved x 72 pJ/B (9.3 J) + 1.0e10 VALU wave-instructions x ~1.05 nJ "
                                           "(10.5 J) = 36.6 J against 38.6 J measured; one stage of butterflies = 0.27 J = 0.33 ms; residency (4 -> 8 waves "
                                           "per SIMD), barriers (6 -> 1), tiles per CU (4 -> 8), software pipelining: all built bit-exact, none changed the "
                                           "time (profiles/r06_sweeps.md sections 1-2, profiles/r06_energybench.json)",
                           "source": f"profiles/{sqrec} (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES ... on this workload) "
                                     "x this run's hipEvent kernel time"}
            except Exception as e:
                binding = {"bound": "valu", "error": f"no SQ-counter record: {type(e).__name__}"}
            # `frac` is the TRANSFORM-level figure north_star's target is about (one read + one write per transform; a
            # transform is two launches that each move the whole tower) = achieved / peak with achieved = the step's algorithmic bytes over
            # its time; the dominant launch on its own (its bytes moved over its duration) is `kernel_achieved` / `kernel_frac`.
            roof = {"bound": "hbm", "kernel": "ntt_static_kernel/" + dom[:-3], "achieved": round(value / world, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(value / world / HBM_PEAK_GBPS, 4),
                    "kernel_achieved": round(ach, 1), "kernel_frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_source": tsrc, "kernel_source_sha": source_sha(),
                    "algorithmic_bytes_per_launch": alg, "per_kernel_ms": per_kernel,
                    # the dominant kernel is HBM-priced above; what actually binds it is the integer pipe:
                    "binding": binding,
                    # transform level (north_star's target is fwd+inv against HBM): a transform = 2 launches, each moving the tower
                    "transform_frac": round(value / world / HBM_PEAK_GBPS, 4), "wasted_traffic_ratio": wasted}
        else:
            lib.check(lib.L.fhe_time_ntt(ctx.h, x, None, L, B, 0, 5, None, C.byref(ms)))
            alg = 2.0 * 8 * N * L * B
            ach = alg / (ms.value * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "ntt_static_kernel(single pass)", "achieved": round(ach, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None}
        # restore a valid tower for anything that follows (single-pass timing scrambles it): not needed further
    ctx.free(x)

    em = None
    if not a.no_evalmult:
        em = evalmult_leg(lib, device, a.evalmult_logn, a.evalmult_batch, max(20, a.steps * 2), 6, gpu_sync, dist,
                          sizeQ=a.evalmult_limbs, parity=not a.no_parity, tdev=tdev)
        if dist is not None:
            tt = torch.tensor([em["ops_per_s_per_gpu"]], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            em["ops_per_s_total"] = round(float(tt.item()), 1)
        else:
            em["ops_per_s_total"] = em["ops_per_s_per_gpu"]

    # ---- multi-rank only: replication of a bootstrapping-sized rotation-key set: rank 0 holds the keys, every
    # rank ends up with a full copy through scatter + all-gather (each xGMI link carries 1/N of the set per step instead of
    # a link-bound ring broadcast of the whole set)
    keyrep = None
    if dist is not None and not os.environ.get("FHE_BENCH_NO_KEYREP"):
        from openfhe_amd import shard
        nkeys = int(os.environ.get("FHE_BENCH_KEYREP_KEYS", "14"))
        words = 3 * (a.evalmult_limbs + 7) * (1 << a.evalmult_logn)  # one key half: [dnum][l+k][N]
        try:
            src_t = torch.zeros((nkeys, 2, words), dtype=torch.int64, device=tdev) if rank == 0 else None  # resident on rank 0
            dist.barrier()
            t0 = time.perf_counter()
            full = shard.allgather_words(src_t, (nkeys, 2, words), tdev)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dist.barrier()
            sec = time.perf_counter() - t0
            gb = nkeys * 2 * words * 8 / 1e9
            keyrep = {"keys": nkeys, "GB": round(gb, 3), "seconds": round(sec, 4), "GB_per_s": round(gb / sec, 1),
                      "how": f"rank 0 -> {world} rank(s): scatter of 1/{world} slices + all-gather ({backend})"}
            del full
        except Exception as e:
            keyrep = {"error": f"{type(e).__name__}: {e}"}

    # ---- what the run looked like from inside (so that the first SCALE record verifies itself): the process group as torch.distributed
    # reports it, and every rank's device as the HIP runtime names it
    dinfo = {"launcher_world_size": world, "backend_requested": backend if dist is not None else None, "process_group": None, "ranks": None}
    mine = {"rank": rank, "local_rank": local, "hip_device": device, "pid": os.getpid(), "host": os.uname().nodename}
    try:
        if torch is not None and torch.cuda.is_available():
            pr = torch.cuda.get_device_properties(device)
            mine.update({"name": pr.name, "memory_GiB": round(pr.total_memory / 2**30, 1), "compute_units": pr.multi_processor_count})
            mine["pci_bus_id"] = getattr(pr, "pci_bus_id", None)
    except Exception as e:
        mine["device_query_error"] = f"{type(e).__name__}: {e}"
    if dist is not None:
        dinfo["process_group"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank": dist.get_rank(),
                                  "is_nccl_available": bool(dist.is_nccl_available()),
                                  "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None}
        got = [None] * world
        dist.all_gather_object(got, mine)
        dinfo["ranks"] = got
        dinfo["distinct_devices"] = len({(g.get("host"), g.get("hip_device")) for g in got})
    else:
        dinfo["ranks"] = [mine]

    bfv = None
    if not a.no_bfv and logN == 16:
        bfv = bfv_leg(lib, device, a.bfv_batch, max(30, a.steps * 3), 6, gpu_sync,
                      rank == 0 and world == 1 and not a.no_cpu_baseline, parity=not a.no_parity)

    ltr = None
    if not a.no_lt and logN == 16 and world == 1:
        ltr = linear_transform_leg(lib, device, (1, 8), max(10, a.steps), 3, rank == 0 and not a.no_cpu_baseline,
                                   parity=not a.no_parity)

    def emit(boot, ccm, cpu):
        if rank != 0:
            return
        out = {
            "metric": "NTT GB/s vs HBM roofline + CKKS EvalMultKeySwitch/sec, N=2^16 L=30, 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"batched fwd+inv negacyclic NTT, N=2^{logN}, L={L} x 60-bit limbs, "
                                   f"{B} polynomials per GPU (BASELINE configs[1])",
                       "global_batch": B * world, "bytes_per_step_per_gpu": bytes_per_step,
                       "parallelism": f"batch-sharded x{world}, no data-path collective"},
            "hbm_roofline_frac_fwd_inv": round(value / world / HBM_PEAK_GBPS, 4),
            "roofline": roof, "cpu_baseline": cpu, "evalmult": em,
            # the leg runs at the package power cap (profiles/r06_sweeps.md section 2): time = energy / cap
            "power": power,
            # the legs that run the reference's own CryptoContext on the backend need the reference-built test programs
            "hal_build": "present" if os.path.exists(os.path.join(ROOT, "tests", "hal", "_build", "shim_ckks_hip")) else "absent",
            "hadamard": hadamard, "parity_at_full_size": roundtrip, "ms_per_step_per_rank": per_rank_ms,
            # ONE rule for every cpu_baseline of this line: the reference runs with its best OpenMP team out of {8, 16, 32, 64, 128, all
            # logical cores}.  Legs whose probe takes seconds search it live (headline NTT, linear transform, BFV: "cores" = the winner);
            # the legs where one probe is a 10-100 s program (bootstrap, cc->EvalMult) use 32, the team that search selects on this host
            # class (rounds 1-3: 32 every time; 8.9-10.2 s per bootstrap at 32 threads, 45 s at 256).
            "cpu_team_rule": "best OpenMP team of {8,16,32,64,128,all}: searched live by the in-process legs, fixed at 32 (= what the search "
                             "picks on these hosts) for the legs that run the reference as a separate program",
        }
        out["distributed"] = dinfo
        if keyrep is not None:
            out["rotation_key_replication"] = keyrep
            dinfo["key_replication_GB_per_s"] = keyrep.get("GB_per_s")
        if boot is not None and isinstance(boot, dict) and boot.get("key_replication_GBps") is not None:
            dinfo["bootstrap_key_set_replication_GB_per_s"] = boot.get("key_replication_GBps")
        if bfv is not None:
            out["bfv_evalmult"] = bfv
        if ltr is not None:
            out["linear_transform"] = ltr
        if boot is not None:
            out["evalbootstrap"] = boot
        if ccm is not None:
            out["cryptocontext_evalmult"] = ccm
        print(json.dumps(out), flush=True)

    boot = None
    on_emu = os.environ.get("FHE_HIP_LIB", "").endswith("libfhe_emu.so")
    if not a.no_bootstrap and ((logN == 16 and not on_emu) or os.environ.get("FHE_BENCH_EMU_BOOTSTRAP")):
        # The leg is an extra with collectives of its own (the key set travels from rank 0): a rank that fails or stalls inside it must
        # not take the headline line down.  With several ranks a watchdog on every rank ends the process after the limit — rank 0 prints the
        # line (without the leg's figures) first.
        import threading
        limit = float(os.environ.get("FHE_BENCH_BOOTSTRAP_LIMIT_S", "1500" if os.environ.get("FHE_BENCH_STOCK_AT_SCALE") else "480"))

        def give_up():
            emit({"error": f"the sharded bootstrap leg did not finish within {limit:.0f} s on {world} ranks: line printed without it"}, None, None)
            os._exit(3)  # (the headline line is out; a stalled collective is a failure of the run, not a clean exit)

        dog = threading.Timer(limit, give_up) if world > 1 else None
        if dog is not None:
            dog.daemon = True
            dog.start()
        try:  # every rank takes part: the batch is sharded, the key set travels from rank 0
            boot = bootstrap_batch_leg(a.bootstrap_logn, a.bootstrap_batch, a.bootstrap_threads, rank, world, device, dist, tdev,
                                       not a.no_cpu_baseline, lib.path, group=a.bootstrap_group, wide_threads=a.bootstrap_wide_threads)
        except Exception as e:
            boot = {"error": f"{type(e).__name__}: {e}"}
        if dog is not None:
            dog.cancel()

    ccm = None
    if rank == 0 and world == 1 and not a.no_cc_evalmult and logN == 16 and not os.environ.get("FHE_HIP_LIB", "").endswith("libfhe_emu.so"):
        try:
            ccm = cc_evalmult_leg(not a.no_cpu_baseline, lib.path)
        except Exception as e:
            ccm = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(q, psi, logN, L, a.cpu_seconds)

    emit(boot, ccm, cpu)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
