"""Sampled towers on the device (fhe_sample_uniform / _gaussian / _ternary, csrc/sampler_kernels.h — SURVEY.md 8(f)-3).

What is pinned on what:
 * the generator: Philox4x32-10 against the Random123 known-answer vectors (oracle and, through the sampled words, the kernels);
 * the Gaussian inversion table against DiscreteGaussianGeneratorImpl::Initialize (discretegaussiangenerator-impl.h:75-89), recomputed
   here in numpy from its text;
 * the kernels against the oracle's restatement, word for word, on the emulator and on the GPU;
 * the distributions against the reference's: range and uniformity of the uniform words, probabilities of the Gaussian integers against
   the table's own cell masses, P(-1) = P(0) = P(1) for the ternary values.
The reference's Blake2 WORDS are not reproduced (a device sampler needs a counter-based generator: the survey keeps this row optional)."""
import ctypes as C
import math

import numpy as np

import libs
from openfhe_amd import fhe_hip as fh

from test_parity import params


def test_philox_known_answers(oracle):
    """Random123 kat_vectors, philox4x32-10"""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        out = np.zeros(4, np.uint32)
        oracle.orc_philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32), out)
        assert tuple(int(v) for v in out) == want


def reference_dgg_table(sigma):
    """DiscreteGaussianGeneratorImpl::Initialize, discretegaussiangenerator-impl.h:75-89, from its text"""
    fin = int(math.ceil(sigma * 12.00610553538285))
    variance = 2 * sigma * sigma
    vals, cusum = [], 0.0
    for x in range(1, fin + 1):
        cusum += math.exp(-(float(x * x) / variance))
        vals.append(cusum)
    a = 1.0 / (2 * cusum + 1.0)
    return [v * a for v in vals], a


def test_gaussian_table_is_the_references(oracle):
    for sigma in (3.19, 1.5, 20.0):
        want, a = reference_dgg_table(sigma)
        vals = (C.c_double * 4096)()
        av = C.c_double()
        n = oracle.orc_dgg_table(sigma, vals, 4096, C.byref(av))
        assert n == len(want) and av.value == a and list(vals[:n]) == want


def test_sampled_towers_match_the_oracle_word_for_word(backend, oracle):
    o = oracle
    logN, L = (11, 3) if "emulator" in backend.version() else (14, 5)
    N = 1 << logN
    q, psi = params(o, logN, L)
    ctx = fh.Context(backend, logN, q, psi)
    B, seed = 2, 0x0123456789abcdef
    want = np.empty((B, L, N), np.uint64)
    o.orc_sample_uniform(want, q, L, B, N, seed, 7)
    got = ctx.sample("uniform", B, L, seed, 7).to_host()
    assert np.array_equal(got, want)
    assert np.all(got < q[None, :, None])
    # a limb subset draws the same element sub-streams against its own moduli
    sel = np.array([2, 0], np.uint32)
    want2 = np.empty((B, 2, N), np.uint64)
    o.orc_sample_uniform(want2, q[sel], 2, B, N, seed, 9)
    assert np.array_equal(ctx.sample("uniform", B, 2, seed, 9, limb_idx=sel).to_host(), want2)
    for sigma in (3.19, 8.0):
        ints = np.empty(B * N, np.int64)
        o.orc_sample_gaussian(want, ints.ctypes.data_as(C.POINTER(C.c_int64)), q, L, B, N, sigma, seed, 11)
        got = ctx.sample("gaussian", B, L, seed, 11, sigma=sigma).to_host()
        assert np.array_equal(got, want)
        # one integer per coefficient, the same in every limb: k >= 0 as k, k < 0 as q - |k| (dcrtpoly-impl.h:141-145)
        k = ints.reshape(B, N)
        for l in range(L):
            assert np.array_equal(got[:, l, :], np.where(k < 0, q[l] - np.abs(k).astype(np.uint64), k.astype(np.uint64)))
    ints = np.empty(B * N, np.int64)
    o.orc_sample_ternary(want, ints.ctypes.data_as(C.POINTER(C.c_int64)), q, L, B, N, seed, 13)
    assert np.array_equal(ctx.sample("ternary", B, L, seed, 13).to_host(), want)
    assert set(np.unique(ints)) <= {-1, 0, 1}
    # another stream id or seed is another tower
    assert not np.array_equal(ctx.sample("ternary", B, L, seed, 14).to_host(), want)
    assert not np.array_equal(ctx.sample("ternary", B, L, seed + 1, 13).to_host(), want)
    ctx.close()


def test_distributions_are_the_references(oracle):
    """on the oracle's restatement (the kernels equal it word for word): 2^18 draws each"""
    o = oracle
    N, B = 1 << 14, 16
    q = np.array([o.orc_last_prime(60, 2 * N), o.orc_last_prime(36, 2 * N), (1 << 20) + 7], np.uint64)  # (the third only as a range)
    x = np.empty((B, 3, N), np.uint64)
    o.orc_sample_uniform(x, q, 3, B, N, 99, 1)
    for l in range(3):
        v = x[:, l, :].ravel().astype(np.float64) / float(q[l])
        assert np.all(x[:, l, :] < q[l])
        hist = np.histogram(v, bins=64, range=(0, 1))[0]
        exp = v.size / 64
        assert np.sum((hist - exp) ** 2 / exp) < 64 + 6 * math.sqrt(2 * 64)  # chi-square, 6 sigma
        assert abs(v.mean() - 0.5) < 6 / math.sqrt(12 * v.size)
    sigma = 3.19
    ints = np.empty(B * N, np.int64)
    y = np.empty((B, 1, N), np.uint64)
    o.orc_sample_gaussian(y, ints.ctypes.data_as(C.POINTER(C.c_int64)), q[:1], 1, B, N, sigma, 99, 2)
    vals, a = reference_dgg_table(sigma)
    # the inversion's own cell masses: P(0) = a, P(+-k) = (vals[k-1] - vals[k-2]) (vals[-1] := a/2 shifted: tmp = |s| - a/2)
    n = ints.size
    p0 = a
    assert abs((ints == 0).mean() - p0) < 6 * math.sqrt(p0 * (1 - p0) / n)
    prev = 0.0
    for k in range(1, 9):
        pk = vals[k - 1] - prev
        prev = vals[k - 1]
        for sgn in (1, -1):
            f = (ints == sgn * k).mean()
            assert abs(f - pk) < 6 * math.sqrt(pk * (1 - pk) / n), (k, sgn, f, pk)
    assert abs(ints.mean()) < 6 * sigma / math.sqrt(n) and abs(ints.std() - sigma) < 0.02
    z = np.empty((B, 1, N), np.uint64)
    o.orc_sample_ternary(z, ints.ctypes.data_as(C.POINTER(C.c_int64)), q[:1], 1, B, N, 99, 3)
    for v in (-1, 0, 1):
        assert abs((ints == v).mean() - 1 / 3) < 6 * math.sqrt(2 / 9 / n)
