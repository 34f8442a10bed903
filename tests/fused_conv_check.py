"""Run with FHE_KS_FUSE_CONV=1 (read once per process): EvalMult + key switching and a BSGS linear transform on two-pass
rings with the ModUp / ModDown conversions computed inside the forward transforms' column passes (experimental path),
against the oracle.  argv[1] = library path."""
import sys

import numpy as np

import libs
from openfhe_amd import fhe_hip as fh
from test_parity import ckks_like_params
from test_parity_lt import run_oracle

o = libs.load_oracle()
lib = fh.Lib(sys.argv[1])
assert lib.L.fhe_debug_fused_conv_launches() == 0
for logN, sizeQ, dnum, sizeQl, B in [(13, 4, 2, 4, 1), (13, 5, 3, 3, 2), (14, 3, 3, 3, 1)]:
    rng = np.random.default_rng(5)
    N = 1 << logN
    q, psiQ, p, psiP = ckks_like_params(o, logN, sizeQ, dnum)
    hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, len(p), p, psiP, dnum)
    allq = np.concatenate([q, p])
    ctx = fh.Context(lib, logN, allq, np.concatenate([psiQ, psiP]))
    plan = fh.KeySwitchPlan(ctx, sizeQ, len(p), dnum)
    kb, ka = libs.rand_tower(rng, allq, N, dnum), libs.rand_tower(rng, allq, N, dnum)
    plan.upload_key(kb, ka)
    ops = [libs.rand_tower(rng, q[:sizeQl], N, B) for _ in range(4)]
    n0 = lib.L.fhe_debug_fused_conv_launches()
    r0, r1 = plan.EvalMult(*[ctx.tower(x) for x in ops])
    launches = lib.L.fhe_debug_fused_conv_launches() - n0
    numParts = min(-(-sizeQl // (-(-sizeQ // dnum))), dnum)
    assert launches == numParts + 1, (launches, numParts)  # one column pass per ModUp digit + the ModDown's
    w0, w1 = np.empty_like(ops[0]), np.empty_like(ops[0])
    for b in range(B):
        o.orc_ckks_eval_mult_relin(hy, ops[0][b], ops[1][b], ops[2][b], ops[3][b], sizeQl, kb, ka, w0[b], w1[b])
    assert np.array_equal(r0.to_host(), w0) and np.array_equal(r1.to_host(), w1), (logN, sizeQ, dnum)
    if logN == 13 and dnum == 2:  # a 2 x 2 BSGS transform through the same fused conversions
        ks = [o.orc_find_automorphism_index_2n_complex(i, 2 * N) for i in (1, 2)]
        keys = [(libs.rand_tower(rng, allq, N, dnum), libs.rand_tower(rng, allq, N, dnum)) for _ in ks]
        hnd = [plan.make_key(*k) for k in keys]
        extq = np.concatenate([q[:sizeQl], p])
        diag = [[libs.rand_tower(rng, extq, N) for _ in range(2)] for _ in range(2)]
        want = run_oracle(o, hy, ops[0], ops[1], sizeQl, [None, (ks[0],) + keys[0]], [None, (ks[1],) + keys[1]], diag)
        g0, g1 = plan.BsgsTransform(ctx.tower(ops[0]), ctx.tower(ops[1]), [None, (ks[0], hnd[0])], [None, (ks[1], hnd[1])],
                                    [[ctx.upload(d) for d in row] for row in diag])
        assert np.array_equal(g0.to_host(), want[0]) and np.array_equal(g1.to_host(), want[1])
    plan.close()
    ctx.close()
    o.orc_hybrid_destroy(hy)
print("fused_conv_check OK")
