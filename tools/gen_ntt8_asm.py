#!/usr/bin/env python3
"""Generates openfhe-development_amd/csrc/ntt_bfly8_pinned.h: the in-place gfx950 butterflies of the 8-residues-per-lane
row pass (ntt_row8.h, round 6).

Same arithmetic and instruction texts as tools/gen_ntt_asm.py (truncated Shoup quotient, quotient-estimate reductions, lazy
Gentleman-Sande sums — imported from there), on a register file sized for EIGHT waves per SIMD (<= 64 VGPRs per lane):

    v[0:35]   the compiler's: addresses and up to 7 per-lane twiddle pairs (28 registers)
    v[36:47]  the temporaries of ONE butterfly slot (12 registers)
    v[48:63]  the lane's 8 residues (residue k = v[48+2k : 49+2k])

One butterfly at a time: with 8 waves per SIMD the two wait states between a VALU carry write and its reader are filled by
other waves; the second interleaved butterfly of the 16-residue kernel buys nothing there (tools/occbench.hip,
profiles/r06_occbench.json: 24.4 vs 24.7 ns per butterfly and SIMD at 8 waves) and would cost 12 more registers.

The header has two parts: the lazy-inverse PLAN as constexpr tables (every build: the lane emulator's C++ butterflies follow the
same plan and check its bounds value by value) and the asm blocks (device build).  Every block is simulated against python
integers before the header is written.

Usage:  python tools/gen_ntt8_asm.py [--check]
"""
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_ntt_asm as g  # noqa: E402

ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "openfhe-development_amd", "csrc", "ntt_bfly8_pinned.h")

DATA0, TEMP0 = 48, 36
g.DATA0 = DATA0
g.CSUB_TMP = (TEMP0, TEMP0 + 2, TEMP0 + 4, TEMP0 + 6)
M64 = g.M64
R, p, v = g.R, g.p, g.v


class T8:
    """temporaries of the single butterfly slot"""
    slot = 0

    def __init__(self):
        b = TEMP0
        self.L, self.C, self.H, self.Q, self.X, self.Y = b, b + 2, b + 4, b + 6, b + 8, b + 10
        self.Z = self.Y      # exact-quotient streams only (they do not use Y); v[Z+1] must hold 0 there
        self.D = self.Y
        self.c0, self.c1 = "s[42:43]", "s[44:45]"


T = T8()
TEMPS = [f"v{r}" for r in range(TEMP0, TEMP0 + 12)]
SCLOB = [f"s{r}" for r in range(40, 50)]


def stage_pairs(b):
    """butterflies of stage b (register bit b) on the 8 registers, as (k0, k1, twiddle group g)"""
    out = []
    for gq in range(4 >> b):
        for lo in range(1 << b):
            k0 = (gq << (b + 1)) | lo
            out.append((k0, k0 | (1 << b), gq))
    return out


def inv_plan(bLo=0, bHi=2, lazy_out=False, B0=3):
    """tools/gen_ntt_asm.py inv_plan on 8 registers (stages bLo..bHi of a radix-8 step): inputs < B0*q; a' = u + v adds the
    bounds, b' = shoup(u - v + K) is < 3q; a pair whose bounds add up to more than 16 is reduced first by the cheapest of
    {nothing, a conditional subtraction of half the bound, a quotient estimate}; the step ends with every residue below B0*q."""
    bound = [B0] * 8
    pre, K = {}, {}
    for b in range(bLo, bHi + 1):
        pre[b], K[b] = [], []
        for (k0, k1, _g) in stage_pairs(b):
            best = None
            for (c0, n0, o0) in g.reduce_options(bound[k0]):
                for (c1, n1, o1) in g.reduce_options(bound[k1]):
                    if n0 + n1 <= 16 and (best is None or (c0 + c1, n0 + n1) < best[0]):
                        best = ((c0 + c1, n0 + n1), n0, n1, o0, o1)
            _, n0, n1, o0, o1 = best
            for k, o in ((k0, o0), (k1, o1)):
                if o:
                    pre[b].append((o[0], k) + o[1:])
            K[b].append(n1)
            bound[k0], bound[k1] = n0 + n1, g.INV_PROD
    end = []
    if not lazy_out:
        for k in range(8):
            if bound[k] > B0:
                if bound[k] <= 2 * B0:
                    m = (bound[k] + 1) // 2
                    end.append(("c", k, m))
                    bound[k] = m
                else:
                    end.append(("r", k))
                    bound[k] = 2
    return pre, K, end, bound


# ---- simulation ------------------------------------------------------------------------------------------------------------
def check_blocks():
    rnd = random.Random(11)
    run, St, sched = g.run, g.St, g.schedule
    for it in range(3000):
        q = g.rand_modulus(rnd) if it % 4 else (1 << 60) - 16383
        w = [rnd.randrange(q), rnd.randrange(q)]
        S = St()
        g.limb_ops(S, q)
        wp = [g.tw_ops(S, i, w[i], q) for i in (0, 1)]
        # two forward butterflies back to back in the one slot: lazy a < 13q, b any 64-bit value
        a = [rnd.randrange(13 * q), 13 * q - 1]
        b = [rnd.getrandbits(64), M64 if it % 7 == 0 else rnd.randrange(16 * q)]
        prs = ((0, 4), (3, 7))
        for i, (k0, k1) in enumerate(prs):
            S.set64(p(R(k0)), a[i]), S.set64(p(R(k1)), b[i])
        run(sched([g.fwd_stream(T, R(0), R(4), 0) + g.fwd_stream(T, R(3), R(7), 1)]), S)
        for i, (k0, k1) in enumerate(prs):
            Tm = g.shoup_trunc_ref(b[i], w[i], wp[i], q)
            assert Tm < 3 * q and Tm % q == b[i] * w[i] % q
            assert S.g64(p(R(k0))) == a[i] + Tm < 16 * q, "fwd a"
            assert S.g64(p(R(k1))) == a[i] - Tm + 3 * q, "fwd b"
        # exact in-place products (the slot's Z aliases Y: its high word must be zero)
        S.v[T.Z + 1] = 0
        x = [rnd.getrandbits(64), rnd.randrange(4 * q)]
        S.set64(p(R(1)), x[0]), S.set64(p(R(5)), x[1])
        run(sched([g.mul_stream(T, R(1), 0) + g.mul_stream(T, R(5), 1)]), S)
        for i, k in enumerate((1, 5)):
            assert S.g64(p(R(k))) == g.shoup_ref(x[i], w[i], wp[i], q) < 2 * q, "mul"
        assert S.v[T.Z + 1] == 0
        # reductions, 4 chains per block
        m = q << rnd.randrange(0, 4)
        S.ops.update({"m": m, "negm": (-m) & M64})
        xs = [rnd.randrange(2 * m) for _ in range(4)]
        for i, x0 in enumerate(xs):
            S.set64(p(R(4 + i)), x0)
        run(sched([g.csub_stream(i, R(4 + i)) for i in range(4)]), S)
        for i, x0 in enumerate(xs):
            assert S.g64(p(R(4 + i))) == (x0 if x0 < m else x0 - m), "csub"
        xs = [rnd.getrandbits(64), M64, rnd.randrange(16 * q), rnd.randrange(1, 17) * q - rnd.randrange(2)]
        for i, x0 in enumerate(xs):
            S.set64(p(R(i)), x0)
        run(sched([g.red_stream(i, R(i)) for i in range(4)]), S)
        for i, x0 in enumerate(xs):
            got = S.g64(p(R(i)))
            assert got < 2 * q and got % q == x0 % q, ("red", q, x0, got)
        xs = [rnd.randrange(16 * q) for _ in range(4)]
        for i, x0 in enumerate(xs):
            S.set64(p(R(i)), x0)
        run(sched([g.red_slow_stream(i, R(i)) for i in range(4)]), S)
        for i, x0 in enumerate(xs):
            got = S.g64(p(R(i)))
            assert got < 2 * q and got % q == x0 % q, "red (ladder)"
    # lazy inverse steps: every stage range a pass shape can ask for, with and without the closing reductions
    for (bLo, bHi) in ((0, 2), (1, 2), (2, 2)):
        for lazy in (False, True):
            pre, K, end, final = inv_plan(bLo, bHi, lazy)
            for it in range(60):
                q = (1 << 60) - 16383 if it < 4 else g.rand_modulus(rnd)
                slow = it % 5 == 4
                S = St()
                g.limb_ops(S, q)
                x = [rnd.randrange(3 * q) if it else 3 * q - 1 for _ in range(8)]
                ref = [v_ % q for v_ in x]
                bnd = [3] * 8
                for k in range(8):
                    S.set64(p(R(k)), x[k])

                def run_ops(ops):
                    for grp in g.op_blocks(ops):
                        run(sched([g.op_stream(i, op, slow) for i, op in enumerate(grp)]), S)
                        for op in grp:
                            bnd[op[1]] = op[2] if op[0] == "c" else 2
                for b in range(bLo, bHi + 1):
                    run_ops(pre[b])
                    prs = stage_pairs(b)
                    for j in range(0, 4, 2):
                        tw = [rnd.randrange(q), rnd.randrange(q)]
                        for i in (0, 1):
                            g.tw_ops(S, i, tw[i], q)
                        (a0, a1, _), (b0, b1, _) = prs[j], prs[j + 1]
                        for (u_, v_), m in (((a0, a1), K[b][j]), ((b0, b1), K[b][j + 1])):
                            uu, vv = S.g64(p(R(u_))), S.g64(p(R(v_)))
                            assert uu < bnd[u_] * q and vv < bnd[v_] * q and bnd[v_] <= m and bnd[u_] + m <= 16
                        run(sched([g.inv_lazy_stream(T, R(a0), R(a1), 0, f"%[k{K[b][j]}]")
                                   + g.inv_lazy_stream(T, R(b0), R(b1), 1, f"%[k{K[b][j + 1]}]")]), S)
                        for (u_, v_), w_ in (((a0, a1), tw[0]), ((b0, b1), tw[1])):
                            ru, rv = ref[u_], ref[v_]
                            ref[u_], ref[v_] = (ru + rv) % q, (ru - rv) * w_ % q
                            bnd[u_], bnd[v_] = bnd[u_] + bnd[v_], g.INV_PROD
                run_ops(end)
                assert bnd == final, (bnd, final)
                for k in range(8):
                    got = S.g64(p(R(k)))
                    assert got < final[k] * q and got % q == ref[k], ("lazy inverse step", bLo, bHi, lazy, k)
                    assert lazy or final[k] <= 3
                    assert final[k] <= 16
    return True


# ---- emission ----------------------------------------------------------------------------------------------------------------
def operand(name, cls):
    m = re.fullmatch(r"(wl|wh|pl|ph)([01])", name)
    if m:
        return f'[{name}] "{cls}"({m.group(1)[0]}{m.group(2)}{m.group(1)[1]})'
    fixed = {"nql": "c.nql", "nqh": "c.nqh", "twoq": "c.twoq", "ntwoq": "c.ntwoq", "threeq": "c.threeq", "redM": "c.redM",
             "redR": "c.redR", "m": "m", "negm": "negm"}
    if name in fixed:
        return f'[{name}] "s"({fixed[name]})'
    m = re.fullmatch(r"([kn])(\d+)", name)
    assert m, name
    mult = f"c.q * {m.group(2)}ull" if m.group(2) != "1" else "c.q"
    return f'[{name}] "s"({mult})' if m.group(1) == "k" else f'[{name}] "s"(0 - {mult})'


def asm_block(block, outs, cls="s", indent="    ", extra_in=()):
    text = "\n".join(f'{indent}    "{c["t"]}\\n\\t"' for c in block)
    names = []
    for c in block:
        for n in re.findall(r"%\[(\w+)\]", c["t"]):
            if n not in names:
                names.append(n)
    ins_ = [operand(n, cls) for n in names] + list(extra_in)
    clob = ", ".join(f'"{r}"' for r in TEMPS + SCLOB)
    return f"{indent}asm volatile(\n{text}\n{indent}    : {', '.join(outs)}\n{indent}    : {', '.join(ins_)}\n{indent}    : {clob});\n"


def pin(k, var):
    return f'"+{{v[{R(k)}:{R(k) + 1}]}}"({var})'


TW_ARGS = "uint64_t (&r)[8], const TwPair wa, const TwPair wb, const BflyConst c"
TW_BODY = "".join(f"    const uint32_t w{i}l = (uint32_t)w{n}.w, w{i}h = (uint32_t)(w{n}.w >> 32), p{i}l = (uint32_t)w{n}.wp, "
                  f"p{i}h = (uint32_t)(w{n}.wp >> 32);\n" for i, n in ((0, "a"), (1, "b")))


def emit_pair_fn(name, ka, kb, cls, stream_a, stream_b):
    block = g.schedule([stream_a + stream_b])
    outs = [pin(k, f"r[{k}]") for k in (ka[0], ka[1], kb[0], kb[1])]
    return f"__device__ __forceinline__ void {name}({TW_ARGS}) {{\n{TW_BODY}" + asm_block(block, outs, cls) + "}\n"


def emit_ops_body(items):
    body = ""
    for grp in g.op_blocks(items):
        outs = [pin(op[1], f"r[{op[1]}]") for op in grp]
        fast = asm_block(g.schedule([g.op_stream(i, op) for i, op in enumerate(grp)]), outs, indent="        ")
        if any(op[0] == "r" for op in grp):
            slow = asm_block(g.schedule([g.op_stream(i, op, True) for i, op in enumerate(grp)]), outs, indent="        ")
            body += f"    if (c.redR != 255u) {{\n{fast}    }}\n    else {{\n{slow}    }}\n"
        else:
            body += f"    {{\n{fast}    }}\n"
    return body


def emit_reduce_fn(name, items):
    return f"__device__ __forceinline__ void {name}(uint64_t (&r)[8], const BflyConst c) {{\n    (void)c;\n{emit_ops_body(items)}}}\n"


def emit_csub_fn(name, ks):
    block = g.schedule([g.csub_stream(i, R(k)) for i, k in enumerate(ks)])
    outs = [pin(k, f"r[{k}]") for k in ks]
    return (f"__device__ __forceinline__ void {name}(uint64_t (&r)[8], uint64_t m, uint64_t negm) {{\n"
            + asm_block(block, outs) + "}\n")


def emit_mul_fn(name, ka, kb, cls):
    block = g.schedule([g.mul_stream(T, R(ka), 0) + g.mul_stream(T, R(kb), 1)])
    outs = [pin(ka, f"r[{ka}]"), pin(kb, f"r[{kb}]")]
    zero = [f'"{{v{T.Z + 1}}}"(zero)']
    return (f"__device__ __forceinline__ void {name}({TW_ARGS}, uint32_t zero) {{\n{TW_BODY}"
            + asm_block(block, outs, cls, extra_in=zero).replace(f'"v{T.Z + 1}", ', "") + "}\n")


def op_table(ops, n):
    """fixed-size constexpr table of reduction ops: {kind (0 none, 1 csub, 2 estimate), residue, multiple of q}"""
    rows = [f"{{{1 if o[0] == 'c' else 2}, {o[1]}, {o[2] if o[0] == 'c' else 2}}}" for o in ops]
    rows += ["{0, 0, 0}"] * (n - len(rows))
    return "{" + ", ".join(rows) + "}"


RANGES = ((0, 2), (1, 2), (2, 2))  # inverse stage ranges of a radix-8 step: all three bits, or the top 2 / top 1 (the step on the wave field of tiles with 4 / 2 waves)


def main():
    check_blocks()
    H = []
    H.append("""// GENERATED by tools/gen_ntt8_asm.py — do not edit; edit the generator and re-run it.
// In-place gfx950 butterflies of the 8-residues-per-lane row pass (ntt_row8.h): residue k = v[48+2k:49+2k], the one butterfly
// slot's temporaries v[36:47], v[0:35] left to the compiler — 64 VGPRs per lane, eight waves per SIMD.
// Arithmetic: tools/gen_ntt_asm.py (the Shoup multiply of ModMulFastConst, ubintnat.h:1464-1469, inside the butterflies of
// transformnat-impl.h:303-374 / 512-625; truncated quotient, quotient-estimate reductions, lazy Gentleman-Sande sums).
// Part 1 (every build): the lazy-inverse plan as tables; part 2 (device build): the asm blocks, each simulated by the generator.
#ifndef FHE_NTT_BFLY8_PINNED_H
#define FHE_NTT_BFLY8_PINNED_H
namespace fhe {
namespace r8 {
struct RedOp {
    unsigned char kind, k, m;  // kind 0: none, 1: x = x < m q ? x : x - m q, 2: quotient estimate (x below 2q afterwards); k: residue
};
""")
    # plan tables: [range][lazy_out]
    for name, sel in (("Full", (0, 2)), ("Two", (1, 2)), ("One", (2, 2))):
        for lazy in (False, True):
            pre, K, end, final = inv_plan(sel[0], sel[1], lazy)
            tag = f"{name}{'Lazy' if lazy else ''}"
            pre_rows = ", ".join(op_table(pre.get(b, []), 4) for b in range(3))
            k_rows = ", ".join("{" + ", ".join(str(x) for x in (K.get(b, [0, 0, 0, 0]))) + "}" for b in range(3))
            H.append(f"constexpr RedOp kInvPre{tag}[3][4] = {{{pre_rows}}};\n"
                     f"constexpr unsigned char kInvK{tag}[3][4] = {{{k_rows}}};  // K (units of q) of u - v + K, per stage and butterfly\n"
                     f"constexpr RedOp kInvEnd{tag}[8] = {op_table(end, 8)};\n"
                     f"constexpr unsigned char kInvOut{tag}[8] = {{{', '.join(str(x) for x in final)}}};  // bounds on return (units of q)\n")
    H.append("}  // namespace r8\n}  // namespace fhe\n#if defined(__HIP_DEVICE_COMPILE__) && !defined(FHE_NO_BFLY_ASM)\nnamespace fhe {\nnamespace r8 {\n")
    # forward stages
    for b in range(3):
        prs = stage_pairs(b)
        for cls in ("v", "s"):
            fn = []
            for i in range(0, 4, 2):
                (a0, a1, ga), (b0, b1, gb) = prs[i], prs[i + 1]
                name = f"bfly2_fwd_{cls}_b{b}_{i // 2}"
                H.append(emit_pair_fn(name, (a0, a1), (b0, b1), cls, g.fwd_stream(T, R(a0), R(a1), 0), g.fwd_stream(T, R(b0), R(b1), 1)))
                fn.append(f"    {name}(r, w[{ga}], w[{gb}], c);\n")
            H.append(f"// stage on register bit {b}: twiddle g serves the butterflies whose index has (k >> {b + 1}) == g\n"
                     f"__device__ __forceinline__ void stage_fwd_{cls}_b{b}(uint64_t (&r)[8], const TwPair (&w)[4], const BflyConst c) {{\n"
                     + "".join(fn) + "}\n")
    # lazy inverse stages per range
    for name, sel in (("full", (0, 2)), ("two", (1, 2)), ("one", (2, 2))):
        pre, K, _, _ = inv_plan(sel[0], sel[1], True)  # (pre / K do not depend on lazy_out)
        for b in range(sel[0], sel[1] + 1):
            H.append(emit_reduce_fn(f"inv_pre_{name}_b{b}", pre[b]))
            prs = stage_pairs(b)
            for cls in ("v", "s"):
                fn = []
                for i in range(0, 4, 2):
                    (a0, a1, ga), (b0, b1, gb) = prs[i], prs[i + 1]
                    fname = f"bfly2_invl_{cls}_{name}_b{b}_{i // 2}"
                    H.append(emit_pair_fn(fname, (a0, a1), (b0, b1), cls,
                                          g.inv_lazy_stream(T, R(a0), R(a1), 0, f"%[k{K[b][i]}]"),
                                          g.inv_lazy_stream(T, R(b0), R(b1), 1, f"%[k{K[b][i + 1]}]")))
                    fn.append(f"    {fname}(r, w[{ga}], w[{gb}], c);\n")
                H.append(f"__device__ __forceinline__ void stage_invl_{cls}_{name}_b{b}(uint64_t (&r)[8], const TwPair (&w)[4], "
                         f"const BflyConst c) {{\n    inv_pre_{name}_b{b}(r, c);\n" + "".join(fn) + "}\n")
        _, _, end, _ = inv_plan(sel[0], sel[1], False)
        H.append(emit_reduce_fn(f"inv_end_{name}", end))
    # exact products of residues i and i|4 (the transform's last inverse stage, the fused epilogue)
    for i in range(4):
        H.append(emit_mul_fn(f"mul2_s_{i}", i, i | 4, "s"))
    for i in range(2):
        H.append(emit_csub_fn(f"csub4_{i}", [4 * i + j for j in range(4)]))
    H.append("""__device__ __forceinline__ void csub8(uint64_t (&r)[8], uint64_t m) {
    const uint64_t negm = 0 - m;
    csub4_0(r, m, negm);
    csub4_1(r, m, negm);
}
""")
    H.append(emit_reduce_fn("red8", [("r", k) for k in range(8)]))
    for b in range(3):
        H.append(emit_reduce_fn(f"red4_a{b}", [("r", k) for k in range(8) if not (k >> b) & 1]))
    H.append("}  // namespace r8\n}  // namespace fhe\n#endif\n#endif\n")
    text = "".join(H)
    if "--check" in sys.argv:
        if open(OUT).read() != text:
            print("ntt_bfly8_pinned.h is stale: run python tools/gen_ntt8_asm.py")
            return 1
        print("ntt_bfly8_pinned.h is up to date; all blocks simulated OK")
        return 0
    open(OUT, "w").write(text)
    nf = g.count(g.schedule([g.fwd_stream(T, R(0), R(1), 0)]))
    print(f"wrote {OUT}: forward butterfly {nf} VALU in one slot; inverse plans "
          + ", ".join(f"{s}: pre {sum(len(x) for x in inv_plan(*s)[0].values())} end {len(inv_plan(*s)[2])}" for s in RANGES))
    return 0


if __name__ == "__main__":
    sys.exit(main())
