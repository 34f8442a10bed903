#!/usr/bin/env python3
"""Generates openfhe-development_amd/csrc/ntt_bfly_pinned.h: hand-scheduled gfx950 butterflies that work IN PLACE on
the 16 residues of a lane, which are pinned to v[32:63] (residue k = v[32+2k : 33+2k]).

Why generated: gfx950 needs 64-bit VGPR operands in even-aligned pairs and inline asm cannot name the halves of a
compiler-allocated pair, so in-place code must name physical registers; one asm text per register pair is needed,
and two independent butterflies are interleaved per asm block so that the 2 wait states between a VALU carry
write and its reader are filled with useful work.  The generator also SIMULATES every emitted block on random
64-bit inputs against the butterfly arithmetic it replaces (python big ints) before writing the header.

Usage:  python tools/gen_ntt_asm.py        (rewrites the header; exits non-zero if a simulated block is wrong)
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "openfhe-development_amd", "csrc", "ntt_bfly_pinned.h")
M32, M64 = (1 << 32) - 1, (1 << 64) - 1
DATA0 = 32  # residue k lives in v[DATA0+2k : DATA0+2k+1]
DEAD = "s[40:41]"  # carry-outs nobody reads


def R(k):
    return DATA0 + 2 * k


class T:
    """temporaries of one butterfly slot (slot 0 / slot 1 run interleaved)"""

    def __init__(self, slot):
        b = 64 + 16 * slot
        self.Z = b        # v[Z:Z+1]: mul_hi result, Z+1 holds 0 (pinned input)
        self.L = b + 2    # Bm, later L
        self.C = b + 4    # C'
        self.H = b + 6    # {C'.hi, carry}
        self.Q = b + 8
        self.X = b + 10   # cross terms
        self.Y = b + 12   # inverse: y = u - v + 2q
        self.D = b + 14   # inverse: s - 2q
        self.c0 = f"s[{42 + 4 * slot}:{43 + 4 * slot}]"
        self.c1 = f"s[{44 + 4 * slot}:{45 + 4 * slot}]"
        self.slot = slot


def p(r):
    return f"v[{r}:{r + 1}]"


def v(r):
    return f"v{r}"


# instruction = (text, reads_sgpr_set, writes_sgpr_set, simfn)
def ins(text, rd=(), wr=(), sim=None):
    return {"t": text, "rd": set(rd), "wr": set(wr), "sim": sim}


# ---- simulator helpers ------------------------------------------------------------------------------
class St:
    def __init__(self):
        self.v = {}
        self.s = {}
        self.ops = {}

    def g(self, name):  # 32-bit source: register "vN" or operand "%[x]"
        if name.startswith("%["):
            return self.ops[name[2:-1]] & M32
        if name.startswith("v"):
            return self.v[int(name[1:])]
        return int(name)

    def g64(self, name):
        if name.startswith("%["):
            return self.ops[name[2:-1]] & M64
        if name.startswith("v["):
            lo = int(name[2:name.index(":")])
            return self.v[lo] | (self.v[lo + 1] << 32)
        return int(name)

    def set64(self, name, val):
        lo = int(name[2:name.index(":")])
        self.v[lo], self.v[lo + 1] = val & M32, (val >> 32) & M32


def mad(d, sd, a, b, c):
    def sim(S):
        r = S.g(a) * S.g(b) + S.g64(c)
        S.set64(d, r & M64)
        S.s[sd] = r >> 64
    return ins(f"v_mad_u64_u32 {d}, {sd}, {a}, {b}, {c}", wr=[sd], sim=sim)


def mulhi(d, a, b):
    def sim(S):
        S.v[int(d[1:])] = (S.g(a) * S.g(b)) >> 32
    return ins(f"v_mul_hi_u32 {d}, {a}, {b}", sim=sim)


def mullo(d, a, b):
    def sim(S):
        S.v[int(d[1:])] = (S.g(a) * S.g(b)) & M32
    return ins(f"v_mul_lo_u32 {d}, {a}, {b}", sim=sim)


def lshr32(d, amt, a):  # d = a >> amt (amt: scalar operand or literal, low 5 bits)
    def sim(S):
        S.v[int(d[1:])] = S.g(a) >> (S.g(amt) & 31)
    return ins(f"v_lshrrev_b32 {d}, {amt}, {a}", sim=sim)


def mov(d, a):
    def sim(S):
        S.v[int(d[1:])] = S.g(a)
    return ins(f"v_mov_b32 {d}, {a}", sim=sim)


def cnd01(d, sc):
    def sim(S):
        S.v[int(d[1:])] = 1 if S.s[sc] else 0
    return ins(f"v_cndmask_b32_e64 {d}, 0, 1, {sc}", rd=[sc], sim=sim)


def cnd(d, a, b, sc):  # d = sc ? b : a
    def sim(S):
        S.v[int(d[1:])] = S.g(b) if S.s[sc] else S.g(a)
    return ins(f"v_cndmask_b32_e64 {d}, {a}, {b}, {sc}", rd=[sc], sim=sim)


def lshladd(d, a, sh, c):
    def sim(S):
        S.set64(d, ((S.g64(a) << sh) + S.g64(c)) & M64)
    return ins(f"v_lshl_add_u64 {d}, {a}, {sh}, {c}", sim=sim)


def add32(d, a, b):
    def sim(S):
        S.v[int(d[1:])] = (S.g(a) + S.g(b)) & M32
    return ins(f"v_add_u32 {d}, {a}, {b}", sim=sim)


def subco(d, sd, a, b):
    def sim(S):
        r = S.g(a) - S.g(b)
        S.v[int(d[1:])] = r & M32
        S.s[sd] = 1 if r < 0 else 0
    return ins(f"v_sub_co_u32_e64 {d}, {sd}, {a}, {b}", wr=[sd], sim=sim)


def subbco(d, sd, a, b, sc):
    def sim(S):
        r = S.g(a) - S.g(b) - S.s[sc]
        S.v[int(d[1:])] = r & M32
        S.s[sd] = 1 if r < 0 else 0
    return ins(f"v_subb_co_u32_e64 {d}, {sd}, {a}, {b}, {sc}", rd=[sc], wr=[sd], sim=sim)


def addco(d, sd, a, b):
    def sim(S):
        r = S.g(a) + S.g(b)
        S.v[int(d[1:])] = r & M32
        S.s[sd] = r >> 32
    return ins(f"v_add_co_u32_e64 {d}, {sd}, {a}, {b}", wr=[sd], sim=sim)


def addcco(d, sd, a, b, sc):
    def sim(S):
        r = S.g(a) + S.g(b) + S.s[sc]
        S.v[int(d[1:])] = r & M32
        S.s[sd] = r >> 32
    return ins(f"v_addc_co_u32_e64 {d}, {sd}, {a}, {b}, {sc}", rd=[sc], wr=[sd], sim=sim)


def mov64(d, a):
    def sim(S):
        S.set64(d, S.g64(a))
    return ins(f"v_mov_b64 {d}, {a}", sim=sim)


def cmplt64(sd, a, b):
    def sim(S):
        S.s[sd] = 1 if S.g64(a) < S.g64(b) else 0
    return ins(f"v_cmp_lt_u64_e64 {sd}, {a}, {b}", wr=[sd], sim=sim)


# ---- instruction streams -----------------------------------------------------------------------------
# Round 4 arithmetic (profiles/r04_sweeps.md):
#  * TRUNCATED Shoup quotient in the butterflies: Q' = yh*ph + floor((yh*pl + yl*ph) / 2^32) — the low product
#    yl*pl (the slowest instruction of the round-3 butterfly, v_mul_hi_u32) is dropped, so Q' is the exact quotient
#    or one less and the product output lies in [0, 3q) instead of [0, 2q);
#  * reductions of a lazily grown residue (anything below 2^64) to [0, 2q) by ONE quotient estimate
#    k = (x.hi * redM) >> (32 + redR), redM = floor(2^(64+redR) / q), redR = bitlen(q) - 33, x -= k*q
#    (5 instructions; k is floor(x/q) or one less) instead of a ladder of conditional subtractions (4 per level);
#  * the multiplications that end a transform (N^-1, the fused epilogue's constant) keep the exact quotient.
def shoup_tail(t, yl, yh, w, dst, addend, exact=False):
    """dst(pair) = lo64(y*w + Q*nq) + addend, y = (yl, yh) registers; w = operand suffix.
    exact: Q = hi64(y*w') (result < 2q);  else Q = the truncated quotient above (result < 3q)"""
    wl, wh, pl, ph = (f"%[{n}{w}]" for n in ("wl", "wh", "pl", "ph"))
    if exact:
        head = [
            mulhi(v(t.Z), yl, pl),
            mad(p(t.X), DEAD, yl, wh, "0"),
            mad(p(t.L), DEAD, yh, pl, p(t.Z)),
        ]
    else:
        head = [
            mad(p(t.L), DEAD, yh, pl, "0"),
            mad(p(t.X), DEAD, yl, wh, "0"),
        ]
    head += [
        mad(p(t.C), t.c0, yl, ph, p(t.L)),
        mad(p(t.X), DEAD, yh, wl, p(t.X)),
        mov(v(t.H), v(t.C + 1)),
        cnd01(v(t.H + 1), t.c0),
        mad(p(t.Q), DEAD, yh, ph, p(t.H)),
        mad(p(t.L), DEAD, yl, wl, addend),
    ]
    return head, [
        mad(p(t.X), DEAD, v(t.Q), "%[nqh]", p(t.X)),
        mad(p(t.X), DEAD, v(t.Q + 1), "%[nql]", p(t.X)),
        add32(v(t.L + 1), v(t.L + 1), v(t.X)),
        mad(dst, DEAD, v(t.Q), "%[nql]", p(t.L)),
    ]


FWD_GROW = 3  # a forward butterfly adds at most 3q to the bound of its `a` input (truncated quotient)
INV_PROD = 3  # bound of an inverse butterfly's product output


def fwd_stream(t, A, B, w):
    """a' = a + T, b' = a - T + 3q with T = shoup(b, w) in [0,3q); a = v[A:A+1], b = v[B:B+1] (any 64-bit value)"""
    head, tail = shoup_tail(t, v(B), v(B + 1), w, p(A), p(A))
    return head + [lshladd(p(B), p(A), 1, "%[threeq]")] + tail + [
        subco(v(B), t.c0, v(B), v(A)),
        subbco(v(B + 1), t.c0, v(B + 1), v(A + 1), t.c0),
    ]


def inv_lazy_stream(t, A, B, w, kop):
    """Gentleman-Sande butterfly without the reduction of the sum: a' = u + v, b' = shoup(u - v + K, w) < 3q with K = the
    bound of v (a multiple of q given as the scalar operand `kop`); the caller tracks the bounds (inv_plan)"""
    pre = [
        lshladd(p(t.Y), p(A), 0, kop),
        lshladd(p(A), p(A), 0, p(B)),
        subco(v(t.Y), t.c0, v(t.Y), v(B)),
        subbco(v(t.Y + 1), t.c0, v(t.Y + 1), v(B + 1), t.c0),
    ]
    head, tail = shoup_tail(t, v(t.Y), v(t.Y + 1), w, p(B), "0")
    return pre + head + tail


def mul_stream(t, A, w):
    """x = shoup(x, w) in place with the EXACT quotient (< 2q): last inverse stage (the caller forms u+v / u-v+K first)
    and the fused epilogue; x is only overwritten by the last instruction, after every read of its halves"""
    head, tail = shoup_tail(t, v(A), v(A + 1), w, p(A), "0", exact=True)
    return head + tail


CSUB_TMP = (68, 70, 72, 74)  # pairs used by the 4 interleaved reduction chains


def csub_stream(i, A):
    """x = x < m ? x : x - m   (m, -m as scalar pairs); chain i of 4"""
    return csub_chain_stream(i, A, "%[m]", "%[negm]")


def csub_chain_stream(i, A, mop, nop):
    """x = x < m ? x : x - m with explicit operand names for m and -m; chain i of up to 4"""
    D, c = CSUB_TMP[i], f"s[{42 + 2 * i}:{43 + 2 * i}]"
    return [
        lshladd(p(D), p(A), 0, nop),
        cmplt64(c, p(A), mop),
        cnd(v(A), v(D), v(A), c),
        cnd(v(A + 1), v(D + 1), v(A + 1), c),
    ]


def red_stream(i, A):
    """x (any 64-bit value) -> x - k*q in [0, 2q), k = (x.hi * redM) >> (32 + redR) = floor(x/q) or one less; chain i of 4.
    Needs bitlen(q) >= 36 (redR >= 3): the limbs below that take red_slow_stream"""
    K = CSUB_TMP[i]
    return [
        mulhi(v(K), v(A + 1), "%[redM]"),
        lshr32(v(K), "%[redR]", v(K)),
        mad(p(A), DEAD, v(K), "%[nql]", p(A)),
        mullo(v(K), v(K), "%[nqh]"),
        add32(v(A + 1), v(A + 1), v(K)),
    ]


def red_slow_stream(i, A):
    """the same contract for x < 16q by three conditional subtractions (8q, 4q, 2q): moduli below 2^35"""
    out = []
    for m in (8, 4, 2):
        out += csub_chain_stream(i, A, f"%[k{m}]", f"%[n{m}]")
    return out


def op_stream(i, op, slow=False):
    """reduction op ('c', k, m): csub of m*q on residue k;  ('r', k): residue k below 2q"""
    if op[0] == "c":
        return csub_chain_stream(i, R(op[1]), f"%[k{op[2]}]", f"%[n{op[2]}]")
    return (red_slow_stream if slow else red_stream)(i, R(op[1]))


COST_C, COST_R = 18, 23  # issue cycles of a conditional subtraction / a quotient-estimate reduction (profiles/r04_seqbench.json)


def reduce_options(b):
    """ways of lowering a bound b (units of q): (cost, new bound, op without the residue index)"""
    opts = [(0, b, None)]
    if b > 1:
        opts.append((COST_C, (b + 1) // 2, ("c", (b + 1) // 2)))
    if b > 2:
        opts.append((COST_R, 2, ("r",)))
    return opts


def inv_plan(bLo, bHi, ends=False, B0=3, lazy_out=False):
    """Bounds (units of q) of the 16 residues through the inverse stages bLo..bHi: inputs < B0*q; a' = u + v adds the bounds,
    b' = shoup(.) is < INV_PROD*q again.  u + v and u - v + K (K = the bound of v) must stay below 2^64 > 16q: a pair
    whose bounds add up to more than 16 is reduced first, by the cheapest of {nothing, one conditional subtraction of
    half the bound, a quotient-estimate reduction} on either side.  Returns
      pre[b] = reduction ops before stage b,   K[b] = [m per butterfly, in stage_pairs order],
      end    = ops that bring every residue below B0*q after the last stage (none when lazy_out: the next PASS reduces),
      bound  = the bounds on return.
    ends: the last stage is the transform's final multiplication stage (exact products, outputs < 2q)."""
    bound = [B0] * 16
    pre, K = {}, {}
    for b in range(bLo, bHi + 1):
        pre[b], K[b] = [], []
        for (k0, k1, _g) in stage_pairs(b):
            best = None
            for (c0, n0, o0) in reduce_options(bound[k0]):
                for (c1, n1, o1) in reduce_options(bound[k1]):
                    if n0 + n1 <= 16 and (best is None or (c0 + c1, n0 + n1) < best[0]):
                        best = ((c0 + c1, n0 + n1), n0, n1, o0, o1)
            _, n0, n1, o0, o1 = best
            for k, o in ((k0, o0), (k1, o1)):
                if o:
                    pre[b].append((o[0], k) + o[1:])
            K[b].append(n1)
            if ends and b == bHi:
                bound[k0] = bound[k1] = 2
            else:
                bound[k0], bound[k1] = n0 + n1, INV_PROD
    end = []
    if not lazy_out and not ends:
        for k in range(16):
            if bound[k] > B0:
                if bound[k] <= 2 * B0:
                    m = (bound[k] + 1) // 2
                    end.append(("c", k, m))
                    bound[k] = m
                else:
                    end.append(("r", k))
                    bound[k] = 2
    return pre, K, end, bound


def op_blocks(items):
    """reduction ops in execution order -> blocks of <= 4 interleaved chains (a block never holds a residue twice)"""
    blocks, cur = [], []
    for op in items:
        if len(cur) == 4 or any(op[1] == o2[1] for o2 in cur):
            blocks.append(cur)
            cur = []
        cur.append(op)
    if cur:
        blocks.append(cur)
    return blocks


def reduce192_stream():
    """(c0 + c1*2^32 + c2*2^64 + k0*2^64 + k1*2^96) mod p for a sum < 2^128, canonical:  hi*R + lo with R = 2^64 mod p
    (Shoup), lo reduced with mu64 = floor(2^64/p), two conditional subtractions.  Operands: c0l..c2h, k0, k1 (VGPR
    words), scalars Rl,Rh,Rpl,Rph (R and its Shoup precon), mul,muh (mu64), nql,nqh (2^64-p), p, twop, np, ntwop."""
    t = T(0)
    A1, A2, A3 = 76, 78, 79  # a1 | a2, a3 (t.Y = 76..77, t.D = 78..79)
    LO = 76                  # pair {c0l copy, a1}: v76 = c0l, v77 = a1
    A1 = 77
    c0, c1 = t.c0, t.c1
    fold = [
        mov(v(LO), "%[c0l]"),
        addco(v(A1), c0, "%[c0h]", "%[c1l]"),
        addco(v(A2), c1, "%[c2l]", "%[k0]"),
        addcco(v(A2), c0, v(A2), "%[c1h]", c0),
        addcco(v(A3), c1, "%[c2h]", "%[k1]", c1),
        addcco(v(A3), c0, v(A3), "0", c0),
    ]
    # t = shoup(hi, R): temps Z,L,C,H,Q,X of slot 0; result pair -> t.L... use dst = p(t.C) after C is dead? keep separate: slot-1 regs
    u = T(1)
    hw = {"wl": "%[Rl]", "wh": "%[Rh]", "pl": "%[Rpl]", "ph": "%[Rph]"}
    yl, yh = v(A2), v(A3)
    sh = [
        mulhi(v(t.Z), yl, hw["pl"]),
        mad(p(t.X), DEAD, yl, hw["wh"], "0"),
        mad(p(t.L), DEAD, yh, hw["pl"], p(t.Z)),
        mad(p(t.C), c1, yl, hw["ph"], p(t.L)),
        mad(p(t.X), DEAD, yh, hw["wl"], p(t.X)),
        mov(v(t.H), v(t.C + 1)),
        cnd01(v(t.H + 1), c1),
        mad(p(t.Q), DEAD, yh, hw["ph"], p(t.H)),
        mad(p(t.L), DEAD, yl, hw["wl"], "0"),
        mad(p(t.X), DEAD, v(t.Q), "%[nqh]", p(t.X)),
        mad(p(t.X), DEAD, v(t.Q + 1), "%[nql]", p(t.X)),
        add32(v(t.L + 1), v(t.L + 1), v(t.X)),
        mad(p(u.L), DEAD, v(t.Q), "%[nql]", p(t.L)),  # t in u.L, [0,2p)
    ]
    # r = lo - hi64(lo*mu64)*p   (lo = v[76:77])
    lo = [
        mulhi(v(u.Z), v(LO), "%[mul]"),
        mad(p(u.C), DEAD, v(A1), "%[mul]", p(u.Z)),
        mad(p(u.H), u.c0, v(LO), "%[muh]", p(u.C)),
        mov(v(u.Q), v(u.H + 1)),
        cnd01(v(u.Q + 1), u.c0),
        mad(p(u.X), DEAD, v(A1), "%[muh]", p(u.Q)),       # Q = u.X
        mad(p(u.C), DEAD, v(u.X), "%[nql]", p(LO)),       # L = Ql*nql + lo
        mad(p(u.H), DEAD, v(u.X), "%[nqh]", "0"),
        mad(p(u.H), DEAD, v(u.X + 1), "%[nql]", p(u.H)),
        add32(v(u.C + 1), v(u.C + 1), v(u.H)),             # r in u.C, [0,2p)
    ]
    fin = [
        lshladd(p(u.L), p(u.L), 0, p(u.C)),                # s = t + r < 4p
        lshladd(p(u.Y), p(u.L), 0, "%[ntwop]"),
        cmplt64(u.c1, p(u.L), "%[twop]"),
        cnd(v(u.L), v(u.Y), v(u.L), u.c1),
        cnd(v(u.L + 1), v(u.Y + 1), v(u.L + 1), u.c1),
        lshladd(p(u.Y), p(u.L), 0, "%[np]"),
        cmplt64(u.c1, p(u.L), "%[p]"),
        cnd(v(u.L), v(u.Y), v(u.L), u.c1),
        cnd(v(u.L + 1), v(u.Y + 1), v(u.L + 1), u.c1),
        mov64("%[out]", p(u.L)),
    ]
    return fold, sh, lo, fin


def schedule(streams):
    """merge instruction streams round-robin; an SGPR written by a VALU instruction is not read by the next two
    issued instructions (gfx950 needs 2 wait states there)"""
    out, pos, recent = [], [0] * len(streams), []  # recent: sgpr write sets of the last 2 emitted
    nxt = 0
    while any(pos[i] < len(s) for i, s in enumerate(streams)):
        done = False
        for k in range(len(streams)):
            i = (nxt + k) % len(streams)
            if pos[i] >= len(streams[i]):
                continue
            c = streams[i][pos[i]]
            if any(c["rd"] & w for w in recent[-2:]):
                continue
            out.append(c)
            recent.append(c["wr"])
            pos[i] += 1
            nxt = (i + 1) % len(streams)
            done = True
            break
        if not done:
            out.append(ins("s_nop 0", sim=lambda S: None))
            recent.append(set())
    return out


# ---- simulation against the arithmetic being replaced ------------------------------------------------
def shoup_ref(y, w, wp, q):
    Q = (y * wp) >> 64
    return (y * w - Q * q) & M64


def shoup_trunc_ref(y, w, wp, q):
    yl, yh, pl, ph = y & M32, y >> 32, wp & M32, wp >> 32
    Q = yh * ph + ((yh * pl + yl * ph) >> 32)
    return (y * w - Q * q) & M64


def run(block, S):
    for c in block:
        c["sim"](S)


def red_consts(q):
    """(redM, redR) of a modulus, or None below 36 bits (ntt_static.h takes the conditional-subtraction ladder there)"""
    L = q.bit_length()
    if L < 36:
        return None
    r = L - 33
    return (1 << (64 + r)) // q, r


def limb_ops(S, q):
    S.ops.update({"nql": (-q) & M32, "nqh": ((-q) & M64) >> 32, "twoq": 2 * q, "ntwoq": (-2 * q) & M64, "threeq": 3 * q})
    for m in range(1, 17):
        S.ops[f"k{m}"], S.ops[f"n{m}"] = (m * q) & M64, (-m * q) & M64
    rc = red_consts(q)
    if rc:
        S.ops["redM"], S.ops["redR"] = rc


def tw_ops(S, i, w, q):
    wp = (w << 64) // q
    S.ops.update({f"wl{i}": w & M32, f"wh{i}": w >> 32, f"pl{i}": wp & M32, f"ph{i}": wp >> 32})
    return wp


def rand_modulus(rnd, bits=None):
    bits = bits or rnd.choice([60, 60, 59, 59, 55, 50, 45, 40, 36])
    return rnd.getrandbits(bits) | (1 << (bits - 1)) | 1


def check_blocks():
    rnd = random.Random(7)
    for it in range(4000):
        q = rand_modulus(rnd) if it % 4 else (1 << 60) - 16383  # (the reference's largest 60-bit prime at N = 2^16: 16q = 2^64 - 262128)
        w = [rnd.randrange(q), rnd.randrange(q)]
        S = St()
        limb_ops(S, q)
        wp = [tw_ops(S, i, w[i], q) for i in (0, 1)]
        t0, t1 = T(0), T(1)
        S.v[t0.Z + 1] = S.v[t1.Z + 1] = 0
        # forward pair: lazy inputs a < 13q, b any 64-bit value
        a = [rnd.randrange(13 * q), 13 * q - 1]
        b = [rnd.getrandbits(64), M64 if it % 7 == 0 else rnd.randrange(16 * q)]
        for i, (k0, k1) in enumerate(((0, 8), (5, 13))):
            S.set64(p(R(k0)), a[i]), S.set64(p(R(k1)), b[i])
        run(schedule([fwd_stream(t0, R(0), R(8), 0), fwd_stream(t1, R(5), R(13), 1)]), S)
        for i, (k0, k1) in enumerate(((0, 8), (5, 13))):
            Tm = shoup_trunc_ref(b[i], w[i], wp[i], q)
            assert Tm < 3 * q and Tm % q == b[i] * w[i] % q, "truncated Shoup product"
            assert S.g64(p(R(k0))) == a[i] + Tm < 16 * q, "fwd a"
            assert S.g64(p(R(k1))) == a[i] - Tm + 3 * q, "fwd b"
        # in-place multiply pair (exact quotient), any 64-bit input
        x = [rnd.getrandbits(64), rnd.randrange(4 * q)]
        S.set64(p(R(1)), x[0]), S.set64(p(R(9)), x[1])
        run(schedule([mul_stream(t0, R(1), 0), mul_stream(t1, R(9), 1)]), S)
        for i, k in enumerate((1, 9)):
            assert S.g64(p(R(k))) == shoup_ref(x[i], w[i], wp[i], q) < 2 * q
        # conditional subtraction, 4 residues per block
        m = q << rnd.randrange(0, 4)
        S.ops.update({"m": m, "negm": (-m) & M64})
        xs = [rnd.randrange(2 * m) for _ in range(4)]
        for i, x0 in enumerate(xs):
            S.set64(p(R(4 + i)), x0)
        run(schedule([csub_stream(i, R(4 + i)) for i in range(4)]), S)
        for i, x0 in enumerate(xs):
            assert S.g64(p(R(4 + i))) == (x0 if x0 < m else x0 - m), "csub"
        # quotient-estimate reduction: any 64-bit value -> [0, 2q); and the ladder for x < 16q
        xs = [rnd.getrandbits(64), M64, rnd.randrange(16 * q), rnd.randrange(q) + rnd.randrange(16) * q]
        if it % 3 == 0:
            xs[3] = rnd.randrange(1, 17) * q - rnd.randrange(2)  # multiples of q and their predecessors
        for i, x0 in enumerate(xs):
            S.set64(p(R(4 + i)), x0)
        run(schedule([red_stream(i, R(4 + i)) for i in range(4)]), S)
        for i, x0 in enumerate(xs):
            got = S.g64(p(R(4 + i)))
            assert got < 2 * q and got % q == x0 % q, ("red", q, x0, got)
        xs = [rnd.randrange(16 * q) for _ in range(4)]
        for i, x0 in enumerate(xs):
            S.set64(p(R(4 + i)), x0)
        run(schedule([red_slow_stream(i, R(4 + i)) for i in range(4)]), S)
        for i, x0 in enumerate(xs):
            got = S.g64(p(R(4 + i)))
            assert got < 2 * q and got % q == x0 % q, "red (ladder)"
    # lazy inverse steps: every (bLo, bHi), with / without the final multiplication stage, lazy output, fast and ladder reductions
    for bLo in range(4):
        for bHi in range(bLo, 4):
            for ends, lazy in (((False, False), (False, True), (True, False)) if bHi == 3 else ((False, False), (False, True))):
                pre, K, end, final = inv_plan(bLo, bHi, ends, lazy_out=lazy)
                for it in range(40):
                    q = (1 << 60) - 16383 if it < 4 else rand_modulus(rnd)
                    slow = it % 5 == 4
                    S = St()
                    S.v[T(0).Z + 1] = S.v[T(1).Z + 1] = 0
                    limb_ops(S, q)
                    B0 = 3
                    x = [rnd.randrange(B0 * q) if it else B0 * q - 1 for _ in range(16)]
                    ref = [v_ % q for v_ in x]
                    bnd = [B0] * 16
                    for k in range(16):
                        S.set64(p(R(k)), x[k])

                    def run_ops(ops):
                        for grp in op_blocks(ops):
                            run(schedule([op_stream(i, op, slow) for i, op in enumerate(grp)]), S)
                            for op in grp:
                                bnd[op[1]] = op[2] if op[0] == "c" else 2
                    for b in range(bLo, bHi + 1):
                        run_ops(pre[b])
                        prs = stage_pairs(b)
                        for j in range(0, 8, 2):
                            tw = [rnd.randrange(q), rnd.randrange(q)]
                            for i in (0, 1):
                                tw_ops(S, i, tw[i], q)
                            (a0, a1, _), (b0, b1, _) = prs[j], prs[j + 1]
                            for (u_, v_), m in (((a0, a1), K[b][j]), ((b0, b1), K[b][j + 1])):
                                uu, vv = S.g64(p(R(u_))), S.g64(p(R(v_)))
                                assert uu < bnd[u_] * q and vv < bnd[v_] * q and bnd[v_] <= m and bnd[u_] + m <= 16
                            if ends and b == bHi:  # final stage: sums / differences formed by the caller, then multiplied
                                for (u_, v_), m in (((a0, a1), K[b][j]), ((b0, b1), K[b][j + 1])):
                                    uu, vv = S.g64(p(R(u_))), S.g64(p(R(v_)))
                                    S.set64(p(R(u_)), uu + vv), S.set64(p(R(v_)), uu - vv + m * q)
                                run(schedule([mul_stream(T(0), R(a0), 0), mul_stream(T(1), R(b0), 1)]), S)
                                run(schedule([mul_stream(T(0), R(a1), 0), mul_stream(T(1), R(b1), 1)]), S)
                                for (u_, v_), w_ in (((a0, a1), tw[0]), ((b0, b1), tw[1])):
                                    ru, rv = ref[u_], ref[v_]
                                    ref[u_], ref[v_] = (ru + rv) * w_ % q, (ru - rv) * w_ % q
                                    bnd[u_] = bnd[v_] = 2
                            else:
                                run(schedule([inv_lazy_stream(T(0), R(a0), R(a1), 0, f"%[k{K[b][j]}]"),
                                              inv_lazy_stream(T(1), R(b0), R(b1), 1, f"%[k{K[b][j + 1]}]")]), S)
                                for (u_, v_), w_ in (((a0, a1), tw[0]), ((b0, b1), tw[1])):
                                    ru, rv = ref[u_], ref[v_]
                                    ref[u_], ref[v_] = (ru + rv) % q, (ru - rv) * w_ % q
                                    bnd[u_], bnd[v_] = bnd[u_] + bnd[v_], INV_PROD
                    run_ops(end)
                    assert bnd == final, (bnd, final)
                    for k in range(16):
                        got = S.g64(p(R(k)))
                        assert got < final[k] * q and got % q == ref[k], ("lazy inverse step", bLo, bHi, ends, lazy, k)
                        assert lazy or final[k] <= (2 if ends else 3)
    # 192-bit column sums -> canonical residue
    for it in range(4000):
        bits = rnd.choice([28, 45, 59, 60])
        pm = rnd.getrandbits(bits) | (1 << (bits - 1)) | 1
        n = rnd.randrange(1, 17)
        c0 = c1 = c2 = 0
        for _ in range(n):
            a, b = rnd.randrange(pm) if rnd.random() < 0.9 else pm - 1, rnd.randrange(pm) if rnd.random() < 0.9 else pm - 1
            c0 += (a & M32) * (b & M32)
            c1 += (a & M32) * (b >> 32) + (a >> 32) * (b & M32)
            c2 += (a >> 32) * (b >> 32)
        total = c0 + (c1 << 32) + (c2 << 64)
        assert total < 1 << 128
        k0, c0 = c0 >> 64, c0 & M64
        k1, c1 = c1 >> 64, c1 & M64
        assert c2 < 1 << 64
        Rm = (1 << 64) % pm
        Rp = (Rm << 64) // pm
        mu = (1 << 64) // pm
        if mu > M64:
            mu = M64
        S = St()
        S.v[T(0).Z + 1] = S.v[T(1).Z + 1] = 0
        S.ops = {"c0l": c0 & M32, "c0h": c0 >> 32, "c1l": c1 & M32, "c1h": c1 >> 32, "c2l": c2 & M32, "c2h": c2 >> 32,
                 "k0": k0, "k1": k1, "Rl": Rm & M32, "Rh": Rm >> 32, "Rpl": Rp & M32, "Rph": Rp >> 32, "mul": mu & M32,
                 "muh": mu >> 32, "nql": (-pm) & M32, "nqh": ((-pm) & M64) >> 32, "p": pm, "twop": 2 * pm,
                 "np": (-pm) & M64, "ntwop": (-2 * pm) & M64}
        fold, sh, lo, fin = reduce192_stream()
        outreg = {}

        def run_out(block):
            for c in block:
                if c["t"].startswith("v_mov_b64 %[out]"):
                    outreg["v"] = S.g64(p(T(1).L))
                else:
                    c["sim"](S)
        run_out(schedule([fold]) + schedule([sh, lo]) + schedule([fin]))
        assert outreg["v"] == total % pm, "reduce192"
    return True


# ---- emission ----------------------------------------------------------------------------------------
import re


def clobbers(nslots):
    regs = []
    for s in range(nslots):
        t = T(s)
        regs += [f"v{r}" for r in range(t.Z, t.Z + 16) if r != t.Z + 1]
        regs += [f"s{r}" for r in range(42 + 4 * s, 46 + 4 * s)]
    return regs + ["s40", "s41"]


def operand(name, cls):
    """C++ input operand of an asm block for the %[name] used in its text"""
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


def asm_block(block, outs, nslots, cls="s", indent="    "):
    text = "\n".join(f'{indent}    "{c["t"]}\\n\\t"' for c in block)
    names = []
    for c in block:
        for n in re.findall(r"%\[(\w+)\]", c["t"]):
            if n not in names:
                names.append(n)
    ins_ = [operand(n, cls) for n in names]
    if nslots:
        ins_ += ZERO_IN
    regs = clobbers(nslots) if nslots else [f"v{r}" for r in range(68, 76)] + [f"s{r}" for r in range(42, 50)]
    c = ", ".join(f'"{r}"' for r in regs)
    return f"{indent}asm volatile(\n{text}\n{indent}    : {', '.join(outs)}\n{indent}    : {', '.join(ins_)}\n{indent}    : {c});\n"


def pin(k, var):
    return f'"+{{v[{R(k)}:{R(k) + 1}]}}"({var})'


ZERO_IN = ['"{v65}"(z.z0)', '"{v81}"(z.z1)']
TW_ARGS = "uint64_t (&r)[16], const TwPair wa, const TwPair wb, const BflyConst c, const BflyZero z"
TW_BODY = "".join(f"    const uint32_t w{i}l = (uint32_t)w{n}.w, w{i}h = (uint32_t)(w{n}.w >> 32), p{i}l = (uint32_t)w{n}.wp, "
                  f"p{i}h = (uint32_t)(w{n}.wp >> 32);\n" for i, n in ((0, "a"), (1, "b")))


def emit_pair_fn(name, ka, kb, cls):
    """two forward butterflies (r[ka[0]], r[ka[1]]) with twiddle 0 and (r[kb[0]], r[kb[1]]) with twiddle 1"""
    block = schedule([fwd_stream(T(0), R(ka[0]), R(ka[1]), 0), fwd_stream(T(1), R(kb[0]), R(kb[1]), 1)])
    outs = [pin(k, f"r[{k}]") for k in (ka[0], ka[1], kb[0], kb[1])]
    return f"__device__ __forceinline__ void {name}({TW_ARGS}) {{\n{TW_BODY}" + asm_block(block, outs, 2, cls) + "}\n"


def emit_lazy_pair_fn(name, ka, kb, cls, Ka, Kb):
    block = schedule([inv_lazy_stream(T(0), R(ka[0]), R(ka[1]), 0, f"%[k{Ka}]"),
                      inv_lazy_stream(T(1), R(kb[0]), R(kb[1]), 1, f"%[k{Kb}]")])
    outs = [pin(k, f"r[{k}]") for k in (ka[0], ka[1], kb[0], kb[1])]
    return f"__device__ __forceinline__ void {name}({TW_ARGS}) {{\n{TW_BODY}" + asm_block(block, outs, 2, cls) + "}\n"


def emit_ops_body(items):
    """reduction ops as blocks of <= 4 chains; a block with quotient-estimate reductions has a second form (the ladder of
    conditional subtractions) behind a wave-uniform branch for moduli below 2^35 (BflyConst::redR == 255)"""
    body = ""
    for grp in op_blocks(items):
        outs = [pin(op[1], f"r[{op[1]}]") for op in grp]
        fast = asm_block(schedule([op_stream(i, op) for i, op in enumerate(grp)]), outs, 0, indent="        ")
        if any(op[0] == "r" for op in grp):
            slow = asm_block(schedule([op_stream(i, op, True) for i, op in enumerate(grp)]), outs, 0, indent="        ")
            body += f"    if (c.redR != 255u) {{\n{fast}    }}\n    else {{\n{slow}    }}\n"
        else:
            body += f"    {{\n{fast}    }}\n"
    return body


def emit_reduce_fn(name, items):
    return f"__device__ __forceinline__ void {name}(uint64_t (&r)[16], const BflyConst c) {{\n    (void)c;\n{emit_ops_body(items)}}}\n"


def emit_mul_fn(name, ka, kb, cls):
    block = schedule([mul_stream(T(0), R(ka), 0), mul_stream(T(1), R(kb), 1)])
    outs = [pin(ka, f"r[{ka}]"), pin(kb, f"r[{kb}]")]
    return f"__device__ __forceinline__ void {name}({TW_ARGS}) {{\n{TW_BODY}" + asm_block(block, outs, 2, cls) + "}\n"


def emit_csub_fn(name, ks):
    block = schedule([csub_stream(i, R(k)) for i, k in enumerate(ks)])
    outs = [pin(k, f"r[{k}]") for k in ks]
    return (f"__device__ __forceinline__ void {name}(uint64_t (&r)[16], uint64_t m, uint64_t negm) {{\n"
            + asm_block(block, outs, 0) + "}\n")


def stage_pairs(b):
    """butterflies of stage b on the 16 registers, as (k0, k1, twiddle group g)"""
    out = []
    for g in range(8 >> b):
        for lo in range(1 << b):
            k0 = (g << (b + 1)) | lo
            out.append((k0, k0 | (1 << b), g))
    return out


def count(stream_or_block):
    return sum(1 for c in stream_or_block if not c["t"].startswith("s_nop"))


OUT_PLAN = os.path.join(ROOT, "openfhe-development_amd", "csrc", "ntt_inv_plan16.h")


def plan_header():
    """the lazy-inverse plan of the 16-residue kernels as constexpr tables for EVERY build: the lane emulator's C++ butterflies follow it
    op by op and check its bounds value by value (round 6; until then the emulator ran a round-3 restatement of the inverse stages and a
    wrong bound in inv_plan was visible only on the GPU)"""
    def ops(lst, n):
        rows = [f"{{{1 if o[0] == 'c' else 2}, {o[1]}, {o[2] if o[0] == 'c' else 2}}}" for o in lst]
        assert len(rows) <= n, (len(rows), n)
        return "{" + ", ".join(rows + ["{0, 0, 0}"] * (n - len(rows))) + "}"
    pre_t, k_t, end_t, out_t, outl_t = [], [], [], [], []
    for bLo in range(4):
        pre_r, k_r, end_r, out_r, outl_r = [], [], [], [], []
        for b in range(4):
            if b >= bLo:
                pre, K, _, _ = inv_plan(bLo, b)
                pre_r.append(ops(pre[b], 16))
                k_r.append("{" + ", ".join(str(x) for x in K[b]) + "}")
                _, _, end, final = inv_plan(bLo, b)
                _, _, _, lazyf = inv_plan(bLo, b, lazy_out=True)
                end_r.append(ops(end, 16))
                out_r.append("{" + ", ".join(str(x) for x in final) + "}")
                outl_r.append("{" + ", ".join(str(x) for x in lazyf) + "}")
            else:
                pre_r.append(ops([], 16)), k_r.append("{0, 0, 0, 0, 0, 0, 0, 0}"), end_r.append(ops([], 16))
                out_r.append("{" + ", ".join(["0"] * 16) + "}"), outl_r.append("{" + ", ".join(["0"] * 16) + "}")
        pre_t.append("{" + ", ".join(pre_r) + "}"), k_t.append("{" + ", ".join(k_r) + "}"), end_t.append("{" + ", ".join(end_r) + "}")
        out_t.append("{" + ", ".join(out_r) + "}"), outl_t.append("{" + ", ".join(outl_r) + "}")
    return ("""// GENERATED by tools/gen_ntt_asm.py — do not edit; edit the generator and re-run it.
// The lazy-inverse plan (inv_plan) of the 16-residues-per-lane kernels of ntt_static.h, as tables for every build.  [bLo][b]: the step's first
// stage is register bit bLo, the entry belongs to stage b (pre, K) or to a step that ends with stage b (end, out).
#ifndef FHE_NTT_INV_PLAN16_H
#define FHE_NTT_INV_PLAN16_H
namespace fhe {
namespace plan16 {
struct RedOp {
    unsigned char kind, k, m;  // kind 0: none, 1: x = x < m q ? x : x - m q, 2: quotient estimate (x below 2q afterwards); k: residue
};
constexpr RedOp kPre[4][4][16] = {""" + ", ".join(pre_t) + """};  // reductions before stage b
constexpr unsigned char kK[4][4][8] = {""" + ", ".join(k_t) + """};  // K (units of q) of u - v + K, per butterfly in stage order
constexpr RedOp kEnd[4][4][16] = {""" + ", ".join(end_t) + """};  // closing reductions of a step whose last stage is b
constexpr unsigned char kOut[4][4][16] = {""" + ", ".join(out_t) + """};  // bounds (units of q) after them
constexpr unsigned char kOutLazy[4][4][16] = {""" + ", ".join(outl_t) + """};  // bounds when the closing reductions are left to the next pass
}  // namespace plan16
}  // namespace fhe
#endif
""")


def main():
    check_blocks()
    H = []
    H.append("""// GENERATED by tools/gen_ntt_asm.py — do not edit; edit the generator and re-run it.
// In-place gfx950 butterflies on the 16 residues of a lane pinned to v[32:63] (residue k = v[32+2k:33+2k]).
// Arithmetic: the Shoup multiply of the reference's ModMulFastConst (ubintnat.h:1464-1469) inside the butterflies of
// transformnat-impl.h:303-374 / 512-625, with lazy ranges: the butterflies take a TRUNCATED quotient (the low product of
// hi64(y*w') is dropped: one v_mul_hi_u32 less, products in [0,3q) instead of [0,2q)), lazily grown residues come back
// below 2q by one quotient estimate (5 instructions) instead of a ladder of conditional subtractions, the products that
// end a transform keep the exact quotient.  Every block below was simulated against python integers by the generator
// before it was written.  Two butterflies are interleaved per asm block so that the two wait states gfx950 needs
// between a VALU carry write and its reader are filled with the other butterfly.
#ifndef FHE_NTT_BFLY_PINNED_H
#define FHE_NTT_BFLY_PINNED_H
#if defined(__HIP_DEVICE_COMPILE__)
namespace fhe {
struct BflyConst {  // wave-uniform (SGPR) constants of the limb
    uint32_t nql, nqh;             // 2^64 - q
    uint64_t q, twoq, ntwoq, threeq;
    uint32_t redM, redR;           // quotient estimate: k = (x.hi * redM) >> (32 + redR); redR == 255: modulus below 2^35, ladder instead
};
struct BflyZero {   // two VGPRs holding 0 (high halves of the zero-extended mul_hi results)
    uint32_t z0, z1;
};
""")
    for b in range(4):
        prs = stage_pairs(b)
        for cls, tag in (("v", "v"), ("s", "s")):
            fn = []
            for i in range(0, 8, 2):
                (a0, a1, ga), (b0, b1, gb) = prs[i], prs[i + 1]
                name = f"bfly2_fwd_{tag}_b{b}_{i // 2}"
                H.append(emit_pair_fn(name, (a0, a1), (b0, b1), cls))
                fn.append(f"    {name}(r, w[{ga}], w[{gb}], c, z);\n")
            H.append(f"// stage b = {b}: twiddle g serves the butterflies whose index has (k >> {b + 1}) == g\n"
                     f"__device__ __forceinline__ void stage_fwd_{tag}_b{b}(uint64_t (&r)[16], const TwPair (&w)[8], "
                     f"const BflyConst c, const BflyZero z) {{\n" + "".join(fn) + "}\n")
    # lazy inverse stages: the sum output is not reduced; bounds follow inv_plan(bLo, .)
    ktab = [[[0] * 8 for _ in range(4)] for _ in range(4)]
    nred = {}
    for bLo in range(4):
        for b in range(bLo, 4):
            pre, K, _, _ = inv_plan(bLo, b)
            ktab[bLo][b] = K[b]
            H.append(emit_reduce_fn(f"inv_lazy_pre_b{b}_lo{bLo}", pre[b]))
            prs = stage_pairs(b)
            for cls, tag in (("v", "v"), ("s", "s")):
                fn = []
                for i in range(0, 8, 2):
                    (a0, a1, ga), (b0, b1, gb) = prs[i], prs[i + 1]
                    name = f"bfly2_invl_{tag}_b{b}_lo{bLo}_{i // 2}"
                    H.append(emit_lazy_pair_fn(name, (a0, a1), (b0, b1), cls, K[b][i], K[b][i + 1]))
                    fn.append(f"    {name}(r, w[{ga}], w[{gb}], c, z);\n")
                H.append(f"__device__ __forceinline__ void stage_invl_{tag}_b{b}_lo{bLo}(uint64_t (&r)[16], const TwPair (&w)[8], "
                         f"const BflyConst c, const BflyZero z) {{\n    inv_lazy_pre_b{b}_lo{bLo}(r, c);\n" + "".join(fn) + "}\n")
        for bHi in range(bLo, 4):
            pre, _, end, _ = inv_plan(bLo, bHi)
            H.append(emit_reduce_fn(f"inv_lazy_end_lo{bLo}_hi{bHi}", end))
            nred[(bLo, bHi)] = (sum(len(v_) for v_ in pre.values()), len(end))
    H.append("// K multiplier (units of q) of butterfly `pair` of stage b when the step starts at stage bLo: [bLo][b][pair]\n"
             "__device__ constexpr unsigned char kInvLazyK[4][4][8] = {"
             + ", ".join("{" + ", ".join("{" + ", ".join(str(x) for x in ktab[lo][b]) + "}" for b in range(4)) + "}" for lo in range(4))
             + "};\n")
    # last inverse stage (s == 0): residues lo and lo|8 multiplied by N^-1 and w1*N^-1 (exact quotient)
    for i in range(8):
        H.append(emit_mul_fn(f"mul2_s_{i}", i, i | 8, "s"))
    for i in range(4):
        H.append(emit_csub_fn(f"csub4_{i}", [4 * i + j for j in range(4)]))
    fold, sh, lo, fin = reduce192_stream()
    block = schedule([fold]) + schedule([sh, lo]) + schedule([fin])
    regs = [f"v{r}" for r in range(64, 96) if r not in (65, 81)] + [f"s{r}" for r in range(40, 50)]
    text = "\n".join(f'        "{c["t"]}\\n\\t"' for c in block)
    H.append("""// Column sums of mac192 (value < 2^128) -> canonical residue mod p, p and its constants wave-uniform:
// value = hi*2^64 + lo  ==  hi*R + lo (mod p), R = 2^64 mod p as a Shoup pair; lo is reduced with mu64 = floor(2^64/p).
// Any exact reduction equals the reference's BarrettUint128ModUint64 (utilities-int.h:60-99).
struct Reduce192Const {
    uint64_t p, R, Rp, mu64;
};
__device__ __forceinline__ uint64_t reduce192_uniform(uint64_t c0, uint64_t c1, uint64_t c2, uint32_t k0, uint32_t k1,
                                                      const Reduce192Const k, const BflyZero z) {
    uint64_t out;
    const uint64_t nq = 0 - k.p, twop = k.p << 1, ntwop = 0 - twop;
    asm volatile(
""" + text + """
        : [out] "=v"(out)
        : [c0l] "v"((uint32_t)c0), [c0h] "v"((uint32_t)(c0 >> 32)), [c1l] "v"((uint32_t)c1), [c1h] "v"((uint32_t)(c1 >> 32)),
          [c2l] "v"((uint32_t)c2), [c2h] "v"((uint32_t)(c2 >> 32)), [k0] "v"(k0), [k1] "v"(k1),
          [Rl] "s"((uint32_t)k.R), [Rh] "s"((uint32_t)(k.R >> 32)), [Rpl] "s"((uint32_t)k.Rp), [Rph] "s"((uint32_t)(k.Rp >> 32)),
          [mul] "s"((uint32_t)k.mu64), [muh] "s"((uint32_t)(k.mu64 >> 32)), [nql] "s"((uint32_t)nq), [nqh] "s"((uint32_t)(nq >> 32)),
          [p] "s"(k.p), [twop] "s"(twop), [np] "s"(nq), [ntwop] "s"(ntwop), "{v65}"(z.z0), "{v81}"(z.z1)
        : """ + ", ".join(f'"{r}"' for r in regs) + """);
    return out;
}
""")
    H.append("""__device__ __forceinline__ void csub16(uint64_t (&r)[16], uint64_t m) {
    const uint64_t negm = 0 - m;
    csub4_0(r, m, negm);
    csub4_1(r, m, negm);
    csub4_2(r, m, negm);
    csub4_3(r, m, negm);
}
""")
    # every residue below 2q (quotient estimate), and the 8 residues that are the `a` inputs of a stage on field bit b
    # (index bit b clear): only those bound the outputs of a forward butterfly, so the lazy-reduction sweep before a
    # forward step touches just them
    H.append(emit_reduce_fn("red16", [("r", k) for k in range(16)]))
    for b in range(4):
        H.append(emit_reduce_fn(f"red8_a{b}", [("r", k) for k in range(16) if not (k >> b) & 1]))
    H.append("""}  // namespace fhe
#endif
#endif
""")
    text = "".join(H)
    plan = plan_header()
    if "--check" in sys.argv:  # tests: the committed headers must be what this generator (and its simulation) produces
        if open(OUT).read() != text or open(OUT_PLAN).read() != plan:
            print("ntt_bfly_pinned.h / ntt_inv_plan16.h is stale: run python tools/gen_ntt_asm.py")
            return 1
        print("ntt_bfly_pinned.h and ntt_inv_plan16.h are up to date; all blocks simulated OK")
        return 0
    open(OUT, "w").write(text)
    open(OUT_PLAN, "w").write(plan)
    n = count(fwd_stream(T(0), R(0), R(1), 0))
    ni = count(inv_lazy_stream(T(0), R(0), R(1), 0, "%[k3]"))
    print(f"wrote {OUT}: forward butterfly {n} VALU, lazy inverse butterfly {ni} VALU, reduction {count(red_stream(0, R(0)))} VALU; "
          f"inverse steps (bLo, bHi): (reductions before stages, at the end) = {nred}; all blocks simulated OK")


if __name__ == "__main__":
    sys.exit(main())
