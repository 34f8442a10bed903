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
def shoup_tail(t, yl, yh, w, dst, addend):
    """dst(pair) = lo64(y*w + Q*nq) + addend, Q = hi64(y*w'); y = (yl, yh) registers; w = operand suffix"""
    wl, wh, pl, ph = (f"%[{n}{w}]" for n in ("wl", "wh", "pl", "ph"))
    return [
        mulhi(v(t.Z), yl, pl),
        mad(p(t.X), DEAD, yl, wh, "0"),
        mad(p(t.L), DEAD, yh, pl, p(t.Z)),
        mad(p(t.C), t.c0, yl, ph, p(t.L)),
        mad(p(t.X), DEAD, yh, wl, p(t.X)),
        mov(v(t.H), v(t.C + 1)),
        cnd01(v(t.H + 1), t.c0),
        mad(p(t.Q), DEAD, yh, ph, p(t.H)),
        mad(p(t.L), DEAD, yl, wl, addend),
    ], [
        mad(p(t.X), DEAD, v(t.Q), "%[nqh]", p(t.X)),
        mad(p(t.X), DEAD, v(t.Q + 1), "%[nql]", p(t.X)),
        add32(v(t.L + 1), v(t.L + 1), v(t.X)),
        mad(dst, DEAD, v(t.Q), "%[nql]", p(t.L)),
    ]


def fwd_stream(t, A, B, w):
    """a' = a + T, b' = a - T + 2q with T = shoup(b, w) in [0,2q); a = v[A:A+1], b = v[B:B+1]"""
    head, tail = shoup_tail(t, v(B), v(B + 1), w, p(A), p(A))
    return head + [lshladd(p(B), p(A), 1, "%[twoq]")] + tail + [
        subco(v(B), t.c0, v(B), v(A)),
        subbco(v(B + 1), t.c0, v(B + 1), v(A + 1), t.c0),
    ]


def inv_stream(t, A, B, w):
    """a' = (u+v) mod 2q, b' = shoup(u - v + 2q, w); u = v[A:A+1], v = v[B:B+1]"""
    pre = [
        lshladd(p(t.Y), p(A), 0, "%[twoq]"),
        lshladd(p(A), p(A), 0, p(B)),
        subco(v(t.Y), t.c0, v(t.Y), v(B)),
        lshladd(p(t.D), p(A), 0, "%[ntwoq]"),
        cmplt64(t.c1, p(A), "%[twoq]"),
        subbco(v(t.Y + 1), t.c0, v(t.Y + 1), v(B + 1), t.c0),
    ]
    head, tail = shoup_tail(t, v(t.Y), v(t.Y + 1), w, p(B), "0")
    sel = [cnd(v(A), v(t.D), v(A), t.c1), cnd(v(A + 1), v(t.D + 1), v(A + 1), t.c1)]
    return pre + head[:2] + sel + head[2:] + tail


def inv_lazy_stream(t, A, B, w, kop):
    """Gentleman-Sande butterfly without the reduction of the sum: a' = u + v, b' = shoup(u - v + K, w) with K = the
    bound of v (a multiple of 2q given as the scalar operand `kop`); the caller tracks the bounds (inv_lazy_plan)"""
    pre = [
        lshladd(p(t.Y), p(A), 0, kop),
        lshladd(p(A), p(A), 0, p(B)),
        subco(v(t.Y), t.c0, v(t.Y), v(B)),
        subbco(v(t.Y + 1), t.c0, v(t.Y + 1), v(B + 1), t.c0),
    ]
    head, tail = shoup_tail(t, v(t.Y), v(t.Y + 1), w, p(B), "0")
    return pre + head + tail


def inv_lazy_plan(bLo, bHi, ends=False):
    """Bounds (units of q) of the 16 residues through the inverse stages bLo..bHi when the sum output is not reduced:
    inputs < 2q; a' = u + v doubles the bound, b' = shoup(.) is < 2q again.  A pair whose bound reached 16q is brought
    back to 8q first (u + v and u - v + K must stay below 2^64 > 16q).  Returns
      pre[b]  = [(k, m)]: residue k gets csub(m*q) before stage b,
      K[b]    = [m per butterfly, in stage_pairs order]: K = m*q,
      end     = [(k, m), ...] in order: csub(m*q) bringing every residue below 2q after the last stage.
    ends: the last stage is the transform's final multiplication stage (outputs < 2q, no end reductions)."""
    bound = [2] * 16
    pre, K = {}, {}
    for b in range(bLo, bHi + 1):
        pre[b], K[b] = [], []
        for (k0, k1, _g) in stage_pairs(b):
            assert bound[k0] == bound[k1]
            if bound[k0] >= 16:
                pre[b] += [(k0, 8), (k1, 8)]
                bound[k0] = bound[k1] = 8
            K[b].append(bound[k0])
            if ends and b == bHi:
                bound[k0] = bound[k1] = 2
            else:
                bound[k0], bound[k1] = 2 * bound[k0], 2
    end = []
    lvl = 8
    while lvl >= 2:
        for k in range(16):
            if bound[k] > lvl:
                end.append((k, lvl))
                bound[k] = lvl
        lvl //= 2
    assert all(b == 2 for b in bound)
    return pre, K, end


def csub_blocks(items):
    """[(k, m)] in execution order -> blocks of <= 4 conditional subtractions; a block never holds the same residue twice
    (the levels of one residue are sequential) and keeps the order between levels"""
    blocks, cur = [], []
    for k, m in items:
        if len(cur) == 4 or any(k == k2 for k2, _ in cur):
            blocks.append(cur)
            cur = []
        cur.append((k, m))
    if cur:
        blocks.append(cur)
    return blocks


def csub_chain_stream(i, A, mop, nop):
    """x = x < m ? x : x - m with explicit operand names for m and -m; chain i of up to 4"""
    D, c = CSUB_TMP[i], f"s[{42 + 2 * i}:{43 + 2 * i}]"
    return [
        lshladd(p(D), p(A), 0, nop),
        cmplt64(c, p(A), mop),
        cnd(v(A), v(D), v(A), c),
        cnd(v(A + 1), v(D + 1), v(A + 1), c),
    ]


def mul_stream(t, A, w):
    """x = shoup(x, w) in place (last inverse stage: the caller forms u+v / u-v+2q first); x is only overwritten by
    the last instruction, after every read of its halves"""
    head, tail = shoup_tail(t, v(A), v(A + 1), w, p(A), "0")
    return head + tail


CSUB_TMP = (68, 70, 72, 74)  # pairs used by the 4 interleaved conditional subtractions


def csub_stream(i, A):
    """x = x < m ? x : x - m   (m, -m as scalar pairs); chain i of 4"""
    D, c = CSUB_TMP[i], f"s[{42 + 2 * i}:{43 + 2 * i}]"
    return [
        lshladd(p(D), p(A), 0, "%[negm]"),
        cmplt64(c, p(A), "%[m]"),
        cnd(v(A), v(D), v(A), c),
        cnd(v(A + 1), v(D + 1), v(A + 1), c),
    ]


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


def run(block, S):
    for c in block:
        c["sim"](S)


def check_blocks():
    rnd = random.Random(7)
    for it in range(4000):
        q = rnd.getrandbits(60) | (1 << 59) | 1
        w = [rnd.randrange(q), rnd.randrange(q)]
        wp = [(x << 64) // q for x in w]
        S = St()
        S.ops = {"nql": (-q) & M32, "nqh": ((-q) & M64) >> 32, "twoq": 2 * q, "ntwoq": (-2 * q) & M64}
        for i in (0, 1):
            S.ops.update({f"wl{i}": w[i] & M32, f"wh{i}": w[i] >> 32, f"pl{i}": wp[i] & M32, f"ph{i}": wp[i] >> 32})
        t0, t1 = T(0), T(1)
        S.v[t0.Z + 1] = S.v[t1.Z + 1] = 0
        # forward pair: lazy inputs a < 14q, b < 2^64
        a = [rnd.randrange(14 * q), rnd.randrange(14 * q)]
        b = [rnd.getrandbits(64), rnd.randrange(16 * q)]
        for i, (k0, k1) in enumerate(((0, 8), (5, 13))):
            S.set64(p(R(k0)), a[i]), S.set64(p(R(k1)), b[i])
        run(schedule([fwd_stream(t0, R(0), R(8), 0), fwd_stream(t1, R(5), R(13), 1)]), S)
        for i, (k0, k1) in enumerate(((0, 8), (5, 13))):
            Tm = shoup_ref(b[i], w[i], wp[i], q)
            assert Tm < 2 * q
            assert S.g64(p(R(k0))) == (a[i] + Tm) & M64, "fwd a"
            assert S.g64(p(R(k1))) == (a[i] - Tm + 2 * q) & M64, "fwd b"
        # inverse pair: inputs < 2q
        u = [rnd.randrange(2 * q), rnd.randrange(2 * q)]
        vv = [rnd.randrange(2 * q), rnd.randrange(2 * q)]
        for i, (k0, k1) in enumerate(((2, 3), (14, 6))):
            S.set64(p(R(k0)), u[i]), S.set64(p(R(k1)), vv[i])
        run(schedule([inv_stream(t0, R(2), R(3), 0), inv_stream(t1, R(14), R(6), 1)]), S)
        for i, (k0, k1) in enumerate(((2, 3), (14, 6))):
            s = u[i] + vv[i]
            assert S.g64(p(R(k0))) == (s if s < 2 * q else s - 2 * q), "inv a"
            assert S.g64(p(R(k1))) == shoup_ref(u[i] - vv[i] + 2 * q, w[i], wp[i], q), "inv b"
        # in-place multiply pair, any 64-bit input
        x = [rnd.getrandbits(64), rnd.randrange(4 * q)]
        S.set64(p(R(1)), x[0]), S.set64(p(R(9)), x[1])
        run(schedule([mul_stream(t0, R(1), 0), mul_stream(t1, R(9), 1)]), S)
        assert S.g64(p(R(1))) == shoup_ref(x[0], w[0], wp[0], q) and S.g64(p(R(9))) == shoup_ref(x[1], w[1], wp[1], q)
        # conditional subtraction, 4 residues per block
        m = q << rnd.randrange(0, 4)
        S.ops.update({"m": m, "negm": (-m) & M64})
        xs = [rnd.randrange(2 * m) for _ in range(4)]
        for i, x0 in enumerate(xs):
            S.set64(p(R(4 + i)), x0)
        run(schedule([csub_stream(i, R(4 + i)) for i in range(4)]), S)
        for i, x0 in enumerate(xs):
            assert S.g64(p(R(4 + i))) == (x0 if x0 < m else x0 - m), "csub"
    # lazy inverse steps: every (bLo, bHi), with and without the final multiplication stage
    for bLo in range(4):
        for bHi in range(bLo, 4):
            for ends in ((False, True) if bHi == 3 else (False,)):
                pre, K, end = inv_lazy_plan(bLo, bHi, ends)
                for it in range(40):
                    q = rnd.getrandbits(60) | (1 << 59) | 1
                    S = St()
                    S.v[T(0).Z + 1] = S.v[T(1).Z + 1] = 0
                    S.ops = {"nql": (-q) & M32, "nqh": ((-q) & M64) >> 32}
                    for m in (2, 4, 8):
                        S.ops[f"k{m}"], S.ops[f"n{m}"] = m * q, (-m * q) & M64
                    x = [rnd.randrange(2 * q) if it else 2 * q - 1 for _ in range(16)]
                    ref = [v_ % q for v_ in x]
                    for k in range(16):
                        S.set64(p(R(k)), x[k])
                    for b in range(bLo, bHi + 1):
                        for grp in csub_blocks(pre[b]):
                            run(schedule([csub_chain_stream(i, R(k), f"%[k{m}]", f"%[n{m}]") for i, (k, m) in enumerate(grp)]), S)
                        prs = stage_pairs(b)
                        for j in range(0, 8, 2):
                            tw = [rnd.randrange(q), rnd.randrange(q)]
                            for i in (0, 1):
                                wp_ = (tw[i] << 64) // q
                                S.ops.update({f"wl{i}": tw[i] & M32, f"wh{i}": tw[i] >> 32, f"pl{i}": wp_ & M32, f"ph{i}": wp_ >> 32})
                            (a0, a1, _), (b0, b1, _) = prs[j], prs[j + 1]
                            if ends and b == bHi:  # final stage: sums / differences formed by the caller, then multiplied
                                for (u_, v_), m in (((a0, a1), K[b][j]), ((b0, b1), K[b][j + 1])):
                                    uu, vv = S.g64(p(R(u_))), S.g64(p(R(v_)))
                                    assert uu + vv < 1 << 64 and uu - vv + m * q < 1 << 64
                                    S.set64(p(R(u_)), uu + vv), S.set64(p(R(v_)), uu - vv + m * q)
                                run(schedule([mul_stream(T(0), R(a0), 0), mul_stream(T(1), R(b0), 1)]), S)
                                run(schedule([mul_stream(T(0), R(a1), 0), mul_stream(T(1), R(b1), 1)]), S)
                                for (u_, v_), w_ in (((a0, a1), tw[0]), ((b0, b1), tw[1])):
                                    ru, rv = ref[u_], ref[v_]
                                    ref[u_], ref[v_] = (ru + rv) * w_ % q, (ru - rv) * w_ % q
                            else:
                                for (u_, v_) in ((a0, a1), (b0, b1)):
                                    assert S.g64(p(R(u_))) + S.g64(p(R(v_))) < 1 << 64
                                run(schedule([inv_lazy_stream(T(0), R(a0), R(a1), 0, f"%[k{K[b][j]}]"),
                                              inv_lazy_stream(T(1), R(b0), R(b1), 1, f"%[k{K[b][j + 1]}]")]), S)
                                for (u_, v_), w_ in (((a0, a1), tw[0]), ((b0, b1), tw[1])):
                                    ru, rv = ref[u_], ref[v_]
                                    ref[u_], ref[v_] = (ru + rv) % q, (ru - rv) * w_ % q
                    for grp in csub_blocks(end):
                        run(schedule([csub_chain_stream(i, R(k), f"%[k{m}]", f"%[n{m}]") for i, (k, m) in enumerate(grp)]), S)
                    for k in range(16):
                        got = S.g64(p(R(k)))
                        assert got < 2 * q and got % q == ref[k], ("lazy inverse step", bLo, bHi, ends, k)
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
def clobbers(nslots):
    regs = []
    for s in range(nslots):
        t = T(s)
        regs += [f"v{r}" for r in range(t.Z, t.Z + 16) if r != t.Z + 1]
        regs += [f"s{r}" for r in range(42 + 4 * s, 46 + 4 * s)]
    return regs + ["s40", "s41"]


def asm_block(block, outs, ins_, nslots):
    text = "\n".join(f'        "{c["t"]}\\n\\t"' for c in block)
    o = ", ".join(outs)
    i = ", ".join(ins_)
    regs = clobbers(nslots) if nslots else [f"v{r}" for r in range(68, 76)] + [f"s{r}" for r in range(42, 50)]
    c = ", ".join(f'"{r}"' for r in regs)
    return f"    asm volatile(\n{text}\n        : {o}\n        : {i}\n        : {c});\n"


def pin(k, var):
    return f'"+{{v[{R(k)}:{R(k) + 1}]}}"({var})'


def tw_in(i, cls):
    return [f'[wl{i}] "{cls}"(w{i}l)', f'[wh{i}] "{cls}"(w{i}h)', f'[pl{i}] "{cls}"(p{i}l)', f'[ph{i}] "{cls}"(p{i}h)']


ZERO_IN = ['"{v65}"(z.z0)', '"{v81}"(z.z1)']
CONST_IN = ['[nql] "s"(c.nql)', '[nqh] "s"(c.nqh)', '[twoq] "s"(c.twoq)', '[ntwoq] "s"(c.ntwoq)']


def emit_pair_fn(name, kind, ka, kb, cls):
    """two butterflies (r[ka[0]], r[ka[1]]) with twiddle 0 and (r[kb[0]], r[kb[1]]) with twiddle 1"""
    mk = fwd_stream if kind == "fwd" else inv_stream
    block = schedule([mk(T(0), R(ka[0]), R(ka[1]), 0), mk(T(1), R(kb[0]), R(kb[1]), 1)])
    args = "uint64_t (&r)[16], const TwPair wa, const TwPair wb, const BflyConst c, const BflyZero z"
    body = "".join(f"    const uint32_t w{i}l = (uint32_t)w{n}.w, w{i}h = (uint32_t)(w{n}.w >> 32), p{i}l = (uint32_t)w{n}.wp, "
                   f"p{i}h = (uint32_t)(w{n}.wp >> 32);\n" for i, n in ((0, "a"), (1, "b")))
    outs = [pin(k, f"r[{k}]") for k in (ka[0], ka[1], kb[0], kb[1])]
    return (f"__device__ __forceinline__ void {name}({args}) {{\n{body}"
            + asm_block(block, outs, tw_in(0, cls) + tw_in(1, cls) + CONST_IN + ZERO_IN, 2) + "}\n")


LAZY_CONST_IN = ['[nql] "s"(c.nql)', '[nqh] "s"(c.nqh)', '[k2] "s"(c.twoq)', '[k4] "s"(c.twoq << 1)', '[k8] "s"(c.twoq << 2)']
LAZY_RED_IN = ['[k2] "s"(c.twoq)', '[k4] "s"(c.twoq << 1)', '[k8] "s"(c.twoq << 2)', '[n2] "s"(0 - c.twoq)',
               '[n4] "s"(0 - (c.twoq << 1))', '[n8] "s"(0 - (c.twoq << 2))']


def emit_lazy_pair_fn(name, ka, kb, cls, Ka, Kb):
    block = schedule([inv_lazy_stream(T(0), R(ka[0]), R(ka[1]), 0, f"%[k{Ka}]"),
                      inv_lazy_stream(T(1), R(kb[0]), R(kb[1]), 1, f"%[k{Kb}]")])
    args = "uint64_t (&r)[16], const TwPair wa, const TwPair wb, const BflyConst c, const BflyZero z"
    body = "".join(f"    const uint32_t w{i}l = (uint32_t)w{n}.w, w{i}h = (uint32_t)(w{n}.w >> 32), p{i}l = (uint32_t)w{n}.wp, "
                   f"p{i}h = (uint32_t)(w{n}.wp >> 32);\n" for i, n in ((0, "a"), (1, "b")))
    outs = [pin(k, f"r[{k}]") for k in (ka[0], ka[1], kb[0], kb[1])]
    return (f"__device__ __forceinline__ void {name}({args}) {{\n{body}"
            + asm_block(block, outs, tw_in(0, cls) + tw_in(1, cls) + LAZY_CONST_IN + ZERO_IN, 2) + "}\n")


def emit_reduce_fn(name, items):
    """conditional subtractions [(k, m)] (threshold m*q) as blocks of <= 4"""
    body = ""
    for grp in csub_blocks(items):
        block = schedule([csub_chain_stream(i, R(k), f"%[k{m}]", f"%[n{m}]") for i, (k, m) in enumerate(grp)])
        body += asm_block(block, [pin(k, f"r[{k}]") for k, _ in grp], LAZY_RED_IN, 0)
    return f"__device__ __forceinline__ void {name}(uint64_t (&r)[16], const BflyConst c) {{\n(void)c;\n{body}}}\n"


def emit_mul_fn(name, ka, kb, cls):
    block = schedule([mul_stream(T(0), R(ka), 0), mul_stream(T(1), R(kb), 1)])
    args = "uint64_t (&r)[16], const TwPair wa, const TwPair wb, const BflyConst c, const BflyZero z"
    body = "".join(f"    const uint32_t w{i}l = (uint32_t)w{n}.w, w{i}h = (uint32_t)(w{n}.w >> 32), p{i}l = (uint32_t)w{n}.wp, "
                   f"p{i}h = (uint32_t)(w{n}.wp >> 32);\n" for i, n in ((0, "a"), (1, "b")))
    outs = [pin(ka, f"r[{ka}]"), pin(kb, f"r[{kb}]")]
    return (f"__device__ __forceinline__ void {name}({args}) {{\n{body}"
            + asm_block(block, outs, tw_in(0, cls) + tw_in(1, cls) + CONST_IN[:2] + ZERO_IN, 2) + "}\n")


def emit_csub_fn(name, ks):
    block = schedule([csub_stream(i, R(k)) for i, k in enumerate(ks)])
    outs = [pin(k, f"r[{k}]") for k in ks]
    return (f"__device__ __forceinline__ void {name}(uint64_t (&r)[16], uint64_t m, uint64_t negm) {{\n"
            + asm_block(block, outs, ['[m] "s"(m)', '[negm] "s"(negm)'], 0) + "}\n")


def stage_pairs(b):
    """butterflies of stage b on the 16 registers, as (k0, k1, twiddle group g)"""
    out = []
    for g in range(8 >> b):
        for lo in range(1 << b):
            k0 = (g << (b + 1)) | lo
            out.append((k0, k0 | (1 << b), g))
    return out


def main():
    check_blocks()
    H = []
    H.append("""// GENERATED by tools/gen_ntt_asm.py — do not edit; edit the generator and re-run it.
// In-place gfx950 butterflies on the 16 residues of a lane pinned to v[32:63] (residue k = v[32+2k:33+2k]).
// Arithmetic = bfly_fwd_fast / bfly_inv_fast of ntt_kernels.h (Shoup multiply of the reference's
// ModMulFastConst, transformnat-impl.h:303-374 / 512-625, with lazy ranges); every block below was simulated against
// that arithmetic by the generator before it was written.  Two butterflies are interleaved per asm block so that
// the two wait states gfx950 needs between a VALU carry write and its reader are filled with the other butterfly.
#ifndef FHE_NTT_BFLY_PINNED_H
#define FHE_NTT_BFLY_PINNED_H
#if defined(__HIP_DEVICE_COMPILE__)
namespace fhe {
struct BflyConst {  // wave-uniform (SGPR) constants of the limb
    uint32_t nql, nqh;      // 2^64 - q
    uint64_t twoq, ntwoq;   // 2q, 2^64 - 2q
};
struct BflyZero {   // two VGPRs holding 0 (high halves of the zero-extended mul_hi results)
    uint32_t z0, z1;
};
""")
    for kind in ("fwd", "inv"):
        for b in range(4):
            prs = stage_pairs(b)
            for cls, tag in (("v", "v"), ("s", "s")):
                fn = []
                for i in range(0, 8, 2):
                    (a0, a1, ga), (b0, b1, gb) = prs[i], prs[i + 1]
                    name = f"bfly2_{kind}_{tag}_b{b}_{i // 2}"
                    H.append(emit_pair_fn(name, kind, (a0, a1), (b0, b1), cls))
                    fn.append(f"    {name}(r, w[{ga}], w[{gb}], c, z);\n")
                H.append(f"// stage b = {b}: twiddle g serves the butterflies whose index has (k >> {b + 1}) == g\n"
                         f"__device__ __forceinline__ void stage_{kind}_{tag}_b{b}(uint64_t (&r)[16], const TwPair (&w)[8], "
                         f"const BflyConst c, const BflyZero z) {{\n" + "".join(fn) + "}\n")
    # lazy inverse stages: the sum output is not reduced; bounds follow inv_lazy_plan(bLo, .)
    ktab = [[[0] * 8 for _ in range(4)] for _ in range(4)]
    for bLo in range(4):
        for b in range(bLo, 4):
            pre, K, _ = inv_lazy_plan(bLo, b)
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
            _, _, end = inv_lazy_plan(bLo, bHi)
            H.append(emit_reduce_fn(f"inv_lazy_end_lo{bLo}_hi{bHi}", end))
    H.append("// K multiplier (units of q) of butterfly `pair` of stage b when the step starts at stage bLo: [bLo][b][pair]\n"
             "__device__ constexpr unsigned char kInvLazyK[4][4][8] = {"
             + ", ".join("{" + ", ".join("{" + ", ".join(str(x) for x in ktab[lo][b]) + "}" for b in range(4)) + "}" for lo in range(4))
             + "};\n")
    # last inverse stage (s == 0): residues lo and lo|8 multiplied by N^-1 and w1*N^-1
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
    # the 8 residues that are the `a` inputs of a stage on field bit b (index bit b clear): only those bound the
    # outputs of a forward butterfly, so the lazy-reduction sweep before a step touches just them
    for b in range(4):
        ks = [k for k in range(16) if not (k >> b) & 1]
        H.append(emit_csub_fn(f"csub4_a{b}_0", ks[:4]))
        H.append(emit_csub_fn(f"csub4_a{b}_1", ks[4:]))
        H.append(f"""__device__ __forceinline__ void csub8_a{b}(uint64_t (&r)[16], uint64_t m) {{
    const uint64_t negm = 0 - m;
    csub4_a{b}_0(r, m, negm);
    csub4_a{b}_1(r, m, negm);
}}
""")
    H.append("""}  // namespace fhe
#endif
#endif
""")
    text = "".join(H)
    if "--check" in sys.argv:  # tests: the committed header must be what this generator (and its simulation) produces
        if open(OUT).read() != text:
            print("ntt_bfly_pinned.h is stale: run python tools/gen_ntt_asm.py")
            return 1
        print("ntt_bfly_pinned.h is up to date; all blocks simulated OK")
        return 0
    open(OUT, "w").write(text)
    n = sum(1 for c in schedule([fwd_stream(T(0), R(0), R(1), 0), fwd_stream(T(1), R(2), R(3), 1)]) if not c["t"].startswith("s_nop"))
    ni = sum(1 for c in schedule([inv_stream(T(0), R(0), R(1), 0), inv_stream(T(1), R(2), R(3), 1)]) if not c["t"].startswith("s_nop"))
    print(f"wrote {OUT}: forward pair {n} VALU, inverse pair {ni} VALU; all blocks simulated OK")


if __name__ == "__main__":
    sys.exit(main())
