#!/usr/bin/env python3
"""Issue-cost measurements of gfx950 instruction sequences (measurement tool, not part of the product library).

Generates tools/abl/seqbench.hip from the instruction streams of tools/gen_ntt_asm.py (the same texts the NTT kernels are
built from) plus single instructions, and — on the GPU box — builds and runs it:  every sequence is repeated inside one
asm block, a loop runs it `iters` times in every wave of a grid that puts W waves on each SIMD, s_memtime brackets the
loop.  Reported per sequence: shader cycles per wave and per instance, divided by W = issue cycles one instance costs the
SIMD when W waves share it (what bounds the integer-issue-bound row passes), and the same per instruction.

  python tools/seqbench.py gen          writes tools/abl/seqbench.hip (runs anywhere)
  python tools/seqbench.py run [out]    hipcc + run on the GPU box, JSON to out (default stdout)
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_ntt_asm as g  # noqa: E402

SRC = os.path.join(HERE, "abl", "seqbench.hip")
EXE = os.path.join(HERE, "abl", "seqbench")

VOPS = {f"{n}{i}": f"v{96 + 4 * i + j}" for i in (0, 1) for j, n in enumerate(("wl", "wh", "pl", "ph"))}
SOPS = {f"{n}{i}": f"s{52 + 4 * i + j}" for i in (0, 1) for j, n in enumerate(("wl", "wh", "pl", "ph"))}
FIXED = {"nql": "s60", "nqh": "s61", "twoq": "s[62:63]", "threeq": "s[64:65]", "ntwoq": "s[66:67]", "redM": "s68", "redR": "s69",
         "m": "s[70:71]", "negm": "s[72:73]"}


def subst(text, cls):
    def f(m):
        n = m.group(1)
        if n in VOPS:
            return (VOPS if cls == "v" else SOPS)[n]
        if n in FIXED:
            return FIXED[n]
        if re.fullmatch(r"k\d+", n):
            return "s[70:71]"
        if re.fullmatch(r"n\d+", n):
            return "s[72:73]"
        raise KeyError(n)
    return re.sub(r"%\[(\w+)\]", f, text)


def block_text(block, cls="s"):
    return [subst(c["t"], cls) for c in block]


def fwd_r3(t, A, B, w):  # round-3 forward butterfly: exact quotient, + 2q
    head, tail = g.shoup_tail(t, g.v(B), g.v(B + 1), w, g.p(A), g.p(A), exact=True)
    return head + [g.lshladd(g.p(B), g.p(A), 1, "%[twoq]")] + tail + [
        g.subco(g.v(B), t.c0, g.v(B), g.v(A)), g.subbco(g.v(B + 1), t.c0, g.v(B + 1), g.v(A + 1), t.c0)]


def inv_r3(t, A, B, w):  # round-3 lazy inverse butterfly: exact quotient
    pre = [g.lshladd(g.p(t.Y), g.p(A), 0, "%[k2]"), g.lshladd(g.p(A), g.p(A), 0, g.p(B)),
           g.subco(g.v(t.Y), t.c0, g.v(t.Y), g.v(B)), g.subbco(g.v(t.Y + 1), t.c0, g.v(t.Y + 1), g.v(B + 1), t.c0)]
    head, tail = g.shoup_tail(t, g.v(t.Y), g.v(t.Y + 1), w, g.p(B), "0", exact=True)
    return pre + head + tail


def single(text_fn, n=8):
    """n independent instances of one instruction (or short chain) on distinct registers"""
    out = []
    for i in range(n):
        out += text_fn(i)
    return out


def ins(text, rd=(), wr=()):
    return g.ins(text, rd, wr, None)


R = g.R
T0, T1 = g.T(0), g.T(1)
TESTS = []  # (name, [instruction texts], instances per body, instructions per instance)


def add(name, texts, inst):
    n = sum(1 for t in texts if not t.startswith("s_nop"))
    TESTS.append((name, texts, inst, n / inst))


def add_block(name, streams, inst, cls="s"):
    add(name, block_text(g.schedule(streams), cls), inst)


P = lambda i: f"v[{32 + 2 * i}:{33 + 2 * i}]"  # noqa: E731
V = lambda i: f"v{32 + i}"  # noqa: E731
add("v_mad_u64_u32 (v*v+v64)", [f"v_mad_u64_u32 {P(i)}, s[40:41], v{96 + i % 8}, v{97 + i % 7}, {P(i)}" for i in range(8)], 8)
add("v_mad_u64_u32 (v*s+v64)", [f"v_mad_u64_u32 {P(i)}, s[40:41], v{96 + i % 8}, s60, {P(i)}" for i in range(8)], 8)
add("v_mad_u64_u32 (v*s+0)", [f"v_mad_u64_u32 {P(i)}, s[40:41], v{96 + i % 8}, s60, 0" for i in range(8)], 8)
add("v_mul_hi_u32 (v*s)", [f"v_mul_hi_u32 {V(i)}, v{96 + i % 8}, s60" for i in range(8)], 8)
add("v_mul_hi_u32 (v*v)", [f"v_mul_hi_u32 {V(i)}, v{96 + i % 8}, v{64 + i}" for i in range(8)], 8)
add("v_mul_lo_u32 (v*s)", [f"v_mul_lo_u32 {V(i)}, v{96 + i % 8}, s60" for i in range(8)], 8)
add("v_mov_b32", [f"v_mov_b32 {V(i)}, v{64 + i}" for i in range(8)], 8)
add("v_mov_b64", [f"v_mov_b64 {P(i)}, v[{64 + 2 * i}:{65 + 2 * i}]" for i in range(8)], 8)
add("v_add_u32", [f"v_add_u32 {V(i)}, {V(i)}, v{64 + i}" for i in range(8)], 8)
add("v_sub_u32", [f"v_sub_u32 {V(i)}, {V(i)}, v{64 + i}" for i in range(8)], 8)
add("v_and_b32", [f"v_and_b32 {V(i)}, {V(i)}, v{64 + i}" for i in range(8)], 8)
add("v_min_u32", [f"v_min_u32 {V(i)}, {V(i)}, v{64 + i}" for i in range(8)], 8)
add("v_lshrrev_b32 (s amount)", [f"v_lshrrev_b32 {V(i)}, s69, v{64 + i}" for i in range(8)], 8)
add("v_ashrrev_i32 (31)", [f"v_ashrrev_i32 {V(i)}, 31, v{64 + i}" for i in range(8)], 8)
add("v_alignbit_b32", [f"v_alignbit_b32 {V(i)}, v{64 + i}, v{72 + i}, 28" for i in range(8)], 8)
add("v_add3_u32", [f"v_add3_u32 {V(i)}, {V(i)}, v{64 + i}, v{72 + i}" for i in range(8)], 8)
add("v_lshl_add_u64 (v, 0, s64)", [f"v_lshl_add_u64 {P(i)}, {P(i)}, 0, s[62:63]" for i in range(8)], 8)
add("v_lshl_add_u64 (v, 0, v64)", [f"v_lshl_add_u64 {P(i)}, {P(i)}, 0, v[{64 + 2 * i}:{65 + 2 * i}]" for i in range(8)], 8)
add("v_lshlrev_b64 (2)", [f"v_lshlrev_b64 {P(i)}, 2, v[{64 + 2 * i}:{65 + 2 * i}]" for i in range(8)], 8)
add("v_lshrrev_b64 (32)", [f"v_lshrrev_b64 {P(i)}, 32, v[{64 + 2 * i}:{65 + 2 * i}]" for i in range(8)], 8)
add("v_cvt_f32_u32", [f"v_cvt_f32_u32 {V(i)}, v{64 + i}" for i in range(8)], 8)
add("v_mul_f32", [f"v_mul_f32 {V(i)}, {V(i)}, v{64 + i}" for i in range(8)], 8)
add("v_cvt_u32_f32", [f"v_cvt_u32_f32 {V(i)}, v{64 + i}" for i in range(8)], 8)
add("v_mad_u32_u24", [f"v_mad_u32_u24 {V(i)}, v{64 + i}, v{72 + i}, {V(i)}" for i in range(8)], 8)
add("v_cmp_lt_u64_e64 (v, s64 -> sgpr)", [f"v_cmp_lt_u64_e64 s[{42 + 2 * (i % 4)}:{43 + 2 * (i % 4)}], {P(i)}, s[62:63]" for i in range(8)], 8)
add("v_cmp_lt_u32_e32 (-> vcc)", [f"v_cmp_lt_u32_e32 vcc, v{64 + i}, {V(i)}" for i in range(8)], 8)
add("v_cndmask_b32_e64 (sgpr mask, set long before)", [f"v_cndmask_b32_e64 {V(i)}, {V(i)}, v{64 + i}, s[42:43]" for i in range(8)], 8)
add("v_cndmask_b32_e32 (vcc, set long before)", [f"v_cndmask_b32_e32 {V(i)}, {V(i)}, v{64 + i}, vcc" for i in range(8)], 8)
add("v_cndmask_b32_e64 0/1 (carry -> word)", [f"v_cndmask_b32_e64 {V(i)}, 0, 1, s[42:43]" for i in range(8)], 8)
add_block("v_sub_co_u32 + v_subb_co_u32 (e64, sgpr carry), 4 chains",
          [[g.subco(g.v(R(i)), f"s[{42 + 2 * i}:{43 + 2 * i}]", g.v(R(i)), g.v(64 + 2 * i)),
            g.subbco(g.v(R(i) + 1), f"s[{42 + 2 * i}:{43 + 2 * i}]", g.v(R(i) + 1), g.v(65 + 2 * i), f"s[{42 + 2 * i}:{43 + 2 * i}]")] for i in range(4)], 4)
add("v_sub_co_u32 + v_subb_co_u32 (e32, vcc), 4 chains back to back",
    sum(([f"v_sub_co_u32_e32 {V(2 * i)}, vcc, {V(2 * i)}, v{64 + 2 * i}", f"v_subb_co_u32_e32 {V(2 * i + 1)}, vcc, {V(2 * i + 1)}, v{65 + 2 * i}, vcc"]
         for i in range(4)), []), 4)
# --- sequences of the NTT kernels ---
add_block("r3 forward butterfly pair (exact quotient, s twiddles)", [fwd_r3(T0, R(0), R(8), 0), fwd_r3(T1, R(5), R(13), 1)], 2)
add_block("r4 forward butterfly pair (truncated quotient, s twiddles)", [g.fwd_stream(T0, R(0), R(8), 0), g.fwd_stream(T1, R(5), R(13), 1)], 2)
add_block("r3 forward butterfly pair (v twiddles)", [fwd_r3(T0, R(0), R(8), 0), fwd_r3(T1, R(5), R(13), 1)], 2, "v")
add_block("r4 forward butterfly pair (v twiddles)", [g.fwd_stream(T0, R(0), R(8), 0), g.fwd_stream(T1, R(5), R(13), 1)], 2, "v")
add_block("r3 lazy inverse butterfly pair (exact quotient)", [inv_r3(T0, R(0), R(8), 0), inv_r3(T1, R(5), R(13), 1)], 2)
add_block("r4 lazy inverse butterfly pair (truncated quotient)",
          [g.inv_lazy_stream(T0, R(0), R(8), 0, "%[k3]"), g.inv_lazy_stream(T1, R(5), R(13), 1, "%[k3]")], 2)
add_block("exact Shoup multiply pair (mul_stream)", [g.mul_stream(T0, R(1), 0), g.mul_stream(T1, R(9), 1)], 2)
add_block("conditional subtraction, 4 chains", [g.csub_stream(i, R(4 + i)) for i in range(4)], 4)
add_block("quotient-estimate reduction (red), 4 chains", [g.red_stream(i, R(4 + i)) for i in range(4)], 4)
add_block("ladder 8q/4q/2q, 4 chains", [g.red_slow_stream(i, R(4 + i)) for i in range(4)], 4)
# csub variant: sign of x - m decides (32-bit compare into vcc, e32 selects); one chain at a time (vcc is one register)
add("conditional subtraction via sign of the difference (vcc), 4 chains back to back",
    sum(([f"v_lshl_add_u64 v[{68 + 2 * i}:{69 + 2 * i}], {P(4 + i)}, 0, s[72:73]", f"v_cmp_gt_i32_e32 vcc, 0, v{69 + 2 * i}", "s_nop 1",
          f"v_cndmask_b32_e32 {V(8 + 2 * i)}, v{68 + 2 * i}, {V(8 + 2 * i)}, vcc", f"v_cndmask_b32_e32 {V(9 + 2 * i)}, v{69 + 2 * i}, {V(9 + 2 * i)}, vcc"]
         for i in range(4)), []), 4)


def gen():
    os.makedirs(os.path.dirname(SRC), exist_ok=True)
    clob = ", ".join(f'"v{r}"' for r in range(32, 104)) + ", " + ", ".join(f'"s{r}"' for r in range(40, 74)) + ', "vcc"'
    init = "".join(f'"v_add_u32 v{r}, {hex((2654435761 * (r + 1)) % 2**32)}, %0\\n\\t"\n' for r in range(32, 104))
    init += "".join(f'"s_mov_b32 s{r}, {0x9E3779B1 ^ (r * 0x01000193) & 0x7fffffff}\\n\\t"\n' for r in range(40, 74))
    init += '"s_mov_b32 s69, 27\\n\\t"\n'
    ker = []
    for k, (name, texts, inst, npi) in enumerate(TESTS):
        reps = max(1, 256 // max(1, len(texts)))
        body = "\n".join(f'            "{t}\\n\\t"' for t in texts * reps)
        ker.append(f"""
__global__ void __launch_bounds__(256) k{k}(unsigned long long* out, unsigned seed, int iters) {{
    unsigned sd = seed + threadIdx.x * 977u + blockIdx.x;
    asm volatile({init}        : : "v"(sd) : {clob});
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
        asm volatile(
{body}
            : : : {clob});
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {{
        out[2 * (blockIdx.x * 4 + threadIdx.x / 64)]     = t1 - t0;
        out[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = w1 - w0;
    }}
}}""")
    names = ",\n".join(f'    {{"{n}", {len(t) and max(1, 256 // len(t))}, {inst}, {npi}}}' for (n, t, inst, npi) in TESTS)
    launches = "\n".join(f"        case {k}: hipLaunchKernelGGL(k{k}, dim3(blocks), dim3(256), 0, 0, out, 12345u, iters); break;" for k in range(len(TESTS)))
    src = f"""// GENERATED by tools/seqbench.py — measurement tool, not part of the product library
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do {{ hipError_t e = (x); if (e != hipSuccess) {{ printf("HIP error %s at line %d\\n", hipGetErrorString(e), __LINE__); exit(1); }} }} while (0)
{''.join(ker)}
struct Test {{ const char* name; int reps; int inst; double instrPerInst; }};
static const Test tests[] = {{
{names}
}};
int main(int argc, char** argv) {{
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    unsigned long long* out; CK(hipMalloc(&out, 1 << 24));
    std::vector<unsigned long long> h(1 << 21);
    printf("{{\\"device\\": \\"%s\\", \\"cus\\": %d, \\"iters\\": %d, \\"results\\": [\\n", p.name, p.multiProcessorCount, iters);
    const int nt = sizeof(tests) / sizeof(tests[0]);
    for (int W : {{1, 4}}) {{
        const int blocks = p.multiProcessorCount * W;  // 4 waves per block = 1 per SIMD; W blocks per CU
        for (int k = 0; k < nt; ++k) {{
            for (int rep = 0; rep < 2; ++rep) {{
                switch (k) {{
{launches}
                }}
                CK(hipDeviceSynchronize());
            }}
            CK(hipMemcpy(h.data(), out, (size_t)blocks * 4 * 16, hipMemcpyDeviceToHost));
            std::vector<double> cyc, mhz;
            for (int i = 0; i < blocks * 4; ++i) {{ cyc.push_back((double)h[2 * i]); mhz.push_back(h[2 * i + 1] ? 100.0 * h[2 * i] / h[2 * i + 1] : 0); }}
            std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
            const double c = cyc[cyc.size() / 2] / ((double)iters * tests[k].reps * tests[k].inst);
            printf(" {{\\"name\\": \\"%s\\", \\"waves_per_simd\\": %d, \\"cycles_per_instance_per_wave\\": %.2f, \\"issue_cycles_per_instance\\": %.2f, "
                   "\\"issue_cycles_per_instruction\\": %.2f, \\"instructions\\": %.2f, \\"clock_mhz\\": %.0f}}%s\\n",
                   tests[k].name, W, c, c / W, c / W / tests[k].instrPerInst, tests[k].instrPerInst, mhz[mhz.size() / 2],
                   (W == 4 && k == nt - 1) ? "" : ",");
        }}
    }}
    printf("]}}\\n");
    return 0;
}}
"""
    open(SRC, "w").write(src)
    print(f"wrote {SRC}: {len(TESTS)} sequences")


def run(out=None):
    gen()
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", SRC, "-o", EXE])
    r = subprocess.run([EXE], capture_output=True, text=True, check=True)
    json.loads(r.stdout)
    if out:
        open(out, "w").write(r.stdout)
    else:
        print(r.stdout)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else None)
    else:
        gen()
