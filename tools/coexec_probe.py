#!/usr/bin/env python3
"""Round 5, VERDICT item 1(i) "overlap": do a column pass (HBM-bound, VALU idle 60 %) and a row pass (VALU-heavy) run faster side by
side than one after the other?  Measurement tool (timing only; the data is not meaningful).  Two host threads, one HIP stream each,
half of the headline batch each: thread A loops the forward column pass over half 1, thread B the forward row pass over half 2
(fhe_time_ntt dir 10 / 11).  Reported: each pass alone on its half, both at once (wall time of the pair), and the same for the
inverse pair, for the two column passes together and the two row passes together (controls: same-kind kernels should not gain).

  python tools/coexec_probe.py [B per half = 512] [iters = 6]      (GPU box; prints one JSON object)
"""
import ctypes as C
import importlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fh = importlib.import_module("openfhe-development_amd.fhe_hip")


def main():
    half = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    logN, L = 16, 30
    lib = fh.Lib()
    q, psi = lib.dcrt_chain(logN, L, 60)
    ctx = fh.Context(lib, logN, q, psi)
    nbytes = half * L * (1 << logN) * 8
    bufs = [ctx.malloc(nbytes) for _ in range(2)]
    streams = []
    for _ in range(2):
        s = C.c_void_p()
        lib.check(lib.L.fhe_stream_create(ctx.h, C.byref(s)))
        streams.append(s)

    def run(buf, d, st, out, key):
        ms = C.c_float()
        lib.check(lib.L.fhe_time_ntt(ctx.h, buf, None, L, half, d, iters, st, C.byref(ms)))
        out[key] = ms.value

    names = {10: "fwd_col", 11: "fwd_row", 12: "inv_row", 13: "inv_col"}
    res = {"half_batch": half, "iters": iters, "alone_ms": {}, "pairs": []}
    for d in names:  # warm-up + alone
        o = {}
        run(bufs[0], d, streams[0], o, "x")
        run(bufs[0], d, streams[0], o, "x")
        res["alone_ms"][names[d]] = round(o["x"], 4)
    for da, db in ((10, 11), (13, 12), (10, 13), (11, 12), (10, 12), (11, 13)):
        o = {}
        ctx.sync()
        ta = threading.Thread(target=run, args=(bufs[0], da, streams[0], o, "a"))
        tb = threading.Thread(target=run, args=(bufs[1], db, streams[1], o, "b"))
        t0 = time.perf_counter()
        ta.start(), tb.start()
        ta.join(), tb.join()
        wall = (time.perf_counter() - t0) * 1e3 / iters
        serial = res["alone_ms"][names[da]] + res["alone_ms"][names[db]]
        res["pairs"].append({"a": names[da], "b": names[db], "a_ms": round(o["a"], 4), "b_ms": round(o["b"], 4),
                             "wall_ms_per_pair": round(wall, 4), "serial_ms": round(serial, 4), "ratio": round(wall / serial, 4)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
