#!/usr/bin/env python3
"""Kernel-time census of the LOCKSTEP bootstrap (config 4's headline path): run under `rocprofv3 --kernel-trace`, then summarise.

  rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/boot_wide_profile.py run [cts] [group] [reps] [host threads] [threads of the narrow pass]
  python tools/boot_wide_profile.py summarise OUT/.../*_kernel_trace.csv [reps] > profiles/r04_bootstrap_wide_kernels.txt

`run` bootstraps `cts` ciphertexts once over host threads (first use of every composite), then `reps` + 1 times in lockstep groups;
`summarise` takes the launches of the last lockstep pass (the trace's tail, cut at the longest pause before it) and prints kernel
totals per bootstrap."""
import csv
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


MARKER = "checksum_kernel"  # a kernel the bootstrap never launches: run() brackets the lockstep passes with it


def run(cts, group, reps, threads=1, narrow_threads=8):
    sys.path.insert(0, ROOT)
    import numpy as np
    from openfhe_amd import boot_batch as bb
    from openfhe_amd import fhe_hip as fh
    prng = os.path.join(ROOT, "tests", "hal", "_build", "libdetprng.so")
    os.environ.setdefault("FHE_HIP_LIB", os.path.join(ROOT, "openfhe-development_amd", "csrc", "libfhe_hip.so"))
    r = bb.run_rank(17, 1 << 16, cts, narrow_threads, 1, 0, prng, warmup=0, key_threads=8)
    h = r.pop("handle")
    h.save_outputs()
    # phase markers for the trace readers below (summarise / pmc): one launch of fhe_checksum's kernel before and after the lockstep passes
    mlib = fh.Lib()
    mq, mpsi = mlib.dcrt_chain(12, 1, 60)
    mctx = fh.Context(mlib, 12, mq, mpsi)
    mt = mctx.tower(np.zeros((1, 1, 4096), np.uint64))

    def mark():
        mctx.checksum(mt)
        mctx.sync()
    mark()
    c0 = h.counters()
    sec = h.bootstrap_wide(group, reps, threads)
    c1 = h.counters()
    mark()
    n = (reps + 1) * cts
    print(f"lockstep: {cts / sec:.2f} bootstraps/s, groups of {group} over {threads} host thread(s); differing outputs {h.compare_saved()}; per bootstrap: "
          f"{(c1['launches'] - c0['launches']) / n:.1f} launches, "
          f"{(c1['operand_read_bytes'] + c1['operand_write_bytes'] - c0['operand_read_bytes'] - c0['operand_write_bytes']) / n / 1e9:.2f} GB of operands")
    h.close()


def sweep(cts, configs):
    """the same ciphertexts in lockstep under several (group, host threads) settings; every setting's outputs compared with the narrow pass"""
    sys.path.insert(0, ROOT)
    from openfhe_amd import boot_batch as bb
    prng = os.path.join(ROOT, "tests", "hal", "_build", "libdetprng.so")
    os.environ.setdefault("FHE_HIP_LIB", os.path.join(ROOT, "openfhe-development_amd", "csrc", "libfhe_hip.so"))
    r = bb.run_rank(17, 1 << 16, cts, 8, 1, 0, prng, warmup=0, key_threads=8)
    print(f"threaded narrow pass: {r['bootstraps_per_s']:.2f} bootstraps/s over 8 host threads (first pass: includes first-use checks)")
    h = r.pop("handle")
    h.save_outputs()
    import ctypes as C

    def alloc_stats():
        try:
            out = (C.c_uint64 * 6)()
            h.L.fhe_hal_alloc_stats(out)
            return list(out)
        except Exception:
            return [0] * 6
    for group, threads in configs:
        try:
            a0 = alloc_stats()
            sec = h.bootstrap_wide(group, 2, threads)
            a1 = alloc_stats()
            print(f"groups of {group} over {threads} host thread(s): {cts / sec:.2f} bootstraps/s; differing outputs {h.compare_saved()}; device memory in use "
                  f"{(a1[5] - a1[4]) / 2**30:.0f} GiB of {a1[5] / 2**30:.0f} ({a1[0] / 2**30:.0f} GiB of it cached released buffers); during the passes: "
                  f"{a1[2] - a0[2]} requests reached the device, {a1[1] - a0[1]} were served from another thread's cache, the caches went back {a1[3] - a0[3]} times",
                  flush=True)
        except Exception as e:
            print(f"groups of {group} over {threads} host thread(s): {type(e).__name__}: {str(e)[-260:]}", flush=True)
    h.close()


def summarise(path, cts, reps):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if MARKER in r[2]]
    if len(marks) >= 2:  # (run() brackets the lockstep passes with marker launches: the LAST two — the first-use checks of the narrow pass
        # launch the same kernel)
        rows = rows[marks[-2] + 1:marks[-1]]
        marked = True
    else:
        marked = False
    # the lockstep passes are the tail: reps + 1 equal passes; take the last one by launch count
    total = len(rows)
    # find the start of the lockstep phase: the longest gap in the second half of the trace precedes pack/unpack of a pass; simpler:
    # the last pass = the last 1/(reps+1) of the launches after the narrow pass, whose launches are the first `narrow` ones
    gaps = [(rows[i + 1][0] - rows[i][1], i) for i in range(total // 4, total - 1)]
    gaps.sort(reverse=True)
    cuts = sorted(i for _, i in gaps[:reps + 1])
    # (the narrow pass ends with the longest pause — the outputs are saved, the first wide ciphertext is packed — so the tail behind
    # the FIRST of the largest gaps holds all reps + 1 lockstep passes)
    last = rows if marked else (rows[cuts[0] + 1:] if cuts else rows)
    passes = reps + 1
    busy = sum(e - s for s, e, _ in last)
    span = last[-1][1] - last[0][0]
    print(f"{passes} lockstep passes: {len(last)} launches, GPU busy {busy / 1e6:.1f} ms, span {span / 1e6:.1f} ms for {passes} x {cts} bootstraps "
          f"= {busy / 1e6 / cts / passes:.2f} ms of kernel time per bootstrap")
    cts *= passes
    per = defaultdict(lambda: [0, 0])
    for s, e, nm in last:
        per[nm][0] += e - s
        per[nm][1] += 1
    print(f"{'ms/bootstrap':>13} {'share':>6} {'calls':>6} {'avg us':>8}  kernel")
    for nm, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"{t / 1e6 / cts:13.3f} {t / busy:6.3f} {c:6d} {t / c / 1e3:8.1f}  {nm[:120]}")


def pmc(trace_csv, counter_csvs, cts, reps, out_json):
    """HBM bytes per bootstrap of the lockstep passes from rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE runs of `run` (one counter
    per run, MI355X_MICROARCH.md: separate passes; FETCH_SIZE doubled on gfx950, units KiB).  The lockstep phase is found in the kernel
    trace of the SAME run as in summarise() (the tail behind the longest pauses) and the counters are joined by dispatch id."""
    import hashlib
    import json
    tot = {}
    by_kernel = {}
    expected = int(os.environ["FHE_PMC_EXPECTED_LAUNCHES"]) if os.environ.get("FHE_PMC_EXPECTED_LAUNCHES") else None
    for path in counter_csvs:
        tr = path.replace("counter_collection.csv", "kernel_trace.csv")
        rows = []
        with open(tr if os.path.exists(tr) else trace_csv, newline="") as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Dispatch_Id"])))
        rows.sort()
        n = len(rows)
        names = {}
        with open(tr if os.path.exists(tr) else trace_csv, newline="") as f:
            for r in csv.DictReader(f):
                if MARKER in r["Kernel_Name"]:
                    names[int(r["Dispatch_Id"])] = 1
        mk = [i for i, r in enumerate(rows) if r[2] in names]
        if len(mk) >= 2:  # run()'s LAST two marker launches bracket the lockstep passes exactly (the narrow pass's first-use checks launch it too)
            tail = {d for _, _, d in rows[mk[-2] + 1:mk[-1]]}
            dur = {d: e - s for s, e, d in rows[mk[-2] + 1:mk[-1]]}  # (kernels run one at a time under --pmc: stand-alone durations)
            with open(path, newline="") as f:
                for r in csv.DictReader(f):
                    d = int(r["Dispatch_Id"])
                    if d in tail:
                        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                        k = by_kernel.setdefault(r["Kernel_Name"], {})
                        k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                        k["calls_" + r["Counter_Name"]] = k.get("calls_" + r["Counter_Name"], 0) + 1
                        k["ns_" + r["Counter_Name"]] = k.get("ns_" + r["Counter_Name"], 0) + dur.get(d, 0)
            tot["launches_" + os.path.basename(os.path.dirname(os.path.dirname(path)))] = len(tail)
            continue
        gaps = sorted(((rows[i + 1][0] - rows[i][1], i) for i in range(n // 4, n - 1)), reverse=True)
        if expected is None:  # the first run: the tail behind the first of the longest pauses, as summarise() takes it
            cut = min(i for _, i in gaps[:reps + 1])
            expected = n - cut - 1
        else:  # the other runs: the pause (of the 12 longest) whose tail has the launch count closest to the first run's — the counter passes
            # serialise kernels and stretch other pauses (key generation, the narrow pass) beyond those between the lockstep passes
            cut = min((i for _, i in gaps[:12]), key=lambda i: abs((n - i - 1) - expected))
        tail = {d for _, _, d in rows[cut + 1:]}
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if int(r["Dispatch_Id"]) in tail:
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        tot["launches_" + os.path.basename(os.path.dirname(os.path.dirname(path)))] = len(tail)
    n = (reps + 1) * cts
    fetch, write = tot.get("FETCH_SIZE", 0.0) * 2048, tot.get("WRITE_SIZE", 0.0) * 1024
    h = hashlib.sha256()
    d = os.path.join(ROOT, "openfhe-development_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) on `python tools/boot_wide_profile.py run "
                   f"{cts} <group> {reps} <threads>` (N = 2^17, 24 Q + 8 P limbs): counters of the launches of the {reps + 1} lockstep passes "
                   "(the trace's tail), FETCH_SIZE x 2048 + WRITE_SIZE x 1024 bytes, per bootstrap",
           "kernel_source_sha": h.hexdigest()[:16], "bootstraps": n, "fetch_bytes_per_bootstrap": fetch / n, "write_bytes_per_bootstrap": write / n,
           "bytes_per_bootstrap": (fetch + write) / n, "detail": tot}
    # per kernel: HBM bytes per launch and the rate over the kernel's stand-alone duration (serialised launches of the counter runs)
    table = []
    for nm, k in by_kernel.items():
        fb, wb = k.get("FETCH_SIZE", 0.0) * 2048, k.get("WRITE_SIZE", 0.0) * 1024
        calls = max(k.get("calls_FETCH_SIZE", 0), k.get("calls_WRITE_SIZE", 0))
        ns = (k.get("ns_FETCH_SIZE", 0) + k.get("ns_WRITE_SIZE", 0)) / max(1, (1 if "ns_FETCH_SIZE" in k else 0) + (1 if "ns_WRITE_SIZE" in k else 0))
        table.append({"kernel": nm[:110], "calls": calls, "ms_per_bootstrap": round(ns / 1e6 / n, 3), "GB_per_bootstrap": round((fb + wb) / 1e9 / n, 3),
                      "fetch_MB_per_call": round(fb / 1e6 / max(1, calls), 2), "write_MB_per_call": round(wb / 1e6 / max(1, calls), 2),
                      "HBM_GBps": round((fb + wb) / ns, 1) if ns else None})
    table.sort(key=lambda r: -r["ms_per_bootstrap"])
    out["by_kernel"] = table
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "by_kernel"}))
    for r in table[:24]:
        print(f"{r['ms_per_bootstrap']:8.3f} ms  {r['GB_per_bootstrap']:7.3f} GB  {str(r['HBM_GBps']):>7} GB/s  {r['calls']:6d}  {r['kernel'][:90]}")


def sq(counter_csv, cts, reps, out_json):
    """per-kernel SQ counters of the lockstep passes (one rocprofv3 --pmc run with several SQ_* counters): VALU instructions per
    launch, the share of wave-time spent issuing VALU work / waiting for memory, and VALU issue slots used per SIMD cycle of the kernel's
    stand-alone duration (1024 SIMDs; a 64-lane VALU instruction occupies its SIMD for at least 4 cycles)"""
    import json
    tr = counter_csv.replace("counter_collection.csv", "kernel_trace.csv")
    rows = []
    with open(tr, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Dispatch_Id"]), r["Kernel_Name"]))
    rows.sort()
    mk = [i for i, r in enumerate(rows) if MARKER in r[3]]
    tailrows = rows[mk[-2] + 1:mk[-1]] if len(mk) >= 2 else rows
    dur = {d: e - s for s, e, d, _ in tailrows}
    per = {}
    seen = {}
    with open(counter_csv, newline="") as f:
        for r in csv.DictReader(f):
            d = int(r["Dispatch_Id"])
            if d not in dur:
                continue
            k = per.setdefault(r["Kernel_Name"], defaultdict(float))
            k[r["Counter_Name"]] += float(r["Counter_Value"])
            if (d, 0) not in seen:
                seen[(d, 0)] = 1
                k["_calls"] += 1
                k["_ns"] += dur[d]
    n = (reps + 1) * cts
    table = []
    for nm, k in per.items():
        ns, waves = k["_ns"], max(1.0, k.get("SQ_WAVES", 0.0))
        table.append({"kernel": nm[:110], "calls": int(k["_calls"]), "ms_per_bootstrap": round(ns / 1e6 / n, 3),
                      "valu_insts_per_wave": round(k.get("SQ_INSTS_VALU", 0.0) / waves, 1),
                      "valu_issue_ns_per_inst_per_simd": round(ns * 1024 / max(1.0, k.get("SQ_INSTS_VALU", 0.0)), 2),
                      "valu_active_share_of_wave_cycles": round(k.get("SQ_ACTIVE_INST_VALU", 0.0) / max(1.0, k.get("SQ_WAVE_CYCLES", 0.0)), 4),
                      "wait_inst_share_of_wave_cycles": round(k.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, k.get("SQ_WAVE_CYCLES", 0.0)), 4),
                      "waves_resident_per_simd": round(k.get("SQ_WAVE_CYCLES", 0.0) / max(1.0, k.get("SQ_BUSY_CYCLES", 0.0)) / 4.0, 2),
                      "raw": {c: v for c, v in k.items() if not c.startswith("_")}})
    table.sort(key=lambda r: -r["ms_per_bootstrap"])
    json.dump({"_how": "rocprofv3 --kernel-trace --pmc SQ_* on `tools/boot_wide_profile.py run` (one host thread): launches of the lockstep passes, by kernel",
               "bootstraps": n, "by_kernel": table}, open(out_json, "w"), indent=1)
    if not any("SQ_INSTS_VALU" in r["raw"] for r in table):  # another counter set: per call and per second
        for r in table[:12]:
            ns = r["ms_per_bootstrap"] * 1e6 * n
            print(f"{r['ms_per_bootstrap']:7.3f} ms  " + "  ".join(f"{c} {v / max(1, r['calls']):.4g}/call {v / ns:.3f}/ns" for c, v in sorted(r["raw"].items())) + f"  {r['kernel'][:50]}")
        return
    for r in table[:16]:
        print(f"{r['ms_per_bootstrap']:7.3f} ms {r['valu_insts_per_wave']:8.1f} VALU/wave {r['valu_issue_ns_per_inst_per_simd']:6.2f} ns/inst/SIMD "
              f"valu {r['valu_active_share_of_wave_cycles']:.3f} wait {r['wait_inst_share_of_wave_cycles']:.3f} waves/SIMD {r['waves_resident_per_simd']:5.2f}  {r['kernel'][:70]}")


if __name__ == "__main__":
    if sys.argv[1] == "sweep":
        sweep(int(sys.argv[2]), [tuple(int(v) for v in a.split("x")) for a in sys.argv[3:]])
    elif sys.argv[1] == "pmc":  # pmc <out.json> <cts> <reps> <counter_collection.csv>...
        pmc(None, sys.argv[5:], int(sys.argv[3]), int(sys.argv[4]), sys.argv[2])
    elif sys.argv[1] == "sq":  # sq <out.json> <cts> <reps> <counter_collection.csv>
        sq(sys.argv[5], int(sys.argv[3]), int(sys.argv[4]), sys.argv[2])
    elif sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 32, int(sys.argv[3]) if len(sys.argv) > 3 else 32, int(sys.argv[4]) if len(sys.argv) > 4 else 2,
            int(sys.argv[5]) if len(sys.argv) > 5 else 1, int(sys.argv[6]) if len(sys.argv) > 6 else 8)
    else:
        summarise(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 32, int(sys.argv[4]) if len(sys.argv) > 4 else 2)
