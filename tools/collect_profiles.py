#!/usr/bin/env python3
"""copies the judged summaries of the last `tools/gpu_session.sh record <tag>` run from gpurun_out/ (scratch) into profiles/"""
import collections, csv, glob, hashlib, json, os, shutil, sys


def newest(pattern):  # gpurun_out accumulates the runs of a round: take the latest file
    return max(glob.glob(pattern), key=os.path.getmtime)

def source_sha():  # the same identity bench.py computes: the kernel sources the measured library was built from
    h = hashlib.sha256()
    d = os.path.join("openfhe-development_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


R = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = os.environ.get("FHE_PROFILE_DIR", "gpurun_out")  # where the record session left the rocprofv3 output directories
# --pmc-only: on the GPU box, between the counter passes and the bench run of a record session — the bench line then quotes the
# counters of the very sources it runs (it keys them by kernel-source identity)
PMC_ONLY = "--pmc-only" in sys.argv
os.makedirs("profiles", exist_ok=True)
if not PMC_ONLY:
    shutil.copy(f"gpurun_out/bench_{R}.json", f"profiles/{R}_bench.json")
    for src, dst in ((f"gpurun_out/prof_{R}/*/*kernel_stats.csv", f"profiles/{R}_rocprof_kernel_stats_bench.csv"),
                     (f"gpurun_out/prof_{R}_ntt/*/*kernel_stats.csv", f"profiles/{R}_rocprof_kernel_stats_ntt_leg.csv"),
                     (f"gpurun_out/prof_{R}_evalmult/*/*kernel_stats.csv", f"profiles/{R}_rocprof_kernel_stats_evalmult256.csv")):
        try:
            shutil.copy(newest(src), dst)
        except ValueError:  # (a pass that was not part of the session)
            pass
names = {"<true, false,": "fwd_column_pass", "<false, false,": "fwd_row_pass", "<false, true,": "inv_row_pass",
         "<true, true,": "inv_column_pass"}
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(f"{SRC}/pmc_{R}_{c}/*/*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "ntt_static_kernel" in r["Kernel_Name"]:
            key = [v for k, v in names.items() if k in r["Kernel_Name"]][0]
            agg[key].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res.setdefault(k, {})[c] = sum(v) / len(v)
out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 2 "
               "--warmup 1 --no-cpu-baseline --no-evalmult` (N=2^16, L=30, B=1024), tools/gpu_session.sh record. Counter units are "
               "KiB; FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md §HBM).",
       "workload": "logN16_L30_B1024", "per_launch_bytes": {},
       # identity of the kernel sources ON THE GPU BOX in that run (bench.py recorded it in its own line)
       "kernel_source_sha": source_sha() if PMC_ONLY else
       ((json.load(open(f"profiles/{R}_bench.json")).get("roofline") or {}).get("kernel_source_sha") or source_sha())}
for k, v in res.items():
    out["per_launch_bytes"][k] = {"fetch_bytes": v["FETCH_SIZE"] * 2048, "write_bytes": v["WRITE_SIZE"] * 1024,
                                  "total": v["FETCH_SIZE"] * 2048 + v["WRITE_SIZE"] * 1024,
                                  "algorithmic": 2 * 8 * 65536 * 30 * 1024}
json.dump(out, open(f"profiles/{R}_pmc_traffic.json", "w"), indent=1)

# SQ counters (one pass): VALU instructions per wave and the share of wave-cycles a VALU instruction was issuing, per kernel
SQ = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY")


def short(name):
    for k, v in names.items():
        if "ntt_static_kernel" in name and k in name:
            return f"{v} ({name.replace('void fhe::', '').split('(fhe::')[0]})"
    return name.replace("void fhe::", "").split("(")[0]


valu = {"_how": "rocprofv3 --kernel-trace --pmc " + " ".join(SQ) + " (one pass, 8 SQ slots) on `python bench.py --steps 2 --warmup 1 ...` (NTT leg: "
                "N=2^16, L=30, B=1024) and on the EvalMult leg at batch 256 (tools/gpu_session.sh record). SQ_*_CYCLES and SQ_ACTIVE_INST_* count "
                "quad-cycles per wave (MI355X_MICROARCH.md); values are averages per launch.",
        "kernel_source_sha": out["kernel_source_sha"], "legs": {}}
for leg in ("ntt", "evalmult"):
    try:
        f = newest(f"{SRC}/pmc_{R}_sq_{leg}/*/*counter_collection.csv")
    except ValueError:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    legd = {}
    for k, c in per.items():
        avg = {n: sum(v) / len(v) for n, v in c.items()}
        wc, waves = avg.get("SQ_WAVE_CYCLES", 0), avg.get("SQ_WAVES", 0)
        if not wc or not waves:
            continue
        legd[k] = {"launches": len(c["SQ_WAVES"]), "waves": waves, "valu_instructions_per_wave": round(avg["SQ_INSTS_VALU"] / waves, 1),
                   "active_valu_over_wave_cycles": round(avg["SQ_ACTIVE_INST_VALU"] / wc, 4),
                   "wait_inst_any_over_wave_cycles": round(avg["SQ_WAIT_INST_ANY"] / wc, 4), "wait_any_over_wave_cycles": round(avg["SQ_WAIT_ANY"] / wc, 4),
                   "active_any_over_wave_cycles": round(avg["SQ_ACTIVE_INST_ANY"] / wc, 4), "sq_busy_cycles": avg["SQ_BUSY_CYCLES"], "wave_cycles": wc}
    valu["legs"][leg] = legd
if valu["legs"]:
    json.dump(valu, open(f"profiles/{R}_pmc_valu.json", "w"), indent=1)
if PMC_ONLY:
    sys.exit(0)
b = json.load(open(f"profiles/{R}_bench.json"))
print(json.dumps({k: b[k] for k in ("value", "ms_per_step", "hbm_roofline_frac_fwd_inv", "roofline", "cpu_baseline", "evalmult")}, indent=1))
