#!/bin/bash
# GPU session r03-o: kernel trace of one wide bootstrap pass (16 ciphertexts in lockstep): where the time of a K-wide bootstrap goes.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
cat > /tmp/bw.py <<PY
import sys
sys.path.insert(0, "$G")
from openfhe_amd import boot_batch as bb
r = bb.run_rank(17, 65536, 16, 8, 1, 0, "$B/libdetprng.so", warmup=0, key_threads=8)
h = r.pop("handle")
t = h.bootstrap_wide(0, 1)
print("wide 16:", t, 16 / t)
h.close()
PY
FHE_HAL_REQUIRE_DEVICE=1 OMP_NUM_THREADS=8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $G/gpurun_out/prof_r03o -- python3 /tmp/bw.py > $G/gpurun_out/prof_r03o.log 2>&1; echo "exit code $?"
grep -a "wide 16" $G/gpurun_out/prof_r03o.log
python3 - <<PY | tee $G/gpurun_out/wide_kernels_o.txt
import csv, glob, collections
f = sorted(glob.glob("$G/gpurun_out/prof_r03o/*/*kernel_trace.csv"))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last wide pass = the last 1/2 of the kernels whose grid is large: take launches after the last pause > 5 ms that hold > 20% of kernel time
cuts, reach = [0], rows[0][1]
for i in range(1, len(rows)):
    if rows[i][0] - reach > 5e6: cuts.append(i)
    reach = max(reach, rows[i][1])
cuts.append(len(rows))
wins = [rows[a:b] for a, b in zip(cuts, cuts[1:])]
tot = sum(e - s for s, e, _ in rows)
win = [w for w in wins if sum(e - s for s, e, _ in w) > tot / 5][-1]
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    k = n.split("(")[0].replace("void fhe::", "")[:70]
    by[k][0] += e - s; by[k][1] += 1
busy = sum(v[0] for v in by.values())
print(f"last window: {len(win)} launches, kernel time {busy / 1e6:.1f} ms, span {(win[-1][1] - win[0][0]) / 1e6:.1f} ms")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"{v[0] / 1e6:9.2f} ms {v[1]:6d} calls {v[0] / v[1] / 1e3:9.1f} us  {k}")
PY
rm -rf $G/gpurun_out/prof_r03o
