cd /tmp; export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT
EM="--no-bootstrap --no-cc-evalmult --no-bfv --no-hadamard --no-lt --no-cpu-baseline --steps 1 --warmup 0 --no-power --no-parity --evalmult-batch 256"
for f in 1 0; do
  FHE_KS_FUSED_MODUP=$f FHE_BENCH_NO_TORCH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_em_$f -- python $G/bench.py $EM > /tmp/prof_em_$f.log 2>&1
  cp $(ls -t /tmp/prof_em_$f/*/*kernel_stats.csv | head -1) $G/gpurun_out/em_stats_$f.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    FHE_KS_FUSED_MODUP=$f FHE_BENCH_NO_TORCH=1 timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_em_${f}_$c -- python $G/bench.py $EM > /tmp/pmc_em_${f}_$c.log 2>&1
    python3 - $f $c <<'PY'
import csv, glob, sys, collections
f, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float)
for fn in glob.glob(f"/tmp/pmc_em_{f}_{c}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].replace("void fhe::", "")[:70]] += float(r["Counter_Value"])
tot = sum(agg.values())
print(f"fused={f} {c} total KiB {tot:.0f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:8]:
    print(f"   {v:14.0f}  {k}")
PY
  done
done
