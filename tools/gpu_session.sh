#!/bin/bash
# One parameterised record script for the round's GPU sessions (replaces the per-session tools/gpu_session_rNN_x.sh files).
#   usage: gpurun --timeout T -- tools/gpu_session.sh <session> [args...]        output: gpurun_out/<session>_*.{json,log}
# (the one-shot session scripts of rounds 2-4 — gpu_session_*.sh, gpu_record.sh — are in the history: `git show 12b9d21:tools/`)
# (the sessions of the rejected round-5 NTT experiments — chunks, persist, t1 — are in commit 3b1f4e6 together with their kernels)
# sessions:
#   mall          tools/mallbench (Infinity-Cache go/no-go) + the headline leg at --batch 2,4,8,16,64,1024
#   ntt [env...]  the headline leg alone (20 steps), with optional FHE_* environment assignments
#   wide          lockstep tests, cc->EvalMult leg, bootstrap (group x threads) sweep
#   ut            the reference's unit tests on the GPU with the per-member mirror counts
#   record [tag]  the round record (PMC passes, bootstrap census + counters, bench line, rocprof kernel stats); then tools/collect_profiles.py tag
#   abl libs...   the headline leg with each tools/abl6/libfhe_hip_<lib>.so (timing-only ablations)
#   tests [k]     pytest -m gpu (optionally -k <k>)
set -u
S=${1:-help}; shift || true
mkdir -p gpurun_out
NTT_ONLY="--no-bootstrap --no-cc-evalmult --no-cpu-baseline --no-evalmult --no-bfv --no-hadamard --no-lt"
case "$S" in
  mall)
    timeout 600 tools/mallbench | tee gpurun_out/mall_bench.json
    for b in 2 4 8 16 64 1024; do
      FHE_BENCH_NO_TORCH=1 timeout 600 python bench.py $NTT_ONLY --no-parity --batch $b --steps 20 --warmup 3 2>gpurun_out/mall_b$b.err | tail -1 | tee gpurun_out/mall_b$b.json | cut -c1-400
    done ;;
  ntt)
    for kv in "$@"; do export "$kv"; done
    FHE_BENCH_NO_TORCH=1 timeout 900 python bench.py $NTT_ONLY --steps 20 --warmup 3 2>gpurun_out/ntt.err | tail -1 | tee gpurun_out/ntt.json | cut -c1-600 ;;
  wide)  # round-5 HAL changes: lockstep shim tests, the cc->EvalMult leg's numbers (views), the bootstrap settings that used to thrash
    timeout 1200 python -m pytest tests/test_hal_shim.py tests/test_multi_gpu_gloo.py tests/test_multi_gpu_rccl_one_rank.py -m gpu -q -x 2>&1 | tail -4
    timeout 900 python - <<'PY' 2>&1 | tail -5 | tee gpurun_out/wide_ccm.json
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
r = bench.cc_evalmult_leg(True, os.path.join(os.getcwd(), "openfhe-development_amd", "csrc", "libfhe_hip.so"))
print(json.dumps(r))
PY
    timeout 1500 python tools/boot_wide_profile.py sweep 64 16x2 16x4 32x2 8x4 16x2 2>&1 | tail -8 | tee gpurun_out/wide_sweep.txt ;;
  ut)    # the reference's own 1729 unit tests on the backend: totals, members with host-mirror executions, decline reasons
    cd /tmp && FHE_HIP_LIB=$GRAFT_REPO_ROOT/openfhe-development_amd/csrc/libfhe_hip.so OMP_NUM_THREADS=8 timeout 900 \
      $GRAFT_REPO_ROOT/tests/hal/_build/ut_hip "--gtest_filter=-*SERIALIZE*:UTBinInt.GetInternalRepresentation" > $GRAFT_REPO_ROOT/gpurun_out/ut_full.log 2>&1
    cd $GRAFT_REPO_ROOT
    grep -E "tests ran|^hal: |halcomposite|haldecline" gpurun_out/ut_full.log | tee gpurun_out/ut_trace.txt
    grep -E "^halmember" gpurun_out/ut_full.log | awk '$4 > 0' | sort -k4 -n -r | tee -a gpurun_out/ut_trace.txt ;;
  record)  # THE round record: counter passes first, their summaries next to the sources on the box, then the bench line (which quotes
           # them by kernel-source identity), rocprof kernel stats of the legs, the lockstep bootstrap's census and HBM counters
    R=${1:-r05}
    export FHE_BENCH_NO_TORCH=1
    cd /tmp && export TMPDIR=/tmp
    G=$GRAFT_REPO_ROOT
    D=/tmp/rec; mkdir -p $D   # (traces and counter files are large: only summaries go to gpurun_out, which is merged back up to 64 MiB)
    NTTLEG="--no-bootstrap --no-cc-evalmult --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-evalmult --no-bfv --no-hadamard --no-lt"
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/pmc_${R}_$c -- python $G/bench.py $NTTLEG > $D/pmc_${R}_$c.log 2>&1
    done
    SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
    timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $D/pmc_${R}_sq_ntt -- python $G/bench.py $NTTLEG > $D/pmc_${R}_sq_ntt.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $D/pmc_${R}_sq_evalmult -- python $G/bench.py --no-bootstrap --no-cc-evalmult --batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-bfv --no-hadamard --no-lt > $D/pmc_${R}_sq_evalmult.log 2>&1
    echo "== bootstrap (64 ciphertexts, groups of 16 on 2 host threads): census, then FETCH / WRITE passes"
    timeout 900 rocprofv3 --kernel-trace --output-format csv -d $D/boot_${R}_trace -- python $G/tools/boot_wide_profile.py run 64 16 2 2 > $D/boot_${R}_trace.log 2>&1
    tail -1 $D/boot_${R}_trace.log
    if ! ls $D/boot_${R}_trace/*/*kernel_trace.csv > /dev/null 2>&1; then  # (rocprofv3 crashes on the two-thread program in some runs: one host thread, same launches)
      timeout 900 rocprofv3 --kernel-trace --output-format csv -d $D/boot_${R}_trace -- python $G/tools/boot_wide_profile.py run 64 16 2 1 1 > $D/boot_${R}_trace.log 2>&1
      tail -1 $D/boot_${R}_trace.log
    fi
    f=$(ls -t $D/boot_${R}_trace/*/*kernel_trace.csv | head -1)
    (echo "# rocprofv3 --kernel-trace of \`tools/boot_wide_profile.py run 64 16 2 2\` (round record: 64 ciphertexts at config 4's shape, lockstep groups of 16 on 2 host threads — bench.py's setting — 3 passes)"; tail -1 $D/boot_${R}_trace.log; python $G/tools/boot_wide_profile.py summarise $f 64 2) > $G/gpurun_out/${R}_bootstrap_wide_kernels.txt
    head -16 $G/gpurun_out/${R}_bootstrap_wide_kernels.txt | cut -c1-150
    for c in FETCH_SIZE WRITE_SIZE; do
      # (ONE host thread throughout: rocprofv3 --pmc segfaults on the multi-threaded program in most runs; same launches, same bytes)
      timeout 1200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/boot_${R}_$c -- python $G/tools/boot_wide_profile.py run 64 16 2 1 1 > $D/boot_${R}_$c.log 2>&1
      tail -1 $D/boot_${R}_$c.log | cut -c1-200
    done
    cd $G
    python tools/boot_wide_profile.py pmc profiles/${R}_bootstrap_pmc.json 64 2 $(ls -t $D/boot_${R}_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls -t $D/boot_${R}_WRITE_SIZE/*/*counter_collection.csv | head -1) | cut -c1-400
    FHE_PROFILE_DIR=$D python tools/collect_profiles.py $R --pmc-only
    cp profiles/${R}_pmc_traffic.json profiles/${R}_pmc_valu.json profiles/${R}_bootstrap_pmc.json gpurun_out/ 2>/dev/null
    unset FHE_BENCH_NO_TORCH
    echo "== bench (default flags, as the driver runs it)"
    timeout 1500 python bench.py 2>gpurun_out/bench_$R.err | tail -1 | tee gpurun_out/bench_$R.json | cut -c1-500
    export FHE_BENCH_NO_TORCH=1
    cd /tmp
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_${R}_ntt -- python $G/bench.py --no-bootstrap --no-cc-evalmult --no-cpu-baseline --no-parity --no-evalmult --no-bfv --no-hadamard --no-lt > $D/prof_${R}_ntt.log 2>&1
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_${R}_evalmult -- python $G/bench.py --no-bootstrap --no-cc-evalmult --batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-bfv --no-hadamard --no-lt > $D/prof_${R}_evalmult.log 2>&1
    cd $G
    f=$(ls -t $D/prof_${R}_ntt/*/*kernel_stats.csv | head -1); cp $f gpurun_out/${R}_rocprof_kernel_stats_ntt_leg.csv; head -6 $f | cut -c1-170
    f=$(ls -t $D/prof_${R}_evalmult/*/*kernel_stats.csv | head -1); cp $f gpurun_out/${R}_rocprof_kernel_stats_evalmult256.csv; head -10 $f | cut -c1-170 ;;
  bootpmc)  # only the lockstep bootstrap's FETCH / WRITE counter passes, on ONE host thread throughout (rocprofv3 --pmc segfaults on the
            # multi-threaded program in most runs; the launches and their bytes are the same whatever the number of host threads)
    R=${1:-r05}; G=$GRAFT_REPO_ROOT; D=/tmp/rec; mkdir -p $D
    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do
      for try in 1 2 3; do
        rm -rf $D/boot_${R}_$c
        timeout 1200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/boot_${R}_$c -- python $G/tools/boot_wide_profile.py run 64 16 2 1 1 > $D/boot_${R}_$c.log 2>&1 && break
        echo "pass $c failed (try $try)"
      done
      tail -1 $D/boot_${R}_$c.log | cut -c1-200
    done
    cd $G
    python tools/boot_wide_profile.py pmc gpurun_out/${R}_bootstrap_pmc.json 64 2 $(ls -t $D/boot_${R}_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls -t $D/boot_${R}_WRITE_SIZE/*/*counter_collection.csv | head -1) | cut -c1-600 ;;
  census)  # kernel census of the lockstep bootstrap alone (one host thread)
    R=${1:-r06}; G=$GRAFT_REPO_ROOT; D=/tmp/rec; mkdir -p $D
    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace --output-format csv -d $D/boot_${R}_trace -- python $G/tools/boot_wide_profile.py run 64 16 2 1 1 > $D/boot_${R}_trace.log 2>&1
    f=$(ls -t $D/boot_${R}_trace/*/*kernel_trace.csv | head -1)
    (echo "# rocprofv3 --kernel-trace of \`tools/boot_wide_profile.py run 64 16 2 1 1\` (64 ciphertexts at config 4's shape, lockstep groups of 16, ONE host thread — rocprofv3 crashes on the two-thread program in some runs; same launches — 3 passes)"; tail -1 $D/boot_${R}_trace.log; python $G/tools/boot_wide_profile.py summarise $f 64 2) > $G/gpurun_out/${R}_bootstrap_wide_kernels.txt
    head -30 $G/gpurun_out/${R}_bootstrap_wide_kernels.txt | cut -c1-150 ;;
  bootsq)  # SQ counters of the lockstep bootstrap by kernel (one host thread, as bootpmc)
    R=${1:-r06}; G=$GRAFT_REPO_ROOT; D=/tmp/rec; mkdir -p $D
    cd /tmp && export TMPDIR=/tmp
    SQ=${FHE_BOOT_COUNTERS:-"SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"}
    for try in 1 2 3; do
      rm -rf $D/boot_${R}_sq
      timeout 1200 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $D/boot_${R}_sq -- python $G/tools/boot_wide_profile.py run 64 16 2 1 1 > $D/boot_${R}_sq.log 2>&1 && break
      echo "SQ pass failed (try $try)"
    done
    tail -1 $D/boot_${R}_sq.log | cut -c1-200
    cd $G
    python tools/boot_wide_profile.py sq gpurun_out/${R}_bootstrap_${2:-sq}.json 64 2 $(ls -t $D/boot_${R}_sq/*/*counter_collection.csv | head -1) ;;
  abl)   # timing-only ablation builds of the library (tools/abl6/*.so, built here with -DFHE_ABL_*; results are wrong: --no-parity)
    for lib in "" "$@"; do
      name=${lib:-default}
      FHE_HIP_LIB=${lib:+$PWD/tools/abl6/libfhe_hip_$lib.so} FHE_BENCH_NO_TORCH=1 timeout 600 python bench.py $NTT_ONLY --no-parity --steps 10 --warmup 2 2>gpurun_out/abl_$name.err | tail -1 > gpurun_out/abl_$name.json
      python - "$name" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/abl_{sys.argv[1]}.json").read())
print(sys.argv[1], d["ms_per_step"], d["roofline"]["per_kernel_ms"])
PY
    done ;;
  tests)
    if [ $# -ge 1 ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$1" 2>&1 | tail -15 | tee gpurun_out/tests.log
    else timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/tests.log; fi ;;
  *) echo "sessions: mall | ntt | tests"; exit 2 ;;
esac
