#!/bin/bash
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
echo "== parity (ntt only)"; timeout 600 python -m pytest tests -m gpu -x -q -k "ntt or config1" 2>&1 | tail -3
echo "== microbench"; timeout 300 ./tools/microbench 4 > gpurun_out/microbench2.json; grep -E "asm_|cache_" gpurun_out/microbench2.json
echo "== T1 sweep"
for t1 in 4 5 6 7 8; do FHE_NTT_T1=$t1 timeout 300 python tools/ntt_sweep.py; done | tee gpurun_out/sweep_t1.jsonl
echo "== chunk sweep (T1=4)"
for ch in 1 2 4 8 16; do FHE_NTT_CHUNK=$ch timeout 300 python tools/ntt_sweep.py; done | tee gpurun_out/sweep_chunk.jsonl
echo "== chunk sweep (T1=8)"
for ch in 2 4 8; do FHE_NTT_T1=8 FHE_NTT_CHUNK=$ch timeout 300 python tools/ntt_sweep.py; done | tee -a gpurun_out/sweep_chunk.jsonl
echo "== rocprof kernel stats"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 256 --no-cpu-baseline --no-evalmult > $GRAFT_REPO_ROOT/gpurun_out/rocprof2.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof2 -name "*kernel_stats.csv" | head -1); echo $f; head -8 $f
echo "== rocprof pmc (SQ)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -- python $GRAFT_REPO_ROOT/tools/ntt_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lds -- python $GRAFT_REPO_ROOT/tools/ntt_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_lds.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python $GRAFT_REPO_ROOT/tools/ntt_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python $GRAFT_REPO_ROOT/tools/ntt_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out -name "*counter_collection.csv" | head; du -sh gpurun_out
