#!/bin/bash
# Round 5, second GPU call: A/B of the window block builds (shipped, exp_spec = speculative gather only, exp_spec2 = + records
# at the previous window's end + first-extension prefetch), PMC passes of the newest one, the new bench lines and the new
# full-size reduce-side tests.   usage (GPU): tools/r5_second_call.sh <tag> "<libs>" [pmc-lib]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
tag=${1:-r05b}
LIBS=${2:-default spec2 spec default}
PMCLIB=${3:-spec2}
O=gpurun_out/$tag; mkdir -p $O
bash tools/ab.sh $tag $LIBS
for n in $PMCLIB; do
  if [ "$n" = default ]; then unset S3S_CODEC_LIB; else export S3S_CODEC_LIB=$R/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$n.so; fi
  P=$R/gpurun_out/$tag/pmc_$n; mkdir -p $P
  CMD="python $R/bench.py --no-cpu-baseline --no-secondary --maps-per-gpu 2 --task-threads 1 --steps 2 --warmup 1"
  i=0
  while read -r line; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $line -d $P/pmc_$i -o p -- $CMD > $P/pmc_$i.log 2>&1)
  done <<'PMC'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
PMC
  python tools/summarize_prof.py $P > $P/summary.txt 2>&1
  echo "== PMC $n"; grep "lz4_compress" $P/summary.txt | cut -c1-600
done
unset S3S_CODEC_LIB
echo "== hbm-bound stages"
timeout 300 python bench.py --hbm-stages-only 2> $O/hbm_err.txt | tee $O/hbm_stages.json | cut -c1-3000
echo "== new reduce-side lines"
for w in terasort-10g-200p-lzf terasort-100g-2000p-zstd; do
  timeout 400 python bench.py --workload $w --direction decompress --maps-per-gpu 4 --steps 8 --warmup 3 --cpu-seconds 2.5 2> $O/err_$w.txt | tee $O/bench_$w.json | cut -c1-1200
done
echo "== new full-size tests"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "zstd or lzf" --durations=10 2>&1 | tail -25 | tee $O/pytest_fullsize_new.txt
