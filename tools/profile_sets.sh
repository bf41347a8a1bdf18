#!/bin/bash
# (GPU) round 4: profile the kernels AS THEY SHIP — rocprofv3 kernel trace + separate PMC passes (FETCH / WRITE / SQ x2 / TCC)
# of the five dominant-kernel commands, with the headline's launch shape (--maps-per-gpu 8: two map tasks per compress launch).
#   gpurun --timeout 900 -- 'bash tools/profile_sets.sh r04b "compress snappy_compress decompress crc2000 snappy_decompress"'
# then here:  python tools/profile_report.py r04b      (writes profiles/r04b_*_rocprofv3_summary.{md,json}, profiles/traffic_latest.json)
tag=${1:-r04b}
SETS=${2:-compress snappy_compress decompress crc2000 snappy_decompress}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python tools/src_stamp.py > $O/kernel_sources_sha256.txt
prof() {  # prof <name> <bench args...>
  local P=$O/prof_$1; shift; mkdir -p $P
  local CMD="python $R/bench.py --no-cpu-baseline --no-secondary --no-image-check --steps 3 --warmup 1 $*"
  if [ "$1" = "--hbm-stages-only" ]; then CMD="python $R/bench.py --hbm-stages-only"; fi
  echo "$CMD" > $P/command.txt
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats -d $P/trace -o t -- $CMD > $P/trace.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o p -- $CMD > $P/pmc_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o p -- $CMD > $P/pmc_write.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $P/pmc_sq1 -o p -- $CMD > $P/pmc_sq1.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES -d $P/pmc_sq2 -o p -- $CMD > $P/pmc_sq2.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $P/pmc_tcc -o p -- $CMD > $P/pmc_tcc.log 2>&1
  cd $R
  python tools/summarize_prof.py $P --md > $P/summary.md 2>&1
  # the trees of CSVs stay on the box: only the summaries travel back
  rm -rf $P/trace $P/pmc_fetch $P/pmc_write $P/pmc_sq1 $P/pmc_sq2 $P/pmc_tcc
}
for s in $SETS; do
  case $s in
    compress) prof compress ;;
    snappy_compress) prof snappy_compress --workload tpcds-wide-100g-200p-snappy --maps-per-gpu 4 ;;
    decompress) prof decompress --direction decompress --maps-per-gpu 8 ;;
    crc2000) prof crc2000 --workload terasort-100g-2000p-lz4-crc32 --maps-per-gpu 4 ;;
    snappy_decompress) prof snappy_decompress --workload tpcds-wide-100g-200p-snappy --direction decompress --maps-per-gpu 8 ;;
    zstd) prof zstd --workload terasort-10g-200p-zstd --direction decompress ;;
    hbm) prof hbm --hbm-stages-only ;;  # round 5: checksum-only / xxHash32 lines on a 1 GiB range
  esac
  head -12 $O/prof_$s/summary.md | cut -c1-200
done
