#!/bin/bash
# Round 5, call 5: xxHash32 pre-pass modes (0 = one wavefront per chunk, 1 = quads + non-temporal, 2 = quads + ordinary loads):
# hash stage time, headline, HBM traffic of the compress launch pair; HBM-bound stage lines; LDS counters of the CRC kernel.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
tag=${1:-r05e}
O=gpurun_out/$tag; mkdir -p $O
for m in 1 0 2 1 0; do
  echo "== headline, S3S_XXH=$m"
  S3S_XXH=$m timeout 300 python bench.py --no-cpu-baseline --no-secondary --verify 2> $O/err_xxh$m.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], 'GB/s', d['stages_ms_per_library_call'])"
done
echo "== hbm-bound stages"
timeout 300 python bench.py --hbm-stages-only 2> $O/hbm_err.txt > $O/hbm_stages.json
python - $O <<'PY'
import json, sys
d=json.load(open(sys.argv[1]+'/hbm_stages.json'))['hbm_bound_stages']
for k,v in d.items():
    if isinstance(v,dict):
        r=v.get('roofline',{})
        print(f"{k:45s} value {v.get('value')}  achieved {r.get('achieved')} frac {r.get('frac')} ms {r.get('avg_kernels_ms', v.get('ms'))} {v.get('matches_zlib','')}")
PY
for m in 1 2 0; do
  P=$R/$O/traffic_xxh$m; mkdir -p $P
  CMD="python $R/bench.py --no-cpu-baseline --no-secondary --maps-per-gpu 2 --task-threads 1 --steps 2 --warmup 1"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && S3S_XXH=$m timeout 300 rocprofv3 --pmc $c -d $P/pmc_$c -o p -- $CMD > $P/pmc_$c.log 2>&1)
  done
  python tools/summarize_prof.py $P > $P/summary.txt 2>&1
  echo "== traffic S3S_XXH=$m"; grep -E "lz4_compress|xxh32|gather_items|checksum_seg" $P/summary.txt | cut -c1-200
done
P=$R/$O/pmc_hbm; mkdir -p $P
CMD="python $R/bench.py --hbm-stages-only"
i=0
while read -r line; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $line -d $P/pmc_$i -o p -- $CMD > $P/pmc_$i.log 2>&1)
done <<'PMC'
FETCH_SIZE
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD
PMC
python tools/summarize_prof.py $P > $P/summary.txt 2>&1
echo "== PMC hbm stages"; grep -E "checksum_segments|xxh32" $P/summary.txt | cut -c1-700
