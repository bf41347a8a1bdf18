#!/bin/bash
# Round 5, call 4: the new checksum / xxHash32 kernels — correctness on the GPU, HBM-bound stage lines, headline with the
# three xxHash32 pre-pass modes, the 1 GiB block both ways.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
tag=${1:-r05d}
O=gpurun_out/$tag; mkdir -p $O
echo "== gpu tests (checksums, compress, full size)"
timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_decompress.py tests/test_gpu_fullsize.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.txt
echo "== hbm-bound stages"
timeout 300 python bench.py --hbm-stages-only 2> $O/hbm_err.txt > $O/hbm_stages.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/'+__import__('sys').argv[1]+'/hbm_stages.json'))['hbm_bound_stages'] if False else json.load(open('gpurun_out/r05d/hbm_stages.json'))['hbm_bound_stages']
for k,v in d.items():
    if isinstance(v,dict):
        r=v.get('roofline',{})
        print(f"{k:45s} value {v.get('value')}  achieved {r.get('achieved')} frac {r.get('frac')} ms {r.get('avg_kernels_ms', v.get('ms'))} {v.get('matches_zlib','')}")
PY
for m in 1 0 2 1; do
  echo "== headline, S3S_XXH=$m"
  S3S_XXH=$m timeout 300 python bench.py --no-cpu-baseline --no-secondary --verify 2> $O/err_xxh$m.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], 'GB/s', d['stages_ms_per_library_call'])"
done
echo "== 1 GiB block"
for dir in compress decompress; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --workload skew-1part-lz4 --direction $dir --map-mib 1024 --maps-per-gpu 1 --steps 8 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dir', d['value'], 'GB/s', d['stages_ms_per_library_call'])"
done
echo "== issue probe"
(cd tools/probe && timeout 120 ./issue_probe 2>&1 | tail -30) | tee $O/issue_probe.txt
