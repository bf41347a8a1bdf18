#!/bin/bash
# scratch: zstd
mkdir -p gpurun_out/$1
python -m pytest tests/test_gpu_zstd.py tests/test_gpu_hardening.py -x -q -m gpu > gpurun_out/$1/pytest.txt 2>&1
tail -2 gpurun_out/$1/pytest.txt
for i in 1 2; do
python bench.py --workload terasort-10g-200p-zstd --direction decompress --maps-per-gpu 8 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/$1/zstd_$i.json 2> gpurun_out/$1/zstd_$i.err
python - <<P
import json
d=json.loads(open("gpurun_out/$1/zstd_$i.json").read().strip().splitlines()[-1])
print("zstd", d["value"], d["ms_per_step"])
P
done
