#!/bin/bash
mkdir -p gpurun_out/$1
for w in tpcds-wide-100g-200p-zstd terasort-10g-200p-zstd; do
for m in 8 10; do
python bench.py --workload $w --direction decompress --maps-per-gpu $m --steps 6 --warmup 2 --cpu-seconds 6 --no-secondary 2>gpurun_out/$1/err_$w.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w maps $m', d['value'], d['ms_per_step'], 'ratio', d['config'].get('compression_ratio'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"
done
done
