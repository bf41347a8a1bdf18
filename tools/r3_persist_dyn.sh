#!/bin/bash
# the persistent compress grid with a block counter (ships): the whole GPU suite, the driver's bench line, then the
# headline again, a launch alone on the chip, and fixed grid sizes (S3S_LZ4_GRID) — gpurun_out/<tag>/
tag=${1:-r03v}
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/$tag; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; python -c "import json; d=json.loads([l for l in open('$O/bench_full.json') if l.startswith('{')][-1]); print('driver-style', d['value'], d['ms_per_step'], {k: v.get('value') for k, v in d['secondary'].items() if isinstance(v, dict)})"
one() { timeout 60 python bench.py --no-cpu-baseline --no-secondary "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['avg_launch_ms'])" || echo "$1 FAILED"; }
one headline --steps 20 --warmup 5 --verify | tee -a $O/ab.txt
one alone_2tasks --maps-per-gpu 2 --task-threads 1 | tee -a $O/ab.txt
one two_threads --task-threads 2 --steps 20 --warmup 5 | tee -a $O/ab.txt
S3S_LZ4_GRID=2560 one headline_grid_2560 --steps 20 --warmup 5 | tee -a $O/ab.txt
one headline --steps 20 --warmup 5 | tee -a $O/ab.txt
