#!/bin/bash
# (GPU) round 6: LZ4 parse block — windows walked by pointer doubling from N tokens on (12 shipped, 16, never)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06n}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_hardening.py tests/test_gpu_batch.py tests/test_gpu_snappy.py tests/test_gpu_lzf.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
run() { python bench.py --direction decompress --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -n 1 | python -c "
import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('LIBTAG'), '$*:', d['value'], 'GB/s ms/step', d['ms_per_step'], 'codec ms', d['stages_ms_per_library_call']['codec'], 'verified', d.get('bytes_verified'))" | tee -a $O/bench.txt; }
for rep in 1 2; do
for lib in default dbl16 dbl99; do
  export LIBTAG=$lib
  if [ $lib = default ]; then unset S3S_CODEC_LIB; else export S3S_CODEC_LIB=$GRAFT_REPO_ROOT/spark-s3-shuffle_amd/lib/libs3shuffle_codec_exp_$lib.so; fi
  run --maps-per-gpu 8
  run --workload tpcds-wide-100g-200p-lz4 --maps-per-gpu 8
done; done
unset S3S_CODEC_LIB; export LIBTAG=default
run --workload terasort-10g-200p-lzf --maps-per-gpu 8
run --workload tpcds-wide-100g-200p-snappy --maps-per-gpu 8
