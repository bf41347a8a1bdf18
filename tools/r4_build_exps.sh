#!/bin/bash
# (CPU, before the first GPU call of round 4) builds the experiment libraries that wait for a measurement
# (DESIGN.md §7.0): each is the shipped source with one macro, bit-exact in the gfx950 interpreter
# (tests/test_isa_kernels.py: rolling_prefetch, persistent_grid_decoder; tests/test_zstd_model.py for the zstd macro).
#   rollpf    -DS3S_X_ROLL_PREFETCH     LZ4 compressor: rolling prefetch through spare lanes of the stream load
#   rollpf6k  ... -DS3S_X_ROLL_DIST=6144  the same, 6 KiB ahead;  rollpfprio = rolling prefetch + issue priority
#   storent   -DS3S_X_STORE_NT          LZ4 compressor: non-temporal sequence stores (was +-0 before the block prefetch)
#   setprio   -DS3S_X_SETPRIO          LZ4 compressor: raised issue priority from a window's entry to its candidate gather
#   decpers   -DS3S_DEC_PERSIST         batch decoder as a persistent grid (S3S_DEC_GRID wavefronts, default 26 per CU)
#   finwide   -DS3S_X_FINISH_WIDE -DS3S_X_SPEC_UNROLL   LZ4 frame discovery: rebase with four records per lane, speculation with 256 positions per step
#   zsfast    -DZS_SEQ_FASTBITS         zstd decoder: one 64-bit bit window per sequence
set -e
cd "$(dirname "$0")/../spark-s3-shuffle_amd/csrc"
make exp EXPNAME=rollpf EXPFLAGS=-DS3S_X_ROLL_PREFETCH &
make exp EXPNAME=storent EXPFLAGS=-DS3S_X_STORE_NT &
wait
make exp EXPNAME=decpers EXPFLAGS=-DS3S_DEC_PERSIST &
make exp EXPNAME=setprio EXPFLAGS=-DS3S_X_SETPRIO &
wait
make exp EXPNAME=rollpf6k EXPFLAGS='-DS3S_X_ROLL_PREFETCH -DS3S_X_ROLL_DIST=6144' &
make exp EXPNAME=rollpfprio EXPFLAGS='-DS3S_X_ROLL_PREFETCH -DS3S_X_SETPRIO' &
make exp EXPNAME=zsfast EXPFLAGS=-DZS_SEQ_FASTBITS &
wait
make exp EXPNAME=finwide EXPFLAGS='-DS3S_X_FINISH_WIDE -DS3S_X_SPEC_UNROLL' &
wait
ls -la ../lib/
