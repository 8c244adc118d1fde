#!/bin/bash
# Run the oracle's golden-fixture tests against an AddressSanitizer + UBSan build of oracle/mincurv_oracle.c
# (CPU only; SURVEY.md section 5 "sanitizers").  GPU sanitizers are not available on this pool.
set -euo pipefail
cd "$(dirname "$0")/.."
make -C oracle -s asan
export ORACLE_SO="$PWD/oracle/libmincurv_oracle_asan.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export OMP_NUM_THREADS=4
python3 -m pytest tests/test_oracle_golden.py tests/test_host_cpu.py -x -q -m "not gpu" -p no:cacheprovider "$@"
