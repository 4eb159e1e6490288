#!/bin/bash
# AddressSanitizer + UBSan run of the CPU oracle against the golden vectors (GPU sanitizers
# are not available on the pool; the oracle is the only native code that runs on the CPU).
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
export OC_ORACLE_LIB="$PWD/oracle/_build/liboc_oracle_asan.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
python -m pytest tests/test_oracle_golden.py tests/test_oracle_invariants.py -x -q "$@"
