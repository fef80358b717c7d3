#!/bin/bash
# The CPU build of the kernels' per-lane code (oracle/hostemu: phc_amd/csrc/*.h compiled by g++) under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the CPU
# tests that call it: stepper (all option combinations, SMPL / H1 / G1), post-physics, reset, motion lookups, AMP observations.  GPU sanitizers are not available on the pool.
#   bash scripts/sanitize_hostemu.sh [pytest args]      -> profiles/r06_sanitize_hostemu.txt
cd "$(dirname "$0")/.."
export PHC_HOSTEMU_SANITIZE=1
export LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export OMP_NUM_THREADS=8
FILES=$(grep -l "hostemu" tests/test_*.py | grep -v _gpu | tr '\n' ' ')
echo "# files: $FILES"
timeout 3000 python -m pytest $FILES -x -q -m "not gpu" -p no:cacheprovider "$@" 2>&1 | tail -15
