#!/bin/bash
mkdir -p gpurun_out/check
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline > /dev/null 2>&1
python profiles/dump_sequence.py $(find /tmp/prof -name '*.db' | head -1) > gpurun_out/check/sequence.txt 2>&1; head -70 gpurun_out/check/sequence.txt
