#!/bin/bash
mkdir -p gpurun_out/learn
O=$GRAFT_REPO_ROOT/gpurun_out/learn
timeout 600 python -m pytest tests/test_learn_gpu.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/prof_step -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --ppo-epochs 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python profiles/dump_step.py $(ls /tmp/prof_step/*/*.db /tmp/prof_step/*.db 2>/dev/null | head -1) 60 > $O/step_kernels.txt 2>&1; head -3 $O/step_kernels.txt; grep -A40 "# aggregated" $O/step_kernels.txt
