#!/bin/bash
# quick GPU check: gpu tests + short env bench with a kernel table
mkdir -p gpurun_out/check
O=gpurun_out/check
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('env-steps/s %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/prof -name '*.db' | head -1) > $O/kernel_stats.txt; head -8 $O/kernel_stats.txt
