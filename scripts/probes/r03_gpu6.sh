#!/bin/bash
# round 3, GPU call: task-kernel math (device elementary functions, exp-map round trip in closed form): device parity + bench
O=gpurun_out/r03_6; mkdir -p $O
timeout 1200 python -m pytest tests/test_task_parity.py tests/test_env_gpu.py tests/test_h1.py -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('env-steps/s %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
python profiles/summarize_rocpd.py $(find /tmp/prof -name '*.db' | head -1) > $O/kernel_stats.txt; head -7 $O/kernel_stats.txt
