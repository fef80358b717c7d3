#!/bin/bash
# post-physics split into two lane-group roles: task parity on hip + env-step timing + kernel stats
OUT=gpurun_out/r04f; mkdir -p $OUT
python -m pytest tests/test_task_parity.py tests/test_env_gpu.py tests/test_h1.py tests/test_config_sizes_gpu.py -m gpu -q -x -k "not graph and not two_rank and not learn" 2>&1 | tail -4 > $OUT/tests.txt; tail -3 $OUT/tests.txt
python bench.py --steps 300 --warmup 30 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $OUT/bench_env.json 2> $OUT/bench_env.err
python -c "
import json; d=json.loads(open('$OUT/bench_env.json').read().strip().splitlines()[-1]); print('M env-steps/s', d['value']/1e6, 'us/step', d['ms_per_step']*1e3, 'stepper us', d['roofline']['kernel_ms']*1e3)"
rm -rf /tmp/prof_p; (cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o p -- python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > /dev/null 2> $OUT/prof.err)
python profiles/summarize_rocpd.py $(find /tmp/prof_p -name '*.db' | head -1) > $OUT/kernel_stats.txt; grep -n "k_im_reset\|k_im_post\|k_sim_step" $OUT/kernel_stats.txt | cut -c1-150
