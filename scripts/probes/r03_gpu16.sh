#!/bin/bash
# round 3: finishing sums of colsum / ppo_loss / weighted_sumsq inside their main launches (ticket counters): learner device suite + PPO bench
O=gpurun_out/r03_16; mkdir -p $O
timeout 1200 python -m pytest tests/test_learn_gpu.py tests/test_learner_parity.py tests/test_env_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --steps 100 --warmup 10 --ppo-epochs 6 --no-cpu-baseline --no-pmc --no-other-workloads 2>$O/bench.err > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print('env %.2f M  update %.1f ms  play %.1f ms  samples/s %.0f' % (d['value']/1e6, d['ppo_update_ms'], d['ppo_play_ms'], d['ppo_samples_per_s']))" || tail -5 $O/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o ppo -- python bench.py --steps 20 --warmup 5 --ppo-epochs 1 --no-cpu-baseline --no-pmc --no-other-workloads > $O/prof_ppo.json 2>> $O/prof.err
python profiles/summarize_rocpd.py $(find /tmp/prof2 -name '*.db' | head -1) > $O/ppo_epoch_kernel_stats.txt; head -30 $O/ppo_epoch_kernel_stats.txt | cut -c1-120
