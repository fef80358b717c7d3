#!/bin/bash
# A/B of the per-frame reset observation table (phc_im_params_t.obs_ref_table) on one box: parity of the reset paths, env-step timing with and without
OUT=gpurun_out/r04d; mkdir -p $OUT
[ -n "$SKIP_PYTEST" ] || python -m pytest tests/test_env_gpu.py tests/test_task_parity.py tests/test_config_sizes_gpu.py -m gpu -q -x -k "reset or table or rollout or step_matches or hist or config3" 2>&1 | tail -60 > $OUT/tests.txt
tail -3 $OUT/tests.txt
for tag in table notable; do
  if [ $tag = notable ]; then export PHC_NO_OBS_REF_TABLE=1; else unset PHC_NO_OBS_REF_TABLE; fi
  python bench.py --steps 300 --warmup 30 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python -c "
import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag', 'M env-steps/s', d['value']/1e6, 'us/step', d['ms_per_step']*1e3, 'stepper us', d['roofline']['kernel_ms']*1e3)"
  rm -rf /tmp/prof_$tag; (cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > /dev/null 2> $OUT/prof_$tag.err)
  python profiles/summarize_rocpd.py $(find /tmp/prof_$tag -name '*.db' | head -1) > $OUT/kernel_stats_$tag.txt; grep -n "k_im_reset\|k_im_post\|k_sim_step" $OUT/kernel_stats_$tag.txt | cut -c1-120
done
