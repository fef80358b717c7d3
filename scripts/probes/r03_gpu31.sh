#!/bin/bash
# round 3: reset roles swapped (task observation, the longer chain, dispatched first): quick A/B
O=gpurun_out/r03_31; mkdir -p $O
timeout 300 python -m pytest tests/test_env_gpu.py -m gpu -x -q -k "oracle or table or rollout" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base new base new; do
  if [ $v = base ]; then export PHC_AMD_LIB=$PWD/phc_amd/_obj/libphc_amd_base.so; else unset PHC_AMD_LIB; fi
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o b -- python bench.py --steps 300 --warmup 30 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_$v.json 2> $O/bench_$v.err
  python profiles/summarize_rocpd.py $(find /tmp/prof_$v -name '*.db' | head -1) > $O/stats_$v.txt
  echo "$v: $(python -c "import json; d=json.load(open('$O/bench_$v.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step')")  $(grep -E 'k_im_reset<3, true' $O/stats_$v.txt | cut -c60-95)"
done
