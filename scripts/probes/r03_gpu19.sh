#!/bin/bash
# round 3: reference lookups issued next to the stepper (phc_im_ref_lookup on a second stream) -- equivalence tests, env suite, A/B by env switch
O=gpurun_out/r03_19; mkdir -p $O
timeout 1200 python -m pytest tests/test_env_gpu.py tests/test_h1.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in off on; do
  if [ $v = off ]; then export PHC_NO_LOOKUP_AHEAD=1; else unset PHC_NO_LOOKUP_AHEAD; fi
  python bench.py --steps 300 --warmup 30 --ppo-epochs 3 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  echo "$v $rep: $(python -c "import json; d=json.load(open('$O/bench_${v}_$rep.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step; stepper', round(d['roofline']['kernel_ms']*1e3,2), 'us; ppo play', round(d['ppo_play_ms'],2), 'ms, samples/s', round(d['ppo_samples_per_s']))" 2>&1 | tail -1)"
done
done
unset PHC_NO_LOOKUP_AHEAD
rocprofv3 --kernel-trace --stats -d /tmp/prof_on -o b -- python bench.py --steps 300 --warmup 30 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/prof_bench.json 2> $O/prof.err
python profiles/summarize_rocpd.py $(find /tmp/prof_on -name '*.db' | head -1) > $O/stats_on.txt; head -8 $O/stats_on.txt | cut -c1-130
