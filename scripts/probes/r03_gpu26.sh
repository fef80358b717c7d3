#!/bin/bash
# round 3: reset in four roles + early clip-table request, post-physics without spills (launch bounds 256 x 2): env suite, same-box A/B against the AMP-table commit
O=gpurun_out/r03_26; mkdir -p $O
timeout 1200 python -m pytest tests/test_env_gpu.py tests/test_h1.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export PHC_AMD_LIB=$PWD/phc_amd/_obj/libphc_amd_base.so; else unset PHC_AMD_LIB; fi
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o b -- python bench.py --steps 300 --warmup 30 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  python profiles/summarize_rocpd.py $(find /tmp/prof_$v -name '*.db' | head -1) > $O/stats_${v}_$rep.txt
  echo "$v $rep: $(python -c "import json; d=json.load(open('$O/bench_${v}_$rep.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step')")"; grep -E "k_im_reset<3, true|k_im_post|k_sim_step<true" $O/stats_${v}_$rep.txt | cut -c1-140
done
done
