#!/bin/bash
# round 3: stepper prologue (state requested before the PD targets are stored): dynamics + env suites, same-box A/B against HEAD
O=gpurun_out/r03_22; mkdir -p $O
timeout 1200 python -m pytest tests/test_dynamics.py tests/test_env_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2 3; do
for v in base new; do
  if [ $v = base ]; then export PHC_AMD_LIB=$PWD/phc_amd/_obj/libphc_amd_base.so; else unset PHC_AMD_LIB; fi
  python bench.py --steps 300 --warmup 30 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  echo "$v $rep: $(python -c "import json; d=json.load(open('$O/bench_${v}_$rep.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step; stepper', round(d['roofline']['kernel_ms']*1e3,2), 'us')")"
done
done
