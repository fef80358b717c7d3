#!/bin/bash
# round 3: stepper block algebra in packed fp32 pairs (Inertia6P): dynamics suite on the device, same-box A/B against the scalar-symmetric version, timeline
O=gpurun_out/r03_17; mkdir -p $O
timeout 900 python -m pytest tests/test_dynamics.py tests/test_env_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export PHC_AMD_LIB=$PWD/phc_amd/_obj/libphc_amd_base.so; else unset PHC_AMD_LIB; fi
  python bench.py --steps 300 --warmup 30 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  echo "$v $rep: $(python -c "import json; d=json.load(open('$O/bench_${v}_$rep.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step; stepper', round(d['roofline']['kernel_ms']*1e3,2), 'us')")"
done
done
unset PHC_AMD_LIB
python scripts/probes/sim_timeline.py 2048 > $O/timeline_2048.txt 2>&1; tail -30 $O/timeline_2048.txt
python scripts/probes/sim_ablation.py 4096 > $O/ablation_4096.txt 2>&1; tail -14 $O/ablation_4096.txt
