#!/bin/bash
# round 3: end-to-end training sanity after the AMP table / post-physics changes (step-in-place clip, 700 epochs) + bench with sampled stepper events
O=gpurun_out/r03_27; mkdir -p $O
for rep in 1 2; do
  python bench.py --steps 300 --warmup 30 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_$rep.json 2> $O/bench_$rep.err
  echo "bench $rep: $(python -c "import json; d=json.load(open('$O/bench_$rep.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step; stepper', round(d['roofline']['kernel_ms']*1e3,2))")"
done
timeout 600 python scripts/learning_curve.py 700 4096 $O/learning_curve_stepinplace_700.json env.motion_file=stepinplace:10 > $O/train.log 2>&1; grep -E "epoch  (  1|100|300|500|600|700)|acceptance" $O/train.log | cut -c1-260
