#!/bin/bash
# same-box A/B: HEAD (LDS hand-over) vs working tree (shuffle hand-over), stepper launch time by HIP events
O=gpurun_out/r03_5; mkdir -p $O
for i in 1 2; do
for v in old new; do
  if [ $v = old ]; then d=_ab_old; else d=.; fi
  ( cd $d && python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v: env-steps/s %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))" )
done; done 2>&1 | tee $O/ab.txt
