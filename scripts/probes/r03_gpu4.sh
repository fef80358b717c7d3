#!/bin/bash
# round 3, GPU call 4: shuffle sweeps -- device parity (dynamics tests), bench
O=gpurun_out/r03_4; mkdir -p $O
timeout 900 python -m pytest tests/test_dynamics.py tests/test_h1.py -m gpu -x -q > $O/pytest_dyn.log 2>&1; tail -5 $O/pytest_dyn.log
python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench.json 2> $O/bench.err || tail -3 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('env-steps/s %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
