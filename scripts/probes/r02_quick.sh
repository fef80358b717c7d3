#!/bin/bash
# quick stepper check: dynamics + env tests, phase profile, bench (env step only)
O=gpurun_out/${1:-r2q}
mkdir -p $O
python -m pytest tests/test_dynamics.py tests/test_env_gpu.py -m gpu -q -x -k "not rccl and not two_rank and not graph and not ppo and not mcp and not getup and not run_entry and not eval" > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-300
python scripts/sim_phase_profile.py run 4096 > $O/phase_profile.txt 2>&1; tail -11 $O/phase_profile.txt
python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('env-steps/s', round(d['value']/1e6,2), 'M; ms/step', round(d['ms_per_step'],4), '; stepper us', round(d['roofline']['kernel_ms']*1e3,1))"
