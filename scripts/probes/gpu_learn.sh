#!/bin/bash
mkdir -p gpurun_out/learn
O=gpurun_out/learn
timeout 900 python -m pytest tests/test_learn_gpu.py tests/test_env_gpu.py -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python scripts/profile_step.py 2>&1 | grep -v Warn | tail -14 | tee $O/profile_step.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print({k: v for k, v in d.items() if 'ppo' in k.lower()}); print(d['value'])"
