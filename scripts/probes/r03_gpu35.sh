#!/bin/bash
# round 3: bench.py keeps stdout to the one JSON line (RCCL's banner goes to stderr): one-rank RCCL run + the multi-rank logic tests
O=gpurun_out/r03_35; mkdir -p $O
timeout 300 python bench.py --force-rccl --steps 50 --warmup 10 --ppo-epochs 2 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_force_rccl.json 2> $O/bench_force_rccl.err
echo "stdout lines: $(wc -l < $O/bench_force_rccl.json); banner on stderr: $(grep -c 'RCCL version' $O/bench_force_rccl.err)"
python -c "
import json; d=json.loads(open('$O/bench_force_rccl.json').read()); print('force-rccl', round(d['ppo_samples_per_s']), d['ppo_comm']['grad_allreduces_per_epoch'], d['ppo_comm']['replicas_identical'])"
timeout 400 python -m pytest tests/test_env_gpu.py -m gpu -x -q -k "bench_multi_rank" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python bench.py --steps 50 --warmup 10 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | wc -l
