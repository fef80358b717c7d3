#!/bin/bash
# Env-count sweep of the env-step bench (random-action protocol, SMPL): throughput and stepper launch time vs envs per GPU.
O=gpurun_out/sweep
mkdir -p $O
for n in 512 1024 2048 4096 8192 16384 32768; do
  python bench.py --envs $n --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads --steps 200 --warmup 20 > $O/envs_$n.json 2>> $O/err.log
done
python - <<'PY'
import json, glob
rows = []
for n in (512, 1024, 2048, 4096, 8192, 16384, 32768):
    try:
        d = json.load(open(f'gpurun_out/sweep/envs_{n}.json'))
    except Exception as e:
        print(n, 'failed', e); continue
    rows.append(dict(envs=n, env_steps_per_s=d['value'], ms_per_step=d['ms_per_step'], stepper_us=d['roofline']['kernel_ms'] * 1e3,
                     stepper_kernel=d['roofline'].get('kernel'), frac=d['roofline']['frac']))
    print(rows[-1])
json.dump(rows, open('gpurun_out/sweep/r02_env_count_sweep.json', 'w'), indent=1)
PY
