#!/bin/bash
O=gpurun_out/${1:-r2g}
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log | cut -c1-300
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time; python - <<PY
import json
d=json.load(open('$O/bench_default.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', {k: d['roofline'][k] for k in ('achieved','frac','traffic','kernel_ms')}, 'traffic_src', str(d['roofline'].get('traffic_source'))[:300])
print({k: v for k, v in d.items() if k.startswith('ppo_') and not isinstance(v, dict)})
print('cpu_baseline', d.get('cpu_baseline', {}).get('value'), 'cpu_reference', d.get('cpu_reference', {}).get('value'))
PY
( time python bench.py --config 3 --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline --no-pmc > $O/bench_cfg3.json 2> $O/bench_cfg3.err ) 2> $O/bench_cfg3.time; tail -3 $O/bench_cfg3.time; tail -2 $O/bench_cfg3.err; python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print(d['value'], d['ms_per_step'], d.get('config3_motion_library'))"
