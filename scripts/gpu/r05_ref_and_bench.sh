#!/bin/bash
# Round 5: the reference on the GPU box (travel copy oracle/_ref): the direct 4096-env parity test, then the full default bench line (live cpu_reference / config0,
# the three roofline objects with live PMC passes, the inertia_lag child line)
O=gpurun_out/${1:-r05_ref}
mkdir -p $O
ls oracle/_ref | head -3 > $O/ref_ls.txt 2>&1
timeout 900 python -m pytest tests/test_reference_direct_gpu.py -m gpu -q -x -s > $O/pytest_ref.log 2>&1; tail -5 $O/pytest_ref.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<PY
import json
d = json.load(open('$O/bench_default.json'))
print('value', round(d['value'] / 1e6, 2), 'M; ms', round(d['ms_per_step'], 4))
for k in ('roofline', 'roofline_post_physics'):
    r = d.get(k) or {}
    print(k, {x: r.get(x) for x in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel_ms')}, r.get('sq_counters'), (r.get('hbm') or {}).get('frac'))
print('cpu_reference', {k: d.get('cpu_reference', {}).get(k) for k in ('value', 'cores', 'host', 'measured_in', 'source')})
print('config0', d.get('config0'))
print('cpu_baseline', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))
print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items() if k.startswith('ppo_') and not isinstance(v, dict)})
print({k: (v.get('value'), v.get('stepper_kernel_ms')) for k, v in d.get('other_workloads', {}).items()})
PY
