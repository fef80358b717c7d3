#!/bin/bash
# final check of the driver's own invocation after the stdout change: default bench.py (children: PMC passes, tracking / H1 lines, cpu baseline)
O=gpurun_out/r03_37; mkdir -p $O
timeout 420 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
echo "stdout lines: $(wc -l < $O/bench_driver_style.json)"
python -c "
import json; d=json.loads(open('$O/bench_driver_style.json').read()); r=d['roofline']
print(round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step', 'traffic', r['traffic'], 'src', (list(r['traffic_source'].keys()) if isinstance(r['traffic_source'], dict) else r['traffic_source']), 'kernel_ms', round(r['kernel_ms'],4))
print('other', {k: round(v['value']/1e6,2) for k, v in d.get('other_workloads', {}).items() if isinstance(v, dict) and 'value' in v}, 'cpu', round(d['cpu_baseline']['value']), 'ppo', round(d['ppo_samples_per_s']))"
