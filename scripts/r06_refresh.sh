#!/bin/bash
# Round-6 measurement refresh on the GPU box: device suite, the default bench line (live reference / PMC / SQ counters inside), driver-style line, other robots,
# kernel-trace summaries (env step, PPO epoch, one optimizer step), PMC traffic + SQ counters, the trained-policy workload on its own.
# (round 6: the lagged stepper is the default; the `_fresh` lines are `--solver inertia_lag=0`)
#   bash scripts/r06_refresh.sh [out-dir-name]       SKIP_PYTEST=1 skips the suite, WITH_CONFIG3=1 adds the 8192-env / 11 313-clip line
O=gpurun_out/${1:-r6r}
mkdir -p $O
if [ -z "$SKIP_PYTEST" ]; then timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; fi
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2>> $O/bench_default.err
python bench.py --actions tracking --no-cpu-baseline --no-pmc --ppo-epochs 0 --no-other-workloads > $O/bench_tracking.json 2>> $O/bench_default.err
python bench.py --config 5 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_h1.json 2>> $O/bench_default.err
python bench.py --config 5 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads --solver inertia_lag=0 > $O/bench_h1_fresh.json 2>> $O/bench_default.err
python bench.py --robot g1 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_g1.json 2>> $O/bench_default.err
python bench.py --robot g1 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads --solver inertia_lag=0 > $O/bench_g1_fresh.json 2>> $O/bench_default.err
python bench.py --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads --solver inertia_lag=0 > $O/bench_fresh.json 2>> $O/bench_default.err
if [ -n "$WITH_CONFIG3" ]; then python bench.py --config 3 --ppo-epochs 0 --no-cpu-baseline --no-pmc > $O/bench_config3.json 2>> $O/bench_default.err; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/prof_bench.json 2> $O/prof.err
python profiles/summarize_rocpd.py $(find /tmp/prof -name '*.db' | head -1) > $O/env_step_kernel_stats.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o ppo -- python bench.py --steps 20 --warmup 5 --ppo-epochs 1 --no-cpu-baseline --no-pmc --no-other-workloads > $O/prof_ppo.json 2>> $O/prof.err
python profiles/summarize_rocpd.py $(find /tmp/prof2 -name '*.db' | head -1) > $O/ppo_epoch_kernel_stats.txt
python profiles/dump_step.py $(find /tmp/prof2 -name '*.db' | head -1) > $O/ppo_optimizer_step_kernels.txt 2>> $O/prof.err || true
bash profiles/collect_pmc.sh > $O/pmc_traffic.txt 2>> $O/prof.err
bash profiles/collect_pmc_valu.sh > $O/pmc_valu.txt 2>> $O/prof.err
python -m phc_amd.learning.bench_policy --train-s 90 --steps 300 > $O/bench_trained_policy.txt 2>> $O/prof.err
python - <<PY
import json
d=json.load(open('$O/bench_default.json'))
print('value', round(d['value']/1e6,2), 'M; ms', round(d['ms_per_step'],4), 'stepper us', round(d['roofline']['kernel_ms']*1e3,1), 'frac', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'])
print({k: (round(v,1) if isinstance(v,float) else v) for k, v in d.items() if k.startswith('ppo_') and not isinstance(v, dict)})
for n in ('driver_style','tracking','h1','h1_fresh','g1','g1_fresh','fresh'):
    t=json.load(open('$O/bench_%s.json'%n)); print(n, round(t['value']/1e6,2), round(t['roofline']['kernel_ms']*1e3,1))
PY
head -7 $O/env_step_kernel_stats.txt | cut -c1-150; cat $O/pmc_traffic.txt | cut -c1-200 | tail -8; tail -1 $O/bench_trained_policy.txt | cut -c1-700
