#!/bin/bash
# Round 5: stepper switches on the device -- parity tests of the new options, then same-box A/B of the env step with and without `inertia_lag`.
O=gpurun_out/${1:-r05_step_ab}
mkdir -p $O
timeout 900 python -m pytest tests/test_stepper_options.py tests/test_dynamics.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads"
for rep in 1 2; do
  $B > $O/smpl_fresh_$rep.json 2>> $O/err.log
  $B --solver inertia_lag=1 > $O/smpl_lag_$rep.json 2>> $O/err.log
done
$B --actions tracking > $O/smpl_track_fresh.json 2>> $O/err.log
$B --actions tracking --solver inertia_lag=1 > $O/smpl_track_lag.json 2>> $O/err.log
$B --config 5 > $O/h1_fresh.json 2>> $O/err.log
$B --config 5 --solver inertia_lag=1 > $O/h1_lag.json 2>> $O/err.log
$B --robot g1 > $O/g1_fresh.json 2>> $O/err.log
$B --robot g1 --solver inertia_lag=1 > $O/g1_lag.json 2>> $O/err.log
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], 'M env-steps/s', round(d['value'] / 1e6, 2), 'ms/step', round(d['ms_per_step'], 4), 'stepper us', round(d['roofline']['kernel_ms'] * 1e3, 1))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -5 $O/err.log
