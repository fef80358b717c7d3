#!/bin/bash
# Round 5: same-box A/B of the stepper builds (default library: LAG as a template parameter, 64-bit contact-point masks; _m32: 32-bit masks) + the per-wavefront spread probe
O=gpurun_out/${1:-r05_step_ab2}
mkdir -p $O
B="python bench.py --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads"
for rep in 1 2; do
  $B > $O/smpl_fresh_$rep.json 2>> $O/err.log
  PHC_AMD_LIB=phc_amd/_obj/libphc_amd_m32.so $B > $O/smpl_fresh_m32_$rep.json 2>> $O/err.log
  $B --solver inertia_lag=1 > $O/smpl_lag_$rep.json 2>> $O/err.log
  PHC_AMD_LIB=phc_amd/_obj/libphc_amd_m32.so $B --solver inertia_lag=1 > $O/smpl_lag_m32_$rep.json 2>> $O/err.log
done
$B --config 5 --solver inertia_lag=1 > $O/h1_lag.json 2>> $O/err.log
$B --robot g1 --solver inertia_lag=1 > $O/g1_lag.json 2>> $O/err.log
python scripts/probes/sim_wave_spread.py 4096 > $O/wave_spread_4096.txt 2>> $O/err.log
python scripts/probes/sim_wave_spread.py 4096 +solver.inertia_lag=1 > $O/wave_spread_4096_lag.txt 2>> $O/err.log
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], 'M env-steps/s', round(d['value'] / 1e6, 2), 'ms/step', round(d['ms_per_step'], 4), 'stepper us', round(d['roofline']['kernel_ms'] * 1e3, 1))
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $O/wave_spread_4096.txt
tail -3 $O/err.log
