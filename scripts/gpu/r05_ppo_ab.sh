#!/bin/bash
# Round 5: device tests of the new learner / stepper pieces, then same-box A/B (i) of the PPO update -- new: deferred column-sum finishes, direct bias
# gradients in the twice-differentiable layers, 16-byte value-head kernels; old: the env toggles + the _nov8 library build -- and (ii) of the ground-contact
# helper lanes on H1 / G1 (default library vs the _nohelp build)
O=gpurun_out/${1:-r05_ppo_ab}
mkdir -p $O
timeout 900 python -m pytest tests/test_stepper_options.py tests/test_learn_gpu.py tests/test_dynamics.py tests/test_h1.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-other-workloads"
for rep in 1 2 3; do
  $B > $O/ppo_new_$rep.json 2>> $O/err.log
  PHC_NO_DEFER_COLSUM=1 PHC_NO_DIRECT_DD_BIAS=1 PHC_AMD_LIB=$PWD/phc_amd/_obj/libphc_amd_nov8.so $B > $O/ppo_old_$rep.json 2>> $O/err.log
done
PHC_NO_DEFER_COLSUM=1 $B > $O/ppo_nodefer_1.json 2>> $O/err.log
PHC_AMD_LIB=$PWD/phc_amd/_obj/libphc_amd_nov8.so $B > $O/ppo_nov8_1.json 2>> $O/err.log
python - <<PY
import glob, json
for f in sorted(glob.glob('$O/ppo_*.json')):
    d = json.load(open(f))
    print(f"{f.split('/')[-1][:-5]:16s} update {d['ppo_update_ms']:.2f} ms  play {d['ppo_play_ms']:.2f} ms  samples/s {d['ppo_samples_per_s'] / 1e6:.3f} M  roofline {d['ppo_roofline']['frac']:.4f}")
PY
bash scripts/gpu/step_ab.sh ${1:-r05_ppo_ab}/helpers -l "default nohelp" -s ";inertia_lag=1" -r "h1 g1" -n 2
