#!/bin/bash
# Round 6: same-box A/B of the PPO update -- new: observation-normaliser moments on the discriminator's stream (IMAmpAgent._split_obs_norm), half-wavefront k_ppo_loss;
# old: PHC_NO_SPLIT_OBS_NORM=1 (the kernel change has no switch: its launch time is read off the rocprofv3 kernel stats of the last run).
#   bash scripts/gpu/r06_ppo_ab.sh [OUT] [REPS]
O=gpurun_out/${1:-r06_ppo_ab}; REPS=${2:-3}
mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-other-workloads"
for rep in $(seq 1 $REPS); do
  $B > $O/ppo_new_$rep.json 2>> $O/err.log
  PHC_NO_SPLIT_OBS_NORM=1 $B > $O/ppo_nosplit_$rep.json 2>> $O/err.log
done
python - <<PY
import glob, json
for f in sorted(glob.glob('$O/ppo_*.json')):
    d = json.load(open(f))
    print(f"{f.split('/')[-1][:-5]:16s} update {d['ppo_update_ms']:.2f} ms  play {d['ppo_play_ms']:.2f} ms  samples/s {d['ppo_samples_per_s'] / 1e6:.3f} M  roofline {d['ppo_roofline']['frac']:.4f}  env {d['value'] / 1e6:.2f} M ({d['ms_per_step'] * 1e3:.1f} us, stepper {d['roofline']['kernel_ms'] * 1e3:.1f})")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o ppo -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-other-workloads > $OLDPWD/$O/prof_run.json 2>> $OLDPWD/$O/err.log
cd $OLDPWD
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 $f > $O/ppo_kernel_stats_head.csv
