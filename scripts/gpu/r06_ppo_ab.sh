#!/bin/bash
# Round 6: the PPO update on one box, REPS alternating runs of two variants (the second one with the environment switches of $VARIANT_ENV, e.g. VARIANT_ENV="PHC_NO_BRANCH_STREAMS=1"),
# then a rocprofv3 kernel-trace of one more run (per-kernel launch times, e.g. k_ppo_loss after its half-wavefront rewrite).
# (First use: the observation normaliser's moments folded on the discriminator's stream -- 54.0 vs 53.1 ms without, 3 x alternating: slower, not kept, profiles/r06_ppo/.)
#   bash scripts/gpu/r06_ppo_ab.sh [OUT] [REPS]
O=gpurun_out/${1:-r06_ppo_ab}; REPS=${2:-3}
mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-other-workloads"
for rep in $(seq 1 $REPS); do
  $B > $O/ppo_new_$rep.json 2>> $O/err.log
  [ -n "$VARIANT_ENV" ] && env $VARIANT_ENV $B > $O/ppo_variant_$rep.json 2>> $O/err.log
done
python - <<PY
import glob, json
for f in sorted(glob.glob('$O/ppo_*.json')):
    d = json.load(open(f))
    print(f"{f.split('/')[-1][:-5]:16s} update {d['ppo_update_ms']:.2f} ms  play {d['ppo_play_ms']:.2f} ms  samples/s {d['ppo_samples_per_s'] / 1e6:.3f} M  roofline {d['ppo_roofline']['frac']:.4f}  env {d['value'] / 1e6:.2f} M ({d['ms_per_step'] * 1e3:.1f} us, stepper {d['roofline']['kernel_ms'] * 1e3:.1f})")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo_ab -o ppo -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-other-workloads > $OLDPWD/$O/prof_run.json 2>> $OLDPWD/$O/err.log
cd $OLDPWD     # (the rocprofv3 database stays in /tmp: gpurun_out/ is capped at 64 MiB)
python profiles/summarize_rocpd.py $(find /tmp/prof_ppo_ab -name '*.db' | head -1) > $O/ppo_epoch_kernel_stats.txt 2>> $O/err.log
python profiles/dump_step.py $(find /tmp/prof_ppo_ab -name '*.db' | head -1) > $O/ppo_optimizer_step_kernels.txt 2>> $O/err.log || true
grep -E "k_ppo_loss|k_running_norm|k_adam|k_sumsq" $O/ppo_epoch_kernel_stats.txt | cut -c1-140
