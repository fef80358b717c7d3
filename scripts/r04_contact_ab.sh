#!/bin/bash
# Round 4: A/B of the ground-contact model at policy level -- the squat clip that PPO does not learn with the penalty contact,
# and the walk clip as the control, with `+solver.contact=penalty` and `+solver.contact=tgs`, same seeds.  One gpurun call.
#   bash scripts/r04_contact_ab.sh [epochs_squat] [epochs_walk] [models...]
set -x
OUT=gpurun_out/r04_contact
mkdir -p $OUT
ES=${1:-2500}; EW=${2:-2500}; shift; shift
MODELS=${@:-"penalty tgs"}
python -m pytest tests -m gpu -q -k "rigid or tgs" 2>&1 | tail -8 > $OUT/gpu_tests.txt
for cm in $MODELS; do
  python scripts/probes/track_probe.py squat:10 300 +solver.contact=$cm > $OUT/squat_pd_tracking_${cm}.txt 2>&1
  python scripts/learning_curve.py $ES 4096 $OUT/squat_${cm}.json env.motion_file=squat:10 +solver.contact=$cm > $OUT/squat_${cm}.log 2>&1
  if [ "$EW" != "0" ]; then python scripts/learning_curve.py $EW 4096 $OUT/walk_${cm}.json env.motion_file=walk:10 +solver.contact=$cm > $OUT/walk_${cm}.log 2>&1; fi
done
tail -2 $OUT/*.log
