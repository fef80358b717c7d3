#!/bin/bash
# Round 6: which network's step count decides the 24-vs-36-optimizer-steps-per-rollout cliff on the 64-clip library?
# 3072 envs (the reference's own count, env_im.yaml:6), everything else as shipped; one stage, SECS of training each, the sweep every 500 epochs.
#   [ENVS=4096] bash scripts/gpu/r06_cliff_ab.sh OUT SECS "<run tags>" ["<seeds>"]      tags: shipped actor24 critic24 disc24 all24 tgs actor24disc24 ...
O=gpurun_out/$1; SECS=${2:-150}; TAGS=${3:-"shipped actor24 critic24 disc24 all24"}; SEEDS=${4:-"0"}
mkdir -p $O
C=learning.params.config
for tag in $TAGS; do
  case $tag in
    shipped)  X="" ;;
    actor24)  X="+$C.debug_actor_steps=24" ;;
    critic24) X="+$C.debug_critic_steps=24" ;;
    disc24)   X="+$C.debug_disc_steps=24" ;;
    actor24disc24) X="+$C.debug_actor_steps=24 +$C.debug_disc_steps=24" ;;
    actor24critic24) X="+$C.debug_actor_steps=24 +$C.debug_critic_steps=24" ;;
    critic24disc24) X="+$C.debug_critic_steps=24 +$C.debug_disc_steps=24" ;;
    all24eager) X="+$C.debug_actor_steps=24 +$C.debug_critic_steps=24 +$C.debug_disc_steps=24" ;;
    all24)    X="$C.mini_epochs=4" ;;
    tgs)      X="+solver.contact=tgs" ;;
    eager)    X="+$C.hip_graph=False" ;;
    oldreset) X="+$C.debug_reset_at_rollout_start=True" ;;
    oldreset24) X="+$C.debug_reset_at_rollout_start=True $C.mini_epochs=4" ;;
    nopower)  X="env.power_reward=False" ;;
    halfpower) X="+env.power_coefficient=0.00025" ;;
    *)        X="$tag" ;;
  esac
  for seed in $SEEDS; do
  t=$tag; [ "$SEEDS" != "0" ] && t=${tag}_s$seed
  PHC_QUIET=1 timeout $((SECS + 240)) python scripts/multi_clip_acceptance.py --envs ${ENVS:-3072} --stage1-s $SECS --stage2-s 0 --eval-every 500 --seed $seed --out $O/$t.json $X > $O/$t.log 2>&1
  echo "== $t: $(grep -c sweep $O/$t.log) sweeps; last: $(grep sweep $O/$t.log | tail -1)"
  grep "primitive 0 after stage 1" $O/$t.log | tail -1
  done
done
