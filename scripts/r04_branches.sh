#!/bin/bash
# A/B of the optimizer step's branch streams (critic / discriminator next to the actor) on one box: learner parity on hip, PPO epoch timing with and without
OUT=gpurun_out/r04e; mkdir -p $OUT
[ -n "$SKIP_PYTEST" ] || python -m pytest tests/test_learner_parity.py tests/test_learn_gpu.py -m gpu -q -x 2>&1 | tail -40 > $OUT/tests.txt
tail -3 $OUT/tests.txt
for tag in ${TAGS:-branches nobranches branches2}; do
    case $tag in q*) export DEBUG_HIP_FORCE_GRAPH_QUEUES=${tag#q};; *) unset DEBUG_HIP_FORCE_GRAPH_QUEUES;; esac
  unset DEBUG_CLR_GRAPH_PACKET_CAPTURE DEBUG_HIP_GRAPH_BATCH_SIZE PHC_NO_BRANCH_STREAMS
  case $tag in *nobr*) export PHC_NO_BRANCH_STREAMS=1;; esac
  case $tag in *pc1*) export DEBUG_CLR_GRAPH_PACKET_CAPTURE=1;; *pc0*) export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0;; esac
  case $tag in *bs*) export DEBUG_HIP_GRAPH_BATCH_SIZE=${tag##*bs};; esac
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-other-workloads ${BENCH_ARGS} > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python -c "
import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag', {k: (round(v,2) if isinstance(v,float) else v) for k, v in d.items() if k.startswith('ppo_') and not isinstance(v, dict)})"
done
