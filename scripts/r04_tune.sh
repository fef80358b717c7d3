#!/bin/bash
# GEMM solution selection for the PPO path on the GPU box (scripts/tune_gemms.py), then the A/B: bench with and without the tuned file
OUT=gpurun_out/tuned; mkdir -p $OUT
timeout 600 python scripts/tune_gemms.py --out $OUT/tuned_gemms_gfx950.csv --iters ${ITERS:-10} --ms ${MS:-10} > $OUT/tune.log 2>&1; tail -3 $OUT/tune.log
ls -la $OUT; wc -l $OUT/tuned_gemms_gfx950.csv
# (the A/B below needs IMAmpAgent to load the file: see profiles/r04_ppo/README.md -- the loader was measured and not kept)
for tag in tuned default tuned2; do
  if [ $tag = default ]; then export PHC_NO_TUNED_GEMMS=1; else unset PHC_NO_TUNED_GEMMS; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-other-workloads > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python -c "
import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag', {k: (round(v,2) if isinstance(v,float) else v) for k, v in d.items() if k.startswith('ppo_') and not isinstance(v, dict)})"
done
python -m pytest tests/test_learner_parity.py tests/test_learn_gpu.py -m gpu -q -x 2>&1 | tail -3
