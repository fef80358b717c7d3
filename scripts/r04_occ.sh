#!/bin/bash
# Round 4: register allocation of the stepper vs env count (2 vs 3 wavefronts per SIMD): env step at 4096 and 8192 envs with --lane-mapping 1 / 3
set -x
OUT=gpurun_out/r04_occ
mkdir -p $OUT
for n in 4096 8192 12288; do for lm in 1 3; do
  python bench.py --envs $n --lane-mapping $lm --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $OUT/envs${n}_lm${lm}.json 2> $OUT/envs${n}_lm${lm}.err
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_occ/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"]/1e6,2), "M env-steps/s", round(d["ms_per_step"]*1e3,1), "us/step  stepper", round(d["roofline"]["kernel_ms"]*1e3,1), "us")
    except Exception as e: print(f, "ERR", e)
P
