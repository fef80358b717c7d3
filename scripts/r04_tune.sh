#!/bin/bash
# GEMM solution selection for the PPO path on the GPU box (scripts/tune_gemms.py), then the A/B: bench with and without the tuned file
OUT=gpurun_out/tuned; mkdir -p $OUT
timeout 600 python scripts/tune_gemms.py --out $OUT/tuned_gemms_gfx950.csv --iters ${ITERS:-10} --ms ${MS:-10} > $OUT/tune.log 2>&1; tail -3 $OUT/tune.log
ls -la $OUT; wc -l $OUT/tuned_gemms_gfx950.csv
# The A/B of round 4 loaded the file in IMAmpAgent (torch.cuda.tunable.enable(True); tuning_enable(False); read_file(...)): 52.4 ms vs 52.1 ms with the
# library defaults -- the loader was not kept (profiles/r04_ppo/README.md).
