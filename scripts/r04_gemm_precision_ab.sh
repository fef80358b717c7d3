#!/bin/bash
# Round 4: policy-level A/B of the GEMM precision -- bf16 MFMA GEMMs (the product) vs fp32 GEMMs (what the reference trains in,
# phc/data/cfg/learning/im.yaml:51 mixed_precision False), same seeds, step-in-place clip (learned within ~600 epochs, r03_learning_curve_stepinplace_clip.json)
set -x
OUT=gpurun_out/r04_gemm; mkdir -p $OUT
E=${1:-900}
python scripts/learning_curve.py $E 4096 $OUT/stepinplace_bf16.json env.motion_file=stepinplace:10 > $OUT/stepinplace_bf16.log 2>&1
python scripts/learning_curve.py $E 4096 $OUT/stepinplace_fp32.json env.motion_file=stepinplace:10 --fp32-gemm > $OUT/stepinplace_fp32.log 2>&1
grep -E "epoch +(100|200|300|400|500|600|700|800|900) |acceptance" $OUT/*.log | cut -c1-260
