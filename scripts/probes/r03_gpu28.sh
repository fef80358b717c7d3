#!/bin/bash
# round 3: does the squat clip's 13-step plateau break with more epochs?  (walk needed ~1900; 3000 were not enough here)
O=gpurun_out/r03_28; mkdir -p $O
timeout 900 python scripts/learning_curve.py 7000 4096 $O/learning_curve_squat_7000.json env.motion_file=squat:10 > $O/train.log 2>&1; grep -E "epoch +(1|500|1000|1500|2000|2500|3000|3500|4000|4500|5000|5500|6000|6500|7000) |acceptance" $O/train.log | cut -c1-200
