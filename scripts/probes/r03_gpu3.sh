#!/bin/bash
# round 3, GPU call 3: stepper phase ablation at 4096 / 2048 envs (+ lifted: no ground contact)
O=gpurun_out/r03_3; mkdir -p $O
python scripts/probes/sim_ablation.py 4096 > $O/ablation_4096.txt 2>&1; cat $O/ablation_4096.txt
python scripts/probes/sim_ablation.py 2048 > $O/ablation_2048.txt 2>&1; cat $O/ablation_2048.txt
