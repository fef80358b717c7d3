#!/bin/bash
# round 3, GPU call 2: extended microbenchmark (SGPR / vcc operands), stepper phase profile (s_memtime), lateral-PNN device test
O=gpurun_out/r03_2; mkdir -p $O
( cd profiles/microbench && ./valu_issue ../../$O/valu_issue.json > ../../$O/valu_issue.txt 2>&1 ); tail -3 $O/valu_issue.txt
python scripts/probes/sim_phase_profile.py run 4096 > $O/phase_4096.txt 2>&1; cat $O/phase_4096.txt
python scripts/probes/sim_phase_profile.py run 2048 > $O/phase_2048.txt 2>&1; tail -12 $O/phase_2048.txt
python scripts/probes/sim_phase_profile.py run 4096 +solver.self_collision=0 > $O/phase_4096_nosc.txt 2>&1; tail -12 $O/phase_4096_nosc.txt
timeout 600 python -m pytest tests/test_learn_gpu.py -m gpu -x -q -k "lateral or fused_relu" > $O/pytest_lateral.log 2>&1; tail -3 $O/pytest_lateral.log
