#!/bin/bash
O=gpurun_out/r03_18; mkdir -p $O
timeout 900 python -m pytest tests/test_env_gpu.py -m gpu -x -q -k "hist_obs or switches" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python scripts/probes/policy_fail_probe.py squat:10 700 > $O/squat_fail.txt 2>&1; tail -120 $O/squat_fail.txt
