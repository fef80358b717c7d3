#!/bin/bash
# round 3: the whole device suite
O=gpurun_out/r03_7; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --durations=12 > $O/pytest_gpu.log 2>&1; tail -40 $O/pytest_gpu.log
