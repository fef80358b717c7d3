#!/bin/bash
O=gpurun_out/r03_36; mkdir -p $O
timeout 300 python -m pytest tests/test_task_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
