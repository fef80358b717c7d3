#!/bin/bash
O=gpurun_out/r03_30; mkdir -p $O
timeout 600 python -m pytest tests/test_task_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python scripts/probes/post_timeline.py > $O/post_timeline.txt 2>&1; tail -13 $O/post_timeline.txt
