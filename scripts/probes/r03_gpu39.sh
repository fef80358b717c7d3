#!/bin/bash
O=gpurun_out/r03_39; mkdir -p $O
timeout 400 python -m pytest tests/test_env_gpu.py -m gpu -x -q -k "bench_multi_rank" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
