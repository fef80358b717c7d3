#!/bin/bash
# quick GPU check: gpu tests + cfg3-shape bench
mkdir -p gpurun_out/check
O=gpurun_out/check
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
( time python bench.py --envs 8192 --motion-clips 2048 --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline > $O/bench_cfg3_shape.json ) 2> $O/bench_cfg3_shape.err
tail -4 $O/bench_cfg3_shape.err; cut -c1-400 $O/bench_cfg3_shape.json
