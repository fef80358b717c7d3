#!/bin/bash
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_results.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
( time python bench.py --steps 20 --warmup 5 --ppo-epochs 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tuning run', {k:(round(v,1) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('ppo_') and k!='ppo_config'})" ) 2>&1 | tail -5
ls -la gpurun_out/tunableop_results*.csv | head; wc -l gpurun_out/tunableop_results*.csv | tail -1
export PYTORCH_TUNABLEOP_TUNING=0
python bench.py --steps 20 --warmup 5 --ppo-epochs 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tuned run', {k:(round(v,1) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('ppo_') and k!='ppo_config'})"
