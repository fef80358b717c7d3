#!/bin/bash
python -m pytest tests/test_env_gpu.py -m gpu -x -q -k "ppo or mcp or h1_ppo" 2>&1 | tail -2
python bench.py --steps 50 --warmup 10 --ppo-epochs 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(v,1) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('ppo_') and k!='ppo_config'})"
