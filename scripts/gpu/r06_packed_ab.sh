#!/bin/bash
# Round 6, last stepper experiment: x / y components of the 3-vector operators and 3x3 products as packed fp32 instructions (phc_math.h, -DPHC_PACKED_F32 / -DPHC_PACKED_MAT),
# same-box A/B against the shipped library: rocprofv3 duration of the stepper launch at 4096 envs, headline, stepper parity tests with the variant.
O=gpurun_out/r6pk; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2; do
for tag in base pkmat pkops pkboth; do
  if [ $tag = base ]; then unset PHC_AMD_LIB; else export PHC_AMD_LIB=$PWD/phc_amd/_obj/libphc_amd_$tag.so; fi
  rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_$round -o b -- python bench.py --steps 300 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_${tag}_$round.json 2> $O/err_${tag}_$round.log
  echo "$tag round $round: $(python profiles/summarize_rocpd.py $(find /tmp/prof_${tag}_$round -name '*.db' | head -1) | grep k_sim_step | cut -c1-110)"
done; done
for tag in pkboth; do
  export PHC_AMD_LIB=$PWD/phc_amd/_obj/libphc_amd_$tag.so
  timeout 900 python -m pytest tests/test_dynamics.py tests/test_stepper_options.py tests/test_env_gpu.py -m gpu -q -x 2>&1 | tail -3
done
