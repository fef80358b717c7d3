python scripts/learning_curve.py 3000 2048 gpurun_out/squat_2048envs.json env.motion_file=squat:10 2>&1 | grep -v amdgpu.ids > gpurun_out/squat_2048envs.log
tail -4 gpurun_out/squat_2048envs.log | cut -c1-400
python scripts/multi_clip_acceptance.py --stage1-s 270 --stage2-s 0 --envs 8192 --eval-every 500 --out gpurun_out/multi_clip_64_8192_mb65536.json learning.params.config.minibatch_size=65536 learning.params.config.amp_minibatch_size=16384 2>&1 | grep -v "amdgpu.ids\|AMP reference table" > gpurun_out/multi_clip_64_8192_mb65536.log
grep "sweep\|primitive" gpurun_out/multi_clip_64_8192_mb65536.log | cut -c1-250 | tail -12
