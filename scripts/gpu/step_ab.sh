#!/bin/bash
# One parameterised GPU-box script for the recurring "parity suite, then same-box A/B of the env step" call (it replaces the ~40 one-off
# scripts/probes/r03_gpu*.sh call logs of round 3 and the r05_step_ab*.sh pair of round 5; what each historical call ran is in the git history).
#
#   bash scripts/gpu/step_ab.sh OUT [-t "<pytest args>"] [-l "<lib tags>"] [-s "<solver switch sets>"] [-r "<robots>"] [-n REPS] [-b "<extra bench args>"] [-p "<probe cmd>"]
#     OUT        results go to gpurun_out/OUT/
#     -t         e.g. "tests/test_dynamics.py tests/test_stepper_options.py"   (run with -m gpu -q -x; skipped when empty)
#     -l         library variants next to the product build: tag X = phc_amd/_obj/libphc_amd_X.so (scripts/probes/build_variant.sh <rev> X,
#                or a -D build of the working tree); "default" = phc_amd/libphc_amd.so.  Default: "default"
#     -s         stepper switch sets to cross with the libraries, ';'-separated, each a list of KEY=VALUE (bench.py --solver): e.g. ";inertia_lag=1"
#                = plain and lagged.  Default: ""  (plain only)
#     -r         robots: smpl h1 g1.  Default: "smpl"
#     -n         alternating repetitions.  Default: 2
#     -b         extra bench.py arguments (e.g. "--actions tracking" or "--envs 8192")
#     -p         a probe command run at the end, output to OUT/probe.txt (e.g. "python scripts/probes/sim_wave_spread.py 4096")
O=gpurun_out/$1; shift
TESTS=""; LIBS="default"; SOLVERS=""; ROBOTS="smpl"; REPS=2; EXTRA=""; PROBE=""
while getopts "t:l:s:r:n:b:p:" o; do case $o in t) TESTS=$OPTARG;; l) LIBS=$OPTARG;; s) SOLVERS=$OPTARG;; r) ROBOTS=$OPTARG;; n) REPS=$OPTARG;; b) EXTRA=$OPTARG;; p) PROBE=$OPTARG;; esac; done
mkdir -p $O
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log; fi
B="python bench.py --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads $EXTRA"
IFS=';' read -ra SETS <<< "${SOLVERS};"
[ ${#SETS[@]} -eq 0 ] && SETS=("")
for rep in $(seq 1 $REPS); do for robot in $ROBOTS; do for lib in $LIBS; do for set in "${SETS[@]}"; do
  args=""; for kv in $set; do args="$args --solver $kv"; done
  [ $robot = h1 ] && args="$args --config 5"; [ $robot = g1 ] && args="$args --robot g1"
  tag=${robot}_${lib}_$(echo "${set:-plain}" | tr ' =' '__')_$rep
  if [ $lib = default ]; then unset PHC_AMD_LIB; else export PHC_AMD_LIB=$PWD/phc_amd/_obj/libphc_amd_$lib.so; fi
  $B $args > $O/$tag.json 2>> $O/err.log
done; done; done; done
unset PHC_AMD_LIB
python - <<PY
import glob, json
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.load(open(f))
        print(f"{f.split('/')[-1][:-5]:44s} {d['value'] / 1e6:7.2f} M env-steps/s  {d['ms_per_step'] * 1e3:7.1f} us/step  stepper {d['roofline']['kernel_ms'] * 1e3:6.1f} us")
    except Exception as e:
        print(f, 'ERR', e)
PY
if [ -n "$PROBE" ]; then $PROBE > $O/probe.txt 2>> $O/err.log; tail -20 $O/probe.txt; fi
grep -v amdgpu.ids $O/err.log | tail -3
