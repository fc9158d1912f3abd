#!/bin/bash
# ONE parameterised GPU-box script (round 5: replaces the 22 one-shot gpu_r4[a-z].sh of round 4 - they are in the git
# history, their outputs are profiles/r04*).  Each sub-command is one well-defined job that writes into gpurun_out/ with
# the given tag; several jobs can be chained in one `gpurun` call:
#
#   gpurun --timeout 1500 -- 'bash tools/gpu_job.sh tests r05a "tblock or ff_fused"; bash tools/gpu_job.sh ab r05a VX_TB_FUSED 0 1'
#
#   box      <tag>                                  GPU / host identity of the box (clocks, CU count, cores)
#   tests    <tag> [-k expression] [file ...]       pytest -m gpu of the given files (default: all of tests/), summary kept
#   tests16  <tag> [-k expression]                  tests/test_gpu_kernels.py with VX_TEST_ELEM=f16 (IEEE-half library) + tests/test_gpu_f16.py
#   smoke    <tag>                                  __graft_entry__.smoke()
#   bench    <tag> [bench.py flags ...]             one bench line -> <tag>_bench.json (+ per-shape GEMM table)
#   ab       <tag> <ENVVAR> <a> <b> [reps] [bench flags ...]   same-box A/B of an environment knob, interleaved reps
#   ablib    <tag> <libA.so> <libB.so> [reps] [bench flags ...] same-box A/B of two builds of the library (VX_LIBRARY)
#   configs  <tag>                                  the other single-GPU configurations (64 / 124 frames, 768x768, ctx 24)
#   profile  <tag>                                  rocprofv3 --kernel-trace --stats of one bench clip -> <tag>_trace_summary.txt
#   pmc      <tag>                                  HBM-side traffic per kernel (separate --pmc passes) -> <tag>_pmc_traffic.{json,txt}
#   final    <tag>                                  the round's validation: box, tests (all), smoke, bench, profile, pmc, bench again
#   scale8   <tag>                                  multi-GPU pre-flight + scaling runs (needs >= 2 GPUs): a 60 s two-rank RCCL probe
#                                                   (tools/rccl_probe.py: init, sub-groups, all-gather, all-to-all, sharded loop vs
#                                                   sequential, bit for bit) FIRST, then bench.py --gpus 1 2 4 8 at F = 124 with
#                                                   per-rank / per-collective timing; stops at the first failure with the step named
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PYTHONPATH
OUT=gpurun_out
JOB=$1; T=${2:-r05}; shift 2

fps_of() {   # last JSON line of a bench run -> "fps ms_per_step"
  python -c "
import sys, json
v = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print(round(v[-1]['value'], 3), round(v[-1]['ms_per_step'], 1)) if v else print('no-result')"
}

case "$JOB" in
box)
  { rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; rocm-smi --showclocks 2>/dev/null | head -12; nproc; free -g | head -2; } > $OUT/${T}_box.log 2>&1
  python tools/lib_id.py > $OUT/${T}_lib_id.txt ;;
tests)
  K=""; if [ "$1" = "-k" ]; then K="$2"; shift 2; fi
  FILES="${*:-tests}"
  if [ -n "$K" ]; then timeout 1500 python -m pytest $FILES -m gpu -q -s --tb=short -p no:cacheprovider -k "$K" > $OUT/${T}_pytest_gpu.log 2>&1
  else timeout 1500 python -m pytest $FILES -m gpu -q --tb=short -p no:cacheprovider --durations=25 > $OUT/${T}_pytest_gpu.log 2>&1; fi
  echo "pytest exit $?" >> $OUT/${T}_pytest_gpu.log
  grep -E "^\[|passed|failed|rror|assert|pytest exit|s call|s setup" $OUT/${T}_pytest_gpu.log | tail -90 > $OUT/${T}_pytest_gpu_summary.log
  tail -4 $OUT/${T}_pytest_gpu_summary.log ;;
tests16)
  # the whole kernel test file against the IEEE-half build (libvexpress_hip_f16.so), tolerances 8x tighter + the always-on f16 cases
  K=""; if [ "$1" = "-k" ]; then K="$2"; shift 2; fi
  VX_TEST_ELEM=f16 timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f16.py -m gpu -q -s --tb=short -p no:cacheprovider ${K:+-k "$K"} > $OUT/${T}_pytest_f16.log 2>&1
  echo "pytest exit $?" >> $OUT/${T}_pytest_f16.log
  grep -E "^\[|passed|failed|rror|assert|pytest exit" $OUT/${T}_pytest_f16.log | tail -80 > $OUT/${T}_pytest_f16_summary.log
  tail -4 $OUT/${T}_pytest_f16_summary.log ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${T}_smoke.log 2>&1; tail -2 $OUT/${T}_smoke.log ;;
bench)
  timeout 900 python bench.py --gemm-shapes $OUT/${T}_gemm_by_shape.txt --detail $OUT/${T}_bench_detail.json "$@" > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err
  python -c "import sys; n = len(open('$OUT/${T}_bench.json').read().strip().splitlines()[-1]); print('bench line bytes:', n); sys.exit(0 if n < 4096 else 1)"
  fps_of < $OUT/${T}_bench.json ;;
ab)
  VAR=$1; A=$2; B=$3; REPS=${4:-2}; shift 4 2>/dev/null || shift $#
  for rep in $(seq 1 $REPS); do for v in $A $B; do
    r=$(env $VAR=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>> $OUT/${T}_ab.err | fps_of)
    echo "$VAR=$v rep $rep: $r" | tee -a $OUT/${T}_ab_${VAR}.txt
  done; done ;;
ablib)
  LA=$1; LB=$2; REPS=${3:-2}; shift 3 2>/dev/null || shift $#
  for rep in $(seq 1 $REPS); do for l in $LA $LB; do
    r=$(VX_LIBRARY=$PWD/$l timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>> $OUT/${T}_ab.err | fps_of)
    echo "lib=$l rep $rep: $r" | tee -a $OUT/${T}_ablib.txt
  done; done ;;
configs)
  for cfg in "F64:--frames 64" "F124:--frames 124" "768:--size 768" "768_fp8:--size 768 --fp8" "ctx24_F44:--frames 44 --context-frames 24"; do
    name=${cfg%%:*}; flags=${cfg#*:}
    timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline $flags > $OUT/${T}_bench_${name}.json 2>> $OUT/${T}_bench.err
    echo "$name: $(fps_of < $OUT/${T}_bench_${name}.json)"
  done ;;
profile)
  SHA=$(python tools/lib_id.py)
  CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- $CMD > $OUT/${T}_prof_bench.log 2> $OUT/${T}_prof_bench.err
  for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp "$f" $OUT/${T}_kernel_stats.csv; done
  for f in $(find /tmp/prof -name "*kernel_trace.csv"); do python tools/trace_summary.py "$f" $SHA $CMD > $OUT/${T}_trace_summary.txt; done
  head -12 $OUT/${T}_trace_summary.txt ;;
pmc)
  bash tools/exp_pmc_bench.sh $T > $OUT/${T}_pmc.log 2>&1; tail -5 $OUT/pmc_${T}_passes.log ;;
final)
  bash tools/gpu_job.sh box $T
  cat /sys/fs/cgroup/cpu.max >> $OUT/${T}_box.log 2>&1
  bash tools/gpu_job.sh tests $T
  bash tools/gpu_job.sh smoke $T
  bash tools/gpu_job.sh bench $T
  bash tools/gpu_job.sh profile $T
  bash tools/gpu_job.sh pmc $T
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${T}_bench2.json 2>> $OUT/${T}_bench.err ;;
scale8)
  NG=$(python -c "import torch; print(torch.cuda.device_count())")
  echo "visible GPUs: $NG" | tee $OUT/${T}_scale8.log
  if [ "$NG" -lt 2 ]; then echo "scale8 needs >= 2 GPUs (RCCL refuses two ranks on one device)" | tee -a $OUT/${T}_scale8.log; exit 3; fi
  export HSA_ENABLE_IPC_MODE_LEGACY=0
  timeout 60 python tools/rccl_probe.py --gpus 2 > $OUT/${T}_rccl_probe_2.json 2> $OUT/${T}_rccl_probe_2.err
  rc=$?; echo "rccl_probe --gpus 2: exit $rc (124 = hung: the last 'step:' line of ${T}_rccl_probe_2.err names the call)" | tee -a $OUT/${T}_scale8.log
  tail -3 $OUT/${T}_rccl_probe_2.err | tee -a $OUT/${T}_scale8.log
  [ $rc -ne 0 ] && exit $rc
  if [ "$NG" -ge 8 ]; then timeout 120 python tools/rccl_probe.py --gpus 8 > $OUT/${T}_rccl_probe_8.json 2> $OUT/${T}_rccl_probe_8.err
    echo "rccl_probe --gpus 8: exit $?" | tee -a $OUT/${T}_scale8.log; fi
  for n in 1 2 4 8; do
    [ "$n" -gt "$NG" ] && break
    timeout 900 python bench.py --gpus $n --frames 124 --steps 2 --warmup 1 --no-cpu-baseline --detail $OUT/${T}_scale_${n}_detail.json \
      > $OUT/${T}_scale_${n}.json 2> $OUT/${T}_scale_${n}.err
    echo "N=$n: exit $? $(fps_of < $OUT/${T}_scale_${n}.json)" | tee -a $OUT/${T}_scale8.log
  done ;;
*)
  echo "usage: gpu_job.sh box|tests|smoke|bench|ab|ablib|configs|profile|pmc|final|scale8 <tag> [...]"; exit 2 ;;
esac
