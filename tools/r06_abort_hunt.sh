#!/bin/bash
# Round 6, VERDICT r5 #1(a): reproduce the SIGABRT of test_no_multi_workgroup_torch_reduction_in_capturable_steps[arxiv-1rank] IN-PROCESS
# (BOT_TEST_ISOLATED_CHILD=1 makes tests/conftest.py run `isolated` bodies in this process) and get the aborting thread's native frames
# (bot_debug_abort_trace) plus whatever the runtime prints on the way down (-s: no capture file to lose it in).
#   tools/r06_abort_hunt.sh <out dir> <runs> [-k expression]
OUT=${1:-gpurun_out/r06/hunt}; RUNS=${2:-8}; KEXPR=${3:-no_multi_workgroup}
mkdir -p "$OUT"
export BOT_TEST_ISOLATED_CHILD=1 TORCH_SHOW_CPP_STACKTRACES=1 PYTHONFAULTHANDLER=1
echo "core_pattern: $(cat /proc/sys/kernel/core_pattern)" > "$OUT/summary.txt"
for i in $(seq 1 "$RUNS"); do
  export BOT_ABORT_TRACE_FILE="$OUT/trace_$i.txt"
  t0=$(date +%s)
  timeout 600 python3 -m pytest tests/test_gpu_parity.py -k "$KEXPR" -x -q -s -p no:cacheprovider $HUNT_EXTRA > "$OUT/run_$i.log" 2>&1
  rc=$?
  echo "run $i ($KEXPR): rc $rc in $(( $(date +%s) - t0 )) s; trace $( [ -s "$OUT/trace_$i.txt" ] && echo yes || echo no ); $(tail -n 1 "$OUT/run_$i.log" | cut -c1-160)" | tee -a "$OUT/summary.txt"
done
