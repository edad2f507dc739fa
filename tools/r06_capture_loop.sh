#!/bin/bash
# Round 6, VERDICT r5 #1 "done" criterion: N consecutive in-process runs (BOT_TEST_ISOLATED_CHILD=1: no child process in between) of the
# tests that capture the 1-rank partitioned step with a live RCCL process group - the sequence that died about one time in ten before
# bot_amd.train.drain_rccl_watchdog.      tools/r06_capture_loop.sh <out dir> <runs>
exec "$(dirname "$0")/r06_abort_hunt.sh" "${1:-gpurun_out/r06/loop}" "${2:-32}" "1rank"
