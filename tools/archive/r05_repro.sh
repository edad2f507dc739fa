#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
fails=0
for i in $(seq 1 ${2:-16}); do
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${1:-side_stream or no_multi_workgroup}" > /tmp/rep_$i.log 2>&1 || { fails=$((fails+1)); grep -a "Fatal\|File \"/tmp/code\|Error" /tmp/rep_$i.log | head -6 | cut -c1-200; }
done
echo "$fails of ${2:-16} runs failed"
