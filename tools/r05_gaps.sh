#!/bin/bash
# idle gaps of the main stream in one headline step:  gpurun -- 'bash tools/r05_gaps.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05gaps; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_g -o g -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --cpu-baseline off > /tmp/g.log 2>&1
DB=$(find /tmp/prof_g -name "*.db" | head -1)
tail -1 /tmp/g.log | cut -c1-150
python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $DB --step 8 > $OUT/gaps.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $DB --step 8 --min-us 0 > $OUT/timeline.txt 2>&1
head -60 $OUT/gaps.txt
