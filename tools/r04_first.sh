#!/bin/bash
# round 4, first GPU pass: the GPU suite without the three long full-size legs, then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -q -x --deselect tests/test_zz_full_size_gpu.py -s 2>&1 | grep -v "^\[Gloo\]" > gpurun_out/r04/first_tests.log
tail -30 gpurun_out/r04/first_tests.log
python bench.py --gpus 1 > gpurun_out/r04/first_bench.json 2> gpurun_out/r04/first_bench.err
tail -c 1500 gpurun_out/r04/first_bench.json
