#!/bin/bash
# round-4 first contact: GPU tests, the bench line, a host cProfile of one edit and the kernel trace of the timed region.  (run on the GPU box)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04a_tests.log
python bench.py --steps 8 --warmup 4 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
python tools/host_profile.py > gpurun_out/r04a_host_profile.log 2>&1
bash tools/profile_bench.sh r04a > gpurun_out/r04a_profile.log 2>&1
tail -5 gpurun_out/r04a_tests.log; cat gpurun_out/r04a_bench.json | cut -c1-600
