#!/bin/bash
set -u -o pipefail
# One GPU-box visit: the -m gpu parity suite, then (unless a step was killed) the headline bench and the conv micro-benchmarks.
# usage: tools/gpu_suite.sh <tag> [pytest-args...]
tag=${1:-run}; shift
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -rf --durations=15 "$@" > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -n 40 gpurun_out/pytest_$tag.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
brc=$?
tail -c 600 gpurun_out/bench_$tag.json; tail -n 5 gpurun_out/bench_$tag.err
if [ $brc -ne 0 ]; then echo "bench rc=$brc"; exit $brc; fi
exit $rc
