#!/bin/bash
set -u -o pipefail
mkdir -p gpurun_out
# 2-rank rehearsal of the multi-GPU step structure on ONE device (gloo exchange): both ranks must finish, report 2 ranks, and
# (same seed-per-rank data) end with finite losses.  (The gloo exchange of the 282 MB gradient arena through the host takes seconds per step: 3 steps.)
CLC_BENCH_WATCHDOG=${WATCHDOG:-150} CLC_SINGLE_DEVICE=1 CLC_DIST_BACKEND=gloo timeout -k 10 ${LIMIT:-400} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-roofline > gpurun_out/bench_v10_2rank.json 2> gpurun_out/bench_v10_2rank.err
echo "2-rank rc=$?"; tail -c 900 gpurun_out/bench_v10_2rank.json; echo; grep -v "^\[W\|amdgpu.ids\|^\*\*\*\|OMP_NUM" gpurun_out/bench_v10_2rank.err | head -80 | cut -c1-200
