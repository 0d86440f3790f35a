#!/bin/bash
mkdir -p gpurun_out
# 2-rank rehearsal of the multi-GPU step structure on ONE device (gloo exchange): both ranks must finish, report 2 ranks, and
# (same seed-per-rank data) end with finite losses
CLC_SINGLE_DEVICE=1 CLC_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --no-roofline > gpurun_out/bench_v10_2rank.json 2> gpurun_out/bench_v10_2rank.err
echo "2-rank rc=$?"; tail -c 900 gpurun_out/bench_v10_2rank.json; echo; tail -n 6 gpurun_out/bench_v10_2rank.err | cut -c1-300
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/bench_v10.json 2> gpurun_out/bench_v10.err && python -c "import json; d=json.load(open('gpurun_out/bench_v10.json')); print('1 rank:', round(d['value'],2), 'img/s')"
