#!/bin/bash
set -u -o pipefail
# Asserting 2-rank rehearsal of the multi-GPU step structure on ONE device (tools/rehearse_2rank.py): 1-rank reference with the same
# three-graph structure first, then two ranks over gloo with identical shards — loss sequence and parameters must be bit-identical.
# The report goes to gpurun_out/rehearse_2rank.txt (copy it to profiles/ for the record).
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
cd "$R"; mkdir -p gpurun_out
timeout -k 10 ${LIMIT:-300} python tools/rehearse_2rank.py --single gpurun_out/rehearse_ref.json > gpurun_out/rehearse_2rank.txt 2> gpurun_out/rehearse_single.err || { echo "1-rank reference failed"; tail -5 gpurun_out/rehearse_single.err; exit 3; }
timeout -k 10 ${LIMIT:-400} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rehearse_2rank.py --check gpurun_out/rehearse_ref.json >> gpurun_out/rehearse_2rank.txt 2> gpurun_out/rehearse_2rank.err
rc=$?
echo "2-rank rc=$rc" >> gpurun_out/rehearse_2rank.txt
grep -h "REHEARSAL\|single-rank\|rc=" gpurun_out/rehearse_2rank.txt | cut -c1-1200
[ $rc -eq 0 ] || { grep -v "^\[W\|amdgpu.ids\|^\*\*\*\|OMP_NUM" gpurun_out/rehearse_2rank.err | tail -30 | cut -c1-300; exit $rc; }
# ... and the SAME structure with ONE rank over RCCL (backend nccl, CLC_FORCE_COLLECTIVES=1): every all-reduce really issued on a 1-rank communicator
timeout -k 10 ${LIMIT:-300} python tools/rehearse_2rank.py --rccl1 gpurun_out/rehearse_ref.json >> gpurun_out/rehearse_2rank.txt 2> gpurun_out/rehearse_rccl1.err
rc1=$?
echo "rccl 1-rank rc=$rc1" >> gpurun_out/rehearse_2rank.txt
grep -h "REHEARSAL" gpurun_out/rehearse_2rank.txt | tail -1 | cut -c1-1200
[ $rc1 -eq 0 ] || { grep -v "^\[W\|amdgpu.ids" gpurun_out/rehearse_rccl1.err | tail -20 | cut -c1-300; exit $rc1; }
