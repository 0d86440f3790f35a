#!/bin/bash
# A/B of the training step between two BUILDS of libclc_hip.so on ONE box, interleaved: tools/ab_lib.sh <other.so> [rounds]   (A = the in-tree build)
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
B="$1"; N=${2:-2}
cd "$R"
for i in $(seq 1 $N); do
  for lib in "" "$B"; do
    v=$(CLC_LIB_PATH="$lib" python bench.py --no-cpu-baseline --no-parity --no-roofline --no-reduced --no-reference-loop --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "round $i  [${lib:-in-tree}]  $v"
  done
done
