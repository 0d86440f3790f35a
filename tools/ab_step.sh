#!/bin/bash
# A/B of the training step under CLC_TUNING settings on ONE box, interleaved: tools/ab_step.sh "13:0" "13:1" [rounds]
# (boxes differ by several % in clock under load: never compare img/s across gpurun calls)
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
A="$1"; B="$2"; N=${3:-3}
cd "$R"
for i in $(seq 1 $N); do
  for cfg in "$A" "$B"; do
    v=$(CLC_TUNING="$cfg" python bench.py --no-cpu-baseline --no-parity --no-roofline --no-reduced --no-reference-loop --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "round $i  [$cfg]  $v"
  done
done
