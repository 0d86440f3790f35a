#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_codec_service_gpu.py -m gpu -q -rf -x > gpurun_out/pytest_v4a.log 2>&1
rc=$?; tail -n 12 gpurun_out/pytest_v4a.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests -m gpu -q -rf --durations=8 --deselect tests/test_codec_service_gpu.py > gpurun_out/pytest_v4.log 2>&1
rc=$?; tail -n 25 gpurun_out/pytest_v4.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 500 python bench.py > gpurun_out/bench_v4.json 2> gpurun_out/bench_v4.err; brc=$?
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_v4.json'))
    print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k not in('roofline','cpu_baseline','config','parity','codec')})
    r=d.get('roofline',{}); print({k:v for k,v in r.items() if k not in ('per_kernel','timing_note')})
    print(d.get('parity')); print(d.get('codec')); print(d.get('cpu_baseline'))
except Exception as e:
    print('bench parse failed', e)
PY
tail -n 5 gpurun_out/bench_v4.err
CLC_FORCE_SPLIT_GRAPHS=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/bench_v4_split.json 2> gpurun_out/bench_v4_split.err && python -c "import json; d=json.load(open('gpurun_out/bench_v4_split.json')); print('forced split graphs', round(d['value'],2), 'img/s')"
exit $rc
