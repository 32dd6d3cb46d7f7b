cd tools/micro && hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../include wino4_stamps.hip -o /tmp/wino4_stamps 2>/dev/null && /tmp/wino4_stamps; cd ../..
python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-train-leg --no-c4 --no-c5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['single_stream_ms_per_step'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'])
for k,v in d['roofline']['by_layer'].items(): print(k, v['us'], v['frac'])
"
