#!/bin/bash
# The denoiser training step (f2): ms per replayed step and the kernel table of its roofline (bench.py --config train).  ~1 GPU-minute.
python bench.py --config train --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('train', round(d['value'], 3), 'ms/step; eager', d.get('eager_ms_per_step'))
r = d['roofline']
print({k: v for k, v in r.items() if k != 'kernels'})
for k in r['kernels']:
    print(f\"{k['share_of_launch_time']:6.3f} {k['calls']:4d} x {k['avg_us']:7.2f} us  {k['kernel'][:60]:60s} {k.get('achieved_tflops')} TF {k.get('achieved_gbs')} GB/s\")
"
