#!/bin/bash
# bench at several interval lengths (steady-state bandwidth of the pileup kernel)
for L in "$@"; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --length $L 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('len',c['interval_bp'],'tile',c['tile'],'segs',c['segments_per_gpu'],'pileup_ms',round(r['kernel_ms'],4),'GB/s',round(r['achieved'],1),'frac',round(r['frac'],4),'ms/step',round(d['ms_per_step'],4))"
done
