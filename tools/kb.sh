#!/bin/bash
# quick kernel timing on the GPU box: per-kernel avg ms of the bench workload
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-host-path "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.4g sf/s  ms/step %.2f' % (d['value'], d['ms_per_step']))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms']): print('  %-22s %8.3f ms  %7.1f GB/s' % (k, v['avg_ms'], v['alg_GBps']))
"
