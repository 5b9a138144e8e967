#!/bin/bash
# one-box sweep of an integer environment switch:  tools/envn.sh VAR v1 v2 ... (value "-" = unset)
V=$1; shift
for r in 1 2; do
for v in "$@"; do
  if [ "$v" == "-" ]; then unset $V; else export $V=$v; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-12s step %.2f ms | ' % ('$V=$v', d['ms_per_step']) + ' '.join('%s %.2f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])[:4]))
"
done; done
