#!/bin/bash
# stretcher-kernel shape sweep: ms of st_td_kernel for batch sizes x candidates-per-thread
for n in 1 16 64 128 256 384 512 768; do
  for nc in 1 2 4; do
    NAE_DEBUG=td_nc=$nc python tests/tools/bench_wsola.py --streams $n --seconds 10 --steps 2 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams %4d nc %d  td %.3f ms  step %.3f ms' % (d['streams'], $nc, d['kernels_ms']['st_td_kernel'], d['ms_per_step']))
"
  done
done
