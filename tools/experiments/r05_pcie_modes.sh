mkdir -p gpurun_out/r5e
for args in "--mode lanes --streams 2048" "--mode queues --streams 2048" "--mode queues --lanes 4 --chunk 32 --streams 2048" "--mode lanes --lanes 4 --chunk 32 --streams 2048" "--mode queues --lanes 6 --chunk 16 --streams 2048" "--mode lanes --streams 512" "--mode queues --lanes 4 --chunk 32 --streams 512"; do
  python tools/bench_pcie.py $args | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$args: %.3e sf/s  down %.1f GB/s up %.1f GB/s | serial h2d %.1f d2h %.1f GB/s graph %.2f ms' % (d['value'], d['down_GBps_in_pipeline'], d['up_GBps_in_pipeline'], d['one_chunk_serial']['h2d_GBps'], d['one_chunk_serial']['d2h_GBps'], d['one_chunk_serial']['graph_ms']))"
done 2>&1 | tee gpurun_out/r5e/pcie2.txt
