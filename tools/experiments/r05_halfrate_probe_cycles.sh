cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in base atanprobe base atanprobe; do
  if [ "$t" == "base" ]; then unset NAE_GPU_LIB; else export NAE_GPU_LIB=$R/nodey-audio-editor_amd/variants/libnae_gpu_$t.so; fi
  rm -rf $R/gpurun_out/r5u/g_$t
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES -d $R/gpurun_out/r5u/g_$t -o pmc --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-pcie --no-host-path --no-kernel-timing > $R/gpurun_out/r5u/g_$t.log 2>&1
  python3 - $R/gpurun_out/r5u/g_$t $t <<'PY'
import csv,sys,glob,collections
d,t=sys.argv[1:3]
acc=collections.defaultdict(float); n=set(); dur={}
for r in csv.DictReader(open(glob.glob(d+'/**/pmc_counter_collection.csv',recursive=True)[0])):
    if 'pv_pipe' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n.add(r['Dispatch_Id'])
        dur[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp'])) if 'End_Timestamp' in r else None
L=len(n)
ms=[v for v in dur.values() if v]
avg_ms=sum(ms)/len(ms)/1e6 if ms else float('nan')
cyc=acc['GRBM_GUI_ACTIVE']/L/8
print(f"{t:10s} pv_pipe: launches {L}, kernel cycles {cyc/1e6:.3f} M, VALU instr {acc['SQ_INSTS_VALU']/L:.4e}, avg ms {avg_ms:.3f}, clock {cyc/avg_ms/1e6:.3f} GHz")
PY
done
