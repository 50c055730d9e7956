export TMPDIR=/tmp
mkdir -p gpurun_out/kt_raster
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_raster -o kt --output-format csv -- python3 tools/bench_passes.py --only "raster" --iters 10 > gpurun_out/kt_raster.log 2>&1 && python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/kt_raster/**/kt_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 2 passes worth of kernels: print the final ~40 kernels with durations and gaps
tail=rows[-44:]
prev=None
for r in tail:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev)/1000 if prev else 0
    print(f"{r['Kernel_Name'][:70]:70s} grid {r.get('Grid_Size_X') or r.get('Grid_Size'):>9s} dur {(e-s)/1000:8.1f} us gap {gap:7.1f}")
    prev=e
PY
