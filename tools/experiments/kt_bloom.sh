export TMPDIR=/tmp
mkdir -p gpurun_out/kt_bloom
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_bloom -o kt --output-format csv -- python3 tools/bench_passes.py --only "bloom" --iters 20 > gpurun_out/kt_bloom.log 2>&1 && python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/kt_bloom/**/kt_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
by=collections.defaultdict(list)
for r in rows:
    g=(r['Kernel_Name'][:60], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'))
    by[g].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for g,v in by.items():
    v=sorted(v); print(g, len(v), 'median us', v[len(v)//2])
# gaps between consecutive kernels
rows.sort(key=lambda r:int(r['Start_Timestamp']))
gaps=[(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1000 for a,b in zip(rows,rows[1:])]
gaps=sorted(gaps); print('gap median us', gaps[len(gaps)//2], 'p90', gaps[int(len(gaps)*.9)])
PY
