export TMPDIR=/tmp
mkdir -p gpurun_out/kt_random
for w in 4k_deferred_gi_random 1080p_64_lights; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_random/$w -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 5 > gpurun_out/kt_random/$w.log 2>&1
python3 - $w <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/kt_random/%s/**/kt_kernel_stats.csv'%sys.argv[1],recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]: print(sys.argv[1], r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
done
