# usage: tools/kt.sh WORKLOAD [STEPS]: rocprofv3 kernel trace of bench.py on one workload; prints the per-kernel table
wl=$1; steps=${2:-50}
root=$(pwd); out=$root/gpurun_out/prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$wl -o kt -- python3 $root/bench.py --steps $steps --no-cpu-baseline --others none --workload $wl > $out/kt_${wl}_bench.json 2> $out/kt_$wl.err || { echo "kernel trace failed"; tail -5 $out/kt_$wl.err; exit 1; }
cd $root
python3 - <<PY
import csv, glob
f = glob.glob('$out/kt_$wl/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:44].ljust(44), r['Calls'].rjust(5), '%8.1f us  min %8.1f  max %8.1f' % (float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
