# L2 requests / hits / misses of every kernel of tools/probe/xcd_probe (one rocprofv3 --pmc pass)
root=$(pwd); out=$root/gpurun_out/prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/xcd_probe -o pmc -- $root/tools/probe/xcd_probe > $out/xcd_probe.out 2> $out/xcd_probe.err || { echo "pmc failed"; tail -5 $out/xcd_probe.err; exit 1; }
cd $root
find $out/xcd_probe -name "*counter_collection.csv" -exec cp {} $out/xcd_probe_l2.csv \;
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/prof/xcd_probe_l2.csv')))
# one line per dispatch in launch order: kernel, counters
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r['Dispatch_Id']), {'k': r['Kernel_Name'].split('(')[0]})
    d[r['Counter_Name']] = float(r['Counter_Value'])
last = None
for i, d in disp.items():
    if d['k'].startswith('fill_') or d['k'].startswith('checksum'): continue
    key = d['k']
    print("%4d %-40s req %7.2f M  hit %7.2f M  miss %7.2f M" % (i, key[:40], d.get('TCC_REQ_sum', 0) / 1e6, d.get('TCC_HIT_sum', 0) / 1e6, d.get('TCC_MISS_sum', 0) / 1e6))
PY
