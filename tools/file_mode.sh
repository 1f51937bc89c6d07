# usage: tools/file_mode.sh N...: bench.py --mode file with N mapping contexts each (one process), value and per-run rates
mkdir -p gpurun_out/r3
for n in "$@"; do
  timeout -k 10 400 python bench.py --mode file --gpus $n --steps 4 > gpurun_out/r3/file_mode_$n.json 2> gpurun_out/r3/file_mode_$n.err || { echo "failed $n"; tail -5 gpurun_out/r3/file_mode_$n.err; exit 1; }
  python - $n <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open("gpurun_out/r3/file_mode_%s.json" % n).read().strip().splitlines()[-1])
print("%s contexts: %.1f Gbases/s  runs %s  %s" % (n, d["value"] / 1e9, [round(r["bases_per_s"] / 1e9, 1) for r in d["runs"]], d["config"]["workload"]))
for l in open("gpurun_out/r3/file_mode_%s.err" % n):
    if l.startswith("[pmap] read-file pipeline"):
        last = l
print("    " + last.strip()[26:])
PY
done
