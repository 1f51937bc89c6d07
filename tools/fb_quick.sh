# usage: tools/fb_quick.sh "ENV=.. ENV2=.." ... : tools/file_bench.py --no-hits once per environment setting, with the pipeline trace
mkdir -p gpurun_out/r3
export PMAP_TRACE_PIPELINE=1
for st in "$@"; do
  env $st timeout -k 10 300 python tools/file_bench.py --copies ${FB_COPIES:-10} --no-hits ${FB_ARGS:-} > gpurun_out/r3/fbq.json 2> gpurun_out/r3/fbq.err || { echo "failed: $st"; tail -5 gpurun_out/r3/fbq.err; exit 1; }
  python - "$st" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r3/fbq.json"))
print("%-40s %s Gbases/s" % (sys.argv[1], [round(r["bases_per_s"] / 1e9, 1) for r in d["runs"]]))
for l in open("gpurun_out/r3/fbq.err"):
    if l.startswith("[pmap] read-file pipeline"):
        print("    " + l.strip()[26:])
PY
done
