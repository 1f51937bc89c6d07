# usage: tools/run_filebench.sh OUT.json: pmap_pangraph_from_read_file on a 2 GB FASTQ of the bench reads (page cache
# warm), a few configurations, three runs each; results as one JSON object
set -e
out=${1:-gpurun_out/r2/file_bench.json}
mkdir -p gpurun_out/r2
export PMAP_TRACE_PIPELINE=1
run() {   # label, arguments of tools/file_bench.py
  label=$1; shift
  timeout -k 10 300 python tools/file_bench.py --copies 10 "$@" > gpurun_out/r2/fb_one.json 2> gpurun_out/r2/fb_one.err
  python - "$label" "$out" <<'PY'
import json, sys
label, out = sys.argv[1], sys.argv[2]
try:
    allr = json.load(open(out))
except Exception:
    allr = {}
d = json.load(open("gpurun_out/r2/fb_one.json"))
d["pipeline_trace"] = [l.strip() for l in open("gpurun_out/r2/fb_one.err") if l.startswith("[pmap] read-file pipeline")]
allr[label] = d
json.dump(allr, open(out, "w"), indent=1)
print(label, [round(r["bases_per_s"] / 1e9, 2) for r in d["runs"]])
PY
}
rm -f $out
run "no hits, 3 contexts (default)" --no-hits
run "hits kept, 3 contexts (default)"
run "no hits, 2 contexts" --no-hits --contexts 2
run "no hits, 4 contexts" --no-hits --contexts 4
run "no hits, 16-Mbase batches" --no-hits --batch-mbases 16
run "no hits, gzip" --no-hits --gz
for t in 4 8 16; do PMAP_PARSE_THREADS=$t run "no hits, $t parser threads" --no-hits; done
