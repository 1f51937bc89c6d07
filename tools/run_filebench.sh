set -e
mkdir -p gpurun_out/r2
export PMAP_TRACE_PIPELINE=1
timeout -k 10 600 python -u -m pytest tests/test_gpu_pangraph.py tests/test_gpu_configs.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
for args in "--no-hits" "" "--no-hits --contexts 3" "--no-hits --batch-mbases 16"; do
  tag=$(echo "$args" | tr -d ' -')
  timeout -k 10 300 python tools/file_bench.py --copies 10 $args > gpurun_out/r2/file2_$tag.json 2> gpurun_out/r2/file2_$tag.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2/file2_$tag.json"))
print("$args", d["file_bytes"], [round(r["bases_per_s"]/1e9,2) for r in d["runs"]], d["runs"][0]["add_node_calls"], d["runs"][0]["hits_kept"])
PY
  grep pmap gpurun_out/r2/file2_$tag.err | tail -1
done
for t in 4 8 16; do
PMAP_PARSE_THREADS=$t timeout -k 10 300 python tools/file_bench.py --copies 10 --no-hits > gpurun_out/r2/file2_t$t.json 2> gpurun_out/r2/file2_t$t.err
python -c "
import json
d=json.load(open('gpurun_out/r2/file2_t$t.json'))
print('parse threads $t', [round(r['bases_per_s']/1e9,2) for r in d['runs']])"
done
