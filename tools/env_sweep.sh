# usage: tools/env_sweep.sh "VAR=v1 VAR2=w1" "VAR=v2" ... -- WORKLOAD...: one bench.py run (headline only) per setting and workload
mkdir -p gpurun_out/r3
settings=()
while [ "$1" != "--" ]; do settings+=("$1"); shift; done
shift
for wl in "$@"; do
  for st in "${settings[@]}"; do
    tag=$(echo "$st" | tr ' =/.' '____')
    env $st timeout -k 10 200 python bench.py --steps 300 --no-cpu-baseline --others none --workload $wl ${BENCH_ARGS:-} > gpurun_out/r3/env_${wl}_$tag.json 2> gpurun_out/r3/env_${wl}_$tag.err || { echo "run $wl $st failed"; tail -5 gpurun_out/r3/env_${wl}_$tag.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("gpurun_out/r3/env_${wl}_$tag.json").read().strip().splitlines()[-1])
print("%-22s %-40s %.4f ms/step  %.1f Gbases/s" % ("$wl", "$st", d["ms_per_step"], d["value"] / 1e9))
PY
  done
done
