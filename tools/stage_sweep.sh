# usage: tools/stage_sweep.sh WORKLOAD...: pipelined step time of bench.py with the path cut after stage 1..4
mkdir -p gpurun_out/r3
for wl in "$@"; do
  for st in 1 2 3 4; do
    timeout -k 10 200 python bench.py --steps 300 --no-cpu-baseline --others none --workload $wl --last-stage $st > gpurun_out/r3/stage_${wl}_$st.json 2> gpurun_out/r3/stage_${wl}_$st.err || { echo "run $wl $st failed"; tail -5 gpurun_out/r3/stage_${wl}_$st.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("gpurun_out/r3/stage_${wl}_$st.json").read().strip().splitlines()[-1])
print("%-22s last stage $st: %.4f ms/step  %.1f Gbases/s" % ("$wl", d["ms_per_step"], d["value"] / 1e9))
PY
  done
done
