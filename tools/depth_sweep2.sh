# usage: tools/depth_sweep2.sh WORKLOAD...: pipelined step time of bench.py with 2..6 steps in flight
mkdir -p gpurun_out/r3
for wl in "$@"; do
  for dp in 2 3 4 6; do
    timeout -k 10 200 python bench.py --steps 300 --no-cpu-baseline --others none --workload $wl --depth $dp > gpurun_out/r3/depth_${wl}_$dp.json 2> gpurun_out/r3/depth_${wl}_$dp.err || { echo "run $wl $dp failed"; tail -5 gpurun_out/r3/depth_${wl}_$dp.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("gpurun_out/r3/depth_${wl}_$dp.json").read().strip().splitlines()[-1])
print("%-22s depth $dp: %.4f ms/step  %.1f Gbases/s" % ("$wl", d["ms_per_step"], d["value"] / 1e9))
PY
  done
done
