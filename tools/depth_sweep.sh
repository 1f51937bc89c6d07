# usage: tools/depth_sweep.sh [workload ...]: bench.py with 2, 3 and 4 steps in flight
mkdir -p gpurun_out/r2
for wl in ${@:-nanopore_1k nanopore_200k illumina_w1 nanopore_1k_families nanopore_20k}; do for d in 2 3 4; do
timeout -k 10 200 python bench.py --steps 200 --no-cpu-baseline --others none --workload $wl --depth $d > gpurun_out/r2/depth.json 2> gpurun_out/r2/depth.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2/depth.json')); print('$wl depth $d', round(d['ms_per_step'],4), 'ms', round(d['value']/1e9,1), 'Gbases/s')"
done; done
