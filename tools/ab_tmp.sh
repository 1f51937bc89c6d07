mkdir -p gpurun_out/r2
run() { PANDORA_AMD_LIB=$1 PMAP_TEST_SKETCH_PAD=$2 timeout -k 10 200 python bench.py --steps 300 --no-cpu-baseline --others none > gpurun_out/r2/ab.json 2> gpurun_out/r2/ab.err || exit 1; python -c "
import json; d=json.load(open('gpurun_out/r2/ab.json')); print('$1 pad $2', round(d['ms_per_step'],4), 'alone', round(d['roofline']['ms_per_launch_alone'],4))"; }
run tools/ab_two.so 0
run tools/ab_one.so 0
run tools/ab_two.so 0
run tools/ab_one.so 0
run tools/ab_two.so 1500
run tools/ab_one.so 1500
