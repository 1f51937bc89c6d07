# usage: tools/pmc_l2.sh WORKLOAD: L2 requests / hits / misses per kernel (one rocprofv3 --pmc pass of bench.py)
wl=$1
root=$(pwd); out=$root/gpurun_out/prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/l2_$wl -o pmc -- python3 $root/bench.py --steps 10 --no-cpu-baseline --others none --workload $wl > /dev/null 2> $out/l2_$wl.err || { echo "pmc failed"; tail -5 $out/l2_$wl.err; exit 1; }
cd $root
find $out/l2_$wl -name "*counter_collection.csv" -exec cp {} $out/l2_$wl.csv \;
python3 tools/pmc_summary.py counters $out/l2_$wl.csv
