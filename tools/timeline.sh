# usage: tools/timeline.sh WORKLOAD [STEPS]: kernel trace of bench.py on one workload, then what every queue did in the
# middle of the run: busy time, gaps between kernels, time with 0 / 1 / 2 / more kernels running
wl=$1; steps=${2:-60}
root=$(pwd); out=$root/gpurun_out/prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/tl_$wl
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out/tl_$wl -o kt -- python3 $root/bench.py --steps $steps --no-cpu-baseline --others none --workload $wl > $out/tl_${wl}_bench.json 2> $out/tl_$wl.err || { echo "kernel trace failed"; tail -5 $out/tl_$wl.err; exit 1; }
cd $root
python3 tools/timeline.py $(find $out/tl_$wl -name "*kernel_trace.csv") ${3:-}
