# usage: tools/profile_driver.sh TAG: rocprofv3 kernel trace of the headline workload under the driver's step counts
# (bench.py --gpus 1 --steps 20 --warmup 5; no other workloads, no CPU baseline, one step of the pass that times
# kernels alone, so that the trace holds the timed region, its warm-up and one more step)
tag=$1
root=$(pwd); out=$root/gpurun_out/prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_driver_kt -o kt -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --others none --no-cpu-baseline --alone-steps 1 > $out/${tag}_driver_cmd_bench_under_rocprof.json 2> $out/${tag}_driver_kt.err || { echo "kernel trace failed"; tail -5 $out/${tag}_driver_kt.err; exit 1; }
find $out/${tag}_driver_kt -name "*kernel_stats.csv" -exec cp {} $out/${tag}_driver_cmd_kernel_stats.csv \;
rm -rf $out/${tag}_driver_kt
cd $root
head -8 $out/${tag}_driver_cmd_kernel_stats.csv | cut -c1-60,300-
