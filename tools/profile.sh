# usage: tools/profile.sh TAG WORKLOAD [STEPS]: rocprofv3 kernel trace + PMC passes (HBM-side traffic, VALU instructions)
# of bench.py on one workload; summaries land in gpurun_out/prof/TAG_WORKLOAD_*.  Run from the repository root.
tag=$1; wl=$2; steps=${3:-100}
root=$(pwd); out=$root/gpurun_out/prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
common="--steps $steps --no-cpu-baseline --others none --workload $wl"
# (one step of the pass that times kernels alone: the trace's averages are those of the timed region and its warm-up)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${wl}_kt -o kt -- python3 $root/bench.py $common --alone-steps 1 > $out/${tag}_${wl}_bench_under_rocprof.json 2> $out/${tag}_${wl}_kt.err || { echo "kernel trace failed"; tail -5 $out/${tag}_${wl}_kt.err; exit 1; }
find $out/${tag}_${wl}_kt -name "*kernel_stats.csv" -exec cp {} $out/${tag}_${wl}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"; do
  name=$(echo $c | cut -d' ' -f1)
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_${wl}_$name -o pmc -- python3 $root/bench.py --steps 10 --no-cpu-baseline --others none --workload $wl > /dev/null 2> $out/${tag}_${wl}_$name.err || { echo "pmc $name failed"; tail -5 $out/${tag}_${wl}_$name.err; exit 1; }
  find $out/${tag}_${wl}_$name -name "*counter_collection.csv" -exec cp {} $out/${tag}_${wl}_$name.csv \;
done
cd $root
python3 tools/pmc_summary.py traffic $wl $out/${tag}_${wl}_FETCH_SIZE.csv $out/${tag}_${wl}_WRITE_SIZE.csv $out/${tag}_pmc_traffic.json
python3 tools/pmc_summary.py valu $wl $out/${tag}_${wl}_SQ_INSTS_VALU.csv $out/${tag}_pmc_traffic.json
python3 tools/pmc_summary.py counters $out/${tag}_${wl}_SQ_INSTS_VALU.csv > $out/${tag}_${wl}_sq_counters.txt
rm -rf $out/${tag}_${wl}_kt $out/${tag}_${wl}_FETCH_SIZE $out/${tag}_${wl}_WRITE_SIZE $out/${tag}_${wl}_SQ_INSTS_VALU
head -12 $out/${tag}_${wl}_kernel_stats.csv | cut -c1-60,300-
