# usage: tools/sweep.sh VAR v1 v2 ... : one bench run per value of the environment variable
var=$1; shift
mkdir -p gpurun_out/r2
for v in "$@"; do
  t=$(basename $v)
  env $var=$v timeout -k 10 400 python bench.py --steps 200 --no-cpu-baseline --others ${OTHERS:-nanopore_1k_families,nanopore_20k,nanopore_200k,illumina_w1} > gpurun_out/r2/sweep_${var}_$t.json 2> gpurun_out/r2/sweep_${var}_$t.err || { echo "run $v failed"; tail -5 gpurun_out/r2/sweep_${var}_$t.err; exit 1; }
  echo "=== $var=$v"
  python tools/show_bench.py gpurun_out/r2/sweep_${var}_$t.json | grep -v "^   M=\|phases:"
done
