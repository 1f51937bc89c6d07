"""usage: tools/bench_table.py BENCH.json: one bench.py line as a markdown table, a row per workload."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rows = [dict(d, workload="nanopore_1k", bases_per_s=d["value"], bases=d["config"]["bases_per_gpu"],
             hits=d["config"]["hits_per_gpu"])]
rows += [dict(w, workload=n) for n, w in d["other_workloads"].items()]
print("| workload | bases / step | hits / step | Gbases/s | ms / step | sketch+probe | emit | sort+cluster (tail + rows) |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    ph = r["phases"]
    sc = ph["sort+cluster"]["kernels"]
    print("| `%s` | %.2e | %.2e | %.1f | %.3f | %.3f | %.3f | %.3f (%s) |" % (
        r["workload"], r["bases"], r["hits"], r["bases_per_s"] / 1e9, r["ms_per_step"], ph["sketch+probe"]["ms"],
        ph["emit"]["ms"], ph["sort+cluster"]["ms"], " + ".join("%.3f" % v for v in sc.values())))
