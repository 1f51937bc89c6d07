#!/usr/bin/env python3
"""Prints the headline and the other workloads of bench.py JSON lines as a table."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:
        print(path, "unreadable:", e)
        continue
    print("== %s" % path)
    r = d["roofline"]
    print("%-22s %8.1f Gbases/s  %.4f ms/step  hits/s %.2e  roofline %s %.0f GB/s frac %.3f (alone %.3f ms)" % (
        d["config"]["workload"].split(":")[0], d["value"] / 1e9, d["ms_per_step"], d["hits_per_s"], r["kernel"],
        r["achieved"], r["frac"], r.get("ms_per_launch_alone", 0)))
    print("   phases: " + "  ".join("%s %.3f ms %.0f GB/s" % (k, v["ms"], v["GBps"]) for k, v in d["phases"].items()))
    if d.get("cpu_baseline"):
        print("   cpu: %.1f Mbases/s x1, %s" % (d["cpu_baseline"]["value"] / 1e6,
                                                d["cpu_baseline"].get("all_cores", {}).get("value")))
    for name, o in (d.get("other_workloads") or {}).items():
        if not isinstance(o, dict):
            print("%-22s %s" % (name, o))
            continue
        km = o["kernel_ms_alone"]
        print("%-22s %8.1f Gbases/s  %.4f ms/step  hits/s %.2e  lookups/s %.2e  | %s | setup %.0f s" % (
            name, o["bases_per_s"] / 1e9, o["ms_per_step"], o["hits_per_s"], o["lookups_per_s"],
            " ".join("%s %.3f" % (k, v) for k, v in km.items() if v > 0), o["setup_s"]))
        print("   M=%.2e H=%.2e C=%.2e  %s" % (o["minimizers"], o["hits"], o["clusters_all"], o["workload"]))
        if o.get("step_traffic"):
            print("   memory traffic of a step (PMC passes): %.0f MB -> %.0f GB/s, %.2f of the HBM peak" % (
                o["step_traffic"]["bytes_per_step"] / 1e6, o["step_traffic"]["GBps"], o["step_traffic"]["frac_of_hbm_peak"]))
