#!/usr/bin/env python3
"""Summaries of rocprofv3 --pmc passes over bench.py.

    python tools/pmc_summary.py counters <counter_collection.csv>...
        per-kernel averages of every counter in the files

    python tools/pmc_summary.py valu <workload> <sq.csv> [out.json]
        SQ_INSTS_VALU per launch of the sketch kernel, merged into out.json (bench.py: roofline.valu)

    python tools/pmc_summary.py traffic <workload> <fetch.csv> <write.csv> [out.json]
        HBM-side bytes per launch of each phase of the path: FETCH_SIZE and WRITE_SIZE are reported in
        KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes, so reads are doubled.
        The result is merged into out.json (default profiles/pmc_traffic.json), which bench.py reads
        for roofline.traffic.
"""
import collections
import csv
import json
import os
import sys

PHASE_OF = [("sketch_wave", "sketch+probe"), ("sketch_dpp", "sketch+probe"), ("probe_wave", "sketch+probe"), ("scan_", None),
            ("scatter_group", "emit"), ("group_reduce", "emit"), ("sort_rank", "sort"), ("sort_wave", "sort"), ("long_presort", "sort"),
            ("long_rank", "sort"), ("cluster_cut", "cluster"), ("cluster_keep", "cluster"), ("tail_", "sort+cluster"),
            ("rows_gather", "sort+cluster")]


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def counters(files):
    for f in files:
        for k, d in per_kernel(f).items():
            print(k[:60], {c: round(sum(v) / len(v)) for c, v in d.items()})


def phase_sums(path, counter):
    """total of `counter` per phase and the number of sketch launches (= steps profiled)"""
    tot = collections.defaultdict(float)
    steps = 0
    for k, d in per_kernel(path).items():
        vals = d.get(counter)
        if not vals:
            continue
        ph = None
        for pat, p in PHASE_OF:
            if pat in k:
                ph = p if p else "scans"
                break
        if ph is None:
            continue
        tot[ph] += sum(vals)
        if ph == "sketch+probe":
            steps += len(vals)
    return tot, max(steps, 1)


def traffic(workload, fetch_csv, write_csv, out):
    f, sf = phase_sums(fetch_csv, "FETCH_SIZE")
    w, sw = phase_sums(write_csv, "WRITE_SIZE")
    res = {}
    for ph in sorted(set(f) | set(w)):
        rd = 2.0 * 1024.0 * f.get(ph, 0.0) / sf
        wr = 1024.0 * w.get(ph, 0.0) / sw
        res[ph] = rd + wr
        res[ph + "_read"] = rd
        res[ph + "_write"] = wr
    cur = {}
    if os.path.exists(out):
        cur = json.load(open(out))
    res["source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes: " + os.path.basename(fetch_csv) + ", " + os.path.basename(write_csv)
    keep = cur.get(workload, {}).get("sketch_valu_insts")
    if keep is not None:
        res["sketch_valu_insts"] = keep
    cur[workload] = res
    cur["_note"] = ("bytes per launch of each phase; reads = 2 x FETCH_SIZE KiB (gfx950 tallies 128-byte requests "
                    "as 64 bytes), writes = WRITE_SIZE KiB; separate --pmc passes; 'scans' (tile and cluster scans) "
                    "are listed apart from the phases they belong to")
    json.dump(cur, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in res.items() if not k.endswith(('_read', '_write'))}, indent=1))


def valu(workload, sq_csv, out):
    """SQ_INSTS_VALU (wavefront instructions) per launch of the sketch kernel, merged into out.json"""
    vals = []
    for k, d in per_kernel(sq_csv).items():
        if ("sketch_wave" in k or "sketch_dpp" in k) and d.get("SQ_INSTS_VALU"):
            vals += d["SQ_INSTS_VALU"]
    cur = json.load(open(out)) if os.path.exists(out) else {}
    cur.setdefault(workload, {})["sketch_valu_insts"] = sum(vals) / max(len(vals), 1)
    json.dump(cur, open(out, "w"), indent=1, sort_keys=True)
    print(cur[workload]["sketch_valu_insts"])


if __name__ == "__main__":
    default_out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
    if sys.argv[1] == "valu":
        valu(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else default_out)
    elif sys.argv[1] == "counters":
        counters(sys.argv[2:])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4],
                sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(__file__), "..", "profiles",
                                                                   "pmc_traffic.json"))
