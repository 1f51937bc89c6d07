#!/usr/bin/env python3
"""End-to-end rate of pmap_pangraph_from_read_file (file on disk -> pangraph summary + coverage):
writes the nanopore_1k workload as a FASTQ file, `--copies` times over, and maps it.

    python tools/file_bench.py [--copies 3] [--gz] [--no-hits]
"""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=10)
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--no-hits", action="store_true")
    ap.add_argument("--batch-mbases", type=int, default=0, help="0 = the library's default")
    ap.add_argument("--contexts", type=int, default=0, help="mapping contexts on GPU 0 (0 = the library's default, four)")
    args = ap.parse_args()
    import pandora_amd
    from pandora_amd import api, synth
    pandora_amd.init(0)
    if args.contexts:
        api.init_devices([0] * args.contexts)
    si, bases, off, prm = synth.workload("nanopore_1k", seed=1)
    d = tempfile.mkdtemp(prefix="pmap_file_bench_")
    path = os.path.join(d, "reads.fq" + (".gz" if args.gz else ""))
    t0 = time.perf_counter()
    opener = (lambda p: gzip.open(p, "wb", compresslevel=1)) if args.gz else (lambda p: open(p, "wb"))
    one = b"".join(b"@r%d\n" % i + bases[int(off[i]):int(off[i + 1])].tobytes() + b"\n+\n" +
                   b"I" * int(off[i + 1] - off[i]) + b"\n" for i in range(len(off) - 1))
    with opener(path) as f:
        for c in range(args.copies):
            f.write(one)
    write_s = time.perf_counter() - t0
    size = os.path.getsize(path)
    gidx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                            si.iv_start, si.iv_len, si.min_path_length)
    n_bases = int(off[-1]) * args.copies
    out = {"file_bytes": size, "bases": n_bases, "reads": (len(off) - 1) * args.copies, "gz": args.gz,
           "keep_hits": not args.no_hits, "write_s": write_s, "contexts": args.contexts or 2,
           "parse_threads": os.environ.get("PMAP_PARSE_THREADS", "default"), "runs": []}
    for rep in range(3):
        t0 = time.perf_counter()
        pg = api.pangraph_from_read_file(path, gidx, prm["w"], prm["k"], prm["max_diff"], prm["e_rate"],
                                         prm["min_cluster_size"], genome_size=5000000, max_covg=1000000,
                                         batch_bases=args.batch_mbases << 20, keep_hits=not args.no_hits,
                                         coverage=True)
        dt = time.perf_counter() - t0
        info = pg.info()
        fwd, rev = pg.coverage()
        out["runs"].append({"s": dt, "bases_per_s": n_bases / dt, "file_MBps": size / dt / 1e6,
                            "add_node_calls": info["n_add_node_calls"], "hits_kept": info["n_hits"],
                            "coverage_sum": int(fwd.sum()) + int(rev.sum())})
        pg.close()
    os.remove(path)
    os.rmdir(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
