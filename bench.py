#!/usr/bin/env python3
"""Benchmark of the read -> PRG minimizer path (sketch + probe + sort + cluster) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the whole device path (pmap_map: sketch, probe/emit, sort, cluster cut) over
one batch of synthetic reads that is already resident in HBM (2-bit packed).  At N > 1 every rank
maps its own, equally sized batch against a replicated index (reads shard, no collective on the data
path); the value is the total number of read bases all ranks mapped per second.  Rank 0 prints one
JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak

# The library keeps four streams busy (two for sketch kernels, one for everything after them, one for
# uploads); torch and the collective library bring their own.  With the HIP default of 4 hardware queues
# per process the two sketch streams then end up on one queue and stop overlapping (measured: 433 instead
# of 463 Gbases/s once the process group exists).  Read by the HIP runtime when it starts; the GPU boxes of
# this pool export 4, so a default would not be enough: anything below 8 is raised.
try:
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
except ValueError:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

# kernels of each phase (names of pmap_kernel_name)
PHASES = {
    "sketch": ["sketch"],          # fused mode: the index lookup runs inside this kernel
    "probe": ["probe", "scatter"],
    "sort": ["sort_rank"],
    "cluster": ["cluster_cut", "cluster_keep"],
}


def algorithmic_bytes(cn, n_prg, n_reads_batch):
    """Algorithmic bytes of each phase for one launch of the path over one batch:
    B read bases, M minimizers, H hits, C clusters.
      sketch   B/4 (2-bit packed read) + 16 M (u64 hash + u32 position + u32 read/strand written)
      probe    2 (16 M + 16 M) (minimizer re-read + one 16-byte slot per probe, count and scatter pass)
               + H (16 + 20) (16-byte record read, 20-byte hit written)
      sort     2 * 20 * H * P, P = 8-bit passes over the (read, prg, strand) key
      cluster  20 H read + 4 H flag/id written + 28 C
    """
    B, M, H, C = cn["n_bases"], cn["n_minimizers"], cn["n_hits"], cn["n_clusters_all"]
    bits = max(1, int(np.ceil(np.log2(max(n_reads_batch, 2))))) + max(1, int(np.ceil(np.log2(max(n_prg, 2))))) + 1
    P = int(np.ceil(bits / 8))
    return {
        "sketch": B / 4 + 16 * M,
        "probe": 2 * (16 * M + 16 * M) + H * 36,
        "sort": 2 * 20 * H * P,
        "cluster": 20 * H + 4 * H + 28 * C,
    }, P


def cpu_baseline(si, bases, offsets, prm, fraction, budget_s=10.0, threads_budget_s=5.0):
    """The oracle (plain-C restatement of the reference path) timed on a bounded sample of the same reads:
    chunks of 250 reads, wrapping around the workload, until the time budget is used.  `value` is one thread,
    as the reference runs; `all_cores` is the same code on independent read chunks in one thread per core of
    this process' share of the host (the reference has no threads; reads are independent)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import threading
    import oracle_lib as O
    oi = O.FlatIndex(si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off, si.iv_start,
                     si.iv_len)
    n_reads = len(offsets) - 1
    chunk = 250

    def run_chunk(a):
        b = min(a + chunk, n_reads)
        sub_off = offsets[a:b + 1]
        ids = np.arange(a, b, dtype=np.uint32)
        sub = bases[int(sub_off[0]):int(sub_off[-1])]
        r = O.oracle_map(oi, sub, sub_off - sub_off[0], ids, prm["w"], prm["k"], prm["max_diff"], fraction,
                         prm["min_cluster_size"], 0xFFFFFFFF, si.min_path_length)
        return int(sub_off[-1] - sub_off[0]), len(r["hit_read"]), b - a

    done_bases, done_reads, hits = 0, 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n_reads:
        nb, nh, nr = run_chunk(done_reads % n_reads)
        done_bases += nb
        hits += nh
        done_reads += nr
    t_used = time.perf_counter() - t0
    out = {"value": done_bases / t_used, "unit": "bases/s", "cores": 1, "kind": "port",
           "sample": "%d reads (%d bases, %d hits) of the same workload in chunks of %d, wrapping around; "
                     "oracle/liboracle.so, one thread, %.1f s" % (done_reads, done_bases, hits, chunk, t_used),
           "hits_per_s": hits / t_used}

    # the calls release the interpreter lock: one thread per core, each on its own chunks
    try:
        n_thr = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")), 64))
    except Exception:
        n_thr = 1
    if n_thr > 1 and n_reads:
        totals = [0] * n_thr
        stop = time.perf_counter() + threads_budget_s

        def worker(t):
            a = (t * chunk * 7) % n_reads
            while time.perf_counter() < stop:
                nb, _, nr = run_chunk(a)
                totals[t] += nb
                a = (a + nr * n_thr) % n_reads

        t0 = time.perf_counter()
        ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        out["all_cores"] = {"value": sum(totals) / dt, "unit": "bases/s", "cores": n_thr,
                            "sample": "%d bases in %.1f s, %d threads on independent chunks of %d reads"
                                      % (sum(totals), dt, n_thr, chunk)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000,
                    help="timed steps (a step takes a quarter of a millisecond: a few hundred of them make the "
                         "barrier, the pipeline fill and the synchronisation at both ends of the timed region negligible)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="nanopore_1k")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=3,
                    help="steps in flight (1 = blocking pmap_map); with 3 both sketch streams always hold a kernel "
                         "while the host waits for a result, and the tail kernels of a step run next to the sketch "
                         "kernels of the following ones")
    ap.add_argument("--last-stage", type=int, default=0,
                    help="stop the path after this stage (1 sketch, 2 hits, 3 sort, 4 clusters = all; 0 = all): for "
                         "experiments, the reported value is then not the metric")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="no HIP events at all inside the timed steps; all kernel times come from the second pass")
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: whatever libraries print there while they start up
    # (the collective library prints a version banner) goes to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    if os.environ.get("PANDORA_BENCH_WATCHDOG"):
        # dumps every thread's Python stack to stderr if the run is still going after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PANDORA_BENCH_WATCHDOG"]), exit=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # PANDORA_BENCH_FORCE_DIST=1: go through torch.distributed (process group, barriers, reductions) even with one
    # rank, to see on a one-GPU box what the collective library's streams do to the timing of the N > 1 runs
    use_dist = world > 1 or os.environ.get("PANDORA_BENCH_FORCE_DIST") == "1"

    import torch
    import torch.distributed as dist
    # PANDORA_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("PANDORA_BENCH_BACKEND", "nccl")
    device = local_rank % max(torch.cuda.device_count(), 1) if backend == "gloo" else local_rank
    torch.cuda.set_device(device)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    import pandora_amd
    from pandora_amd import api, synth
    pandora_amd.init(device)

    # same index on every rank (replicated), a different shard of reads per rank
    def progress(msg):     # a big workload takes minutes to synthesise: say so on stderr
        if rank == 0:
            print("[bench] %s (%.0f s)" % (msg, time.perf_counter() - t_start), file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    progress("synthesising workload %s" % args.workload)
    si, bases, offsets, prm = synth.workload(args.workload, seed=1, scale=args.scale, shard=rank)
    progress("flattening the index: %d keys, %d records" % (len(si.keys), len(si.rec_prg)))
    fraction = api.fraction(prm["e_rate"], prm["k"])
    expected = 0xFFFFFFFF
    if prm["illumina"]:
        expected = 150 * 2 // prm["w"]
    gidx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                            si.iv_start, si.iv_len, si.min_path_length)
    progress("index on the device, uploading %d bases" % len(bases))
    t0 = time.perf_counter()
    reads = api.Reads(bases, offsets)
    upload_s = time.perf_counter() - t0

    def step(res, profile, wait=True):
        return api.map_reads(gidx, reads, prm["w"], prm["k"], prm["max_diff"], fraction, prm["min_cluster_size"],
                             expected, profile=profile, result=res, wait=wait, last_stage=args.last_stage)

    # `depth` result objects take turns, so that the host enqueues a step while the GPU runs the one before
    # (pmap_map_async / pmap_map_wait); depth 1 is the blocking pmap_map
    depth = max(1, args.depth)
    ring = [None] * depth
    for i in range(max(args.warmup, depth)):
        ring[i % depth] = step(ring[i % depth], 0)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[device]) if backend == "nccl" else dist.barrier()

    kernel_ms, sketch_ms = {}, 0.0
    sync()
    t0 = time.perf_counter()
    prof = 0 if args.no_kernel_events else 2
    inflight = [False] * depth

    def retire(j):
        nonlocal sketch_ms
        ring[j].wait()
        inflight[j] = False
        if prof:
            sketch_ms += ring[j].kernel_ms()["sketch"]

    for i in range(args.steps):
        # profile=2: one pair of HIP events rides on the dispatch of the sketch kernel (the dominant one)
        # on the library's stream, inside the timed region
        j = i % depth
        if inflight[j]:
            retire(j)
        ring[j] = step(ring[j], prof, wait=False)
        inflight[j] = True
    for j in range(depth):
        if inflight[j]:
            retire(j)
    sync()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    res = ring[(args.steps - 1) % depth] if args.steps else ring[0]
    cn = res.counts()
    # second, untimed pass with events on every kernel, for the per-phase table
    for _ in range(args.steps):
        res = step(res, 1)
        for name, ms in res.kernel_ms().items():
            kernel_ms[name] = kernel_ms.get(name, 0.0) + ms
    # the sketch kernel's time when it has the machine to itself (second pass, one step at a time); in the timed
    # region above it shares the CUs with the tail kernels of the step before, so its event time there is longer
    sketch_alone_ms = kernel_ms.get("sketch", 0.0)
    if prof:
        kernel_ms["sketch"] = sketch_ms

    red_dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([float(cn["n_bases"]), float(cn["n_hits"])], dtype=torch.float64, device=red_dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_bases, total_hits = float(tot[0].item()), float(tot[1].item())

    if rank == 0:
        steps = max(args.steps, 1)
        kernel_ms = {k: v / steps for k, v in kernel_ms.items()}
        algo, P = algorithmic_bytes(cn, si.n_prg, cn["n_reads"])
        phases = {}
        for ph, ks in PHASES.items():
            ms = sum(kernel_ms.get(k, 0.0) for k in ks)
            phases[ph] = {"ms": ms, "algorithmic_bytes": algo[ph],
                          "GBps": (algo[ph] / (ms * 1e-3) / 1e9) if ms > 0 else 0.0,
                          "kernels": {k: kernel_ms.get(k, 0.0) for k in ks}}
        dom = max(phases, key=lambda p: phases[p]["ms"])
        traffic, valu = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(args.workload, {})
                traffic = rec.get(dom)
                # the sketch kernel is bounded by integer VALU issue: wavefront instructions per launch (PMC
                # SQ_INSTS_VALU, same workload) against 256 CUs x 4 SIMDs x one wave64 instruction per 4 cycles
                if dom == "sketch" and rec.get("sketch_valu_insts") and args.scale == 1.0:
                    peak = 256 * 4 * 2.4e9 / 4
                    rate = rec["sketch_valu_insts"] / (phases[dom]["ms"] * 1e-3)
                    valu = {"wave_insts_per_launch": rec["sketch_valu_insts"], "achieved_insts_per_s": rate,
                            "peak_insts_per_s": peak, "frac": rate / peak,
                            "note": "VALU issue roofline at 2.4 GHz; the kernel's binding limit"}
            except Exception:
                traffic, valu = None, None
        roofline = {"bound": "hbm", "kernel": "%s phase (%s)" % (dom, "+".join(PHASES[dom])),
                    "achieved": phases[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": phases[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                    "ms_per_launch": phases[dom]["ms"], "algorithmic_bytes_per_launch": algo[dom], "valu": valu}
        if dom == "sketch" and sketch_alone_ms > 0:
            alone = sketch_alone_ms / steps
            roofline["ms_per_launch_alone"] = alone
            roofline["achieved_alone"] = algo[dom] / (alone * 1e-3) / 1e9
            if prof:
                # the library alternates the sketch kernels of successive steps between two streams: launches
                # overlap, each lasts longer than it would alone, and together they move this many bytes per second
                roofline["launches_in_flight"] = sketch_ms / (elapsed_local * 1e3)
                roofline["achieved_all_launches"] = algo[dom] * args.steps / elapsed_local / 1e9
            if valu:
                valu["frac_alone"] = valu["wave_insts_per_launch"] / (alone * 1e-3) / valu["peak_insts_per_s"]
        out = {
            "metric": "read bases mapped/sec (sketch+probe+cluster)",
            "value": total_bases * args.steps / elapsed,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32" if prm["k"] <= 15 else "u64",
            "data": "synthetic",
            "config": {"workload": "%s: %d-locus synthetic PRG index, %d reads/GPU (%d bases/GPU), w=%d k=%d"
                                   % (args.workload, si.n_prg, cn["n_reads"], cn["n_bases"], prm["w"], prm["k"]),
                       "reads_per_gpu": cn["n_reads"], "bases_per_gpu": cn["n_bases"],
                       "minimizers_per_gpu": cn["n_minimizers"], "hits_per_gpu": cn["n_hits"],
                       "clusters_all": cn["n_clusters_all"], "clusters_kept": cn["n_clusters"],
                       "index_keys": int(len(si.keys)), "index_records": int(len(si.rec_prg)),
                       "parallelism": "reads sharded over %d GPU(s), index replicated" % world,
                       "steps_in_flight": depth},
            "hits_per_s": total_hits * args.steps / elapsed,
            "roofline": roofline,
            "phases": phases,
            "sort_passes_P": P,
            "upload_and_encode_s": upload_s,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(si, bases, offsets, prm, fraction)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier(device_ids=[device]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
