#!/usr/bin/env python3
"""Benchmark of the read -> PRG minimizer path (sketch + probe + sort + cluster) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the whole device path (pmap_map: sketch with the index lookup fused in, hit
emission, sort, cluster cut) over one batch of synthetic reads that is already resident in HBM (2-bit
packed).  At N > 1 every rank maps its own, equally sized batch against a replicated index (reads shard,
no collective on the data path); the value is the total number of read bases all ranks mapped per
second.  Rank 0 prints one JSON line.  At N = 1 the same run then measures the other workloads (dense
Illumina sketches, the 20 000- and 200 000-locus indexes, an index with several records per key) and
reports them under "other_workloads".
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
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 4    # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz

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

# kernels of each phase (names of pmap_kernel_name).  The index lookup of add_read_hits runs inside the
# sketch kernel, so "sketch+probe" is one kernel; "emit" is the construction of the hits.
# Sort, cluster cut and size threshold are one kernel ("tail", with "rows" behind it placing the rows of the kept
# clusters); "sort_rank", "cluster_cut" and "cluster_keep" are the separate kernels of PMAP_FLAG_TAIL_STAGED.
PHASES = {
    "sketch+probe": ["sketch", "probe"],
    "emit": ["scatter"],
    "sort+cluster": ["tail", "rows", "sort_rank", "cluster_cut", "cluster_keep"],
}
PMC_PHASES = ("sketch+probe", "emit", "sort", "cluster", "sort+cluster")      # keys of profiles/pmc_traffic.json

ALGO_DEF = ("B read bases, M minimizers, H hits, C clusters before the size threshold.  sketch+probe: B/4 "
            "(2-bit packed reads) + 16 M (a minimizer: u64 hash, u32 position, u32 read/strand) + 16 M (one 16-byte "
            "table slot per looked-up minimizer); emit: 36 H (16-byte record read, 20-byte hit written); sort+cluster: "
            "40 H (every 20-byte hit read once and written once) + 24 H + 28 C (what a separate pass over the sorted "
            "hits for the cluster cut would move: the fused kernel is priced as the two phases it replaces)")


def algorithmic_bytes(cn):
    B, M, H, C = cn["n_bases"], cn["n_minimizers"], cn["n_hits"], cn["n_clusters_all"]
    return {
        "sketch+probe": B / 4 + 16 * M + 16 * M,
        "emit": 36 * H,
        "sort+cluster": 40 * H + 24 * H + 28 * C,
    }


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


def rotated_batch(bases, offsets, shift):
    """The same reads starting with read `shift` (a different batch in memory with the same per-read results)."""
    n = len(offsets) - 1
    if shift % max(n, 1) == 0:
        return bases, offsets
    cut = int(offsets[shift % n])
    lens = np.diff(offsets.astype(np.int64))
    new_off = np.concatenate([[0], np.cumsum(np.roll(lens, -(shift % n)))]).astype(offsets.dtype)
    return np.concatenate([bases[cut:], bases[:cut]]), new_off


class Workload:
    """One synthetic workload resident on the GPU: index, `n_batches` batches of reads (the same reads in rotated
    order: steps take the batches in turn, so that what a step reads was last touched n_batches steps ago and the
    batches together exceed the 256 MB Infinity Cache), the call that maps one."""

    def __init__(self, api, synth, name, scale, shard, last_stage=0, n_batches=1):
        self.api, self.name = api, name
        self.si, self.bases, self.offsets, self.prm = synth.workload(name, seed=1, scale=scale, shard=shard)
        si, prm = self.si, self.prm
        self.fraction = api.fraction(prm["e_rate"], prm["k"])
        self.expected = 150 * 2 // prm["w"] if prm["illumina"] else 0xFFFFFFFF
        self.last_stage = last_stage
        self.gidx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                                     si.iv_start, si.iv_len, si.min_path_length)
        self.gidx.upload()
        t0 = time.perf_counter()
        self.reads = api.Reads(self.bases, self.offsets)
        self.upload_s = time.perf_counter() - t0
        n_reads = len(self.offsets) - 1
        self.batches = [self.reads]
        for j in range(1, max(1, n_batches)):
            b, o = rotated_batch(self.bases, self.offsets, j * n_reads // n_batches)
            self.batches.append(api.Reads(b, o))
        self.turn = 0

    def step(self, res, profile, wait=True):
        p = self.prm
        reads = self.batches[self.turn % len(self.batches)]
        self.turn += 1
        return self.api.map_reads(self.gidx, reads, p["w"], p["k"], p["max_diff"], self.fraction,
                                  p["min_cluster_size"], self.expected, profile=profile, result=res, wait=wait,
                                  last_stage=self.last_stage)

    def close(self):
        for r in self.batches:
            r.close()
        self.gidx.close()


def timed_steps(wl, steps, depth, prof, ring, before=None, after=None):
    """`steps` steps with `depth` of them in flight (pmap_map_async / pmap_map_wait; depth 1 = blocking calls).
    Returns (seconds, sum of the sketch kernel's event times in ms, last result)."""
    sketch_ms = 0.0
    inflight = [False] * depth

    def retire(j):
        nonlocal sketch_ms
        ring[j].wait()
        inflight[j] = False
        if prof:
            sketch_ms += ring[j].kernel_ms()["sketch"]

    if before:
        before()
    t0 = time.perf_counter()
    for i in range(steps):
        j = i % depth
        if inflight[j]:
            retire(j)
        ring[j] = wl.step(ring[j], prof, wait=False)
        inflight[j] = True
    for j in range(depth):
        if inflight[j]:
            retire(j)
    if after:
        after()
    elapsed = time.perf_counter() - t0
    return elapsed, sketch_ms, ring[(steps - 1) % depth] if steps else ring[0]


def upload_variants(wl, step_s):
    """What it costs to bring one batch to the device (copy + encode kernels, buffers in place): letters or 2-bit
    words packed on the host beforehand, from pageable or page-locked host memory.  Each is the best of three
    pmap_reads_update* calls on a reads object of its own; bases_per_s adds one step of the timed region."""
    import ctypes as C

    import numpy as np
    import torch

    from pandora_amd import _lib
    lib = _lib.load()
    u64p, u32p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    bases = np.ascontiguousarray(wl.bases, np.uint8)
    offsets = np.ascontiguousarray(wl.offsets, np.uint64)
    n, n_bases = len(offsets) - 1, int(offsets[-1] - offsets[0])
    lens = np.diff(offsets.astype(np.int64))
    words = np.zeros(int(((lens + 15) // 16).sum()) + 1, np.uint32)
    ok = np.zeros(n + 1, np.uint8)
    t0 = time.perf_counter()
    if lib.pmap_pack_reads(bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(u64p), n, words.ctypes.data_as(u32p),
                           ok.ctypes.data_as(u8p)):
        return None
    pack_s = time.perf_counter() - t0

    def pinned(a):
        t = torch.empty(a.shape, dtype=torch.uint8 if a.dtype == np.uint8 else torch.int32, pin_memory=True)
        v = t.numpy().view(a.dtype)
        v[...] = a
        return t, v

    keep = []
    out = {"host_pack_s_one_thread": pack_s}
    for name, packed, pin in (("letters_pageable", False, False), ("letters_pinned", False, True),
                              ("packed_pageable", True, False), ("packed_pinned", True, True)):
        src = words if packed else bases
        if pin:
            t, src = pinned(src)
            keep.append(t)
        h = C.c_void_p()
        if packed:
            args = (src.ctypes.data_as(u32p), offsets.ctypes.data_as(u64p), ok.ctypes.data_as(u8p), n, 0)
            rc = lib.pmap_reads_create_packed(*args, C.byref(h))
            update = lambda: lib.pmap_reads_update_packed(h, *args)
        else:
            args = (src.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(u64p), n, 0)
            rc = lib.pmap_reads_create(*args, C.byref(h))
            update = lambda: lib.pmap_reads_update(h, *args)
        if rc:
            return None
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            if update():
                return None
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        lib.pmap_reads_destroy(h)
        out[name] = {"upload_s": best, "bases_per_s": n_bases / (best + step_s)}
    return out


def kernel_table(wl, res, steps):
    """Per-kernel milliseconds per step from an untimed pass with events on every kernel, one step at a time
    (each kernel with the machine to itself)."""
    kernel_ms = {}
    for _ in range(steps):
        res = wl.step(res, 1)
        for name, ms in res.kernel_ms().items():
            kernel_ms[name] = kernel_ms.get(name, 0.0) + ms
    return {k: v / max(steps, 1) for k, v in kernel_ms.items()}, res


def phase_table(kernel_ms, algo):
    phases = {}
    for ph, ks in PHASES.items():
        ms = sum(kernel_ms.get(k, 0.0) for k in ks)
        gbps = (algo[ph] / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
        phases[ph] = {"ms": ms, "algorithmic_bytes": algo[ph], "GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS,
                      "kernels": {k: kernel_ms.get(k, 0.0) for k in ks if kernel_ms.get(k, 0.0) > 0}}
        # a phase cannot move its algorithmic bytes faster than HBM delivers them unless they came from a cache
        if gbps > HBM_PEAK_GBS:
            phases[ph]["note"] = "above the HBM peak: this phase's working set is served by the caches on this workload"
    return phases


def other_workload(api, synth, name, scale, depth, budget_s, n_batches=1):
    """Throughput and per-kernel table of one more workload (N = 1 only)."""
    t0 = time.perf_counter()
    wl = Workload(api, synth, name, scale, 0, n_batches=n_batches)
    ring = [None] * depth
    for i in range(max(3, depth)):
        ring[i % depth] = wl.step(ring[i % depth], 0)
    one, _, res = timed_steps(wl, depth, depth, 0, ring)
    steps = int(max(20, min(400, budget_s / max(one / depth, 1e-5))))
    elapsed, _, res = timed_steps(wl, steps, depth, 0, ring)
    cn = res.counts()
    kernel_ms, res = kernel_table(wl, res, 10)
    algo = algorithmic_bytes(cn)
    info = wl.gidx.info()
    sk = kernel_ms.get("sketch", 0.0)
    out = {
        "workload": "%s: %d loci, %d keys, %d records, table %.0f MB, filter %s; %d reads, w=%d k=%d"
                    % (name, wl.si.n_prg, info["n_keys"], info["n_recs"], info["table_slots"] * 32 / 1e6,
                       "key-range bitmap" if info["filter_direct"] else "bits/key %d" % info["filter_bits_per_key"],
                       cn["n_reads"], wl.prm["w"], wl.prm["k"]),
        "bases_per_s": cn["n_bases"] * steps / elapsed, "hits_per_s": cn["n_hits"] * steps / elapsed,
        "ms_per_step": elapsed / steps * 1e3, "steps": steps, "steps_in_flight": depth, "resident_batches": n_batches,
        "bases": cn["n_bases"], "minimizers": cn["n_minimizers"], "hits": cn["n_hits"],
        "clusters_all": cn["n_clusters_all"], "clusters_kept": cn["n_clusters"],
        # batches that were mapped again with the single-workgroup cluster kernels (PMAP_FLAG_TAIL_STAGED only)
        "serial_reruns": int(cn["ms"][5]),
        "kernel_ms_alone": kernel_ms, "phases": phase_table(kernel_ms, algo),
        # hash-probe bandwidth: minimizers looked up per second of the kernel that does the lookups
        "lookups_per_s": (cn["n_minimizers"] / (sk * 1e-3)) if sk > 0 else 0.0,
        "setup_s": None,
    }
    for r in ring:
        if r is not None:
            r.close()
    wl.close()
    out["setup_s"] = time.perf_counter() - t0 - elapsed
    out.update(step_traffic(name, out["ms_per_step"]))
    return out


def step_traffic(name, ms_per_step):
    """HBM-side bytes of one step of a workload (all phases) from the committed PMC passes, against this run's step
    time: what the memory system moves per second while the workload runs, wasted parts of 128-byte lines included."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(pmc)).get(name)
    except Exception:
        rec = None
    if not rec or ms_per_step <= 0:
        return {}
    total = sum(rec.get(p, 0.0) for p in PMC_PHASES)
    return {"step_traffic": {"bytes_per_step": total, "GBps": total / (ms_per_step * 1e-3) / 1e9,
                             "frac_of_hbm_peak": total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "source": "not measured in this run: " + str(rec.get("source"))}}


def file_mode(args):
    """--mode file: the drop-in entry point, pmap_pangraph_from_read_file (what src/map_main.cpp:319 and
    src/compare_main.cpp:325 would call), from a FASTQ file on disk (page cache warm) to the pangraph summary, in ONE
    process over `--gpus` mapping contexts (pmap_init_devices): contexts 0..N-1 on GPUs 0..N-1 when the machine has
    that many, else all of them on GPU 0 (a rehearsal of the threads and queues, not of N GPUs).  Not the headline
    metric: the host parses, packs and uploads inside the timed region."""
    import tempfile
    import torch
    import pandora_amd
    from pandora_amd import api, synth
    n_dev = max(torch.cuda.device_count(), 1)
    devices = list(range(args.gpus)) if args.gpus <= n_dev else [0] * args.gpus
    pandora_amd.init(0)
    if args.gpus > 1 or args.file_contexts:
        devices = devices * max(args.file_contexts, 1)
        api.init_devices(devices)
    si, bases, off, prm = synth.workload(args.workload, seed=1, scale=args.scale)
    d = tempfile.mkdtemp(prefix="pmap_bench_file_")
    path = os.path.join(d, "reads.fq")
    one = b"".join(b"@r%d\n" % i + bases[int(off[i]):int(off[i + 1])].tobytes() + b"\n+\n" +
                   b"I" * int(off[i + 1] - off[i]) + b"\n" for i in range(len(off) - 1))
    with open(path, "wb") as f:
        for _ in range(args.file_copies):
            f.write(one)
    size = os.path.getsize(path)
    gidx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                            si.iv_start, si.iv_len, si.min_path_length)
    for dev in sorted(set(devices)):        # the table on every GPU before the clock starts
        pandora_amd.init(dev)
        gidx.upload()
    pandora_amd.init(0)
    n_bases = int(off[-1]) * args.file_copies
    os.environ["PMAP_TRACE_PIPELINE"] = "1"
    runs = []
    for rep in range(max(args.steps, 1) if args.steps < 50 else 4):
        t0 = time.perf_counter()
        pg = api.pangraph_from_read_file(path, gidx, prm["w"], prm["k"], prm["max_diff"], prm["e_rate"],
                                         prm["min_cluster_size"], genome_size=5000000, max_covg=1000000,
                                         illumina=prm["illumina"], keep_hits=args.file_keep_hits, coverage=True)
        dt = time.perf_counter() - t0
        info = pg.info()
        runs.append({"s": dt, "bases_per_s": n_bases / dt, "add_node_calls": info["n_add_node_calls"],
                     "hits_kept": info["n_hits"]})
        pg.close()
    os.remove(path)
    os.rmdir(d)
    steady = sorted(r["bases_per_s"] for r in runs[1:] or runs)
    return {"metric": "read bases mapped/sec (file on disk -> pangraph, pmap_pangraph_from_read_file)",
            "value": steady[len(steady) // 2], "unit": "bases/s", "n_gpus": args.gpus, "higher_is_better": True,
            "data": "synthetic", "dtype": "u32",
            "config": {"workload": "%s as a FASTQ file of %d bytes (%d bases, %d reads), %d mapping contexts on devices %s"
                                   % (args.workload, size, n_bases, (len(off) - 1) * args.file_copies, len(devices), devices),
                       "keep_hits": args.file_keep_hits},
            "runs": runs, "note": "first run includes making the pinned buffers; value = median of the others; the "
                                   "per-thread wait times of every run are on stderr (PMAP_TRACE_PIPELINE)"}


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
    ap.add_argument("--others", default="default",
                    help="comma-separated workloads measured after the headline one at N = 1 ('none' = skip; default: "
                         "nanopore_1k_linear, nanopore_1k_families, illumina_1k, illumina_w1, nanopore_20k, "
                         "illumina_20k_w1, nanopore_200k, nanopore_1k_8g while the time budget lasts)")
    ap.add_argument("--others-budget", type=float, default=240.0, help="seconds the other workloads may take in all")
    ap.add_argument("--resident-batches", type=int, default=6,
                    help="batches of reads kept on the GPU and mapped in turn (the same reads in rotated order): with "
                         "one, a step re-reads what the step before left in the 256 MB Infinity Cache")
    ap.add_argument("--depth", type=int, default=3,
                    help="steps in flight (1 = blocking pmap_map); with 3 both sketch streams always hold a kernel "
                         "while the host waits for a result, and the tail kernels of a step run next to the sketch "
                         "kernels of the following ones")
    ap.add_argument("--last-stage", type=int, default=0,
                    help="stop the path after this stage (1 sketch, 2 hits, 3 sort, 4 clusters = all; 0 = all): for "
                         "experiments, the reported value is then not the metric")
    ap.add_argument("--mode", default="batch", choices=["batch", "file"],
                    help="batch (default): the metric, resident batches through pmap_map; file: the drop-in call "
                         "pmap_pangraph_from_read_file on a FASTQ file, one process over --gpus mapping contexts")
    ap.add_argument("--file-copies", type=int, default=40, help="--mode file: the workload's reads this many times over")
    ap.add_argument("--file-contexts", type=int, default=0, help="--mode file: mapping contexts per GPU (0 = the library's default for one GPU, one per GPU otherwise)")
    ap.add_argument("--file-keep-hits", action="store_true", help="--mode file: keep Read::hits")
    ap.add_argument("--alone-steps", type=int, default=200,
                    help="steps of the second, untimed pass that times every kernel with the machine to itself (at least 1; "
                         "1 under a kernel trace, so that the trace's averages are those of the timed region)")
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

    if args.mode == "file":
        if int(os.environ.get("RANK", "0")) == 0:       # one process drives every GPU
            out = file_mode(args)
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        return

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

    def progress(msg):     # a big workload takes a while to synthesise: say so on stderr
        if rank == 0:
            print("[bench] %s (%.0f s)" % (msg, time.perf_counter() - t_start), file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    progress("synthesising workload %s" % args.workload)
    # same index on every rank (replicated), a different shard of reads per rank
    wl = Workload(api, synth, args.workload, args.scale, rank, args.last_stage, n_batches=args.resident_batches)
    si, prm = wl.si, wl.prm
    progress("index (%d keys, %d records) and %d bases on the device" % (len(si.keys), len(si.rec_prg), len(wl.bases)))

    # `depth` result objects take turns, so that the host enqueues a step while the GPU runs the one before
    depth = max(1, args.depth)
    ring = [None] * depth
    for i in range(max(args.warmup, depth)):
        ring[i % depth] = wl.step(ring[i % depth], 0)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[device]) if backend == "nccl" else dist.barrier()

    # profile=2: one pair of HIP events rides on the dispatch of the sketch kernel (the dominant one) on the
    # library's stream, inside the timed region
    prof = 0 if args.no_kernel_events else 2
    elapsed, sketch_ms, res = timed_steps(wl, args.steps, depth, prof, ring, before=sync, after=sync)
    elapsed_local = elapsed
    cn = res.counts()
    # second, untimed pass with events on every kernel, for the per-phase table: one step at a time, every
    # kernel with the machine to itself (in the timed region the sketch kernels of two steps and the tail
    # kernels of a third share the CUs, so a launch lasts longer there)
    kernel_ms, res = kernel_table(wl, res, max(1, min(args.steps, args.alone_steps)))
    sketch_alone_ms = kernel_ms.get("sketch", 0.0)

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
        algo = algorithmic_bytes(cn)
        phases = phase_table(kernel_ms, algo)
        dom = max(phases, key=lambda p: phases[p]["ms"])
        # the dominant kernel's duration inside the timed region (HIP events on its dispatches) when it is the
        # sketch kernel and events were on; otherwise its duration alone
        dom_ms = sketch_ms / steps if (dom == "sketch+probe" and prof) else phases[dom]["ms"]
        achieved = algo[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, valu, pmc_src = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(args.workload, {})
                traffic = rec.get(dom)
                pmc_src = rec.get("source")
                # the sketch kernel is bounded by integer VALU issue: wavefront instructions per launch (PMC
                # SQ_INSTS_VALU, same workload) against the issue rate of the whole chip
                if dom == "sketch+probe" and rec.get("sketch_valu_insts") and args.scale == 1.0:
                    valu = {"wave_insts_per_launch": rec["sketch_valu_insts"], "peak_insts_per_s": VALU_PEAK_WAVE_INSTS,
                            "frac": rec["sketch_valu_insts"] / (dom_ms * 1e-3) / VALU_PEAK_WAVE_INSTS,
                            "note": "VALU issue roofline at the nominal 2.4 GHz (256 CUs x 4 SIMDs, 4 cycles per wavefront "
                                    "instruction), the kernel's binding limit; instruction count from the committed PMC "
                                    "pass, duration from this run; frac_alone (the kernel with the machine to itself) "
                                    "comes out a few percent above 1 when the engine clock is above nominal"}
            except Exception:
                traffic, valu = None, None
        roofline = {"bound": "hbm", "limited_by": "valu" if dom == "sketch+probe" else "latency",
                    "kernel": "%s (%s)" % (dom, "+".join(k for k in PHASES[dom] if kernel_ms.get(k, 0.0) > 0)),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic,
                    "traffic_source": ("not measured in this run: %s" % pmc_src) if traffic is not None else None,
                    "ms_per_launch": dom_ms, "algorithmic_bytes_per_launch": algo[dom],
                    "algorithmic_bytes_definition": ALGO_DEF, "valu": valu}
        if dom == "sketch+probe" and sketch_alone_ms > 0:
            roofline["ms_per_launch_alone"] = sketch_alone_ms
            roofline["achieved_alone"] = algo[dom] / (sketch_alone_ms * 1e-3) / 1e9
            if prof:
                # the library alternates the sketch kernels of successive steps between two streams: launches
                # overlap, each lasts longer than it would alone, and together they move this many bytes per second
                roofline["launches_in_flight"] = sketch_ms / (elapsed_local * 1e3)
                roofline["achieved_all_launches"] = algo[dom] * args.steps / elapsed_local / 1e9
            if valu:
                valu["frac_alone"] = valu["wave_insts_per_launch"] / (sketch_alone_ms * 1e-3) / VALU_PEAK_WAVE_INSTS
                # a kernel far below the VALU issue rate with the machine to itself waits for its index lookups:
                # requests beyond the L2, which the chip serves at about 55 G per second whatever their size
                if valu["frac_alone"] < 0.6:
                    roofline["limited_by"] = "memory requests beyond the L2 (index lookups)"
        total_algo = sum(algo.values())
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
                       "index_intervals": int(len(si.iv_start)),
                       "parallelism": "reads sharded over %d GPU(s), index replicated" % world,
                       "steps_in_flight": depth, "resident_batches": args.resident_batches},
            "hits_per_s": total_hits * args.steps / elapsed,
            "serial_reruns": int(cn["ms"][5]),
            "roofline": roofline,
            "phases": phases,
            "whole_step": {"algorithmic_bytes": total_algo,
                           "GBps": total_algo / (elapsed_local / steps) / 1e9,
                           "frac_of_hbm_peak": total_algo / (elapsed_local / steps) / 1e9 / HBM_PEAK_GBS},
            **(step_traffic(args.workload, elapsed_local / steps * 1e3) if args.scale == 1.0 else {}),
            "upload_and_encode_s": wl.upload_s,
            "end_to_end": {"note": "one batch from pageable host memory as ASCII: upload + encode + one step (the first "
                                   "upload of the process, device buffers made on the way); variants: the same with "
                                   "the buffers in place, letters or words packed on the host, pageable or pinned",
                           "bases_per_s": cn["n_bases"] / (wl.upload_s + elapsed_local / steps)},
        }
        if world == 1:
            out["end_to_end"]["variants"] = upload_variants(wl, elapsed_local / steps)
        if world == 1 and not args.no_cpu_baseline:
            progress("cpu baseline")
            out["cpu_baseline"] = cpu_baseline(si, wl.bases, wl.offsets, prm, wl.fraction)
        else:
            out["cpu_baseline"] = None
        names = []
        if world == 1 and args.others != "none" and args.last_stage == 0:
            names = ["nanopore_1k_linear", "nanopore_1k_families", "illumina_1k", "illumina_w1", "nanopore_20k",
                     "illumina_20k_w1", "nanopore_200k", "nanopore_1k_8g"] if args.others == "default" else args.others.split(",")
        for r in ring:
            if r is not None:
                r.close()
        res.close()
        wl.close()
        others, t_others = {}, time.perf_counter()
        for name in names:
            if name == args.workload:
                continue
            if time.perf_counter() - t_others > args.others_budget:
                others[name] = "skipped: the time budget of the other workloads was used up"
                continue
            progress("other workload %s" % name)
            try:
                others[name] = other_workload(api, synth, name, 1.0, depth, 0.5, args.resident_batches)
            except Exception as e:      # a side measurement must not lose the headline line
                others[name] = "failed: %r" % (e,)
        if names:
            out["other_workloads"] = others
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier(device_ids=[device]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
