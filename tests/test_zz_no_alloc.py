"""pmap_map_noalloc, the entry point that allocates nothing (its scratch memory is an array inside the
library's device image), through the C ABI against pmap_map on the same batches: two calls in a row,
the second with a smaller batch, so that whatever the first call left in the shared array must not leak
into the second one's results."""
import ctypes as C

import numpy as np
import pytest

from pandora_amd import _lib, api, synth
from parity import gpu_index

pytestmark = pytest.mark.gpu


class Hip:
    """The caller's side of the entry point: device buffers from the HIP runtime that libpandora_amd.so itself
    is linked to (found in the process' own mappings, so that no second runtime gets loaded)."""

    def __init__(self):
        _lib.load()
        path = None
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
        assert path, "libpandora_amd.so did not bring in libamdhip64"
        self.rt = C.CDLL(path)
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def alloc(self, n_words):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), 4 * n_words) == 0
        assert self.rt.hipMemset(p, 0x5A, 4 * n_words) == 0
        assert self.rt.hipDeviceSynchronize() == 0      # the library's streams do not wait for the null stream
        return p

    def fetch(self, p, n_words):
        a = np.zeros(n_words, np.uint32)
        assert self.rt.hipMemcpy(a.ctypes.data_as(C.c_void_p), p, 4 * n_words, 2) == 0      # hipMemcpyDeviceToHost
        return a

    def free(self, p):
        assert self.rt.hipFree(p) == 0


def _noalloc(hip, gidx, rd, prm, hit_cap, cl_cap):
    hits = [hip.alloc(hit_cap) for _ in range(5)]
    rows = hip.alloc(cl_cap * 8)
    out = _lib.DevOut(hit_cap, *hits, cl_cap, rows)
    cn = _lib.Counts()
    rc = _lib.load().pmap_map_noalloc(gidx._h, rd._h, C.byref(prm), C.byref(out), C.byref(cn))
    host = [hip.fetch(h, hit_cap) for h in hits]
    host_rows = hip.fetch(rows, cl_cap * 8).reshape(-1, 8)
    for h in hits + [rows]:
        hip.free(h)
    return rc, cn, host, host_rows


def _check_against_pmap_map(hip, gidx, rd, first_id, prm_args, hit_cap=65536, cl_cap=8192):
    w, k, max_diff, frac, mcs = prm_args
    ref = api.map_reads(gidx, rd, w, k, max_diff, frac, mcs)
    rc_, cn_ = ref.counts(), None
    prm = _lib.Params(w, k, max_diff, frac, mcs, 0xFFFFFFFF, 0, 0, 0)
    rc, cn, (h_read, h_key, h_pos, h_rank, h_rec), rows = _noalloc(hip, gidx, rd, prm, hit_cap, cl_cap)
    assert rc == 0, api._lib.load().pmap_last_error()
    for f in ("n_reads", "n_bases", "n_minimizers", "n_hits", "n_clusters_all", "n_clusters"):
        assert getattr(cn, f) == rc_[f], f
    n = cn.n_hits
    want = ref.hits(sorted=True)
    np.testing.assert_array_equal(h_read[:n] + first_id, want["read"])
    np.testing.assert_array_equal(h_key[:n] >> 1, want["prg"])
    np.testing.assert_array_equal(1 - (h_key[:n] & 1), want["fwd"])
    np.testing.assert_array_equal(h_pos[:n], want["pos"])
    np.testing.assert_array_equal(h_rec[:n], want["rec"])
    np.testing.assert_array_equal(h_rank[:n], gidx.records()["path_rank"][want["rec"]])
    c = ref.clusters()
    m = cn.n_clusters
    np.testing.assert_array_equal(rows[:m, 0], c["begin"])
    np.testing.assert_array_equal(rows[:m, 1], c["size"])
    np.testing.assert_array_equal(rows[:m, 2] + first_id, c["read"])
    np.testing.assert_array_equal(rows[:m, 3] >> 1, c["prg"])
    np.testing.assert_array_equal(1 - (rows[:m, 3] & 1), c["fwd"])
    np.testing.assert_array_equal(rows[:m, 4], c["first_pos"])
    np.testing.assert_array_equal(rows[:m, 5], c["last_pos"])
    np.testing.assert_array_equal(rows[:m, 6], c["first_rank"])
    ref.close()
    return cn


def test_noalloc_matches_pmap_map_twice(gpu):
    hip = Hip()
    rng = np.random.default_rng(99)
    loci = synth.make_loci(30, rng, dup_frac=0.1)
    si = synth.build_index(loci, 14, 15, rng=rng, split_paths=0.2)
    genome = synth.make_genome(loci, rng)
    bases, off = synth.simulate_reads(genome, 90, rng, mean_len=3000, e_rate=0.08, n_frac=0.03, short_frac=0.03)
    gidx = gpu_index(si)
    gidx.upload()
    prm_args = (14, 15, 250, api.fraction(0.11, 15), 5)
    big = api.Reads(bases, off, first_read_id=7)
    cn1 = _check_against_pmap_map(hip, gidx, big, 7, prm_args)
    assert cn1.n_hits > 2000 and cn1.n_clusters > 10
    # the second call: a smaller batch over the same, now dirty, scratch array
    a, b = 20, 45
    small = api.Reads(bases[int(off[a]):int(off[b])], off[a:b + 1] - off[a], first_read_id=1000)
    cn2 = _check_against_pmap_map(hip, gidx, small, 1000, prm_args)
    assert 0 < cn2.n_hits < cn1.n_hits and 0 < cn2.n_clusters < cn1.n_clusters
    # other thresholds on the same index, and the general sketch kernel (another tile geometry)
    _check_against_pmap_map(hip, gidx, small, 1000, (14, 15, 40, api.fraction(0.01, 15), 2))
    _check_against_pmap_map(hip, gidx, big, 7, prm_args)
    si9 = synth.build_index(loci, 19, 15)
    g9 = gpu_index(si9)
    g9.upload()
    _check_against_pmap_map(hip, g9, small, 1000, (19, 15, 250, api.fraction(0.11, 15), 3))
    g9.close()
    gidx.close()


def test_noalloc_limits_and_errors(gpu):
    hip = Hip()
    rng = np.random.default_rng(100)
    loci = synth.make_loci(5, rng)
    si = synth.build_index(loci, 14, 15)
    gidx = gpu_index(si)
    lib = _lib.load()
    prm = _lib.Params(14, 15, 250, api.fraction(0.11, 15), 5, 0xFFFFFFFF, 0, 0, 0)
    rd = api.Reads.from_list([bytes(loci[0]) * 3, bytes(loci[1])])
    # the index has not been uploaded: refused, nothing is allocated on the caller's behalf
    rc, _, _, _ = _noalloc(hip, gidx, rd, prm, 8192, 1024)
    assert rc == 1 and b"pmap_index_upload_now" in lib.pmap_last_error()
    gidx.upload()
    rc, cn, _, _ = _noalloc(hip, gidx, rd, prm, 8192, 1024)
    assert rc == 0 and cn.n_hits > 100
    # output buffers below one block of the cluster kernels
    rc, _, _, _ = _noalloc(hip, gidx, rd, prm, 4096, 1024)
    assert rc == 1
    # a batch beyond the scratch array: a clear error, not a fault
    huge = api.Reads.from_list([bytes(synth.random_dna(1_200_000, rng))])
    rc, _, _, _ = _noalloc(hip, gidx, huge, prm, 8192, 1024)
    assert rc == 4 and b"too large" in lib.pmap_last_error()
    # more hits than the caller's buffers hold: PMAP_ELIMIT, and the counts say how many there are
    many = api.Reads.from_list([bytes(l) * 12 for l in loci] * 6)
    rc, cn, _, _ = _noalloc(hip, gidx, many, prm, 8192, 1024)
    assert rc == 4 and cn.n_hits > 8192
    rc, cn2, _, _ = _noalloc(hip, gidx, many, prm, 65536, 1024)
    assert rc == 0 and cn2.n_hits == cn.n_hits
    # an empty batch
    rc, cn, _, _ = _noalloc(hip, gidx, api.Reads.from_list([]), prm, 8192, 1024)
    assert rc == 0 and cn.n_hits == 0 and cn.n_clusters == 0
    # unfused mode / partial pipelines belong to pmap_map
    rc, _, _, _ = _noalloc(hip, gidx, rd, _lib.Params(14, 15, 250, 0.1, 5, 0xFFFFFFFF, 0, 0, 1), 8192, 1024)
    assert rc == 1
    gidx.close()
