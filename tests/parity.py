"""Shared helper: run one batch through the GPU path and through the oracle and compare every
stage bit for bit (minimizers, sorted hits, kept clusters, filter_clusters survivors).  Both device
modes are checked: the default one (index lookup fused into the sketch kernel) and the unfused one
that keeps the complete minimizer list."""
import numpy as np

import oracle_lib as O
from pandora_amd import api

UINT32_MAX = 0xFFFFFFFF
TAIL_STAGED, TAIL_NARROW = 128, 256      # PMAP_FLAG_TAIL_STAGED, PMAP_FLAG_TEST_TAIL_NARROW


def gpu_index(si):
    return api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                            si.iv_start, si.iv_len, si.min_path_length)


def oracle_index(si):
    return O.FlatIndex(si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off, si.iv_start,
                       si.iv_len)


def _check_hits_and_clusters(res, cn, orc):
    # MinimizerHits in MinimizerHit::operator< order
    gh = res.hits(sorted=True)
    assert cn["n_hits"] == len(orc["hit_read"])
    for a, b in (("read", "hit_read"), ("prg", "hit_prg"), ("fwd", "hit_fwd"), ("pos", "hit_pos"),
                 ("rec", "hit_rec")):
        np.testing.assert_array_equal(gh[a], orc[b], err_msg=a)
    # clusters kept by define_clusters, then clusterComp order and filter_clusters
    assert cn["n_clusters_all"] == orc["n_all_clusters"]
    gc = res.clusters()
    assert cn["n_clusters"] == len(orc["cl_size"])
    order, alive = api.filter_clusters(gc)
    o = order.astype(np.int64)
    np.testing.assert_array_equal(gc["begin"][o], orc["cl_begin"])
    np.testing.assert_array_equal(gc["size"][o], orc["cl_size"])
    np.testing.assert_array_equal(gc["read"][o], orc["cl_read"])
    np.testing.assert_array_equal(gc["prg"][o], orc["cl_prg"])
    np.testing.assert_array_equal(gc["fwd"][o], orc["cl_fwd"])
    np.testing.assert_array_equal(gc["first_pos"][o], orc["cl_first_pos"])
    np.testing.assert_array_equal(gc["last_pos"][o], orc["cl_last_pos"])
    np.testing.assert_array_equal(alive, orc["cl_alive"])


def compare_batch(si, bases, offsets, w, k, max_diff, fraction, min_cluster_size=10, expected_kmers=UINT32_MAX,
                  first_read_id=0, gidx=None):
    """Asserts GPU == oracle for one batch; returns (gpu counts, oracle result dict)."""
    offsets = np.ascontiguousarray(offsets, np.uint64)
    n_reads = len(offsets) - 1
    own = gidx is None
    if own:
        gidx = gpu_index(si)
    rd = api.Reads(bases, offsets, first_read_id)
    ids = np.arange(first_read_id, first_read_id + n_reads, dtype=np.uint32)
    orc = O.oracle_map(oracle_index(si), bases, offsets, ids, w, k, max_diff, fraction, min_cluster_size,
                       expected_kmers, si.min_path_length)
    assert orc["n_dup_hits"] == 0

    # the default sketch kernel and, where (w, k) has more than one, the others
    variants = (None,)
    if k <= 15 and 1 <= w <= 16:
        variants = (None, "np8", "lds") if w == 14 else (None, "lds")
    for sketch in variants:
        # unfused: the complete minimizer set per read (Seq::sketch), then everything downstream
        # (rank-counting sort for every read: PMAP_FLAG_SORT_NO_WAVE)
        res = api.map_reads(gidx, rd, w, k, max_diff, fraction, min_cluster_size, expected_kmers, unfused=True,
                            sketch=sketch, flags=32 | TAIL_STAGED)
        cn = res.counts()
        assert cn["n_minimizers"] == len(orc["mini_kmer"]), sketch
        gm = res.minimizers()
        so = orc["sketch_off"]
        oread = np.repeat(ids, np.diff(so).astype(np.int64))
        np.testing.assert_array_equal(gm["read"], oread)
        np.testing.assert_array_equal(gm["pos"], orc["mini_pos"])
        np.testing.assert_array_equal(gm["kmer"], orc["mini_kmer"])
        np.testing.assert_array_equal(gm["strand"], orc["mini_strand"])
        _check_hits_and_clusters(res, cn, orc)
        res.close()

        # default (fused) mode; the sort variant for very wide keys takes turns with the usual one, and so does
        # the wavefront sort for every read it can take (PMAP_FLAG_TEST_SORT_WAVE_ALL) with the default split
        # between it and the rank counting
        res = api.map_reads(gidx, rd, w, k, max_diff, fraction, min_cluster_size, expected_kmers, sketch=sketch,
                            sort_wide=sketch is None, flags=TAIL_STAGED | (64 if sketch is None else 0))
        cn = res.counts()
        assert cn["n_minimizers"] == len(orc["mini_kmer"]), sketch
        _check_hits_and_clusters(res, cn, orc)
        res.close()

        # the default tail: sort, cluster cut and size threshold in one kernel; with another sketch kernel, the
        # reads of a chunk sorted two at a time (PMAP_FLAG_TEST_TAIL_NARROW)
        res = api.map_reads(gidx, rd, w, k, max_diff, fraction, min_cluster_size, expected_kmers, sketch=sketch,
                            flags=0 if sketch is None else TAIL_NARROW)
        cn = res.counts()
        assert cn["n_minimizers"] == len(orc["mini_kmer"]), sketch
        _check_hits_and_clusters(res, cn, orc)
        res.close()
    rd.close()
    # the same batch packed to 2 bits per base on the host (what pangraph_from_read_file sends)
    rdp = api.Reads.packed(bases, offsets, first_read_id)
    res = api.map_reads(gidx, rdp, w, k, max_diff, fraction, min_cluster_size, expected_kmers)
    cn = res.counts()
    assert cn["n_minimizers"] == len(orc["mini_kmer"])
    _check_hits_and_clusters(res, cn, orc)
    res.close()
    rdp.close()
    if own:
        gidx.close()
    return cn, orc
