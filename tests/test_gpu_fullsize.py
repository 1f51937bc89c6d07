"""The bench workload at its full size (1 000 loci, 10 000 nanopore reads, 1e8 bases) through the C ABI.
The oracle needs seconds per thousand reads, so the whole batch is checked through properties that do
not depend on size, and slices of it against the oracle itself."""
import numpy as np
import pytest

from pandora_amd import api, synth
from parity import compare_batch, gpu_index

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(gpu):
    si, bases, off, prm = synth.workload("nanopore_1k", seed=1)
    gidx = gpu_index(si)
    frac = api.fraction(prm["e_rate"], prm["k"])
    rd = api.Reads(bases, off)
    res = api.map_reads(gidx, rd, prm["w"], prm["k"], prm["max_diff"], frac, prm["min_cluster_size"])
    yield dict(si=si, bases=bases, off=off, prm=prm, gidx=gidx, frac=frac, res=res,
               hits=res.hits(), clusters=res.clusters(), counts=res.counts())
    res.close()
    rd.close()
    gidx.close()


def test_hit_order_and_cluster_invariants(world):
    h, c, cn, si, prm = world["hits"], world["clusters"], world["counts"], world["si"], world["prm"]
    assert cn["n_bases"] > 9e7 and cn["n_hits"] > 1e6 and cn["n_clusters"] > 1e4
    # MinimizerHit::operator< (src/minihit.cpp:44-66): read, prg, forward first, position (then path)
    assert h["prg"].max() < (1 << 11) and h["pos"].max() < (1 << 31) and h["read"].max() < (1 << 18)
    coarse = (h["read"].astype(np.int64) << 44) | (h["prg"].astype(np.int64) << 33) | \
             ((1 - h["fwd"].astype(np.int64)) << 32) | h["pos"].astype(np.int64)
    assert (np.diff(coarse) >= 0).all()
    # every hit is a record of the k-mer's key with the right orientation
    assert (h["prg"] == si.rec_prg[h["rec"]]).all()
    # clusters: disjoint runs of hits of one (read, prg, strand), gaps within max_diff, size above the threshold
    b, s = c["begin"].astype(np.int64), c["size"].astype(np.int64)
    assert (np.diff(b) > 0).all() and (b[:-1] + s[:-1] <= b[1:]).all() and b[-1] + s[-1] <= cn["n_hits"]
    assert (h["read"][b] == c["read"]).all() and (h["prg"][b] == c["prg"]).all() and (h["fwd"][b] == c["fwd"]).all()
    last = b + s - 1
    assert (h["read"][last] == c["read"]).all() and (h["prg"][last] == c["prg"]).all()
    assert (h["pos"][b] == c["first_pos"]).all() and (h["pos"][last] == c["last_pos"]).all()
    thr = np.maximum((np.minimum(si.min_path_length, 0xFFFFFFFF).astype(np.float32)
                      * np.float32(world["frac"])).astype(np.uint32), prm["min_cluster_size"])
    assert (c["size"] > thr[c["prg"]]).all()
    inside = np.ones(cn["n_hits"], bool)
    inside[b] = False                 # first hit of a kept cluster may follow any gap
    same = (h["read"][1:] == h["read"][:-1]) & (h["prg"][1:] == h["prg"][:-1]) & (h["fwd"][1:] == h["fwd"][:-1])
    gap = np.abs(h["pos"][1:].astype(np.int64) - h["pos"][:-1].astype(np.int64))
    member = np.zeros(cn["n_hits"] + 1, np.int64)
    np.add.at(member, b, 1)
    np.add.at(member, b + s, -1)
    member = np.cumsum(member)[:-1] > 0
    link = member[1:] & inside[1:]    # hit i+1 continues the kept cluster of hit i
    assert (same[link]).all() and (gap[link] <= prm["max_diff"]).all()


def test_batch_split_invariance(world):
    """Mapping the reads in three uneven batches gives the hits and clusters of the single batch."""
    off, bases, prm, gidx, frac = world["off"], world["bases"], world["prm"], world["gidx"], world["frac"]
    n = len(off) - 1
    cuts = [0, n // 7, n // 2 + 3, n]
    got_h = {f: [] for f in world["hits"]}
    got_c = {f: [] for f in world["clusters"]}
    base_hit = 0
    res = None
    for a, b in zip(cuts[:-1], cuts[1:]):
        rd = api.Reads(bases[int(off[a]):int(off[b])], off[a:b + 1] - off[a], first_read_id=a)
        res = api.map_reads(gidx, rd, prm["w"], prm["k"], prm["max_diff"], frac, prm["min_cluster_size"], result=res)
        h, c = res.hits(), res.clusters()
        for f in got_h:
            got_h[f].append(h[f])
        c["begin"] = c["begin"] + base_hit
        for f in got_c:
            got_c[f].append(c[f])
        base_hit += len(h["read"])
        rd.close()
    res.close()
    for f in got_h:
        np.testing.assert_array_equal(np.concatenate(got_h[f]), world["hits"][f], err_msg=f)
    for f in got_c:
        np.testing.assert_array_equal(np.concatenate(got_c[f]), world["clusters"][f], err_msg=f)


@pytest.mark.parametrize("first,count", [(0, 150), (4321, 150), (9850, 150)])
def test_slices_against_oracle(world, first, count):
    """Slices of the same reads, against the full-size index, stage by stage against the oracle; and the
    slice's rows inside the full-batch result."""
    off, bases, prm, si = world["off"], world["bases"], world["prm"], world["si"]
    sub_off = off[first:first + count + 1]
    sub = bases[int(sub_off[0]):int(sub_off[-1])]
    cn, orc = compare_batch(si, sub, sub_off - sub_off[0], prm["w"], prm["k"], prm["max_diff"], world["frac"],
                            prm["min_cluster_size"], first_read_id=first, gidx=world["gidx"])
    h = world["hits"]
    sel = (h["read"] >= first) & (h["read"] < first + count)
    assert sel.sum() == cn["n_hits"]
    np.testing.assert_array_equal(h["pos"][sel], orc["hit_pos"])
    np.testing.assert_array_equal(h["rec"][sel], orc["hit_rec"])
    np.testing.assert_array_equal(h["prg"][sel], orc["hit_prg"])


def test_big_index_slice_against_oracle(gpu):
    """20 000 loci: 2.8 M keys, a 128 MB table beyond the L2 caches and a presence filter with two bits per
    key.  A slice of the nanopore reads and a slice of 150-bp reads (w = 1: every k-mer is looked up) against
    the oracle on the same index."""
    si, bases, off, prm = synth.workload("nanopore_20k", seed=1, scale=0.02)
    gidx = gpu_index(si)
    info = gidx.info()
    assert info["n_keys"] > 2_000_000 and info["filter_bits_per_key"] == 2
    n = min(120, len(off) - 1)
    sub_off = off[:n + 1]
    cn, _ = compare_batch(si, bases[:int(sub_off[-1])], sub_off, prm["w"], prm["k"], prm["max_diff"],
                          api.fraction(prm["e_rate"], prm["k"]), prm["min_cluster_size"], gidx=gidx)
    assert cn["n_hits"] > 10000
    # short reads cut from the same ones, every k-mer a minimizer
    rng = np.random.default_rng(3)
    starts = rng.integers(0, int(sub_off[-1]) - 150, 3000)
    short = np.concatenate([bases[int(s):int(s) + 150] for s in starts])
    soff = np.arange(0, 150 * len(starts) + 1, 150, dtype=np.uint64)
    cn, _ = compare_batch(si, short, soff, 1, prm["k"], 31, api.fraction(0.001, prm["k"]), 10, expected_kmers=300,
                          gidx=gidx)
    assert cn["n_minimizers"] > 300000
    gidx.close()


def _check_full_size(name, slices, min_reads, min_minimizers, min_hits):
    """One bench workload at its full size: hit order, record and cluster invariants over the whole batch, the batch
    in three uneven pieces against the single batch, and slices of it stage by stage against the oracle."""
    si, bases, off, prm = synth.workload(name, seed=1)
    gidx = gpu_index(si)
    frac = api.fraction(prm["e_rate"], prm["k"])
    expected = 150 * 2 // prm["w"] if prm["illumina"] else 0xFFFFFFFF
    rd = api.Reads(bases, off)
    res = api.map_reads(gidx, rd, prm["w"], prm["k"], prm["max_diff"], frac, prm["min_cluster_size"], expected)
    cn, h, c = res.counts(), res.hits(), res.clusters()
    assert cn["n_reads"] >= min_reads and cn["n_minimizers"] > min_minimizers and cn["n_hits"] > min_hits
    assert h["prg"].max() < (1 << 19) and h["read"].max() < (1 << 19) and h["pos"].max() < (1 << 24)
    coarse = (h["read"].astype(np.int64) << 44) | (h["prg"].astype(np.int64) << 25) | \
             ((1 - h["fwd"].astype(np.int64)) << 24) | h["pos"].astype(np.int64)
    assert (np.diff(coarse) >= 0).all()
    assert (h["prg"] == si.rec_prg[h["rec"]]).all()
    b, s = c["begin"].astype(np.int64), c["size"].astype(np.int64)
    if len(b):
        assert (np.diff(b) > 0).all() and (b[:-1] + s[:-1] <= b[1:]).all() and b[-1] + s[-1] <= cn["n_hits"]
        last = b + s - 1
        for f in ("read", "prg", "fwd"):
            assert (h[f][b] == c[f]).all() and (h[f][last] == c[f]).all()
        thr = np.maximum((np.minimum(si.min_path_length, expected).astype(np.float32) * np.float32(frac)).astype(np.uint32),
                         prm["min_cluster_size"])
        assert (c["size"] > thr[c["prg"]]).all()
    # three uneven pieces with one result object
    n = len(off) - 1
    cuts = [0, n // 9, n // 2 + 5, n]
    pieces_h = {f: [] for f in h}
    pieces_c = {f: [] for f in c}
    base_hit, r2 = 0, None
    for a, e in zip(cuts[:-1], cuts[1:]):
        sub = api.Reads(bases[int(off[a]):int(off[e])], off[a:e + 1] - off[a], first_read_id=a)
        r2 = api.map_reads(gidx, sub, prm["w"], prm["k"], prm["max_diff"], frac, prm["min_cluster_size"], expected, result=r2)
        hh, cc = r2.hits(), r2.clusters()
        for f in pieces_h:
            pieces_h[f].append(hh[f])
        cc["begin"] = cc["begin"] + base_hit
        for f in pieces_c:
            pieces_c[f].append(cc[f])
        base_hit += len(hh["read"])
        sub.close()
    r2.close()
    for f in pieces_h:
        np.testing.assert_array_equal(np.concatenate(pieces_h[f]), h[f], err_msg=f)
    for f in pieces_c:
        np.testing.assert_array_equal(np.concatenate(pieces_c[f]), c[f], err_msg=f)
    # slices against the oracle, and their rows inside the full-batch result
    for first, count in slices:
        sub_off = off[first:first + count + 1]
        sub = bases[int(sub_off[0]):int(sub_off[-1])]
        cns, orc = compare_batch(si, sub, sub_off - sub_off[0], prm["w"], prm["k"], prm["max_diff"], frac,
                                 prm["min_cluster_size"], expected_kmers=expected, first_read_id=first, gidx=gidx)
        sel = (h["read"] >= first) & (h["read"] < first + count)
        assert sel.sum() == cns["n_hits"] > 10000
        np.testing.assert_array_equal(h["pos"][sel], orc["hit_pos"])
        np.testing.assert_array_equal(h["rec"][sel], orc["hit_rec"])
    res.close()
    rd.close()
    gidx.close()


def test_illumina_w1_full_size(gpu):
    """The benchmark's Illumina configuration at its full size: 200 000 reads of 150 bases, w = 1 (every k-mer is
    looked up), 13 M hits."""
    _check_full_size("illumina_w1", ((0, 1200), (123456, 1200)), 200000, 2.5e7, 1e7)


def test_illumina_20k_w1_full_size(gpu):
    """The benchmark's Illumina configuration on its 20 000-locus index, at full size: a w = 1 index
    of 22.6 M keys spread evenly over the key range (2 GB table, no presence filter), 400 000 reads of 150 bases,
    every k-mer looked up, 29 M hits; the reference's thresholds (src/map_main.cpp:267-272, src/utils.cpp:413-416)
    reject every cluster."""
    _check_full_size("illumina_20k_w1", ((0, 1200), (234567, 1200)), 400000, 5e7, 2.5e7)


def test_200k_locus_full_size(gpu):
    """The 200 000-locus index (2 GB table behind the key-range bitmap) with its full batch: 15 000 nanopore reads,
    7 M hits in 5 M clusters before the size threshold, reads of every hit-count class of the sort."""
    _check_full_size("nanopore_200k", ((0, 60), (9000, 60)), 15000, 1.5e7, 5e6)
