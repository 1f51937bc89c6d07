"""GPU parity on the index shapes and inputs the benchmark configurations name: PRGs with variant sites
(multi-interval paths) and gene families (several records per key), a 200 000-locus index whose table is
beyond every cache (with the key-range bitmap in front of it and with no filter at all), the real reads the reference ships, and an index that
arrives as the reference's `.idx` text and as its binary cache."""
import os

import numpy as np
import pytest

import oracle_lib as O
import test_gpu_pangraph as TP
from pandora_amd import api, synth
from parity import compare_batch, gpu_index

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(__file__), "golden", "ref_fixtures")


def _variant_world(seed, n_loci, n_reads, w=14, k=15, family_frac=0.3, **sim):
    rng = np.random.default_rng(seed)
    loci = synth.make_loci(n_loci, rng, dup_frac=0.05, family_frac=family_frac, family_max=9)
    vloci = synth.make_variant_loci(loci, rng)
    si = synth.build_variant_index(vloci, w, k)
    genome = synth.make_variant_genome(vloci, rng)
    bases, off = synth.simulate_reads(genome, n_reads, rng, **sim)
    return si, bases, off


def test_variant_sites_and_gene_families(gpu):
    si, bases, off = _variant_world(41, 60, 80, mean_len=3000, e_rate=0.06, n_frac=0.03, short_frac=0.03)
    n_iv = np.diff(si.path_off.astype(np.int64))
    per_key = np.diff(si.rec_off.astype(np.int64))
    assert (n_iv == 3).sum() > 500 and (n_iv == 2).sum() > 50 and (si.iv_len == 0).sum() > 10
    assert per_key.max() >= 8 and len(si.rec_prg) > 1.25 * len(si.keys)
    cn, orc = compare_batch(si, bases, off, 14, 15, 250, api.fraction(0.11, 15))
    assert cn["n_hits"] > 10000 and cn["n_clusters"] > 50 and 0 < orc["cl_alive"].sum() < len(orc["cl_alive"])
    # dense sketches of short reads over the same loci: w = 1 index, Illumina thresholds
    si1, b1, o1 = _variant_world(42, 25, 500, w=1, fixed_len=150, e_rate=0.002)
    cn, _ = compare_batch(si1, b1, o1, 1, 15, 31, api.fraction(0.001, 15), 10, expected_kmers=300)
    assert cn["n_hits"] > 20000
    cn, _ = compare_batch(si1, b1, o1, 1, 15, 31, api.fraction(0.001, 15), 3, expected_kmers=20)   # keeps clusters
    assert cn["n_clusters"] > 100


def test_workload_families_slice(gpu):
    """A slice of the bench sibling with 1.5 records per key, against the oracle."""
    si, bases, off, prm = synth.workload("nanopore_1k_families", seed=1, scale=0.012)
    assert len(si.rec_prg) > 1.45 * len(si.keys)
    cn, _ = compare_batch(si, bases, off, prm["w"], prm["k"], prm["max_diff"], api.fraction(prm["e_rate"], prm["k"]),
                          prm["min_cluster_size"])
    assert cn["n_hits"] > 20000


@pytest.mark.parametrize("direct", [None, "0"])
def test_200k_locus_index_slices(gpu, monkeypatch, direct):
    """200 000 loci: 27 M keys, a 2 GB slot table in HBM.  No hashed presence filter is worth having at that
    many keys: the library puts a bitmap over the whole key range in front of the table (minimizer keys crowd
    the low end of the range), or, with PMAP_FILTER_DIRECT=0, nothing (every minimizer probes the table).
    A slice of the nanopore reads and 3 000 reads of 150 bases with w = 1 (every k-mer probes) against the
    oracle over the same index."""
    if direct is not None:
        monkeypatch.setenv("PMAP_FILTER_DIRECT", direct)
    si, bases, off, prm = synth.workload("nanopore_200k", seed=1, scale=0.01)
    gidx = gpu_index(si)
    info = gidx.info()
    assert info["n_keys"] > 25_000_000 and info["table_slots"] * 32 >= (2 << 30)
    assert (info["filter_direct"], info["filter_bits_per_key"]) == ((1, 1) if direct is None else (0, 0))
    n = min(150, len(off) - 1)
    sub_off = off[:n + 1]
    frac = api.fraction(prm["e_rate"], prm["k"])
    cn, _ = compare_batch(si, bases[:int(sub_off[-1])], sub_off, prm["w"], prm["k"], prm["max_diff"], frac,
                          prm["min_cluster_size"], gidx=gidx)
    assert cn["n_hits"] > 10000 and cn["n_clusters"] > 100
    rng = np.random.default_rng(3)
    starts = rng.integers(0, int(sub_off[-1]) - 150, 3000)
    short = np.concatenate([bases[int(s):int(s) + 150] for s in starts])
    soff = np.arange(0, 150 * len(starts) + 1, 150, dtype=np.uint64)
    cn, _ = compare_batch(si, short, soff, 1, prm["k"], 31, api.fraction(0.001, prm["k"]), 10, expected_kmers=300,
                          gidx=gidx)
    assert cn["n_minimizers"] > 300000 and cn["n_hits"] > 1000
    gidx.close()


def test_8gb_table_slice(gpu):
    """A slot table of 2^28 slots (8 GB: beyond what one level of address translation covers): the 1 000-locus
    index plus 7e7 keys spread evenly over the key range on 199 000 more PRGs, no presence filter, so that every
    minimizer of a read probes the table.  A slice of the reads stage by stage against the oracle over the same
    arrays (clusters come from the real loci, the filler keys add scattered hits), then the whole batch of
    10 000 reads: hits of the slice's reads the same inside it."""
    si, bases, off, prm = synth.workload("nanopore_1k_8g", seed=1)
    gidx = gpu_index(si)
    info = gidx.info()
    assert info["n_keys"] > 70_000_000 and info["table_slots"] * 32 >= (8 << 30) and info["filter_bits"] == 0
    n = 60
    sub_off = off[:n + 1]
    frac = api.fraction(prm["e_rate"], prm["k"])
    cn, orc = compare_batch(si, bases[:int(sub_off[-1])], sub_off, prm["w"], prm["k"], prm["max_diff"], frac,
                            prm["min_cluster_size"], gidx=gidx)
    assert cn["n_hits"] > 10000 and cn["n_clusters"] > 100
    filler = orc["hit_prg"] >= 1000
    assert filler.sum() > 1000 and (~filler).sum() > 1000
    rd = api.Reads(bases, off)
    res = api.map_reads(gidx, rd, prm["w"], prm["k"], prm["max_diff"], frac, prm["min_cluster_size"])
    h = res.hits()
    sel = h["read"] < n
    assert sel.sum() == cn["n_hits"]
    np.testing.assert_array_equal(h["pos"][sel], orc["hit_pos"])
    np.testing.assert_array_equal(h["rec"][sel], orc["hit_rec"])
    np.testing.assert_array_equal(h["prg"][sel], orc["hit_prg"])
    res.close()
    rd.close()
    gidx.close()


@pytest.mark.parametrize("fname,w,k,max_diff,min_cluster", [
    ("oxa9_read.fasta", 14, 15, 250, 10),          # pandora map's defaults
    ("oxa9_read.fasta", 5, 15, 250, 10),           # w of test/test_cases/oxa/run_test.sh
    ("loman_k12_merged_pass.mm2.sorted_1196-1216.fastq", 14, 15, 250, 10),
    ("loman_k12_merged_pass.mm2.sorted_1196-1216.fastq", 3, 7, 15, 2)])
def test_reference_real_reads_file(gpu, fname, w, k, max_diff, min_cluster):
    """The real nanopore reads among the reference's test files (test/test_cases/), through
    pangraph_from_read_file.  The index the reference would map them against comes out of `pandora index`
    (src/localPRG.cpp:333-540, not part of this path and not buildable here), so the loci are cut from the
    reads themselves, on one strand or the other, overlapping, with variant sites where they are long enough.
    The fastq is a pile-up slice of reads of 15 to 27 bases, several with broken header lines and tabs that
    the reader turns into longer reads with letters outside ACGT: at the default w and k none of them has a
    sketch (the reference maps nothing from this file either, every read still takes an id); the short w
    and k make them map."""
    path = os.path.join(FIX, fname)
    reads = [r for _, r in O.fastaq_iterate(open(path, "rb").read())]
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    loci, seen = [], set()
    for r in reads:
        seq = np.frombuffer(r.upper().encode(), np.uint8)
        if len(seq) < 18 or not np.isin(seq, acgt).all() or r in seen or len(loci) >= 15:
            continue
        seen.add(r)
        for a in range(0, max(len(seq) - 100, 1), 250):
            piece = seq[a:a + 350]
            loci.append(piece.copy() if rng.random() < 0.5 else synth.revcomp(piece))
    assert len(loci) >= 3
    vloci = synth.make_variant_loci(loci, rng)
    si = synth.build_variant_index(vloci, w, k)
    idx = gpu_index(si)
    res, covg, _ = TP._expected_from_oracle(si, open(path, "rb").read(), w, k, max_diff, 0.11, min_cluster, 5000000,
                                            False, 300)
    want_fwd, want_rev = O.oracle_coverage(res, len(si.rec_prg))
    n_calls = 0
    for batch in (0, 3000):
        pg = api.pangraph_from_read_file(path, idx, w, k, max_diff, 0.11, min_cluster, batch_bases=batch, coverage=True)
        n_calls = TP._check_pangraph(pg, res, covg)
        got_fwd, got_rev = pg.coverage()
        np.testing.assert_array_equal(got_fwd, want_fwd)
        np.testing.assert_array_equal(got_rev, want_rev)
    if fname.startswith("oxa9") or k == 7:
        assert n_calls >= 2
    # the batch path on the same reads: minimizers, hits and clusters of every read against the oracle
    bases, off = O.pack_reads(reads)
    compare_batch(si, bases, off, w, k, max_diff, api.fraction(0.11, k), min_cluster, gidx=idx)
    idx.close()


def _write_idx_text(si, path):
    """Index::save (src/index.cpp:51-68)."""
    with open(path, "w") as f:
        f.write("%d\n" % len(si.keys))
        for i, key in enumerate(si.keys):
            recs = []
            for j in range(int(si.rec_off[i]), int(si.rec_off[i + 1])):
                a, b = int(si.path_off[j]), int(si.path_off[j + 1])
                ivs = "".join("[%d, %d)" % (si.iv_start[q], si.iv_start[q] + si.iv_len[q]) for q in range(a, b))
                recs.append("(%d, %d{%s}, %d, %d)" % (si.rec_prg[j], b - a, ivs, si.rec_knode[j], si.rec_strand[j]))
            f.write("%d\t%d\t%s\t\n" % (key, len(recs), "\t".join(recs)))


def test_idx_text_and_cache_to_map(gpu, tmp_path):
    """Index::load's text format -> pmap_index_load_cached -> map == oracle, then the same through the
    binary cache the first load left behind."""
    si, bases, off = _variant_world(43, 40, 60, mean_len=2500, e_rate=0.07)
    txt, cache = tmp_path / "prgs.fa.k15.w14.idx", tmp_path / "prgs.fa.k15.w14.idx.pmapcache"
    _write_idx_text(si, txt)
    frac = api.fraction(0.11, 15)
    from_text, hit = api.Index.load_cached(txt, 15, si.min_path_length, cache)
    assert not hit and cache.exists()
    cn1, _ = compare_batch(si, bases, off, 14, 15, 250, frac, gidx=from_text)
    from_cache, hit = api.Index.load_cached(txt, 15, si.min_path_length, cache)
    assert hit
    cn2, _ = compare_batch(si, bases, off, 14, 15, 250, frac, gidx=from_cache)
    assert cn1["n_hits"] == cn2["n_hits"] > 5000 and cn1["n_clusters"] == cn2["n_clusters"] > 20
    # and the file-level entry point over the cached index
    p = tmp_path / "reads.fq"
    TP._write_fastq(p, bases, off, None)
    res, covg, _ = TP._expected_from_oracle(si, p.read_bytes(), 14, 15, 250, 0.11, 10, 5000, False, 300)
    TP._check_pangraph(api.pangraph_from_read_file(p, from_cache, 14, 15, 250, 0.11, 10, 5000), res, covg)


def test_one_result_object_across_indexes(gpu):
    """A result object reused for an index that was created after the one it served before was destroyed
    (the allocator may hand out the same address): thresholds and record tables must follow the new index."""
    rng = np.random.default_rng(44)
    loci_a = synth.make_loci(6, rng)
    loci_b = synth.make_loci(40, rng)
    genome = synth.make_genome(loci_a + loci_b, rng)
    bases, off = synth.simulate_reads(genome, 60, rng, mean_len=2500, e_rate=0.05)
    rd = api.Reads(bases, off)
    frac = api.fraction(0.11, 15)
    res = None
    for loci in (loci_a, loci_b, loci_a[:2], loci_b):
        si = synth.build_index(loci, 14, 15)
        si.min_path_length[:] = rng.integers(20, 400, len(loci))      # thresholds that differ from index to index
        gidx = gpu_index(si)
        res = api.map_reads(gidx, rd, 14, 15, 250, frac, 3, result=res)
        fresh = api.map_reads(gidx, rd, 14, 15, 250, frac, 3)
        assert res.counts()["n_clusters"] == fresh.counts()["n_clusters"]
        a, b = res.clusters(), fresh.clusters()
        for f in a:
            np.testing.assert_array_equal(a[f], b[f], err_msg=f)
        orc = O.oracle_map(O.FlatIndex(si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                                       si.iv_start, si.iv_len), bases, off, np.arange(60, dtype=np.uint32), 14, 15,
                           250, frac, 3, 0xFFFFFFFF, si.min_path_length)
        assert res.counts()["n_clusters"] == len(orc["cl_size"])
        fresh.close()
        res.wait()
        gidx.close()
