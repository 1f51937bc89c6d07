"""pangraph_from_read_file through the C ABI: the reference's own expectations for read2.fa /
read2.fq, and file-level parity with the oracle (reader quirks, read-id assignment, coverage stop,
batch-size invariance, filter_clusters, add_node sequence)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import ref_cases as RC
from pandora_amd import api, synth

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(__file__), "golden", "ref_fixtures")


def _index_from_records(records, mpl, k):
    fi = O.FlatIndex.from_records(records)
    return fi, api.Index.create(k, fi.keys, fi.rec_off, fi.rec_prg, fi.rec_knode, fi.rec_strand, fi.path_off,
                                fi.iv_start, fi.iv_len, mpl)


@pytest.mark.parametrize("fname", ["read2.fa", "read2.fq"])
def test_reference_pangraph_from_read_file(gpu, fname):
    # test/utils_test.cpp:708-944: nodes 0, 1, 2, 3 appear
    case = RC.PANGRAPH_FROM_READ_FILE
    fi, idx = _index_from_records(case["records"], RC.case_mpl(case), 3)
    pg = api.pangraph_from_read_file(os.path.join(FIX, fname), idx, 1, 3, 1, 0.1, 1)
    assert set(pg.nodes()) == case["expected_nodes"]
    assert set(pg.reads()) == {0, 1}          # read2 and read2_prime


def _expected_from_oracle(si, path_bytes, w, k, max_diff, e_rate, min_cluster_size, genome_size, illumina, max_covg):
    reads = O.fastaq_iterate(path_bytes)
    bases, off = O.pack_reads([r for _, r in reads])
    used, take, ids, covg, exp = O.read_loop(bases, off, w, k, genome_size, max_covg, illumina)
    # untaken reads produce nothing: blank them so one oracle call covers the file
    keep = [reads[i][1] if take[i] else "" for i in range(len(reads))]
    kb, ko = O.pack_reads(keep)
    fi = O.FlatIndex(si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off, si.iv_start,
                     si.iv_len)
    frac = O.lib().orc_fraction(e_rate, k)
    res = O.oracle_map(fi, kb, ko, ids, w, k, max_diff, frac, min_cluster_size, exp, si.min_path_length)
    return res, covg, fi


def _check_pangraph(pg, res, covg):
    alive = res["cl_alive"].astype(bool)
    calls = pg.calls()
    np.testing.assert_array_equal(calls["prg"], res["cl_prg"][alive])
    np.testing.assert_array_equal(calls["read"], res["cl_read"][alive])
    np.testing.assert_array_equal(calls["size"], res["cl_size"][alive])
    np.testing.assert_array_equal(calls["fwd"], res["cl_fwd"][alive])
    assert pg.covg == covg
    # node coverage = number of add_node calls per prg; read node lists drop immediate repeats
    exp_covg, exp_reads = {}, {}
    for p, r, f in zip(res["cl_prg"][alive], res["cl_read"][alive], res["cl_fwd"][alive]):
        exp_covg[int(p)] = exp_covg.get(int(p), 0) + 1
        lst = exp_reads.setdefault(int(r), [])
        if not lst or lst[-1] != (int(p), int(f)):
            lst.append((int(p), int(f)))
    assert pg.nodes() == exp_covg
    assert pg.reads() == exp_reads
    # Read::hits = the hits of every surviving cluster
    exp_hits = set()
    for b, s in zip(res["cl_begin"][alive], res["cl_size"][alive]):
        for i in range(int(b), int(b) + int(s)):
            exp_hits.add((int(res["hit_read"][i]), int(res["hit_prg"][i]), int(res["hit_pos"][i]),
                          int(res["hit_rec"][i]), int(res["hit_fwd"][i])))
    h = pg.hits()
    got = list(zip(h["read"].tolist(), h["prg"].tolist(), h["pos"].tolist(), h["rec"].tolist(), h["fwd"].tolist()))
    assert len(got) == len(exp_hits) and set(got) == exp_hits
    # the cluster of every call, as add_clusters_to_pangraph hands it to add_node (src/utils.cpp:346-351):
    # the call's slice of the oracle's sorted hits, in pComp order
    off, cpos, crec = pg.call_hits()
    assert len(off) == len(calls["prg"]) + 1 and off[0] == 0 and off[-1] == len(got)
    np.testing.assert_array_equal(np.diff(off.astype(np.int64)), res["cl_size"][alive])
    ob = res["cl_begin"][alive].astype(np.int64)
    idx = np.concatenate([np.arange(b, b + s) for b, s in zip(ob, res["cl_size"][alive])]) if len(ob) else \
        np.zeros(0, np.int64)
    np.testing.assert_array_equal(cpos, res["hit_pos"][idx])
    np.testing.assert_array_equal(crec, res["hit_rec"][idx])
    # replay of the calls the way a binding does it (pangenome::Graph::add_node, src/pangenome/pangraph.cpp:
    # 108-140): Read::add_hits must find every hit new, the orientation comes from the cluster's first hit
    read_hits, node_covg, read_nodes = {}, {}, {}
    for j in range(len(calls["prg"])):
        p, r, f = int(calls["prg"][j]), int(calls["read"][j]), int(calls["fwd"][j])
        have = read_hits.setdefault((r, p), set())
        for i in range(int(off[j]), int(off[j + 1])):
            hit = (int(cpos[i]), int(crec[i]), f)
            assert hit not in have                       # src/pangenome/panread.cpp:25-27
            have.add(hit)
        node_covg[p] = node_covg.get(p, 0) + 1
        lst = read_nodes.setdefault(r, [])
        if not lst or lst[-1] != (p, f):
            lst.append((p, f))
    assert node_covg == exp_covg and read_nodes == exp_reads
    assert {(r, p, pos, rec, f) for (r, p), hs in read_hits.items() for pos, rec, f in hs} == exp_hits
    return len(calls["prg"])


def _write_fastq(path, bases, off, rng, blank_tail=True):
    with open(path, "wb") as f:
        for i in range(len(off) - 1):
            seq = bytes(bases[int(off[i]):int(off[i + 1])])
            if i % 3 == 0:
                f.write(b">r%d\n" % i)
                for j in range(0, len(seq), 70):     # multi-line FASTA record
                    f.write(seq[j:j + 70] + b"\n")
            else:
                f.write(b"@r%d\n" % i + seq + b"\n+\n" + b"@" * len(seq) + b"\n")
        if blank_tail:
            f.write(b"\n\n")                          # the reference maps the last read again per blank line


def test_file_parity_nanopore(gpu, tmp_path):
    rng = np.random.default_rng(21)
    loci = synth.make_loci(25, rng, dup_frac=0.1)
    si = synth.build_index(loci, 14, 15)
    genome = synth.make_genome(loci, rng)
    bases, off = synth.simulate_reads(genome, 80, rng, mean_len=2500, e_rate=0.06, n_frac=0.05, short_frac=0.05)
    p = tmp_path / "reads.fq"
    _write_fastq(p, bases, off, rng)
    idx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                           si.iv_start, si.iv_len, si.min_path_length)
    res, covg, _ = _expected_from_oracle(si, p.read_bytes(), 14, 15, 250, 0.11, 10, 5000, False, 300)
    want_fwd, want_rev = O.oracle_coverage(res, len(si.rec_prg))
    assert want_fwd.sum() + want_rev.sum() > 1000 and want_fwd.max() > 1
    for batch in (0, 20000, 3000):                    # batch size must not change anything
        pg = api.pangraph_from_read_file(p, idx, 14, 15, 250, 0.11, 10, 5000, False, False, 300, batch_bases=batch,
                                         coverage=True)
        n = _check_pangraph(pg, res, covg)
        assert n > 20
        got_fwd, got_rev = pg.coverage()
        np.testing.assert_array_equal(got_fwd, want_fwd)
        np.testing.assert_array_equal(got_rev, want_rev)
    # without Read::hits: same graph, same coverage, no hits kept
    pg = api.pangraph_from_read_file(p, idx, 14, 15, 250, 0.11, 10, 5000, False, False, 300, batch_bases=7000,
                                     keep_hits=False, coverage=True)
    assert pg.info()["n_hits"] == 0 and len(pg.calls()["prg"]) == n
    got_fwd, got_rev = pg.coverage()
    np.testing.assert_array_equal(got_fwd, want_fwd)
    np.testing.assert_array_equal(got_rev, want_rev)
    with pytest.raises(api.PmapError):
        api.pangraph_from_read_file(p, idx, 14, 15, 250, 0.11, 10, 5000).coverage()     # built without coverage


def test_file_parity_coverage_stop_and_illumina(gpu, tmp_path):
    rng = np.random.default_rng(22)
    loci = synth.make_loci(10, rng)
    si = synth.build_index(loci, 14, 15)
    genome = synth.make_genome(loci, rng)
    bases, off = synth.simulate_reads(genome, 300, rng, fixed_len=150, e_rate=0.002, n_frac=0.02)
    p = tmp_path / "ill.fa"
    _write_fastq(p, bases, off, rng, blank_tail=False)
    idx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                           si.iv_start, si.iv_len, si.min_path_length)
    # genome_size 1000, max_covg 20: stops after ~21 kb of sketched reads
    res, covg, _ = _expected_from_oracle(si, p.read_bytes(), 14, 15, 31, 0.001, 2, 1000, True, 20)
    pg = api.pangraph_from_read_file(p, idx, 14, 15, 31, 0.001, 2, 1000, True, False, 20, batch_bases=5000,
                                     coverage=True)
    n = _check_pangraph(pg, res, covg)
    want_fwd, want_rev = O.oracle_coverage(res, len(si.rec_prg))
    got_fwd, got_rev = pg.coverage()
    np.testing.assert_array_equal(got_fwd, want_fwd)
    np.testing.assert_array_equal(got_rev, want_rev)
    assert covg == 21 and 0 < n and max(pg.reads()) < 200
    with pytest.raises(api.PmapError):
        api.pangraph_from_read_file(p, idx, 14, 15, 31, 0.001, 2, 1000, True, True, 20)      # clean=true
    with pytest.raises(api.PmapError) as e:
        api.pangraph_from_read_file(tmp_path / "missing.fa", idx, 14, 15, 31, 0.001)
    assert e.value.code == 3


def test_gzip_and_repeated_clusters_coverage(gpu, tmp_path):
    """A gzip file, and reads that cross the same locus twice (two surviving clusters on one prg): every hit
    of such a read counts twice in the kmer node coverage, as in add_hits_to_kmergraphs."""
    import gzip
    rng = np.random.default_rng(23)
    loci = synth.make_loci(6, rng)
    si = synth.build_index(loci, 14, 15)
    spacer = lambda n: bytes(synth.random_dna(n, rng))
    reads = []
    for i in range(12):
        l = bytes(loci[i % 6])
        reads.append(spacer(300) + l + spacer(800) + l + spacer(200))        # the same locus twice
        reads.append(spacer(100) + l + spacer(50) + bytes(synth.revcomp(loci[(i + 1) % 6])))
    bases, off = O.pack_reads(reads)
    p = tmp_path / "twice.fq.gz"
    plain = tmp_path / "twice.fq"
    _write_fastq(plain, bases, off, rng, blank_tail=False)
    with gzip.open(p, "wb") as f:
        f.write(plain.read_bytes())
    idx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                           si.iv_start, si.iv_len, si.min_path_length)
    res, covg, _ = _expected_from_oracle(si, plain.read_bytes(), 14, 15, 250, 0.11, 10, 5000, False, 300)
    want_fwd, want_rev = O.oracle_coverage(res, len(si.rec_prg))
    alive = res["cl_alive"].astype(bool)
    pairs = list(zip(res["cl_read"][alive].tolist(), res["cl_prg"][alive].tolist()))
    assert len(pairs) > len(set(pairs))               # some (read, prg) has two surviving clusters
    pg = api.pangraph_from_read_file(p, idx, 14, 15, 250, 0.11, 10, 5000, coverage=True, batch_bases=4000)
    _check_pangraph(pg, res, covg)
    got_fwd, got_rev = pg.coverage()
    np.testing.assert_array_equal(got_fwd, want_fwd)
    np.testing.assert_array_equal(got_rev, want_rev)
    assert got_fwd.max() >= 2


def test_contexts_and_parsers_give_one_result(gpu, tmp_path, monkeypatch):
    """pangraph_from_read_file spreads the pieces of a file over parser threads and its batches over mapping
    contexts (here three of them on the one GPU, the default two, and one); a gzip-free file can also go
    through the single parsing thread that gzip files use.  Same pangraph every way, equal to the oracle's."""
    from pandora_amd import sharding
    rng = np.random.default_rng(31)
    loci = synth.make_loci(30, rng, dup_frac=0.1, family_frac=0.3)
    vloci = synth.make_variant_loci(loci, rng)
    si = synth.build_variant_index(vloci, 14, 15)
    genome = synth.make_variant_genome(vloci, rng)
    bases, off = synth.simulate_reads(genome, 150, rng, mean_len=2000, e_rate=0.06, n_frac=0.05, short_frac=0.05)
    p = tmp_path / "reads.fq"
    _write_fastq(p, bases, off, rng)          # quality lines start with '@': most cuts of the file are wrong ones
    idx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                           si.iv_start, si.iv_len, si.min_path_length)
    res, covg, _ = _expected_from_oracle(si, p.read_bytes(), 14, 15, 250, 0.11, 10, 5000, False, 300)
    want_fwd, want_rev = O.oracle_coverage(res, len(si.rec_prg))

    def run(batch):
        pg = api.pangraph_from_read_file(p, idx, 14, 15, 250, 0.11, 10, 5000, batch_bases=batch, coverage=True)
        n = _check_pangraph(pg, res, covg)
        got_fwd, got_rev = pg.coverage()
        np.testing.assert_array_equal(got_fwd, want_fwd)
        np.testing.assert_array_equal(got_rev, want_rev)
        return n

    try:
        for devices in ([0, 0, 0], [0], []):
            api.init_devices(devices)
            for batch in (2500, 40000, 0):
                assert run(batch) > 50
        monkeypatch.setenv("PMAP_PARSE_THREADS", "0")
        assert run(7000) > 50
        monkeypatch.setenv("PMAP_PARSE_THREADS", "3")
        assert run(900) > 50
    finally:
        api.init_devices([])
    # objects made before the context juggling still work, and so does the batch API from this thread
    rd = api.Reads(bases, off)
    whole = api.map_reads(idx, rd, 14, 15, 250, api.fraction(0.11, 15))
    # the same batch sharded in two as N GPUs would take it, merged in read-id order
    parts, counts = [], []
    for rank in range(2):
        sb, so, first = sharding.shard(bases, off, rank, 2)
        r = api.map_reads(idx, api.Reads(sb, so, first_read_id=first), 14, 15, 250, api.fraction(0.11, 15))
        parts.append(r.clusters())
        counts.append(r.counts()["n_hits"])
    merged = sharding.merge_tables(parts, counts)
    want = whole.clusters()
    assert len(want["size"]) > 50
    for f in want:
        np.testing.assert_array_equal(merged[f], want[f], err_msg=f)
