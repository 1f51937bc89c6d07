"""CPU-side checks of libpandora_amd.so that need no GPU: exported symbols, index construction
(table, CSR, path ranks, text parser), the FASTA/FASTQ reader, min_path_length, clusterComp +
filter_clusters, read sharding under gloo."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

import oracle_lib as O
import ref_cases as RC
from pandora_amd import _lib, api, sharding, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "pandora_amd.h")).read()
    declared = set(re.findall(r"\b(pmap_[a-z0-9_]+)\s*\(", header))
    declared -= {"pmap_index_desc_t"}
    lib = _lib.load()
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.PmapError) as e:
        api.init(0)
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(api.PmapError):
        api.Reads.from_list([b"ACGT"])


def test_fraction_bits():
    for e, bits in [(0.11, 0x3DC4A8BB), (0.001, 0x3EFC304B), (0.0001, 0x3EFF9DC6)]:
        assert struct.unpack("<I", struct.pack("<f", api.fraction(e, 15)))[0] == bits


def _ranks_by_python(si):
    """Rank of every record's (prg, path) in (prg, prg::Path::operator<) order, independently in Python."""
    keys = []
    for j in range(len(si.rec_prg)):
        a, b = int(si.path_off[j]), int(si.path_off[j + 1])
        keys.append((int(si.rec_prg[j]), [(int(si.iv_start[q]), int(si.iv_len[q])) for q in range(a, b)]))
    order = sorted(range(len(keys)), key=lambda j: keys[j])
    rank = np.zeros(len(keys), np.uint32)
    r = 0
    for i, j in enumerate(order):
        if i and keys[order[i - 1]] != keys[j]:
            r += 1
        rank[j] = r
    return rank


def test_index_create_path_ranks_and_accessors():
    rng = np.random.default_rng(0)
    loci = [synth.random_dna(int(rng.integers(60, 200)), rng) for _ in range(6)]
    loci.append(loci[2].copy())
    si = synth.build_index(loci, 3, 7, rng=rng, split_paths=0.5)
    idx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                           si.iv_start, si.iv_len, si.min_path_length)
    info = idx.info()
    assert info["n_keys"] == len(si.keys) and info["n_recs"] == len(si.rec_prg) and info["n_prg"] == len(loci)
    assert info["table_slots"] >= 2 * info["n_keys"] and info["table_slots"] & (info["table_slots"] - 1) == 0
    rec = idx.records()
    np.testing.assert_array_equal(rec["prg"], si.rec_prg)
    np.testing.assert_array_equal(rec["knode"], si.rec_knode)
    np.testing.assert_array_equal(rec["strand"], si.rec_strand)
    np.testing.assert_array_equal(rec["path_off"], si.path_off)
    np.testing.assert_array_equal(rec["iv_start"], si.iv_start)
    np.testing.assert_array_equal(rec["key"], np.repeat(si.keys, np.diff(si.rec_off).astype(np.int64)))
    np.testing.assert_array_equal(rec["path_rank"], _ranks_by_python(si))
    # the ranks order paths exactly as the reference build's prg::Path::operator<
    r = O.ref()
    if r is not None:
        for _ in range(300):
            a, b = (int(x) for x in rng.integers(0, len(si.rec_prg), 2))
            if si.rec_prg[a] != si.rec_prg[b]:
                continue
            pa = [int(si.path_off[a]), int(si.path_off[a + 1])]
            pb = [int(si.path_off[b]), int(si.path_off[b + 1])]
            c = r.ref_path_cmp(si.iv_start[pa[0]:pa[1]].ctypes.data_as(O.u32p), si.iv_len[pa[0]:pa[1]].ctypes.data_as(O.u32p),
                               pa[1] - pa[0], si.iv_start[pb[0]:pb[1]].ctypes.data_as(O.u32p),
                               si.iv_len[pb[0]:pb[1]].ctypes.data_as(O.u32p), pb[1] - pb[0])
            assert np.sign(int(rec["path_rank"][a]) - int(rec["path_rank"][b])) == c


def test_index_create_rejects_bad_input():
    ok = dict(k=3, keys=[5], rec_off=[0, 1], rec_prg=[0], rec_knode=[1], rec_strand=[1], path_off=[0, 1],
              iv_start=[3], iv_len=[3], min_path_length=[4])
    api.Index.create(**ok)
    for bad, code in [(dict(ok, k=32), 1), (dict(ok, k=0), 1), (dict(ok, keys=[64]), 1),       # key >= 4^k
                      (dict(ok, rec_prg=[1]), 1),                                                 # prg >= n_prg
                      (dict(ok, iv_len=[4]), 1),                                                  # path length != k
                      (dict(ok, rec_off=[0, 2], rec_prg=[0, 0], rec_knode=[1, 2], rec_strand=[1, 1],
                            path_off=[0, 1, 2], iv_start=[3, 3], iv_len=[3, 3]), 5)]:            # duplicate record
        with pytest.raises(api.PmapError) as e:
            api.Index.create(**bad)
        assert e.value.code == code
    # same path on both strands is legal (MiniRecord::operator== includes the strand)
    api.Index.create(**dict(ok, rec_off=[0, 2], rec_prg=[0, 0], rec_knode=[1, 2], rec_strand=[1, 0],
                            path_off=[0, 1, 2], iv_start=[3, 3], iv_len=[3, 3]))


def test_index_load_reference_fixture_files():
    """test/test_cases/prg{1,2,3}.fa.idx (reference data files, w=1 k=3)."""
    idx = api.Index.load(os.path.join(FIX, "prg1.fa.idx"), 3, [0, 3])
    rec = idx.records()
    assert list(rec["key"]) == [30, 30] and list(rec["prg"]) == [1, 1]
    assert list(rec["knode"]) == [1, 2] and list(rec["strand"]) == [1, 0]
    assert list(zip(rec["iv_start"], rec["iv_len"])) == [(0, 3), (1, 3)]
    assert 30 == min(O.kmerhash("AGC", 3))
    idx = api.Index.load(os.path.join(FIX, "prg3.fa.idx"), 3, [0, 0, 0, 5])
    rec = idx.records()
    assert idx.info()["n_keys"] == 2 and list(rec["key"]) == [6, 30, 30]
    assert list(rec["path_off"]) == [0, 4, 8, 12]
    assert list(zip(rec["iv_start"][:4], rec["iv_len"][:4])) == [(4, 1), (12, 1), (16, 0), (23, 1)]
    assert rec["path_rank"][1] < rec["path_rank"][2] < rec["path_rank"][0]
    with pytest.raises(api.PmapError) as e:
        api.Index.load(os.path.join(FIX, "does_not_exist.idx"), 3, [0])
    assert e.value.code == 3 and "Have you run pandora index?" in str(e.value)


def test_index_text_round_trip(tmp_path):
    """Index::save format (src/index.cpp:51-68) written by the test, parsed by the library."""
    rng = np.random.default_rng(1)
    si = synth.build_index([synth.random_dna(150, rng) for _ in range(5)], 4, 9, rng=rng, split_paths=0.4)
    lines = [str(len(si.keys))]
    for i, key in enumerate(si.keys):
        recs = []
        for j in range(int(si.rec_off[i]), int(si.rec_off[i + 1])):
            a, b = int(si.path_off[j]), int(si.path_off[j + 1])
            path = "%d{%s}" % (b - a, "".join("[%d, %d)" % (si.iv_start[q], si.iv_start[q] + si.iv_len[q])
                                              for q in range(a, b)))
            recs.append("(%d, %s, %d, %d)" % (si.rec_prg[j], path, si.rec_knode[j], si.rec_strand[j]))
        lines.append("%d\t%d\t%s" % (key, len(recs), "\t".join(recs)))
    p = tmp_path / "x.k9.w4.idx"
    p.write_text("\n".join(lines) + "\n")
    idx = api.Index.load(p, 9, si.min_path_length)
    rec = idx.records()
    np.testing.assert_array_equal(rec["prg"], si.rec_prg)
    np.testing.assert_array_equal(rec["knode"], si.rec_knode)
    np.testing.assert_array_equal(rec["strand"], si.rec_strand)
    np.testing.assert_array_equal(rec["iv_start"], si.iv_start)
    np.testing.assert_array_equal(rec["iv_len"], si.iv_len)


def _write_idx_text(si, path):
    lines = [str(len(si.keys))]
    for i, key in enumerate(si.keys):
        recs = []
        for j in range(int(si.rec_off[i]), int(si.rec_off[i + 1])):
            a, b = int(si.path_off[j]), int(si.path_off[j + 1])
            ivs = "".join("[%d, %d)" % (si.iv_start[q], si.iv_start[q] + si.iv_len[q]) for q in range(a, b))
            recs.append("(%d, %d{%s}, %d, %d)" % (si.rec_prg[j], b - a, ivs, si.rec_knode[j], si.rec_strand[j]))
        lines.append("%d\t%d\t%s" % (key, len(recs), "\t".join(recs)))
    path.write_text("\n".join(lines) + "\n")


def _same_index(a, b):
    ia, ib = a.info(), b.info()
    for f in ("k", "n_keys", "n_recs", "n_paths", "table_slots", "n_prg"):
        assert ia[f] == ib[f], f
    ra, rb = a.records(), b.records()
    for f in ra:
        np.testing.assert_array_equal(ra[f], rb[f], err_msg=f)


def test_index_binary_cache(tmp_path):
    rng = np.random.default_rng(4)
    si = synth.build_index([synth.random_dna(400, rng) for _ in range(8)], 5, 11, rng=rng, split_paths=0.3)
    txt, cache = tmp_path / "p.idx", tmp_path / "p.idx.pmapcache"
    _write_idx_text(si, txt)
    ref = api.Index.load(txt, 11, si.min_path_length)
    a, hit = api.Index.load_cached(txt, 11, si.min_path_length, cache)
    assert not hit and cache.exists()
    _same_index(ref, a)
    b, hit = api.Index.load_cached(txt, 11, si.min_path_length, cache)
    assert hit
    _same_index(ref, b)
    _same_index(ref, api.Index.load_cache(cache))
    # other parameters or a changed text file: the cache is not used, and is rewritten
    mpl2 = np.array(si.min_path_length, np.uint32)
    mpl2[0] += 1
    c, hit = api.Index.load_cached(txt, 11, mpl2, cache)
    assert not hit
    c, hit = api.Index.load_cached(txt, 11, mpl2, cache)
    assert hit
    os.utime(txt, ns=(1, 1))
    c, hit = api.Index.load_cached(txt, 11, mpl2, cache)
    assert not hit
    # explicit save / load, damaged files
    other = tmp_path / "other.cache"
    ref.save_cache(other)
    _same_index(ref, api.Index.load_cache(other))
    raw = other.read_bytes()
    for bad in (raw[:len(raw) // 2], b"PMAPIDX0" + raw[8:], raw[:-8] + b"XXXXXXXX", raw + b"\0" * 64, b""):
        (tmp_path / "bad.cache").write_bytes(bad)
        with pytest.raises(api.PmapError) as e:
            api.Index.load_cache(tmp_path / "bad.cache")
        assert e.value.code == 3
    with pytest.raises(api.PmapError):
        api.Index.load_cache(tmp_path / "missing.cache")
    # a damaged cache next to a good text file: falls back to the text
    cache.write_bytes(raw[:100])
    d, hit = api.Index.load_cached(txt, 11, mpl2, cache)
    assert not hit
    assert api.Index.load_cached(txt, 11, mpl2, cache)[1]


def test_presence_filter_choice(tmp_path, monkeypatch):
    """Which presence filter an index gets: a hashed bitmap of at most 2^24 bits while the keys are few enough
    for it (one bit per key, two on denser indexes); beyond that, for k <= 15, a bitmap over the whole key range
    when the keys crowd the low end of the range as window minima do (w = 14), none when they are spread evenly
    (w = 1); the bitmap survives the binary cache."""
    rng = np.random.default_rng(9)
    loci = [synth.random_dna(3000, rng) for _ in range(12)]
    si14 = synth.build_index(loci, 14, 15)
    si1 = synth.build_index(loci, 1, 15)

    def info(si):
        idx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off,
                               si.iv_start, si.iv_len, si.min_path_length)
        i = idx.info()
        return idx, (i["filter_bits"], i["filter_bits_per_key"], i["filter_direct"])

    assert info(si14)[1][1:] == (2, 0)                    # few keys: hashed, 8 to 16 bits per key, two of them set
    monkeypatch.setenv("PMAP_FILTER_MAX_BITS", "12")      # as if the keys outnumbered what a hashed filter takes
    idx, got = info(si14)
    assert got == (30, 1, 1)
    assert info(si1)[1] == (0, 0, 0)
    monkeypatch.setenv("PMAP_FILTER_DIRECT", "0")
    assert info(si14)[1] == (0, 0, 0)
    cache = tmp_path / "direct.cache"
    idx.save_cache(cache)
    i = api.Index.load_cache(cache).info()
    assert (i["filter_bits"], i["filter_bits_per_key"], i["filter_direct"]) == (30, 1, 1)


def test_release_host_cache_without_buffers():
    """Nothing page-locked is kept before any read file has been mapped."""
    assert _lib.load().pmap_release_host_cache() == 0


def test_gfa_min_path_length(tmp_path):
    cases = list(RC.BIGGER_INFER["graphs"].values()) + list(RC.SIMPLE_INFER["graphs"].values())
    for i, (paths, edges) in enumerate(cases):
        p = tmp_path / ("g%d.gfa" % i)
        p.write_text(RC.gfa_text(paths, edges))
        assert api.gfa_min_path_length(p) == RC.min_path_length(paths, edges)
    # linear kmer graph: number of minimizer nodes + 1
    paths = [[(0, 0)]] + [[(j, j + 5)] for j in range(0, 40, 3)] + [[(45, 45)]]
    edges = [(i, i + 1) for i in range(len(paths) - 1)]
    p = tmp_path / "lin.gfa"
    p.write_text(RC.gfa_text(paths, edges))
    assert api.gfa_min_path_length(p) == len(paths) - 1
    with pytest.raises(api.PmapError):
        api.gfa_min_path_length(tmp_path / "missing.gfa")


def test_read_fastaq_reference_fixtures():
    for name in ["reads.fa", "reads.fq", "reads.fa.gz", "reads.fq.gz", "read2.fa", "read2.fq", "read0.fa",
                 "test.fastq", "oxa9_read.fasta", "loman_k12_merged_pass.mm2.sorted_1196-1216.fastq"]:
        path = os.path.join(FIX, name)
        raw = open(path, "rb").read()
        if name.endswith(".gz"):
            import gzip
            raw = gzip.decompress(raw)
        assert api.read_fastaq(path) == O.fastaq_iterate(raw), name
    got = api.read_fastaq(os.path.join(FIX, "reads.fa.gz"))
    assert got[:2] == [("read0", "to be ignored"), ("read1", "should copy the phrase *should*")]
    assert len(api.read_fastaq(os.path.join(FIX, "loman_k12_merged_pass.mm2.sorted_1196-1216.fastq"))) >= 2
    with pytest.raises(api.PmapError) as e:
        api.read_fastaq(os.path.join(FIX, "nope.fa"))
    assert e.value.code == 3


def test_read_fastaq_quirks(tmp_path):
    for i, buf in enumerate([b">a\nACGT\n\n\n", b">a\nACGT\nAC", b"@a\nACGT\n+\n@@@@\n@b\nGG\n+\n!!\n", b"",
                             b">a\n>b\nAC\nGT\n>c\n", b"ACGT\n>x\nGG\n", b"\n\n>a\nAC\n"]):
        p = tmp_path / ("q%d.fa" % i)
        p.write_bytes(buf)
        assert api.read_fastaq(p) == O.fastaq_iterate(buf), buf


def test_filter_clusters_matches_oracle():
    rng = np.random.default_rng(5)
    for trial in range(5):
        k, w = 7, 2
        loci = [synth.random_dna(int(rng.integers(150, 400)), rng) for _ in range(6)]
        loci += [loci[0][20:].copy(), synth.revcomp(loci[1])]
        si = synth.build_index(loci, w, k, rng=rng, split_paths=0.3)
        fi = O.FlatIndex(si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off, si.iv_start,
                         si.iv_len)
        genome = np.concatenate(loci + [synth.revcomp(x) for x in loci[:3]])
        bases, off = synth.simulate_reads(genome, 30, rng, mean_len=600, e_rate=0.02)
        res = O.oracle_map(fi, bases, off, np.arange(30, dtype=np.uint32), w, k, 15, 0.02, 2, 0xFFFFFFFF,
                           si.min_path_length)
        n = len(res["cl_size"])
        assert n > 10 and 0 < res["cl_alive"].sum() < n
        rank = _ranks_by_python(si)
        perm = rng.permutation(n)
        first_rec = res["hit_rec"][res["cl_begin"].astype(np.int64)]
        cl = {"read": res["cl_read"][perm], "first_pos": res["cl_first_pos"][perm], "size": res["cl_size"][perm],
              "prg": res["cl_prg"][perm], "first_rank": rank[first_rec][perm], "fwd": res["cl_fwd"][perm],
              "last_pos": res["cl_last_pos"][perm]}
        order, alive = api.filter_clusters(cl)
        np.testing.assert_array_equal(perm[order.astype(np.int64)], np.arange(n))
        np.testing.assert_array_equal(alive, res["cl_alive"])


def test_shard_bounds():
    off = np.array([0, 10, 10, 30, 35, 100, 101], np.uint64)
    for world in (1, 2, 3, 4, 8):
        b = sharding.shard_bounds(off, world)
        assert len(b) == world and b[0][0] == 0 and b[-1][1] == 6
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    bases = np.arange(101, dtype=np.uint8)
    sb, so, first = sharding.shard(bases, off, 1, 2)
    assert first == sharding.shard_bounds(off, 2)[1][0] and so[0] == 0 and len(sb) == int(so[-1])


def _gloo_worker(rank, world, port, si_arrays, bases, off, prm, q):
    """One rank of a 2-rank CPU rehearsal of the multi-GPU flow: shard -> map own shard (the oracle
    stands in for the GPU kernels here) -> gather tables on rank 0 in read-id order."""
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    fi = O.FlatIndex(*si_arrays[:-1])
    sb, so, first = sharding.shard(bases, off, rank, world)
    ids = np.arange(first, first + len(so) - 1, dtype=np.uint32)
    r = O.oracle_map(fi, sb, so, ids, *prm, si_arrays[-1])
    alive = r["cl_alive"].astype(bool)
    local = {k: r[k][alive] for k in ("cl_read", "cl_prg", "cl_fwd", "cl_size", "cl_first_pos", "cl_last_pos")}
    merged = sharding.gather_tables(local, dst=0)
    dist.barrier()
    if rank == 0:
        q.put(merged)
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    import torch.multiprocessing as mp
    rng = np.random.default_rng(6)
    loci = synth.make_loci(15, rng)
    si = synth.build_index(loci, 14, 15)
    genome = synth.make_genome(loci, rng)
    bases, off = synth.simulate_reads(genome, 40, rng, mean_len=3000, e_rate=0.05)
    arrays = (si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off, si.iv_start, si.iv_len,
              si.min_path_length)
    prm = (14, 15, 250, api.fraction(0.11, 15), 10, 0xFFFFFFFF)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + int(rng.integers(0, 2000))
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, arrays, bases, off, prm, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = O.oracle_map(O.FlatIndex(*arrays[:-1]), bases, off, np.arange(40, dtype=np.uint32), *prm, arrays[-1])
    alive = one["cl_alive"].astype(bool)
    assert alive.sum() > 5
    for k in merged:
        np.testing.assert_array_equal(merged[k], one[k][alive])


def _piece_cases():
    rng = np.random.default_rng(12)
    seqs = [bytes(synth.random_dna(int(rng.integers(0, 300)), rng)) for _ in range(60)]
    fastq = b"".join(b"@r%d extra\n%s\n+\n%s\n" % (i, s, bytes(rng.choice(np.frombuffer(b"@>+!Ia5", np.uint8), len(s))))
                     for i, s in enumerate(seqs))
    fasta = b"".join(b">r%d\n" % i + b"".join(s[j:j + 60] + b"\n" for j in range(0, len(s), 60)) for i, s in enumerate(seqs))
    mixed = b"".join((b"@q%d\n%s\n+\n%s\n" % (i, s, b"@" * len(s))) if i % 2 else (b">a%d\n%s\n" % (i, s))
                     for i, s in enumerate(seqs))
    blanks = b"\n\n>a\nACGT\n\n>b\nGG\nTT\n\n\n@c\nAC\n+\n@@\n\n>d\n>e\nACG\n" * 7
    return [fastq, fasta, mixed, blanks, fastq[:-1], fasta + b"\n\n", b">only\nACGT", b"\n", b"@a\nAC\n+\n",
            b"ACGT\nGG\n>x\nTT\n", b"@a\nACGT\n+\n@CGT\n@b\nGGGG\n+\n>>>>\n" * 30]


def test_read_fastaq_in_pieces(tmp_path):
    """The reader of pangraph_from_read_file for plain files parses pieces of the file independently; however
    the file is cut (quality lines that start with '@' or '>' make wrong cuts that must be noticed), the reads
    are those of the sequential reader."""
    for i, buf in enumerate(_piece_cases()):
        p = tmp_path / ("pieces%d.fq" % i)
        p.write_bytes(buf)
        want = [r for _, r in O.fastaq_iterate(buf)]
        assert [r for _, r in api.read_fastaq(p)] == want
        for piece in (1, 2, 3, 7, 19, 64, 257, 1000, 5000, 1 << 20):
            assert api.read_fastaq_pieces(p, piece) == want, (i, piece)
    for name in ["reads.fa", "reads.fq", "read2.fa", "read2.fq", "read0.fa", "test.fastq", "oxa9_read.fasta",
                 "loman_k12_merged_pass.mm2.sorted_1196-1216.fastq"]:
        path = os.path.join(FIX, name)
        want = [r for _, r in O.fastaq_iterate(open(path, "rb").read())]
        for piece in (1, 5, 50, 333, 4096):
            assert api.read_fastaq_pieces(path, piece) == want, (name, piece)
    empty = tmp_path / "empty.fq"
    empty.write_bytes(b"")
    assert api.read_fastaq_pieces(empty, 10) == [] == [r for _, r in O.fastaq_iterate(b"")]
    with pytest.raises(api.PmapError):
        api.read_fastaq_pieces(tmp_path / "missing.fq", 10)


def test_read_fastaq_in_pieces_random_files(tmp_path):
    """Files made of random lines of every kind the reader distinguishes (headers of both kinds, '+' lines,
    blank lines, sequence, lines that only look like headers), with and without a final newline."""
    rng = np.random.default_rng(2025)
    kinds = [b"@h", b">h", b"+", b"", b"ACGT", b"GGA", b"@@@@", b">>", b"+x", b"\t", b"+", b"ACGTACGTAC"]
    weights = np.array([3, 3, 3, 2, 6, 3, 2, 1, 1, 1, 2, 3], float)
    weights /= weights.sum()
    for case in range(150):
        n = int(rng.integers(0, 40))
        lines = [kinds[i] for i in rng.choice(len(kinds), n, p=weights)]
        buf = b"\n".join(lines) + (b"\n" if rng.random() < 0.6 and n else b"")
        p = tmp_path / ("rnd%d.fq" % case)
        p.write_bytes(buf)
        want = [r for _, r in O.fastaq_iterate(buf)]
        if buf:
            assert [r for _, r in api.read_fastaq(p)] == want, buf
        for piece in (1, 2, 4, 9, 30, 1000):
            assert api.read_fastaq_pieces(p, piece) == want, (buf, piece)


def test_pack_reads_is_nt4():
    """pmap_pack_reads (what the parser threads of pangraph_from_read_file do to every read): nt4
    (src/inthash.cpp:42-63) of every letter, 16 to a word with the first base on top; a read with a letter that
    maps to 4 is flagged (it has no sketch, src/seq.cpp:49-53)."""
    import ctypes as C
    from pandora_amd import _lib
    from pandora_amd._lib import u8p, u32p, u64p
    lib = _lib.load()
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"ACGTacgtNnUu\x00\x01\x02\x03-", np.uint8)
    weight = np.array([1, 1, 1, 1, .3, .3, .3, .3, .02, .02, .02, .02, .05, .05, .05, .05, .01])
    for trial in range(200):
        n = int(rng.integers(0, 40))
        lens = rng.integers(0, 90, n)
        bases = rng.choice(alphabet, int(lens.sum()), p=weight / weight.sum()).astype(np.uint8)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        words = np.zeros(int(((lens + 15) // 16).sum()) + 1, np.uint32)
        ok = np.zeros(n + 1, np.uint8)
        assert lib.pmap_pack_reads(bases.ctypes.data_as(C.c_char_p), off.ctypes.data_as(u64p), n,
                                   words.ctypes.data_as(u32p), ok.ctypes.data_as(u8p)) == 0
        w = 0
        for r in range(n):
            codes = np.array([O.lib().orc_nt4(int(c)) for c in bases[int(off[r]):int(off[r + 1])]], np.uint64)
            good = bool((codes < 4).all())
            assert bool(ok[r]) == good
            nw = (len(codes) + 15) // 16
            if good:
                for j in range(nw):
                    want = sum(int(x) << (30 - 2 * i) for i, x in enumerate(codes[16 * j:16 * j + 16]))
                    assert int(words[w + j]) == want
            w += nw
