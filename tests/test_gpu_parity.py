"""GPU parity tests: every stage of the HIP path against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

import oracle_lib as O
from pandora_amd import api, synth
from parity import compare_batch, gpu_index

pytestmark = pytest.mark.gpu


def _small_world(seed, w, k, n_loci=40, n_reads=60, split_paths=0.0, mean_len=3000, e_rate=0.08):
    rng = np.random.default_rng(seed)
    loci = synth.make_loci(n_loci, rng, dup_frac=0.1)
    si = synth.build_index(loci, w, k, rng=rng, split_paths=split_paths)
    genome = synth.make_genome(loci, rng)
    bases, off = synth.simulate_reads(genome, n_reads, rng, mean_len=mean_len, e_rate=e_rate, n_frac=0.05,
                                      short_frac=0.05)
    return si, bases, off


def test_nanopore_small(gpu):
    si, bases, off = _small_world(1, 14, 15)
    cn, orc = compare_batch(si, bases, off, 14, 15, 250, api.fraction(0.11, 15))
    assert cn["n_hits"] > 1000 and cn["n_clusters"] > 10


def test_split_paths_and_first_read_id(gpu):
    si, bases, off = _small_world(2, 14, 15, split_paths=0.5)
    cn, _ = compare_batch(si, bases, off, 14, 15, 250, api.fraction(0.11, 15), first_read_id=1000)
    assert cn["n_clusters"] > 0


@pytest.mark.parametrize("w,k", [(1, 15), (5, 15), (10, 11), (14, 15), (19, 19), (31, 31), (7, 16), (12, 21),
                                 (1, 31), (14, 3), (3, 5), (2, 7), (4, 9), (5, 9), (6, 6), (8, 10), (17, 17),
                                 (33, 12), (64, 15), (63, 31), (13, 15), (15, 15), (16, 15)])
def test_wk_grid(gpu, w, k):
    si, bases, off = _small_world(10 + w + 100 * k, w, k, n_loci=12, n_reads=25, mean_len=1500, e_rate=0.03)
    compare_batch(si, bases, off, w, k, 100, api.fraction(0.05, k), min_cluster_size=3)


def test_low_complexity_ties(gpu):
    """Homopolymers and short repeats: long runs of equal hashes, every tie must be kept."""
    rng = np.random.default_rng(5)
    reads = [b"A" * 500, b"AC" * 300, b"ACG" * 200, b"T" * 40 + b"ACGT" * 50 + b"G" * 100,
             bytes(synth.random_dna(50, rng)) * 20, b"AAAAAAAAAAAAAAAAAAAAC" * 30]
    loci = [np.frombuffer(r, np.uint8).copy() for r in reads[:4]]
    for w, k in [(1, 3), (2, 3), (5, 4), (12, 5), (14, 15), (8, 8)]:
        si = synth.build_index(loci, w, k)
        bases, off = O.pack_reads(reads)
        compare_batch(si, bases, off, w, k, 50, api.fraction(0.01, k), min_cluster_size=1)


def test_tiles_with_more_minimizers_than_one_pass(gpu):
    """Long low-complexity stretches: nearly every position of a tile is a minimizer.  The sparse sketch
    kernels (w >= 8) keep a tile's minimizers in a list shorter than the tile and take a second pass for
    such tiles; stretches start and end inside tiles, so both passes of a tile hold minimizers."""
    rng = np.random.default_rng(15)
    rnd = lambda n: bytes(synth.random_dna(n, rng))
    reads = [b"A" * 3000, b"AC" * 2500, rnd(700) + b"A" * 1700 + rnd(900), rnd(300) + b"GT" * 400 + rnd(1000) + b"C" * 2200,
             b"ACG" * 1500, rnd(2500), b"T" * 641 + rnd(100) + b"T" * 1300, (b"A" * 20 + b"C") * 150]
    loci = [np.frombuffer(r, np.uint8).copy() for r in (reads[0][:300], reads[1][:300], reads[2][500:1000], reads[5][:800])]
    for w, k in [(14, 15), (8, 8), (16, 15), (9, 5), (14, 11)]:
        si = synth.build_index(loci, w, k)
        bases, off = O.pack_reads(reads)
        cn, _ = compare_batch(si, bases, off, w, k, 50, api.fraction(0.01, k), min_cluster_size=1)
        assert cn["n_minimizers"] > 9000


def test_edge_reads(gpu):
    """Lengths around w+k-1, non-ACGT letters, lower case, raw 0..3 bytes, empty reads."""
    rng = np.random.default_rng(6)
    w, k = 14, 15
    locus = synth.random_dna(400, rng)
    si = synth.build_index([locus], w, k)
    body = bytes(locus)
    reads = [
        body[:w + k - 2],            # one base too short: no sketch
        body[:w + k - 1],            # exactly one window
        body[:w + k],
        b"",
        body[:200].lower(),
        body[:100] + b"N" + body[101:300],
        body[:100] + b"U" + body[101:300],
        bytes(synth._CODE[np.frombuffer(body[:300], np.uint8)]),   # bytes 0,1,2,3 are valid letters
        body[50:51],
        body,
        b"",
        body[:16], body[:17], body[:31], body[:32], body[:33],     # word-boundary lengths
        bytes(synth.revcomp(locus)),
    ]
    bases, off = O.pack_reads(reads)
    cn, orc = compare_batch(si, bases, off, w, k, 250, api.fraction(0.11, k), min_cluster_size=1)
    so = orc["sketch_off"]
    assert so[1] - so[0] == 0 and so[2] - so[1] >= 1
    assert so[6] - so[5] == 0 and so[7] - so[6] == 0


def test_illumina_thresholds(gpu):
    """150 bp reads: w=1 dense sketch where the reference's own threshold rejects every cluster
    (expected_kmers = 300 -> threshold 147 > 136 k-mers per read), and w=14 where clusters survive."""
    rng = np.random.default_rng(8)
    loci = synth.make_loci(30, rng)
    genome = synth.make_genome(loci, rng)
    bases, off = synth.simulate_reads(genome, 400, rng, fixed_len=150, e_rate=0.001)
    k = 15
    for w in (1, 14):
        si = synth.build_index(loci, w, k)
        frac = api.fraction(0.001, k)
        cn, orc = compare_batch(si, bases, off, w, k, 2 * k + 1, frac, 10, expected_kmers=150 * 2 // w)
        if w == 1:
            assert cn["n_hits"] > 10000 and cn["n_clusters"] == 0
        else:
            assert cn["n_clusters"] > 0


def test_large_segment_and_reuse(gpu):
    """One read with far more hits than the sort kernel's LDS chunk, and result reuse across calls."""
    rng = np.random.default_rng(9)
    w, k = 1, 9
    locus = synth.random_dna(6000, rng)
    si = synth.build_index([locus, locus[1000:4000].copy()], w, k)
    reads = [bytes(locus), bytes(locus[:3000]), bytes(synth.revcomp(locus))]
    bases, off = O.pack_reads(reads)
    cn, _ = compare_batch(si, bases, off, w, k, 20, api.fraction(0.01, k), min_cluster_size=2)
    assert cn["n_hits"] > 3 * 2048
    gidx = gpu_index(si)
    rd = api.Reads(bases, off)
    res = api.map_reads(gidx, rd, w, k, 20, 0.3, 2)
    first = res.hits()
    small = api.Reads.from_list([bytes(locus[:500])])
    res = api.map_reads(gidx, small, w, k, 20, 0.3, 2, result=res)
    assert res.counts()["n_hits"] < len(first["read"])
    res = api.map_reads(gidx, rd, w, k, 20, 0.3, 2, result=res)
    again = res.hits()
    for f in first:
        np.testing.assert_array_equal(first[f], again[f])


def test_empty_batches(gpu):
    rng = np.random.default_rng(11)
    si = synth.build_index([synth.random_dna(300, rng)], 14, 15)
    gidx = gpu_index(si)
    for reads in ([], [b""], [b"ACGT"], [b"N" * 100]):
        rd = api.Reads.from_list(reads)
        res = api.map_reads(gidx, rd, 14, 15, 250, 0.1)
        c = res.counts()
        assert c["n_minimizers"] == 0 and c["n_hits"] == 0 and c["n_clusters"] == 0
    # reads that sketch but hit nothing
    rd = api.Reads.from_list([bytes(synth.random_dna(2000, rng))])
    c = api.map_reads(gidx, rd, 14, 15, 250, 0.1).counts()
    assert c["n_minimizers"] > 0 and c["n_hits"] == 0 and c["n_clusters"] == 0


def test_stage_entry_points(gpu):
    si, bases, off = _small_world(3, 14, 15, n_loci=10, n_reads=10)
    gidx = gpu_index(si)
    rd = api.Reads(bases, off)
    r1 = api.map_reads(gidx, rd, 14, 15, 250, 0.1, last_stage=api.STAGE_SKETCH)
    c1 = r1.counts()
    assert c1["n_minimizers"] > 0 and c1["n_hits"] == 0
    r2 = api.map_reads(gidx, rd, 14, 15, 250, 0.1, last_stage=api.STAGE_HITS)
    c2 = r2.counts()
    assert c2["n_hits"] > 0 and c2["n_clusters_all"] == 0
    # emission order: (read, position) ascending
    h = r2.hits(sorted=False)
    key = h["read"].astype(np.int64) * (1 << 32) + h["pos"]
    assert (np.diff(key) >= 0).all()


def test_errors(gpu):
    rng = np.random.default_rng(12)
    si = synth.build_index([synth.random_dna(300, rng)], 14, 15)
    gidx = gpu_index(si)
    rd = api.Reads.from_list([b"ACGT" * 100])
    with pytest.raises(api.PmapError):
        api.map_reads(gidx, rd, 14, 16, 250, 0.1)       # k differs from the index
    with pytest.raises(api.PmapError):
        api.map_reads(gidx, rd, 0, 15, 250, 0.1)        # w = 0 (the reference asserts w != 0)
    # duplicate record: MinimizerHits::sort would abort in the reference
    with pytest.raises(api.PmapError) as e:
        api.Index.create(15, [5], [0, 2], [0, 0], [1, 2], [1, 1], [0, 1, 2], [3, 3], [15, 15], [4])
    assert e.value.code == 5
    with pytest.raises(api.PmapError):
        api.Index.create(32, [], [0], [], [], [], [0], [], [], [])


def test_async_batches_in_flight(gpu):
    """pmap_map_async / pmap_map_wait: several batches enqueued before any is waited for, results
    (including one that overflows its first buffers and is mapped again) equal to the blocking call."""
    si, bases, off = _small_world(21, 14, 15, n_loci=30, n_reads=80)
    gidx = gpu_index(si)
    frac = api.fraction(0.11, 15)
    parts = [(0, 10), (10, 50), (50, 80), (0, 80)]
    reads = [api.Reads(bases[int(off[a]):int(off[b])], off[a:b + 1] - off[a], first_read_id=a) for a, b in parts]
    want = []
    for rd in reads:
        r = api.map_reads(gidx, rd, 14, 15, 250, frac)
        want.append((r.counts(), r.hits(), r.clusters()))
    # small batch first: the carried-over capacities are too small for the batches after it
    flying = [api.map_reads(gidx, rd, 14, 15, 250, frac, wait=False) for rd in reads]
    for r, (cn, hits, cl) in zip(flying, want):
        got = r.wait().counts()
        for f in ("n_minimizers", "n_hits", "n_clusters_all", "n_clusters"):
            assert got[f] == cn[f]
        gh, gc = r.hits(), r.clusters()
        for f in hits:
            np.testing.assert_array_equal(gh[f], hits[f])
        for f in cl:
            np.testing.assert_array_equal(gc[f], cl[f])
    # one result object reused while its run is still in flight: the second call waits for the first
    r = api.map_reads(gidx, reads[3], 14, 15, 250, frac, wait=False)
    r = api.map_reads(gidx, reads[1], 14, 15, 250, frac, result=r, wait=False)
    assert r.counts()["n_hits"] == want[1][0]["n_hits"]


def test_random_small_worlds(gpu):
    """Many small random configurations: w, k, read lengths around the window size and around the
    tile size (480..512 positions), reads with bad letters, repeated loci, tiny and empty reads."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        k = int(rng.integers(3, 32))
        w = int(rng.integers(1, 65))
        n_loci = int(rng.integers(1, 6))
        loci = [synth.random_dna(int(rng.integers(k + w + 5, 700)), rng) for _ in range(n_loci)]
        if rng.random() < 0.5:
            loci.append(loci[0].copy())                      # a duplicated locus: keys with two records
        si = synth.build_index(loci, w, k, rng=rng, split_paths=float(rng.choice([0.0, 0.3])))
        reads = []
        for _ in range(int(rng.integers(1, 30))):
            kind = rng.integers(0, 6)
            if kind == 0:
                r = bytes(synth.random_dna(int(rng.integers(0, w + k + 3)), rng))       # around w+k-1
            elif kind == 1:
                r = bytes(synth.random_dna(int(rng.integers(440, 560)), rng))           # around one tile
            else:
                l = loci[int(rng.integers(0, len(loci)))]
                a = int(rng.integers(0, max(1, len(l) - 20)))
                r = bytes(l[a:a + int(rng.integers(20, 900))])
                if kind == 2:
                    r = bytes(synth.revcomp(np.frombuffer(r, np.uint8)))
                if kind == 3 and len(r) > 5:
                    p = int(rng.integers(0, len(r)))
                    r = r[:p] + b"N" + r[p + 1:]
                if kind == 4:
                    r = r.lower()
            reads.append(r)
        bases, off = O.pack_reads(reads)
        compare_batch(si, bases, off, w, k, int(rng.integers(1, 300)), api.fraction(float(rng.random()) * 0.2, k),
                      min_cluster_size=int(rng.integers(0, 4)), first_read_id=int(rng.integers(0, 1000)))


def test_register_window_kernels(gpu):
    """w = 14 with k <= 15 has three sketch kernels (16 or 8 positions per lane with the windows in
    registers, and the general one): compare_batch runs all of them.  Read lengths sit around the tile
    sizes (480 and 992 output positions), around w+k-1, and reads carry bad letters, so that tiles
    start and end inside, at and between reads in every alignment."""
    rng = np.random.default_rng(77)
    for case in range(24):
        k = int(rng.choice([15, 15, 13, 11, 8, 5]))
        loci = [synth.random_dna(int(rng.integers(400, 2500)), rng) for _ in range(int(rng.integers(1, 5)))]
        si = synth.build_index(loci, 14, k, rng=rng, split_paths=float(rng.choice([0.0, 0.3])))
        reads = []
        for _ in range(int(rng.integers(1, 40))):
            kind = int(rng.integers(0, 7))
            if kind == 0:
                r = bytes(synth.random_dna(int(rng.integers(0, 14 + k + 3)), rng))
            elif kind == 1:
                r = bytes(synth.random_dna(int(rng.choice([480, 992, 1024])) + int(rng.integers(-20, 21)), rng))
            else:
                l = loci[int(rng.integers(0, len(loci)))]
                a = int(rng.integers(0, max(1, len(l) - 30)))
                r = bytes(l[a:a + int(rng.integers(30, 2400))])
                if kind == 2:
                    r = bytes(synth.revcomp(np.frombuffer(r, np.uint8)))
                if kind == 3 and len(r) > 5:
                    p = int(rng.integers(0, len(r)))
                    r = r[:p] + b"N" + r[p + 1:]
                if kind == 4:
                    r = r * 2           # the same locus twice in one read
            reads.append(r)
        bases, off = O.pack_reads(reads)
        compare_batch(si, bases, off, 14, k, int(rng.integers(50, 300)), api.fraction(0.11, k),
                      min_cluster_size=int(rng.integers(0, 4)))


def test_sort_segment_sizes(gpu):
    """Reads with a few up to > 1024 hits next to each other (the sort works per read segment and stages
    segments through LDS in chunks).  Two loci that share most of their sequence make keys with two
    records, so that hits of both loci and both strands interleave in position order."""
    rng = np.random.default_rng(5)
    base = synth.random_dna(2400, rng)
    other = base.copy()
    other[::97] = synth.random_dna(len(other[::97]), rng)
    si = synth.build_index([base, other, synth.random_dna(900, rng)], 14, 15)
    reads = []
    for n_bp in (60, 150, 200, 330, 460, 520, 900, 1000, 1800, 2300, 3700, 4700, 7300, 8100, 40, 9000, 250):
        unit = bytes(base) if n_bp % 200 else bytes(synth.revcomp(base))
        r = (unit * (n_bp // len(unit) + 1))[:n_bp]
        reads.append(r)
    bases, off = O.pack_reads(reads)
    cn, orc = compare_batch(si, bases, off, 14, 15, 250, api.fraction(0.11, 15), min_cluster_size=2)
    per_read = np.bincount(orc["hit_read"], minlength=len(reads))
    assert per_read.max() > 1024 and (per_read < 24).any() and ((per_read > 512) & (per_read <= 1024)).any()


def test_presence_filter_modes(gpu, monkeypatch):
    """The presence filter in front of the table has one bit per key, two bits per key (dense indexes),
    is absent, or is a bitmap over the whole key range (k <= 15; what big indexes of sparse sketches get).
    PMAP_FILTER_MAX_BITS shrinks it so that a small index reaches every hashed mode and PMAP_FILTER_DIRECT
    chooses the bitmap; results must not depend on it."""
    si, bases, off = _small_world(31, 14, 15, n_loci=12, n_reads=40)
    seen = set()
    for max_bits, direct in (("24", "0"), ("16", "0"), ("14", "0"), ("12", "0"), ("12", "1"), ("12", None)):
        monkeypatch.setenv("PMAP_FILTER_MAX_BITS", max_bits)
        if direct is None:
            monkeypatch.delenv("PMAP_FILTER_DIRECT")     # the library's own choice: minimizer keys crowd the low end
        else:
            monkeypatch.setenv("PMAP_FILTER_DIRECT", direct)
        gidx = gpu_index(si)
        info = gidx.info()
        assert info["filter_direct"] == (direct != "0") and (not info["filter_direct"] or info["filter_bits"] == 30)
        seen.add((info["filter_bits_per_key"], info["filter_direct"]))
        compare_batch(si, bases, off, 14, 15, 250, api.fraction(0.11, 15), gidx=gidx)
        compare_batch(si, bases, off, 9, 15, 250, api.fraction(0.11, 15), gidx=gidx)     # general sketch kernel
        gidx.close()
    assert seen == {(0, 0), (1, 0), (2, 0), (1, 1)}, seen
    # keys below 2^(2k) for a smaller k: the bitmap has 2^(2k) bits
    si, bases, off = _small_world(32, 5, 7, n_loci=6, n_reads=30)
    monkeypatch.setenv("PMAP_FILTER_DIRECT", "1")
    gidx = gpu_index(si)
    assert gidx.info()["filter_direct"] == 1 and gidx.info()["filter_bits"] == 14
    compare_batch(si, bases, off, 5, 7, 30, api.fraction(0.11, 7), min_cluster_size=2, gidx=gidx)
    gidx.close()


def test_megabase_reads(gpu):
    """Reads of 100 kb and 1 Mb on an index with several records per key: 10^4 to 10^5 hits per read.  Such
    reads leave the rank-counting sort (a comparison per pair of hits of a read) for the long-segment
    kernels; shorter reads in the same batch stay where they are."""
    rng = np.random.default_rng(88)
    loci = synth.make_loci(150, rng, family_frac=0.5, family_max=8)
    vloci = synth.make_variant_loci(loci, rng)
    si = synth.build_variant_index(vloci, 14, 15)
    genome = synth.make_variant_genome(vloci, rng)
    big = np.concatenate([genome, synth.revcomp(genome), genome, genome])[:1_000_000]
    reads = [bytes(genome[:3000]), bytes(big), bytes(genome[1000:101_000]), bytes(synth.revcomp(genome[5000:9000])),
             bytes(big[200_000:700_000]), bytes(genome[:800])]
    bases, off = O.pack_reads(reads)
    cn, orc = compare_batch(si, bases, off, 14, 15, 250, api.fraction(0.11, 15))
    per_read = np.bincount(orc["hit_read"], minlength=len(reads))
    assert per_read[1] > 100_000 and 10_000 < per_read[2] < 100_000 and per_read[0] < 4096
    # what the sort costs with such reads in the batch (for the record: -s shows it)
    gidx = gpu_index(si)
    rd = api.Reads(bases, off)
    res = api.map_reads(gidx, rd, 14, 15, 250, api.fraction(0.11, 15), profile=1)
    res = api.map_reads(gidx, rd, 14, 15, 250, api.fraction(0.11, 15), profile=1, result=res)
    print("megabase reads: %d hits, kernel ms %s" % (res.counts()["n_hits"], res.kernel_ms()))
    assert res.kernel_ms()["tail"] < 20.0
    res = api.map_reads(gidx, rd, 14, 15, 250, api.fraction(0.11, 15), profile=1, flags=128)     # the separate sort kernels
    res = api.map_reads(gidx, rd, 14, 15, 250, api.fraction(0.11, 15), profile=1, result=res, flags=128)
    assert res.kernel_ms()["sort_rank"] < 20.0


def test_wavefront_sort_comes_and_goes(gpu):
    """The fused tail kernel packs reads of up to 128 hits, sorts reads of up to 1024 hits one per wavefront,
    leaves those of up to 2048 to a second kernel and longer ones to the long-segment kernels, which a result
    object launches only while its batches hold such reads.  The separate sort kernels (PMAP_FLAG_TAIL_STAGED)
    split at 192, 2048 and 4096 hits.  One result object of each kind over batches with and without long reads,
    every hit count from a few to beyond 4096 side by side: same hits and clusters as the oracle each time."""
    from parity import _check_hits_and_clusters, oracle_index
    rng = np.random.default_rng(77)
    loci = synth.make_loci(40, rng, family_frac=0.5, family_max=6)
    vloci = synth.make_variant_loci(loci, rng)
    si = synth.build_variant_index(vloci, 14, 15)
    genome = synth.make_variant_genome(vloci, rng)
    g2 = np.concatenate([genome, synth.revcomp(genome)])
    small = [bytes(g2[a:a + n]) for a, n in ((0, 400), (3000, 900), (9000, 1500), (100, 60), (20000, 2000))]
    mixed = [bytes(g2[a:a + n]) for a, n in ((0, 700), (500, 4000), (2000, 7000), (100, 11000), (7000, 16000),
                                              (1000, 23000), (300, 30000), (5000, 45000), (0, 250), (4000, 60000))]
    gidx = gpu_index(si)
    frac = api.fraction(0.11, 15)
    res = res_staged = None
    seen = set()
    for reads in (small, mixed, mixed, small, small, mixed):
        bases, off = O.pack_reads(reads)
        ids = np.arange(len(reads), dtype=np.uint32)
        orc = O.oracle_map(oracle_index(si), bases, off, ids, 14, 15, 250, frac, 10, 0xFFFFFFFF, si.min_path_length)
        per_read = np.bincount(orc["hit_read"], minlength=len(reads))
        seen.update("packed" if n <= 128 else "small" if n <= 192 else "wave" if n <= 1024 else "wave32" if n <= 2048
                    else "rank" if n <= 4096 else "long" for n in per_read)
        rd = api.Reads(bases, off)
        res = api.map_reads(gidx, rd, 14, 15, 250, frac, result=res)
        _check_hits_and_clusters(res, res.counts(), orc)
        res_staged = api.map_reads(gidx, rd, 14, 15, 250, frac, result=res_staged, flags=128)
        _check_hits_and_clusters(res_staged, res_staged.counts(), orc)
        rd.close()
    assert seen == {"packed", "small", "wave", "wave32", "rank", "long"}, seen
    res.close()
    res_staged.close()
    gidx.close()


def test_few_hits_per_read_switch_tails(gpu):
    """A result object whose last batch had fewer than 16 hits per read (many short reads under a sparse sketch)
    maps the next batch with the separate sort and cluster kernels, and goes back to the fused tail kernel after
    a batch with more hits per read; same hits and clusters as the oracle each time."""
    from parity import _check_hits_and_clusters, oracle_index
    rng = np.random.default_rng(91)
    loci = synth.make_loci(30, rng)
    genome = synth.make_genome(loci, rng)
    si = synth.build_index(loci, 14, 15)
    few = synth.simulate_reads(genome, 1500, rng, fixed_len=150, e_rate=0.001)
    many = synth.simulate_reads(genome, 1100, rng, fixed_len=700, e_rate=0.001)
    gidx = gpu_index(si)
    frac = api.fraction(0.001, 15)
    res = None
    tails = []
    for bases, off in (few, few, many, many, few):
        ids = np.arange(len(off) - 1, dtype=np.uint32)
        orc = O.oracle_map(oracle_index(si), bases, off, ids, 14, 15, 31, frac, 3, 30, si.min_path_length)
        rd = api.Reads(bases, off)
        res = api.map_reads(gidx, rd, 14, 15, 31, frac, min_cluster_size=3, expected_kmers=30, result=res, profile=1)
        cn = res.counts()
        _check_hits_and_clusters(res, cn, orc)
        per_read = cn["n_hits"] / (len(off) - 1)
        assert (per_read < 16) == (bases is few[0]), per_read
        ms = res.kernel_ms()
        tails.append("separate" if ms.get("sort_rank", 0) > 0 else "fused")
        assert (ms.get("tail", 0) > 0) != (ms.get("sort_rank", 0) > 0), ms
        rd.close()
    assert tails == ["fused", "separate", "separate", "fused", "fused"], tails
    res.close()
    gidx.close()


def test_chained_scan_fallback(gpu):
    """When a chained prefix sum of the cluster kernels gives up waiting, the batch is mapped again with one
    workgroup per kernel that carries the prefix itself; same clusters.  The test hook makes the chain give up
    at its first wait."""
    si, bases, off = _small_world(51, 14, 15, n_loci=60, n_reads=1500, mean_len=2000)
    gidx = gpu_index(si)
    rd = api.Reads(bases, off)
    frac = api.fraction(0.11, 15)
    want = api.map_reads(gidx, rd, 14, 15, 250, frac)
    assert want.counts()["n_hits"] > 3 * 8192 and want.counts()["n_clusters_all"] > 1024     # several blocks in each chain
    got = api.map_reads(gidx, rd, 14, 15, 250, frac, flags=16 | 128)       # (the chains belong to the separate kernels)
    assert got.counts()["ms"][5] >= 1, "the chain did not give up: the fallback did not run"
    for f in ("n_hits", "n_clusters_all", "n_clusters"):
        assert got.counts()[f] == want.counts()[f]
    a, b = want.clusters(), got.clusters()
    for f in a:
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    ha, hb = want.hits(), got.hits()
    for f in ha:
        np.testing.assert_array_equal(ha[f], hb[f], err_msg=f)


def test_two_processes_share_the_gpu(gpu, tmp_path):
    """Two fresh processes map on the same GPU at the same time (they once hung in their first call when the
    library gave its streams different priorities)."""
    import subprocess
    import sys
    import os
    script = tmp_path / "child.py"
    script.write_text('''
import sys
import numpy as np
sys.path.insert(0, %r)
import pandora_amd
from pandora_amd import api, synth
pandora_amd.init(0)
rng = np.random.default_rng(int(sys.argv[1]))
loci = synth.make_loci(40, rng)
si = synth.build_index(loci, 14, 15)
genome = synth.make_genome(loci, rng)
bases, off = synth.simulate_reads(genome, 300, rng, mean_len=4000, e_rate=0.08)
gidx = api.Index.create(si.k, si.keys, si.rec_off, si.rec_prg, si.rec_knode, si.rec_strand, si.path_off, si.iv_start,
                        si.iv_len, si.min_path_length)
rd = api.Reads(bases, off)
frac = api.fraction(0.11, 15)
first = api.map_reads(gidx, rd, 14, 15, 250, frac).counts()
ring = [None, None, None]
for i in range(600):
    j = i %% 3
    if ring[j] is not None:
        c = ring[j].wait().counts()
        assert (c["n_hits"], c["n_clusters"]) == (first["n_hits"], first["n_clusters"])
    ring[j] = api.map_reads(gidx, rd, 14, 15, 250, frac, result=ring[j], wait=False)
for r in ring:
    c = r.wait().counts()
    assert (c["n_hits"], c["n_clusters"]) == (first["n_hits"], first["n_clusters"])
print("child ok", first["n_hits"], int(ring[0].counts()["ms"][5]))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    procs = [subprocess.Popen([sys.executable, str(script), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for seed in (1, 2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=100)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two processes on one GPU: a child did not finish")
        outs.append(out.decode())
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "child ok" in o, o



def test_two_tail_streams(gpu):
    """With at least 6 hardware queues granted (GPU_MAX_HW_QUEUES, as bench.py sets it) a mapping context runs the
    kernels behind the sketch kernel of successive batches on two streams in turn.  The library reads that when it
    is loaded, so a fresh process runs the tests that keep several batches in flight, reuse result objects and map
    file after file; same results."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", PMAP_TAIL_STREAMS="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(root, "tests", "test_gpu_parity.py"), os.path.join(root, "tests", "test_gpu_pangraph.py"),
           "-k", "async_batches_in_flight or large_segment_and_reuse or nanopore_small or wavefront_sort_comes or "
                 "contexts_and_parsers or batch_size or coverage"]
    p = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and " passed" in out, out[-3000:]
