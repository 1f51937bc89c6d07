"""Deterministic synthetic inputs for tests and bench.py: random linear-locus PRGs, their
minimizer index (the records `pandora index` writes for a linear PRG: one record per sketch
minimizer of the locus, path [j, j+k), see test/localPRG_test.cpp:651-737 of the reference),
a genome built from the loci, and simulated nanopore / Illumina reads.

This is input generation (numpy on the host); nothing here is on the mapping path.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", np.uint8)
_CODE = np.full(256, 4, np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
    _CODE[_c + 32] = _i
for _i in range(4):
    _CODE[_i] = _i


def random_dna(n, rng):
    return _ACGT[rng.integers(0, 4, n, dtype=np.uint8)]


def revcomp(seq):
    return _ACGT[3 - _CODE[seq[::-1]]]


def hash64_np(key, mask):
    """Thomas Wang's 64-bit mix masked to 2k bits (src/inthash.cpp:115-125), vectorised."""
    key = key.astype(np.uint64)
    m = np.uint64(mask)
    with np.errstate(over="ignore"):
        key = (~key + (key << np.uint64(21))) & m
        key = key ^ (key >> np.uint64(24))
        key = ((key + (key << np.uint64(3))) + (key << np.uint64(8))) & m
        key = key ^ (key >> np.uint64(14))
        key = ((key + (key << np.uint64(2))) + (key << np.uint64(4))) & m
        key = key ^ (key >> np.uint64(28))
        key = (key + (key << np.uint64(31))) & m
    return key


def np_sketch(seq, w, k):
    """(positions, hashes, strands) of the (w,k)-minimizer sketch of one ASCII sequence, all ties kept."""
    seq = np.asarray(seq, np.uint8)
    n = len(seq) - k + 1
    empty = (np.zeros(0, np.int64), np.zeros(0, np.uint64), np.zeros(0, np.uint8))
    if n < w or n <= 0:
        return empty
    c = _CODE[seq].astype(np.uint64)
    if (c > 3).any():
        return empty
    fwd = np.zeros(n, np.uint64)
    rc = np.zeros(n, np.uint64)
    for i in range(k):
        fwd = (fwd << np.uint64(2)) | c[i:i + n]
        rc = rc | ((np.uint64(3) - c[i:i + n]) << np.uint64(2 * i))
    mask = (1 << (2 * k)) - 1
    hf, hr = hash64_np(fwd, mask), hash64_np(rc, mask)
    h = np.minimum(hf, hr)
    strand = (hf <= hr).astype(np.uint8)
    win = np.lib.stride_tricks.sliding_window_view(h, w).min(axis=1)          # minimum of window i, i in [0, n-w]
    pad = np.zeros(w - 1, np.uint64)
    padded = np.concatenate([pad, win, pad])                                    # windows containing j: padded[j .. j+w-1]
    best = np.lib.stride_tricks.sliding_window_view(padded, w).max(axis=1)
    flag = best == h
    pos = np.nonzero(flag)[0]
    return pos, h[pos], strand[pos]


class SynthIndex:
    """Flat arrays of an index (the layout pmap_index_desc_t and the oracle take)."""

    def __init__(self, k, w, keys, rec_off, rec_prg, rec_knode, rec_strand, path_off, iv_start, iv_len,
                 min_path_length):
        self.k, self.w = k, w
        self.keys = np.ascontiguousarray(keys, np.uint64)
        self.rec_off = np.ascontiguousarray(rec_off, np.uint64)
        self.rec_prg = np.ascontiguousarray(rec_prg, np.uint32)
        self.rec_knode = np.ascontiguousarray(rec_knode, np.uint32)
        self.rec_strand = np.ascontiguousarray(rec_strand, np.uint8)
        self.path_off = np.ascontiguousarray(path_off, np.uint64)
        self.iv_start = np.ascontiguousarray(iv_start, np.uint32)
        self.iv_len = np.ascontiguousarray(iv_len, np.uint32)
        self.min_path_length = np.ascontiguousarray(min_path_length, np.uint32)

    @property
    def n_prg(self):
        return len(self.min_path_length)


def build_index(loci, w, k, rng=None, split_paths=0.0):
    """Index of linear loci.  With split_paths > 0 that fraction of records gets its path [j, j+k)
    written as several intervals separated by zero-length or shifted pieces, the shape paths take in
    PRGs with variant sites (include/prg/path.h); coordinates stay unique per locus."""
    prg_l, key_l, pos_l, strand_l, knode_l = [], [], [], [], []
    mpl = np.zeros(len(loci), np.uint32)
    for p, locus in enumerate(loci):
        pos, h, s = np_sketch(locus, w, k)
        mpl[p] = len(pos) + 1          # linear kmer graph: #minimizer nodes + 1 (src/kmergraph.cpp:664-684)
        prg_l.append(np.full(len(pos), p, np.uint32))
        key_l.append(h)
        pos_l.append(pos.astype(np.uint32))
        strand_l.append(s)
        knode_l.append(np.arange(1, len(pos) + 1, dtype=np.uint32))
    prg = np.concatenate(prg_l) if prg_l else np.zeros(0, np.uint32)
    key = np.concatenate(key_l) if key_l else np.zeros(0, np.uint64)
    pos = np.concatenate(pos_l) if pos_l else np.zeros(0, np.uint32)
    strand = np.concatenate(strand_l) if strand_l else np.zeros(0, np.uint8)
    knode = np.concatenate(knode_l) if knode_l else np.zeros(0, np.uint32)
    # group by key (stable: records of one key stay in (prg, position) order, as add_record appends them)
    order = np.argsort(key, kind="stable")
    key, prg, pos, strand, knode = key[order], prg[order], pos[order], strand[order], knode[order]
    ukeys, first = np.unique(key, return_index=True)
    rec_off = np.concatenate([first, [len(key)]]).astype(np.uint64)
    n = len(key)
    if split_paths > 0 and n:
        rng = rng or np.random.default_rng(7)
        nsplit = np.where(rng.random(n) < split_paths, rng.integers(2, 5, n), 1)
        path_off = np.concatenate([[0], np.cumsum(nsplit)]).astype(np.uint64)
        ivs = np.zeros(int(path_off[-1]), np.uint32)
        ivl = np.zeros(int(path_off[-1]), np.uint32)
        for j in np.nonzero(nsplit > 1)[0]:
            m = int(nsplit[j])
            cuts = np.sort(rng.integers(0, k + 1, m - 1))
            lens = np.diff(np.concatenate([[0], cuts, [k]]))
            gaps = rng.integers(0, 4, m) * 5
            start = int(pos[j]) * 64       # spread so split paths of neighbouring k-mers do not collide
            o = int(path_off[j])
            for q in range(m):
                ivs[o + q] = start
                ivl[o + q] = lens[q]
                start += int(lens[q]) + int(gaps[q])
        single = np.nonzero(nsplit == 1)[0]
        ivs[path_off[single].astype(np.int64)] = pos[single] * 64
        ivl[path_off[single].astype(np.int64)] = k
    else:
        path_off = np.arange(n + 1, dtype=np.uint64)
        ivs = pos.astype(np.uint32)
        ivl = np.full(n, k, np.uint32)
    return SynthIndex(k, w, ukeys, rec_off, prg, knode, strand, path_off, ivs, ivl, mpl)


def make_loci(n_loci, rng, median_len=1000, min_len=300, max_len=5000, dup_frac=0.02, dup_div=0.01):
    """Random loci, lognormal length; dup_frac of them are diverged copies of another locus (shared k-mers)."""
    lens = np.clip(rng.lognormal(np.log(median_len), 0.35, n_loci), min_len, max_len).astype(np.int64)
    loci = [random_dna(int(l), rng) for l in lens]
    n_dup = int(n_loci * dup_frac)
    for _ in range(n_dup):
        a, b = rng.integers(0, n_loci, 2)
        if a == b:
            continue
        src = loci[a].copy()
        mut = rng.random(len(src)) < dup_div
        src[mut] = random_dna(int(mut.sum()), rng)
        loci[b] = src
    return loci


def make_genome(loci, rng, spacer=(200, 2000)):
    """One copy of every locus in random order and orientation, separated by random spacers."""
    parts = []
    for i in rng.permutation(len(loci)):
        parts.append(random_dna(int(rng.integers(spacer[0], spacer[1])), rng))
        parts.append(loci[i] if rng.random() < 0.5 else revcomp(loci[i]))
    parts.append(random_dna(int(rng.integers(spacer[0], spacer[1])), rng))
    return np.concatenate(parts)


def simulate_reads(genome, n_reads, rng, mean_len=10000, sigma=0.5, e_rate=0.11, fixed_len=None, n_frac=0.001,
                   short_frac=0.001, short_len=20):
    """Reads sampled uniformly from the genome.  fixed_len=150 gives Illumina-style reads (substitutions
    only).  n_frac of the reads get one 'N' (the reference drops such reads, src/seq.cpp:49-53),
    short_frac are shorter than any w+k-1."""
    G = len(genome)
    if n_reads == 0:
        return np.zeros(0, np.uint8), np.zeros(1, np.uint64)
    mu = np.log(mean_len) - sigma * sigma / 2
    if fixed_len:
        lens = np.full(n_reads, fixed_len, np.int64)
    else:
        lens = np.clip(rng.lognormal(mu, sigma, n_reads), 200, G - 1).astype(np.int64)
    lens[rng.random(n_reads) < short_frac] = short_len
    starts = (rng.random(n_reads) * (G - lens)).astype(np.int64)
    flip = rng.random(n_reads) < 0.5
    gcode = _CODE[genome]
    parts = []
    for s, L, f in zip(starts, lens, flip):
        c = gcode[s:s + L]
        parts.append(3 - c[::-1] if f else c)
    clean = _ACGT[np.concatenate(parts)]
    offsets = np.zeros(n_reads + 1, np.int64)
    offsets[1:] = np.cumsum(lens)
    n = len(clean)
    u = rng.random(n, dtype=np.float32)
    if fixed_len:
        sub = u < e_rate
        clean[sub] = random_dna(int(sub.sum()), rng)
        bases = clean
    else:
        # substitutions, insertions and deletions, a third each, over the whole batch at once
        sub = u < e_rate / 3
        ins = (u >= e_rate / 3) & (u < 2 * e_rate / 3)
        dele = (u >= 2 * e_rate / 3) & (u < e_rate)
        clean[sub] = random_dna(int(sub.sum()), rng)
        reps = np.ones(n, np.int8)
        reps[ins] = 2
        reps[dele] = 0
        bases = np.repeat(clean, reps)
        new_end = np.concatenate([[0], np.cumsum(reps, dtype=np.int64)])
        # the first copy of every doubled base becomes the inserted random base
        bases[new_end[:-1][ins]] = random_dna(int(ins.sum()), rng)
        offsets = new_end[offsets]
    # one 'N' in a few reads
    for r in np.nonzero(rng.random(n_reads) < n_frac)[0]:
        if offsets[r + 1] > offsets[r]:
            bases[int(rng.integers(offsets[r], offsets[r + 1]))] = ord("N")
    return bases, offsets.astype(np.uint64)


def workload(name, seed=1, scale=1.0, shard=0):
    """Named synthetic workloads.  Returns (SynthIndex, bases, offsets, params dict).

    nanopore_1k   1 000 loci, 10 000 reads of ~10 kb at 11 % error, w=14 k=15 (scale multiplies reads)
    illumina_1k   1 000 loci, 150 bp reads at 0.1 % substitutions, w=14 k=15, illumina thresholds
    illumina_w1   the same with w=1 (dense minimizers; the reference's threshold rejects every cluster)
    nanopore_20k  20 000 loci, 15 000 nanopore reads (30x of the genome), w=14 k=15
    illumina_20k_w1  20 000 loci, 400 000 reads of 150 bp, w=1 k=15
    nanopore_200k 200 000 loci (27 M keys, multi-GB table), 15 000 nanopore reads, w=14 k=15

    `shard` selects an independent batch of reads over the same index and genome (one per GPU rank).
    """
    rng = np.random.default_rng(seed)
    if name == "nanopore_1k":
        w, k, n_loci, n_reads = 14, 15, 1000, int(10000 * scale)
        loci = make_loci(n_loci, rng)
        idx = build_index(loci, w, k)
        genome = make_genome(loci, np.random.default_rng(seed + 1))
        bases, off = simulate_reads(genome, n_reads, np.random.default_rng([seed + 2, shard]))
        prm = dict(w=w, k=k, max_diff=250, e_rate=0.11, min_cluster_size=10, illumina=False)
    elif name in ("illumina_1k", "illumina_w1"):
        w = 1 if name == "illumina_w1" else 14
        k, n_loci, n_reads = 15, 1000, int(200000 * scale)
        loci = make_loci(n_loci, rng)
        idx = build_index(loci, w, k)
        genome = make_genome(loci, np.random.default_rng(seed + 1))
        bases, off = simulate_reads(genome, n_reads, np.random.default_rng([seed + 3, shard]), fixed_len=150, e_rate=0.001)
        prm = dict(w=w, k=k, max_diff=2 * k + 1, e_rate=0.001, min_cluster_size=10, illumina=True)
    elif name in ("nanopore_20k", "illumina_20k_w1", "nanopore_200k"):
        # E. coli-scale index: 20 000 loci, a 128 MB slot table (beyond every cache but the Infinity Cache);
        # multi-species scale: 200 000 loci, a 2 GB slot table in HBM
        ill = name == "illumina_20k_w1"
        w, k, n_loci = (1 if ill else 14), 15, (200000 if name == "nanopore_200k" else 20000)
        loci = make_loci(n_loci, rng)
        idx = build_index(loci, w, k)
        genome = make_genome(loci, np.random.default_rng(seed + 1))
        if ill:
            bases, off = simulate_reads(genome, int(400000 * scale), np.random.default_rng([seed + 3, shard]),
                                        fixed_len=150, e_rate=0.001)
            prm = dict(w=w, k=k, max_diff=2 * k + 1, e_rate=0.001, min_cluster_size=10, illumina=True)
        else:
            bases, off = simulate_reads(genome, int(15000 * scale), np.random.default_rng([seed + 2, shard]))
            prm = dict(w=w, k=k, max_diff=250, e_rate=0.11, min_cluster_size=10, illumina=False)
    else:
        raise ValueError(name)
    return idx, bases, off, prm
