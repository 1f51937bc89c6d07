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


def _sliding(v, w, op):
    """op over every window of w consecutive elements, by doubling: result[i] = op(v[i .. i+w-1])."""
    cover = 1
    while 2 * cover <= w:
        v = op(v[:-cover], v[cover:])
        cover *= 2
    if cover < w:
        v = op(v[:len(v) - (w - cover)], v[w - cover:])
    return v


def _kmers(c, k, n):
    """Forward and reverse-complement k-mers (2 bits per base) at the first n positions of the code array c,
    built by doubling: pieces of 1, 2, 4, ... bases are joined into k."""
    c = c.astype(np.uint64)
    f = {1: c}
    r = {1: np.uint64(3) - c}
    size = 1
    while 2 * size <= k:
        m = len(f[size]) - size
        f[2 * size] = (f[size][:m] << np.uint64(2 * size)) | f[size][size:size + m]
        r[2 * size] = r[size][:m] | (r[size][size:size + m] << np.uint64(2 * size))
        size *= 2
    fwd, rc, have = None, None, 0
    while size >= 1:
        if have + size <= k:
            if fwd is None:
                fwd, rc = f[size][:n], r[size][:n]
            else:
                fwd = (fwd << np.uint64(2 * size)) | f[size][have:have + n]
                rc = rc | (r[size][have:have + n] << np.uint64(2 * have))
            have += size
        size //= 2
    return fwd, rc


def _sketch_chunk(seqs, first, w, k):
    mask = (1 << (2 * k)) - 1
    lens = np.array([len(x) for x in seqs], np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)])
    cat = np.concatenate([np.asarray(x, np.uint8) for x in seqs]) if lens.sum() else np.zeros(0, np.uint8)
    n = len(cat) - k + 1
    if n < w or n <= 0:
        return None
    c = _CODE[cat]
    bad = np.concatenate([[0], np.cumsum(c > 3)])
    fwd, rc = _kmers(c & 3, k, n)
    hf, hr = hash64_np(fwd, mask), hash64_np(rc, mask)
    del fwd, rc
    h = np.minimum(hf, hr)
    strand = hf <= hr
    del hf, hr
    loc = np.repeat(np.arange(len(seqs)), lens)                   # sequence of every base
    ok_seq = (lens + 1 >= w + k) & (bad[starts[1:]] == bad[starts[:-1]])
    kvalid = (loc[:n] == loc[k - 1:]) & ok_seq[loc[:n]]           # a k-mer of one sketchable sequence
    win = _sliding(h, w, np.minimum)                              # min of window i = k-mers i .. i+w-1
    if win is h:
        win = h.copy()
    wvalid = kvalid[:len(win)] & kvalid[w - 1:] & (loc[:len(win)] == loc[w - 1:w - 1 + len(win)])
    win[~wvalid] = 0
    pad = np.zeros(w - 1, np.uint64)
    best = _sliding(np.concatenate([pad, win, pad]), w, np.maximum)  # over the windows that contain k-mer j
    at = np.nonzero(kvalid & (best == h))[0]
    return loc[at] + first, at - starts[loc[at]], h[at], strand[at].astype(np.uint8)


def np_sketch_many(seqs, w, k, chunk_bases=1 << 23, threads=None):
    """np_sketch of every sequence of a list, computed over concatenated chunks (a few numpy passes per chunk
    instead of per sequence; chunks on several threads, numpy releases the interpreter lock).  Returns flat
    arrays (sequence index, position, hash, strand) in (sequence, position) order."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    chunks = []
    i = 0
    while i < len(seqs):
        j, tot = i, 0
        while j < len(seqs) and (j == i or tot + len(seqs[j]) <= chunk_bases):
            tot += len(seqs[j])
            j += 1
        chunks.append((i, j))
        i = j
    if threads is None:
        try:
            threads = min(len(os.sched_getaffinity(0)), 16)
        except Exception:
            threads = 4
    with ThreadPoolExecutor(max(1, min(threads, len(chunks) or 1))) as ex:
        parts = [p for p in ex.map(lambda ab: _sketch_chunk(seqs[ab[0]:ab[1]], ab[0], w, k), chunks) if p is not None]
    if not parts:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.uint64), np.zeros(0, np.uint8)
    return tuple(np.concatenate([p[f] for p in parts]) for f in range(4))


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
    seq_i, pos, key, strand = np_sketch_many(loci, w, k)
    per_locus = np.bincount(seq_i, minlength=len(loci)).astype(np.int64)
    mpl = (per_locus + 1).astype(np.uint32)    # linear kmer graph: #minimizer nodes + 1 (src/kmergraph.cpp:664-684)
    prg = seq_i.astype(np.uint32)
    pos = pos.astype(np.uint32)
    first = np.concatenate([[0], np.cumsum(per_locus)])[:-1]
    knode = (np.arange(len(prg), dtype=np.int64) - first[seq_i] + 1).astype(np.uint32)
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


class VariantLocus:
    """A locus with variant sites, the shape of a PRG with one level of bubbles: plain sequence pieces
    alternate with sites, each site a list of alleles (an allele may be empty).  Coordinates are those of
    the PRG string `pre 5 allele 6 allele 5 post ...` (test/test_cases/prg0123.fa), markers included."""

    def __init__(self, plains, sites):
        assert len(plains) == len(sites) + 1
        self.plains, self.sites = plains, sites

    def haplotype(self, choice):
        """Sequence with allele choice[i] (clipped to the site's allele count) at site i."""
        parts = [self.plains[0]]
        for i, al in enumerate(self.sites):
            parts.append(al[min(int(choice[i]), len(al) - 1)])
            parts.append(self.plains[i + 1])
        return np.concatenate(parts)

    def prg_string(self):
        out = [bytes(self.plains[0])]
        for i, al in enumerate(self.sites):
            odd, even = b" %d " % (5 + 2 * i), b" %d " % (6 + 2 * i)
            out.append(odd + even.join(bytes(a) for a in al) + odd)
            out.append(bytes(self.plains[i + 1]))
        return b"".join(out)

    def segments(self, h):
        """(bases, prg_start) of every piece of haplotype h (allele min(h, n-1) at every site), in order."""
        segs, c = [(self.plains[0], 0)], len(self.plains[0])
        for i, al in enumerate(self.sites):
            mo, me = len(str(5 + 2 * i)) + 2, len(str(6 + 2 * i)) + 2
            c += mo
            pick = min(h, len(al) - 1)
            for a, allele in enumerate(al):
                if a == pick:
                    segs.append((allele, c))
                c += len(allele) + (me if a + 1 < len(al) else mo)
            segs.append((self.plains[i + 1], c))
            c += len(self.plains[i + 1])
        return segs


def make_variant_loci(loci, rng, var_frac=0.8, site_every=100, min_gap=48, max_alleles=3):
    """Turns linear loci into loci with variant sites on var_frac of them: a site about every site_every
    bases (never closer than min_gap, so that a (w, k) window with w + k - 1 <= min_gap meets at most one
    site), SNPs mostly, some short insertions and deletions, two or three alleles."""
    out = []
    for seq in loci:
        L = len(seq)
        if rng.random() >= var_frac or L < 3 * min_gap:
            out.append(VariantLocus([seq], []))
            continue
        plains, sites, at = [], [], 0
        p = int(rng.integers(min_gap, min_gap + site_every))
        while p + 4 + min_gap < L:
            kind = rng.random()
            if kind < 0.7:                                  # SNP
                ref = seq[p:p + 1]
                alts = [b for b in _ACGT if b != ref[0]]
                rng.shuffle(alts)
                n_al = 2 if rng.random() < 0.8 else min(3, max_alleles)
                alleles = [ref.copy()] + [np.array([alts[i]], np.uint8) for i in range(n_al - 1)]
                used = 1
            elif kind < 0.85:                               # insertion after one reference base
                ref = seq[p:p + 1]
                alleles = [ref.copy(), np.concatenate([ref, random_dna(int(rng.integers(1, 4)), rng)])]
                used = 1
            else:                                           # deletion of 1..3 bases: the other allele is empty
                used = int(rng.integers(1, 4))
                alleles = [seq[p:p + used].copy(), np.zeros(0, np.uint8)]
            plains.append(seq[at:p].copy())
            sites.append(alleles)
            at = p + used
            p = at + int(rng.integers(min_gap, 2 * site_every - min_gap + 1))
        plains.append(seq[at:].copy())
        out.append(VariantLocus(plains, sites))
    return out


def build_variant_index(vloci, w, k, max_alleles=3):
    """Index of loci with variant sites: the records `pandora index` writes are the minimizer k-mers of every
    path through the PRG (src/localPRG.cpp:333-540).  With sites at least w + k - 1 apart a window meets at
    most one site, so the windows of the haplotypes "allele h at every site", h = 0 .. max_alleles-1, are all
    the windows there are: the index is the union of those haplotypes' sketches, each k-mer with its path in
    PRG coordinates (one interval per piece it touches; an empty allele inside gives a zero-length
    interval), duplicates of the same (prg, path, strand) dropped as Index::add_record does
    (src/index.cpp:20-37)."""
    n_loci = len(vloci)
    hap_seqs, seg_hap_start, seg_len, seg_prg, seg_off, base_seg = [], [], [], [], [0], []
    for v in vloci:
        for h in range(max_alleles if v.sites else 1):
            segs = v.segments(h)
            lens = np.array([len(b) for b, _ in segs], np.int64)
            hap_seqs.append(np.concatenate([b for b, _ in segs]))
            seg_hap_start.append(np.concatenate([[0], np.cumsum(lens)[:-1]]))
            seg_len.append(lens)
            seg_prg.append(np.array([c for _, c in segs], np.int64))
            base_seg.append(np.repeat(np.arange(len(segs), dtype=np.int32) + np.int32(seg_off[-1]), lens))
            seg_off.append(seg_off[-1] + len(segs))
    hap_locus = np.concatenate([np.full(max_alleles if v.sites else 1, i) for i, v in enumerate(vloci)]) \
        if n_loci else np.zeros(0, np.int64)
    hap_first = np.concatenate([[1] + [0] * ((max_alleles if v.sites else 1) - 1) for v in vloci]) \
        if n_loci else np.zeros(0, np.int64)
    base_off = np.concatenate([[0], np.cumsum([len(x) for x in hap_seqs])]).astype(np.int64)
    seg_hap_start = np.concatenate(seg_hap_start) if seg_hap_start else np.zeros(0, np.int64)
    seg_len = np.concatenate(seg_len) if seg_len else np.zeros(0, np.int64)
    seg_prg = np.concatenate(seg_prg) if seg_prg else np.zeros(0, np.int64)
    base_seg = np.concatenate(base_seg) if base_seg else np.zeros(0, np.int32)

    q, pos, key, strand = np_sketch_many(hap_seqs, w, k)
    prg = hap_locus[q]
    s0 = base_seg[base_off[q] + pos].astype(np.int64)
    s1 = base_seg[base_off[q] + pos + k - 1].astype(np.int64)
    n_iv = s1 - s0 + 1
    assert len(n_iv) == 0 or n_iv.max() <= 3, "sites closer than k: a k-mer touches more than three pieces"
    c0 = seg_prg[s0] + (pos - seg_hap_start[s0])
    c_last = seg_prg[s1] + (pos + k - 1 - seg_hap_start[s1])
    mid = np.where(n_iv == 3, seg_prg[np.minimum(s0 + 1, len(seg_prg) - 1)], 0xFFFFFFFF)
    # drop the k-mers that several haplotypes share: same prg, same path, same strand (then also the same key)
    a = (prg.astype(np.uint64) << np.uint64(32)) | c0.astype(np.uint64)
    b = (c_last.astype(np.uint64) << np.uint64(33)) | (mid.astype(np.uint64) << np.uint64(1)) | strand.astype(np.uint64)
    order = np.lexsort((b, a))
    keep = np.ones(len(order), bool)
    keep[1:] = (a[order][1:] != a[order][:-1]) | (b[order][1:] != b[order][:-1])
    sel = order[keep]                                       # in (prg, path) order
    q, pos, key, strand, prg, s0, s1, n_iv, c0 = (x[sel] for x in (q, pos, key, strand, prg, s0, s1, n_iv, c0))
    per_locus = np.bincount(prg, minlength=n_loci).astype(np.int64)
    first = np.concatenate([[0], np.cumsum(per_locus)])[:-1]
    knode = (np.arange(len(prg), dtype=np.int64) - first[prg] + 1).astype(np.uint32)
    # min_path_length is an input of the path: here the node count of the first haplotype + 1, what the
    # reference's computation gives for a linear graph
    n_first = np.bincount(prg[hap_first[q] == 1], minlength=n_loci)
    mpl = (n_first + 1).astype(np.uint32)
    # intervals
    n = len(prg)
    path_off = np.concatenate([[0], np.cumsum(n_iv)]).astype(np.int64)
    ivs = np.zeros(int(path_off[-1]), np.uint32)
    ivl = np.zeros(int(path_off[-1]), np.uint32)
    for t in range(3):
        m = n_iv > t
        seg = s0[m] + t
        lo = np.maximum(pos[m], seg_hap_start[seg])
        hi = np.minimum(pos[m] + k, seg_hap_start[seg] + seg_len[seg])
        ivs[path_off[:-1][m] + t] = seg_prg[seg] + (lo - seg_hap_start[seg])
        ivl[path_off[:-1][m] + t] = hi - lo
    # group by key
    order = np.argsort(key, kind="stable")
    ukeys, firstk = np.unique(key[order], return_index=True)
    rec_off = np.concatenate([firstk, [n]]).astype(np.uint64)
    new_off = np.concatenate([[0], np.cumsum(n_iv[order])]).astype(np.int64)
    gather = np.repeat(path_off[:-1][order] - new_off[:-1], n_iv[order]) + np.arange(int(new_off[-1]))
    return SynthIndex(k, w, ukeys, rec_off, prg[order].astype(np.uint32), knode[order], strand[order],
                      new_off.astype(np.uint64), ivs[gather], ivl[gather], mpl)


def make_variant_genome(vloci, rng, spacer=(200, 2000)):
    """As make_genome, with a random allele at every site of every locus."""
    loci = [v.haplotype(rng.integers(0, 3, len(v.sites))) for v in vloci]
    return make_genome(loci, rng, spacer)


def make_loci(n_loci, rng, median_len=1000, min_len=300, max_len=5000, dup_frac=0.02, dup_div=0.01,
              family_frac=0.0, family_max=10, family_div=0.005):
    """Random loci, lognormal length; dup_frac of them are diverged copies of another locus (shared k-mers).
    family_frac of the loci come in families of 2..family_max diverged copies of one founder (paralogous
    genes, insertion sequences): their k-mers are index keys with several records."""
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
    at, n_fam = 0, int(n_loci * family_frac)
    while at + 1 < n_fam:
        size = min(int(rng.integers(2, family_max + 1)), n_fam - at)
        for m in range(1, size):
            src = loci[at].copy()
            mut = rng.random(len(src)) < family_div
            src[mut] = random_dna(int(mut.sum()), rng)
            loci[at + m] = src
        at += size
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


_BIG_CACHE = {}


def workload(name, seed=1, scale=1.0, shard=0):
    """Named synthetic workloads.  Returns (SynthIndex, bases, offsets, params dict).

    nanopore_1k       1 000 loci, 80 % of them with variant sites about every 100 bases (SNPs, short indels:
                      index paths of one to three intervals), 2 % diverged duplicates; 10 000 reads of ~10 kb
                      at 11 % error sampled from a genome with a random allele at every site; w=14 k=15
                      (scale multiplies the reads)
    nanopore_1k_linear    the same loci without variant sites (every path one interval)
    nanopore_1k_families  variant sites, and 50 % of the loci in families of 2..10 paralogs: 1.5 records per
                      index key on average, up to 10
    illumina_1k       the nanopore_1k index, 150 bp reads at 0.1 % substitutions, w=14 k=15, illumina thresholds
    illumina_w1       the same with w=1 (dense minimizers; the reference's threshold rejects every cluster)
    nanopore_20k      20 000 loci with variant sites, 15 000 nanopore reads (30x of the genome), w=14 k=15
    illumina_20k_w1   20 000 loci, 400 000 reads of 150 bp, w=1 k=15
    nanopore_200k     200 000 loci, 80 % of them with a SNP site about every 100 bases (build_snp_index: paths of one
                      to three intervals), multi-GB table, 15 000 nanopore reads, w=14 k=15
    nanopore_200k_linear  the same loci without sites
    nanopore_1k_8g    the nanopore_1k index plus 7e7 keys spread evenly over the key range, one record each, on
                      199 000 more PRGs (add_filler_records): a table of 2^28 slots (8 GB), no presence filter;
                      the nanopore_1k reads (scale multiplies the extra keys as well as the reads)

    `shard` selects an independent batch of reads over the same index and genome (one per GPU rank).
    """
    rng = np.random.default_rng(seed)
    ill = name.startswith("illumina")
    w = 1 if name.endswith("_w1") else 14
    k = 15
    n_loci = 200000 if "200k" in name else 20000 if "20k" in name else 1000
    linear = name.endswith("_linear")
    if name not in ("nanopore_1k", "nanopore_1k_linear", "nanopore_1k_families", "illumina_1k", "illumina_w1",
                    "nanopore_20k", "nanopore_20k_linear", "illumina_20k_w1", "nanopore_200k", "nanopore_200k_linear",
                    "nanopore_1k_8g"):
        raise ValueError(name)
    # (a 200 000-locus index takes a minute to make: the last one is kept for the next call with the same name and seed)
    cached = _BIG_CACHE.get((name, seed))
    if cached is not None:
        idx, genome = cached
        loci = None
    else:
        loci = make_loci(n_loci, rng, family_frac=0.5 if name.endswith("_families") else 0.0)
    if cached is not None:
        pass
    elif linear:
        idx = build_index(loci, w, k)
        genome = make_genome(loci, np.random.default_rng(seed + 1))
    elif n_loci >= 100000:
        idx, alt_loci, sites = build_snp_index(loci, w, k, np.random.default_rng(seed + 5))
        genome = snp_genome(loci, alt_loci, sites, np.random.default_rng(seed + 1))
    else:
        vloci = make_variant_loci(loci, np.random.default_rng(seed + 5))
        idx = build_variant_index(vloci, w, k)
        genome = make_variant_genome(vloci, np.random.default_rng(seed + 1))
    if cached is None and n_loci >= 100000:
        _BIG_CACHE.clear()
        _BIG_CACHE[(name, seed)] = (idx, genome)
    if name.endswith("_8g"):
        idx = add_filler_records(idx, int(7e7 * scale), 200000, np.random.default_rng(seed + 7))
    if ill:
        n_reads = int((400000 if "20k" in name else 200000) * scale)
        bases, off = simulate_reads(genome, n_reads, np.random.default_rng([seed + 3, shard]), fixed_len=150,
                                    e_rate=0.001)
        prm = dict(w=w, k=k, max_diff=2 * k + 1, e_rate=0.001, min_cluster_size=10, illumina=True)
    else:
        n_reads = int((10000 if n_loci == 1000 else 15000) * scale)
        bases, off = simulate_reads(genome, n_reads, np.random.default_rng([seed + 2, shard]))
        prm = dict(w=w, k=k, max_diff=250, e_rate=0.11, min_cluster_size=10, illumina=False)
    return idx, bases, off, prm


# ---------------------------------------------------------------------------------------------------------
# Big indexes.  make_variant_loci / build_variant_index walk every locus and site in Python: fine for thousands
# of loci, minutes for 200 000.  The two generators below work on whole arrays.

def build_snp_index(loci, w, k, rng, var_frac=0.8, site_every=100, min_gap=48):
    """Loci with biallelic SNP sites on var_frac of them, a site about every site_every bases and never closer
    than min_gap (>= w + k - 1: a window meets at most one site), as the PRG string
    `pre 5 A 6 C 5 post 7 G 8 T 7 ...` (test/test_cases/prg0123.fa).  Returns (SynthIndex, alt_loci): the index
    holds the minimizer k-mers of every path (src/localPRG.cpp:333-540) - with one site per window those of
    the haplotype with the first allele everywhere and of the one with the second allele everywhere - each
    with its path in PRG coordinates: one interval, or three (piece, allele, piece) for a k-mer over a site;
    records of the same (prg, path, strand) once, as Index::add_record keeps them (src/index.cpp:20-37).
    alt_loci[i] is locus i with the second allele at every site (a genome picks per site, see snp_genome)."""
    assert min_gap >= w + k - 1
    n_loci = len(loci)
    lens = np.array([len(x) for x in loci], np.int64)
    base_off = np.concatenate([[0], np.cumsum(lens)])
    ref = np.concatenate(loci) if n_loci else np.zeros(0, np.uint8)
    total = int(base_off[-1])
    # sites: candidates at random gaps of min_gap .. 2 site_every - min_gap along the concatenation, kept where
    # they lie inside one varied locus, min_gap away from its ends
    n_cand = int(total / site_every * 1.3) + 16
    cand = np.cumsum(rng.integers(min_gap, 2 * site_every - min_gap + 1, n_cand))
    cand = cand[cand < total]
    loc = np.searchsorted(base_off, cand, side="right") - 1
    varied = rng.random(n_loci) < var_frac
    inside = cand - base_off[loc]
    ok = varied[loc] & (inside >= min_gap) & (inside + 1 + min_gap < lens[loc])
    site_pos, site_loc = cand[ok], loc[ok]                       # position in the concatenation, locus
    alt = ref.copy()
    alt[site_pos] = _ACGT[(_CODE[ref[site_pos]] + rng.integers(1, 4, len(site_pos))) & 3]
    # PRG coordinates.  Site number s (0-based inside its locus) is written " (5+2s) R (6+2s) A (5+2s) ": the
    # first allele starts after the odd marker, the second after first allele + even marker, the sequence
    # goes on after the closing odd marker
    first_site = np.searchsorted(site_loc, np.arange(n_loci), side="left")
    s_no = np.arange(len(site_pos)) - first_site[site_loc]
    def digits(v):
        return 1 + (v >= 10).astype(np.int64) + (v >= 100) + (v >= 1000) + (v >= 10000)

    odd = digits(5 + 2 * s_no) + 2
    even = digits(6 + 2 * s_no) + 2
    extra_before_allele0 = odd                                   # marker text before the first allele
    extra_before_allele1 = odd + 1 + even
    extra_total = 2 * odd + even + 1                             # everything a site adds to the string
    # PRG coordinate of a base outside any site: its offset in the locus + what the sites of its locus before it
    # have added (looked up per position asked for, never as arrays over all bases)
    site_cum = np.concatenate([[0], np.cumsum(extra_total)])        # exclusive prefix over the sites

    def coord_plain(x, x_locus, sites_before):
        return x - base_off[x_locus] + site_cum[sites_before] - site_cum[first_site[x_locus]]

    recs = []
    for h, seq in ((0, ref), (1, alt)):
        hap = [seq[base_off[i]:base_off[i + 1]] for i in range(n_loci)]
        q, pos, key, strand = np_sketch_many(hap, w, k)
        g = base_off[q] + pos                                    # first base of the k-mer in the concatenation
        s = np.searchsorted(site_pos, g, side="left")            # the first site at or after it
        over = (s < len(site_pos)) & (site_pos[np.minimum(s, len(site_pos) - 1)] < g + k)      # the k-mer covers site s
        recs.append((h, q, pos, key, strand, g, s, over))
    prg = np.concatenate([r[1] for r in recs]).astype(np.int64)
    hcol = np.concatenate([np.full(len(r[1]), r[0], np.int64) for r in recs])
    key = np.concatenate([r[3] for r in recs])
    strand = np.concatenate([r[4] for r in recs])
    g = np.concatenate([r[5] for r in recs])
    s = np.concatenate([r[6] for r in recs])
    over = np.concatenate([r[7] for r in recs])
    # a k-mer away from any site has the same path in both haplotypes (it may be a minimizer in one of them only:
    # its window can reach a site): one record
    ident = (g.astype(np.uint64) << np.uint64(2)) | (strand.astype(np.uint64) << np.uint64(1)) | (over & (hcol == 1)).astype(np.uint64)
    _, keep = np.unique(ident, return_index=True)
    prg, hcol, key, strand, g, s, over = (x[keep] for x in (prg, hcol, key, strand, g, s, over))
    n = len(prg)
    sp = site_pos[np.minimum(s, max(len(site_pos) - 1, 0))] if len(site_pos) else np.zeros(n, np.int64)
    n_iv = np.where(over, 3, 1)
    # a k-mer whose first (last) base is the site itself has an empty first (last) piece: the reference keeps
    # no empty interval at the ends of a path, only inside (src/localPRG.cpp:333-540 builds paths from nodes it
    # walked), so such k-mers have two intervals
    starts_at = over & (sp == g)
    ends_at = over & (sp == g + k - 1)
    n_iv = n_iv - starts_at - ends_at
    path_off = np.concatenate([[0], np.cumsum(n_iv)]).astype(np.int64)
    ivs = np.zeros(int(path_off[-1]), np.uint32)
    ivl = np.zeros(int(path_off[-1]), np.uint32)
    plain = ~over
    ivs[path_off[:-1][plain]] = coord_plain(g[plain], prg[plain], s[plain])
    ivl[path_off[:-1][plain]] = k
    o = np.nonzero(over)[0]
    if len(o):
        so, go, spo, ho, po = s[o], g[o], sp[o], hcol[o], prg[o]
        at = path_off[:-1][o]
        pre_len = spo - go                                       # bases before the site
        post_len = go + k - 1 - spo                              # bases after it
        site_coord = coord_plain(spo, po, so)                    # where the site's text starts
        allele_coord = site_coord + np.where(ho == 0, extra_before_allele0[so], extra_before_allele1[so])
        post_coord = site_coord + extra_total[so] + 1
        has_pre, has_post = pre_len > 0, post_len > 0
        ivs[at[has_pre]] = coord_plain(go[has_pre], po[has_pre], so[has_pre])
        ivl[at[has_pre]] = pre_len[has_pre]
        a2 = at + has_pre
        ivs[a2] = allele_coord
        ivl[a2] = 1
        a3 = a2 + 1
        ivs[a3[has_post]] = post_coord[has_post]
        ivl[a3[has_post]] = post_len[has_post]
    # per locus: records in (prg, path) order get the kmer node ids 1, 2, ...; min_path_length as for a linear
    # graph: nodes of the first haplotype + 1
    first_iv = ivs[path_off[:-1]].astype(np.uint64)
    order = np.lexsort((hcol, first_iv, prg))
    per_locus = np.bincount(prg, minlength=n_loci).astype(np.int64)
    first = np.concatenate([[0], np.cumsum(per_locus)])[:-1]
    knode = np.zeros(n, np.uint32)
    knode[order] = (np.arange(n, dtype=np.int64) - first[prg[order]] + 1).astype(np.uint32)
    mpl = (np.bincount(prg[hcol == 0], minlength=n_loci) + 1).astype(np.uint32)
    # group by key
    order = np.argsort(key, kind="stable")
    ukeys, firstk = np.unique(key[order], return_index=True)
    rec_off = np.concatenate([firstk, [n]]).astype(np.uint64)
    new_off = np.concatenate([[0], np.cumsum(n_iv[order])]).astype(np.int64)
    gather = np.repeat(path_off[:-1][order] - new_off[:-1], n_iv[order]) + np.arange(int(new_off[-1]))
    si = SynthIndex(k, w, ukeys, rec_off, prg[order].astype(np.uint32), knode[order], strand[order],
                    new_off.astype(np.uint64), ivs[gather], ivl[gather], mpl)
    alt_loci = [alt[base_off[i]:base_off[i + 1]] for i in range(n_loci)]
    return si, alt_loci, (site_pos, base_off)


def snp_genome(loci, alt_loci, sites, rng, spacer=(200, 2000)):
    """As make_genome, with a random allele at every SNP site of every locus."""
    site_pos, base_off = sites
    ref = np.concatenate(loci)
    alt = np.concatenate(alt_loci)
    pick = rng.random(len(site_pos)) < 0.5
    mixed = ref.copy()
    mixed[site_pos[pick]] = alt[site_pos[pick]]
    return make_genome([mixed[base_off[i]:base_off[i + 1]] for i in range(len(loci))], rng, spacer)


def add_filler_records(si, n_fill, n_prg_total, rng, filler_mpl=100):
    """The index `si` plus n_fill more keys spread evenly over the key range, one record each, on PRGs
    [si.n_prg, n_prg_total): what makes a table of many gigabytes out of a small index without sketching
    gigabases (the lookups of a read then mostly miss or hit one of these records; the loci of `si` still
    give the reads their clusters)."""
    k = si.k
    draw = rng.integers(0, 1 << (2 * k), int(n_fill * 1.15) + 1024, dtype=np.uint64)
    draw = np.unique(draw)
    draw = draw[~np.isin(draw, si.keys, assume_unique=True)]
    keys_new = draw[rng.permutation(len(draw))[:n_fill]]
    n = len(keys_new)
    n0 = len(si.rec_prg)
    prg = rng.integers(si.n_prg, n_prg_total, n, dtype=np.uint32)
    start = rng.integers(0, 4000, n, dtype=np.uint32)
    return SynthIndex(
        k, si.w, np.concatenate([si.keys, keys_new]),
        np.concatenate([si.rec_off, si.rec_off[-1] + np.arange(1, n + 1, dtype=np.uint64)]),
        np.concatenate([si.rec_prg, prg]), np.concatenate([si.rec_knode, np.arange(1, n + 1, dtype=np.uint32)]),
        np.concatenate([si.rec_strand, rng.integers(0, 2, n, dtype=np.uint8)]),
        np.concatenate([si.path_off, si.path_off[-1] + np.arange(1, n + 1, dtype=np.uint64)]),
        np.concatenate([si.iv_start, start]), np.concatenate([si.iv_len, np.full(n, k, np.uint32)]),
        np.concatenate([si.min_path_length, np.full(n_prg_total - si.n_prg, filler_mpl, np.uint32)]))
