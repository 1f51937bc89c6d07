"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and for the
reference-derived checker library (oracle/_ref/libpandora_ref.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the pandora_amd package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def _p(a, t):
    return a.ctypes.data_as(t)


class OrcIndex(C.Structure):
    _fields_ = [("n_keys", C.c_uint64), ("keys", u64p), ("rec_off", u64p), ("n_recs", C.c_uint64),
                ("rec_prg", u32p), ("rec_knode", u32p), ("rec_strand", u8p), ("path_off", u64p),
                ("iv_start", u32p), ("iv_len", u32p)]


class OrcParams(C.Structure):
    _fields_ = [("w", C.c_uint32), ("k", C.c_uint32), ("max_diff", C.c_int32), ("fraction", C.c_float),
                ("min_cluster_size", C.c_uint32), ("expected_kmers", C.c_uint32), ("n_prg", C.c_uint32),
                ("min_path_length", u32p)]


class OrcResult(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("sketch_off", u64p), ("n_mini", C.c_uint64), ("mini_kmer", u64p),
                ("mini_pos", u32p), ("mini_strand", u8p), ("n_hits", C.c_uint64), ("hit_read", u32p),
                ("hit_pos", u32p), ("hit_prg", u32p), ("hit_rec", u32p), ("hit_fwd", u8p),
                ("n_dup_hits", C.c_uint64), ("n_clusters", C.c_uint64), ("cl_begin", u64p), ("cl_size", u32p),
                ("cl_read", u32p), ("cl_prg", u32p), ("cl_first_pos", u32p), ("cl_last_pos", u32p),
                ("cl_fwd", u8p), ("cl_alive", u8p), ("n_all_clusters", C.c_uint64)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_nt4.restype = C.c_uint32
        L.orc_nt4.argtypes = [C.c_uint8]
        L.orc_hash64.restype = C.c_uint64
        L.orc_hash64.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_kmerhash.argtypes = [C.c_char_p, C.c_uint32, u64p, u64p]
        L.orc_sketch.restype = C.c_int64
        L.orc_sketch.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, u64p, u32p, u8p, C.c_uint64]
        L.orc_path_cmp.restype = C.c_int
        L.orc_path_cmp.argtypes = [u32p, u32p, C.c_uint64, u32p, u32p, C.c_uint64]
        L.orc_map.restype = C.c_int
        L.orc_map.argtypes = [C.POINTER(OrcIndex), C.POINTER(OrcParams), C.c_char_p, u64p, u32p, C.c_uint32,
                              C.POINTER(OrcResult)]
        L.orc_result_free.argtypes = [C.POINTER(OrcResult)]
        L.orc_fraction.restype = C.c_float
        L.orc_fraction.argtypes = [C.c_float, C.c_uint32]
        L.orc_threshold.restype = C.c_uint32
        L.orc_threshold.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_uint32]
        L.orc_read_loop.restype = C.c_uint32
        L.orc_read_loop.argtypes = [C.c_char_p, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_int, u8p, u32p, u32p, u32p]
        L.orc_fastaq_iterate.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def ref():
    """The reference's own boost-free translation units (built in the build container)."""
    global _ref
    if _ref is None:
        path = os.path.join(_ORACLE_DIR, "_ref", "libpandora_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_nt4.restype = C.c_uint32
        R.ref_nt4.argtypes = [C.c_uint8]
        R.ref_hash64.restype = C.c_uint64
        R.ref_hash64.argtypes = [C.c_uint64, C.c_uint64]
        R.ref_kmerhash.argtypes = [C.c_char_p, C.c_uint32, u64p, u64p]
        R.ref_path_cmp.restype = C.c_int
        R.ref_path_cmp.argtypes = [u32p, u32p, C.c_uint64, u32p, u32p, C.c_uint64]
        R.ref_minimizer_cmp.restype = C.c_int
        R.ref_minimizer_cmp.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, C.c_uint32,
                                        C.c_uint32, C.c_int]
        R.ref_minirecord_parse.restype = C.c_int
        R.ref_minirecord_parse.argtypes = [C.c_char_p, u32p, u32p, C.POINTER(C.c_int), u32p, u32p, C.c_uint32,
                                           u32p, C.c_char_p, C.c_uint32]
        R.ref_sort_hits.restype = C.c_int64
        R.ref_sort_hits.argtypes = [C.c_uint64, u32p, u32p, u32p, u8p, u64p, u32p, u32p, C.c_uint32, u64p]
        R.ref_order_clusters.restype = C.c_int64
        R.ref_order_clusters.argtypes = [C.c_uint64, u64p, u32p, u32p, u32p, u8p, u64p, u32p, u32p, C.c_uint32,
                                         u64p]
        _ref = R
    return _ref


def hash64(key, k):
    return int(lib().orc_hash64(C.c_uint64(key), C.c_uint64((1 << (2 * k)) - 1)))


def kmerhash(s, k):
    a, b = C.c_uint64(), C.c_uint64()
    lib().orc_kmerhash(s.encode(), k, C.byref(a), C.byref(b))
    return a.value, b.value


def sketch(seq, w, k):
    """Sketch of one read in std::set<Minimizer> order: list of (kmer, start, strand)."""
    if isinstance(seq, str):
        seq = seq.encode("latin-1")
    n = max(len(seq), 1)
    km = np.zeros(n, np.uint64)
    st = np.zeros(n, np.uint32)
    sd = np.zeros(n, np.uint8)
    c = lib().orc_sketch(seq, len(seq), w, k, _p(km, u64p), _p(st, u32p), _p(sd, u8p), n)
    assert c >= 0
    return [(int(km[i]), int(st[i]), int(sd[i])) for i in range(c)]


class FlatIndex:
    """Flat arrays describing Index::minhash; shared by the oracle and the product bindings."""

    def __init__(self, keys, rec_off, rec_prg, rec_knode, rec_strand, path_off, iv_start, iv_len):
        self.keys = np.ascontiguousarray(keys, np.uint64)
        self.rec_off = np.ascontiguousarray(rec_off, np.uint64)
        self.rec_prg = np.ascontiguousarray(rec_prg, np.uint32)
        self.rec_knode = np.ascontiguousarray(rec_knode, np.uint32)
        self.rec_strand = np.ascontiguousarray(rec_strand, np.uint8)
        self.path_off = np.ascontiguousarray(path_off, np.uint64)
        self.iv_start = np.ascontiguousarray(iv_start, np.uint32)
        self.iv_len = np.ascontiguousarray(iv_len, np.uint32)
        assert len(self.rec_off) == len(self.keys) + 1
        assert len(self.path_off) == len(self.rec_prg) + 1

    @property
    def n_prg(self):
        return int(self.rec_prg.max()) + 1 if len(self.rec_prg) else 0

    @staticmethod
    def from_records(records):
        """records: iterable of (key, prg_id, [(start, length), ...], knode_id, strand) in
        Index::add_record order (src/index.cpp:20-37: per key, duplicates by (prg, path, strand) dropped)."""
        by_key = {}
        for key, prg, path, knode, strand in records:
            lst = by_key.setdefault(int(key), [])
            path = tuple((int(s), int(l)) for s, l in path)
            if any(r[0] == prg and r[1] == path and r[3] == bool(strand) for r in lst):
                continue
            lst.append((int(prg), path, int(knode), bool(strand)))
        keys, rec_off, rp, rk, rs, po, ivs, ivl = [], [0], [], [], [], [0], [], []
        for key, lst in by_key.items():
            keys.append(key)
            for prg, path, knode, strand in lst:
                rp.append(prg); rk.append(knode); rs.append(1 if strand else 0)
                for s, l in path:
                    ivs.append(s); ivl.append(l)
                po.append(len(ivs))
            rec_off.append(len(rp))
        return FlatIndex(keys, rec_off, rp, rk, rs, po, ivs, ivl)

    def as_oracle(self):
        oi = OrcIndex(len(self.keys), _p(self.keys, u64p), _p(self.rec_off, u64p), len(self.rec_prg),
                      _p(self.rec_prg, u32p), _p(self.rec_knode, u32p), _p(self.rec_strand, u8p),
                      _p(self.path_off, u64p), _p(self.iv_start, u32p), _p(self.iv_len, u32p))
        return oi

    def path_of(self, rec):
        a, b = int(self.path_off[rec]), int(self.path_off[rec + 1])
        return tuple((int(self.iv_start[i]), int(self.iv_len[i])) for i in range(a, b))


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


def pack_reads(reads):
    """list of bytes/str -> (bases bytes, offsets uint64[n+1])."""
    bs = [r.encode("latin-1") if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(bs) + 1, np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    return b"".join(bs), off


def oracle_map(index, bases, offsets, read_ids, w, k, max_diff, fraction, min_cluster_size, expected_kmers,
               min_path_length):
    """Runs orc_map; returns a dict of numpy arrays."""
    offsets = np.ascontiguousarray(offsets, np.uint64)
    read_ids = np.ascontiguousarray(read_ids, np.uint32)
    mpl = np.ascontiguousarray(min_path_length, np.uint32)
    n_reads = len(offsets) - 1
    prm = OrcParams(w, k, max_diff, fraction, min_cluster_size, expected_kmers, len(mpl), _p(mpl, u32p))
    oi = index.as_oracle()
    res = OrcResult()
    if isinstance(bases, np.ndarray):
        bases = bases.tobytes()
    rc = lib().orc_map(C.byref(oi), C.byref(prm), bases, _p(offsets, u64p), _p(read_ids, u32p), n_reads,
                       C.byref(res))
    assert rc == 0
    out = {
        "sketch_off": _arr(res.sketch_off, n_reads + 1, np.uint64),
        "mini_kmer": _arr(res.mini_kmer, res.n_mini, np.uint64),
        "mini_pos": _arr(res.mini_pos, res.n_mini, np.uint32),
        "mini_strand": _arr(res.mini_strand, res.n_mini, np.uint8),
        "hit_read": _arr(res.hit_read, res.n_hits, np.uint32),
        "hit_pos": _arr(res.hit_pos, res.n_hits, np.uint32),
        "hit_prg": _arr(res.hit_prg, res.n_hits, np.uint32),
        "hit_rec": _arr(res.hit_rec, res.n_hits, np.uint32),
        "hit_fwd": _arr(res.hit_fwd, res.n_hits, np.uint8),
        "n_dup_hits": int(res.n_dup_hits),
        "cl_begin": _arr(res.cl_begin, res.n_clusters, np.uint64),
        "cl_size": _arr(res.cl_size, res.n_clusters, np.uint32),
        "cl_read": _arr(res.cl_read, res.n_clusters, np.uint32),
        "cl_prg": _arr(res.cl_prg, res.n_clusters, np.uint32),
        "cl_first_pos": _arr(res.cl_first_pos, res.n_clusters, np.uint32),
        "cl_last_pos": _arr(res.cl_last_pos, res.n_clusters, np.uint32),
        "cl_fwd": _arr(res.cl_fwd, res.n_clusters, np.uint8),
        "cl_alive": _arr(res.cl_alive, res.n_clusters, np.uint8),
        "n_all_clusters": int(res.n_all_clusters),
    }
    lib().orc_result_free(C.byref(res))
    return out


def read_loop(bases, offsets, w, k, genome_size, max_covg, illumina):
    offsets = np.ascontiguousarray(offsets, np.uint64)
    n = len(offsets) - 1
    take = np.zeros(max(n, 1), np.uint8)
    ids = np.zeros(max(n, 1), np.uint32)
    covg, exp = C.c_uint32(), C.c_uint32()
    used = lib().orc_read_loop(bases, _p(offsets, u64p), n, w, k, genome_size, max_covg, int(illumina),
                               _p(take, u8p), _p(ids, u32p), C.byref(covg), C.byref(exp))
    return int(used), take[:n], ids[:n], covg.value, exp.value


_CB = C.CFUNCTYPE(None, C.POINTER(C.c_char), C.c_size_t, C.POINTER(C.c_char), C.c_size_t, C.c_void_p)


def fastaq_iterate(buf):
    """[(name, read), ...] exactly as the reference's `while(!fh.eof()) fh.get_next()` loop yields them."""
    out = []

    def cb(name, nl, read, rl, user):
        out.append((C.string_at(name, nl).decode("latin-1"), C.string_at(read, rl).decode("latin-1")))

    cbf = _CB(cb)
    lib().orc_fastaq_iterate(buf, len(buf), C.cast(cbf, C.c_void_p), None)
    return out


def oracle_coverage(res, n_records):
    """KmerNode coverage after pangenome::Graph::add_hits_to_kmergraphs (src/pangenome/pangraph.cpp:347-383)
    from an oracle_map result: Node::reads is a multiset with one entry per add_node call (:80-89), and for
    every entry each hit of Read::hits[prg] (the union of the read's surviving clusters on that prg,
    src/pangenome/panread.cpp:21-28) increments its kmer node's counter of the hit's direction
    (src/kmernode.cpp:42-49).  Returns (forward, reverse) counters per index record."""
    alive = res["cl_alive"].astype(bool)
    calls = {}
    for r, p in zip(res["cl_read"][alive], res["cl_prg"][alive]):
        calls[(int(r), int(p))] = calls.get((int(r), int(p)), 0) + 1
    fwd = np.zeros(n_records, np.uint32)
    rev = np.zeros(n_records, np.uint32)
    for b, s, r, p, f in zip(res["cl_begin"][alive], res["cl_size"][alive], res["cl_read"][alive],
                             res["cl_prg"][alive], res["cl_fwd"][alive]):
        recs = res["hit_rec"][int(b):int(b) + int(s)]
        np.add.at(fwd if f else rev, recs, calls[(int(r), int(p))])
    return fwd, rev
