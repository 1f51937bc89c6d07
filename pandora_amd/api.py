"""Python face of the C ABI, used by the tests and by bench.py.

Class and function names follow the reference's own types (Index, Seq sketch,
MinimizerHits, define_clusters, filter_clusters, pangraph_from_read_file); every
call goes through libpandora_amd.so — nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import u8p, u32p, u64p, f32p

STAGE_SKETCH, STAGE_HITS, STAGE_SORT, STAGE_CLUSTER = 1, 2, 3, 4
UINT32_MAX = 0xFFFFFFFF


class PmapError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pmap error %d: %s" % (code, msg))
        self.code = code


def _check(rc):
    if rc != 0:
        raise PmapError(rc, _lib.load().pmap_last_error().decode("utf-8", "replace"))


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def init(device=-1):
    _check(_lib.load().pmap_init(device))


def init_devices(devices):
    """The mapping contexts pangraph_from_read_file spreads its batches over: one per entry (a GPU may appear
    more than once); [] = the default."""
    arr = (C.c_int * max(len(devices), 1))(*devices)
    _check(_lib.load().pmap_init_devices(arr, len(devices)))


def fraction(e_rate, k):
    """fraction_kmers_required_for_cluster = 0.5 / exp(e_rate * k) (src/utils.cpp:394)."""
    return float(_lib.load().pmap_fraction(C.c_float(e_rate), k))


class Index:
    """Index::minhash (include/index.h:14-37) flattened and resident in HBM."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def create(k, keys, rec_off, rec_prg, rec_knode, rec_strand, path_off, iv_start, iv_len, min_path_length):
        lib = _lib.load()
        keys = np.ascontiguousarray(keys, np.uint64)
        rec_off = np.ascontiguousarray(rec_off, np.uint64)
        rec_prg = np.ascontiguousarray(rec_prg, np.uint32)
        rec_knode = np.ascontiguousarray(rec_knode, np.uint32)
        rec_strand = np.ascontiguousarray(rec_strand, np.uint8)
        path_off = np.ascontiguousarray(path_off, np.uint64)
        iv_start = np.ascontiguousarray(iv_start, np.uint32)
        iv_len = np.ascontiguousarray(iv_len, np.uint32)
        mpl = np.ascontiguousarray(min_path_length, np.uint32)
        d = _lib.IndexDesc(k, len(keys), _p(keys, u64p), _p(rec_off, u64p), len(rec_prg), _p(rec_prg, u32p),
                           _p(rec_knode, u32p), _p(rec_strand, u8p), _p(path_off, u64p), _p(iv_start, u32p),
                           _p(iv_len, u32p), len(mpl), _p(mpl, u32p))
        h = C.c_void_p()
        _check(lib.pmap_index_create(C.byref(d), C.byref(h)))
        return Index(h)

    @staticmethod
    def load(idx_path, k, min_path_length):
        """Index::load (src/index.cpp:75-124) of the reference's text format."""
        lib = _lib.load()
        mpl = np.ascontiguousarray(min_path_length, np.uint32)
        h = C.c_void_p()
        _check(lib.pmap_index_load(str(idx_path).encode(), k, len(mpl), _p(mpl, u32p), C.byref(h)))
        return Index(h)

    @staticmethod
    def load_cached(idx_path, k, min_path_length, cache_path):
        """Index.load through a binary cache next to the text file; returns (index, came_from_cache)."""
        lib = _lib.load()
        mpl = np.ascontiguousarray(min_path_length, np.uint32)
        h = C.c_void_p()
        hit = C.c_int(0)
        _check(lib.pmap_index_load_cached(str(idx_path).encode(), k, len(mpl), _p(mpl, u32p), str(cache_path).encode(),
                                          C.byref(h), C.byref(hit)))
        return Index(h), bool(hit.value)

    @staticmethod
    def load_cache(cache_path):
        h = C.c_void_p()
        _check(_lib.load().pmap_index_load_cache(str(cache_path).encode(), C.byref(h)))
        return Index(h)

    def save_cache(self, cache_path, source_path=None):
        _check(_lib.load().pmap_index_save_cache(self._h, str(cache_path).encode(),
                                                 str(source_path).encode() if source_path else None))

    def upload(self):
        _check(_lib.load().pmap_index_upload_now(self._h))

    def info(self):
        i = _lib.IndexInfo()
        _check(_lib.load().pmap_index_info(self._h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in i._fields_}

    def records(self):
        n = self.info()["n_recs"]
        key = np.zeros(n, np.uint64)
        prg = np.zeros(n, np.uint32)
        knode = np.zeros(n, np.uint32)
        strand = np.zeros(n, np.uint8)
        rank = np.zeros(n, np.uint32)
        poff = np.zeros(n + 1, np.uint64)
        _check(_lib.load().pmap_index_records(self._h, _p(key, u64p), _p(prg, u32p), _p(knode, u32p),
                                              _p(strand, u8p), _p(rank, u32p), _p(poff, u64p)))
        nint = C.c_uint64()
        _check(_lib.load().pmap_index_intervals(self._h, None, None, C.byref(nint)))
        ivs = np.zeros(nint.value, np.uint32)
        ivl = np.zeros(nint.value, np.uint32)
        _check(_lib.load().pmap_index_intervals(self._h, _p(ivs, u32p), _p(ivl, u32p), None))
        return {"key": key, "prg": prg, "knode": knode, "strand": strand, "path_rank": rank, "path_off": poff,
                "iv_start": ivs, "iv_len": ivl}

    def close(self):
        if self._h:
            _lib.load().pmap_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gfa_min_path_length(path):
    out = C.c_uint32()
    _check(_lib.load().pmap_gfa_min_path_length(str(path).encode(), C.byref(out)))
    return out.value


class Reads:
    """A batch of reads resident in HBM (2-bit packed on upload)."""

    def __init__(self, bases, offsets, first_read_id=0):
        lib = _lib.load()
        if isinstance(bases, np.ndarray):
            self._bases = np.ascontiguousarray(bases, np.uint8)
            bptr = self._bases.ctypes.data_as(C.c_char_p)
        else:
            self._bases = bytes(bases)
            bptr = self._bases
        offsets = np.ascontiguousarray(offsets, np.uint64)
        self.n_reads = len(offsets) - 1
        self.n_bases = int(offsets[-1] - offsets[0])
        self.first_read_id = first_read_id
        h = C.c_void_p()
        _check(lib.pmap_reads_create(bptr, _p(offsets, u64p), self.n_reads, first_read_id, C.byref(h)))
        self._h = h
        self._bases = None

    @staticmethod
    def packed(bases, offsets, first_read_id=0):
        """The same batch packed to 2 bits per base on the host (pmap_pack_reads) and uploaded as such
        (pmap_reads_create_packed): what pangraph_from_read_file does with the batches of a file."""
        lib = _lib.load()
        bases = np.ascontiguousarray(np.frombuffer(bytes(bases), np.uint8) if not isinstance(bases, np.ndarray) else bases,
                                     np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        lens = np.diff(offsets.astype(np.int64))
        words = np.zeros(int(((lens + 15) // 16).sum()) + 1, np.uint32)
        ok = np.zeros(n + 1, np.uint8)
        _check(lib.pmap_pack_reads(bases.ctypes.data_as(C.c_char_p), _p(offsets, u64p), n, _p(words, u32p), _p(ok, u8p)))
        self = Reads.__new__(Reads)
        self._bases = None
        self.n_reads, self.n_bases, self.first_read_id = n, int(offsets[-1] - offsets[0]), first_read_id
        h = C.c_void_p()
        _check(lib.pmap_reads_create_packed(_p(words, u32p), _p(offsets, u64p), _p(ok, u8p), n, first_read_id, C.byref(h)))
        self._h = h
        self.packed_words, self.acgt_only = words[:-1], ok[:n]
        return self

    @staticmethod
    def from_list(reads, first_read_id=0):
        bs = [r.encode("latin-1") if isinstance(r, str) else bytes(r) for r in reads]
        off = np.zeros(len(bs) + 1, np.uint64)
        if bs:
            off[1:] = np.cumsum([len(b) for b in bs])
        return Reads(b"".join(bs), off, first_read_id)

    def repack(self):
        _check(_lib.load().pmap_reads_repack(self._h))

    def close(self):
        if self._h:
            _lib.load().pmap_reads_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MapResult:
    """Minimizers, hits and clusters of one batch (device resident until fetched)."""

    def __init__(self):
        self._h = C.c_void_p()
        self._keep = None

    def wait(self):
        """Blocks until a map_reads(..., wait=False) has finished."""
        _check(_lib.load().pmap_map_wait(self._h))
        return self

    def counts(self):
        c = _lib.Counts()
        _check(_lib.load().pmap_result_counts(self._h, C.byref(c)))
        d = {f: getattr(c, f) for f, _ in c._fields_ if f != "ms"}
        d["ms"] = list(c.ms)
        return d

    def kernel_ms(self):
        ms = np.zeros(_lib.PMAP_N_KERNELS, np.float32)
        _check(_lib.load().pmap_result_kernel_ms(self._h, _p(ms, f32p)))
        out = {}
        for i in range(_lib.PMAP_N_KERNELS):
            name = _lib.load().pmap_kernel_name(i).decode()
            if name:
                out[name] = float(ms[i])
        return out

    def minimizers(self):
        n = self.counts()["n_minimizers"]
        kmer = np.zeros(n, np.uint64)
        read = np.zeros(n, np.uint32)
        pos = np.zeros(n, np.uint32)
        strand = np.zeros(n, np.uint8)
        _check(_lib.load().pmap_result_minimizers(self._h, _p(kmer, u64p), _p(read, u32p), _p(pos, u32p),
                                                  _p(strand, u8p)))
        return {"kmer": kmer, "read": read, "pos": pos, "strand": strand}

    def hits(self, sorted=True):
        n = self.counts()["n_hits"]
        read = np.zeros(n, np.uint32)
        pos = np.zeros(n, np.uint32)
        prg = np.zeros(n, np.uint32)
        rec = np.zeros(n, np.uint32)
        fwd = np.zeros(n, np.uint8)
        _check(_lib.load().pmap_result_hits(self._h, 1 if sorted else 0, _p(read, u32p), _p(pos, u32p),
                                            _p(prg, u32p), _p(rec, u32p), _p(fwd, u8p)))
        return {"read": read, "pos": pos, "prg": prg, "rec": rec, "fwd": fwd}

    def clusters(self):
        n = self.counts()["n_clusters"]
        begin = np.zeros(n, np.uint64)
        a = {k: np.zeros(n, np.uint32) for k in ("size", "read", "prg", "first_pos", "last_pos", "first_rank")}
        fwd = np.zeros(n, np.uint8)
        _check(_lib.load().pmap_result_clusters(self._h, _p(begin, u64p), _p(a["size"], u32p), _p(a["read"], u32p),
                                                _p(a["prg"], u32p), _p(fwd, u8p), _p(a["first_pos"], u32p),
                                                _p(a["last_pos"], u32p), _p(a["first_rank"], u32p)))
        a["begin"] = begin
        a["fwd"] = fwd
        return a

    def close(self):
        if self._h:
            _lib.load().pmap_result_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def map_reads(index, reads, w, k, max_diff, fraction_required, min_cluster_size=10, expected_kmers=UINT32_MAX,
              last_stage=0, profile=0, result=None, unfused=False, wait=True, sketch=None, sort_wide=False, flags=0):
    """Seq::minimizer_sketch -> add_read_hits -> MinimizerHits::sort -> define_clusters on the GPU.
    wait=False only enqueues the kernels (pmap_map_async): call result.wait(), or any accessor, later;
    index and reads must stay alive until then.  sketch="np8" | "lds" picks another of the equivalent
    sketch kernels (PMAP_FLAG_SKETCH_*)."""
    res = result if result is not None else MapResult()
    flags |= (1 if unfused else 0) | {None: 0, "np8": 2, "lds": 4}[sketch] | (8 if sort_wide else 0)
    prm = _lib.Params(w, k, max_diff, fraction_required, min_cluster_size, expected_kmers, last_stage, profile,
                      flags)
    fn = _lib.load().pmap_map if wait else _lib.load().pmap_map_async
    _check(fn(index._h, reads._h, C.byref(prm), C.byref(res._h)))
    res._keep = (index, reads)
    return res


def filter_clusters(cl):
    """clusterComp order + filter_clusters (src/utils.cpp:251-288) over a dict from MapResult.clusters().
    Returns (order, alive): order[j] = index of the j-th cluster in clusterComp order."""
    n = len(cl["size"])
    order = np.zeros(n, np.uint64)
    alive = np.zeros(n, np.uint8)
    c = {k: np.ascontiguousarray(v) for k, v in cl.items()}
    _check(_lib.load().pmap_filter_clusters(n, _p(c["read"], u32p), _p(c["first_pos"], u32p), _p(c["size"], u32p),
                                            _p(c["prg"], u32p), _p(c["first_rank"], u32p), _p(c["fwd"], u8p),
                                            _p(c["last_pos"], u32p), _p(order, u64p), _p(alive, u8p)))
    return order, alive


def read_fastaq(path):
    """[(name, read), ...] as the reference's `while (!fh.eof()) fh.get_next()` loop yields them."""
    lib = _lib.load()
    bases, names = C.c_void_p(), C.c_void_p()
    off, noff = u64p(), u64p()
    n = C.c_uint32()
    _check(lib.pmap_read_fastaq(str(path).encode(), C.byref(bases), C.byref(off), C.byref(n), C.byref(names),
                                C.byref(noff)))
    out = []
    try:
        for i in range(n.value):
            r = C.string_at(bases.value + off[i], off[i + 1] - off[i]).decode("latin-1")
            nm = C.string_at(names.value + noff[i], noff[i + 1] - noff[i]).decode("latin-1")
            out.append((nm, r))
    finally:
        lib.pmap_free(bases)
        lib.pmap_free(names)
        lib.pmap_free(C.cast(off, C.c_void_p))
        lib.pmap_free(C.cast(noff, C.c_void_p))
    return out


def read_fastaq_pieces(path, piece_bytes):
    """The reads of read_fastaq (without names) through the piece-wise parser of pangraph_from_read_file."""
    lib = _lib.load()
    bases = C.c_void_p()
    off = u64p()
    n = C.c_uint32()
    _check(lib.pmap_read_fastaq_pieces(str(path).encode(), piece_bytes, C.byref(bases), C.byref(off), C.byref(n)))
    try:
        return [C.string_at(bases.value + off[i], off[i + 1] - off[i]).decode("latin-1") for i in range(n.value)]
    finally:
        lib.pmap_free(bases)
        lib.pmap_free(C.cast(off, C.c_void_p))


class Pangraph:
    """Summary of the pangenome::Graph that pangraph_from_read_file fills."""

    def __init__(self, handle, covg, n_records=0):
        self._h = handle
        self.covg = covg
        self._n_records = n_records

    def coverage(self):
        """KmerNode coverage after add_hits_to_kmergraphs, per index record: (forward, reverse)."""
        fwd = np.zeros(self._n_records, np.uint32)
        rev = np.zeros(self._n_records, np.uint32)
        _check(_lib.load().pmap_pangraph_coverage(self._h, _p(fwd, u32p), _p(rev, u32p)))
        return fwd, rev

    def info(self):
        i = _lib.PangraphInfo()
        _check(_lib.load().pmap_pangraph_info(self._h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in i._fields_}

    def nodes(self):
        n = self.info()["n_nodes"]
        prg = np.zeros(n, np.uint32)
        covg = np.zeros(n, np.uint32)
        _check(_lib.load().pmap_pangraph_nodes(self._h, _p(prg, u32p), _p(covg, u32p)))
        return {int(p): int(c) for p, c in zip(prg, covg)}

    def reads(self):
        i = self.info()
        rid = np.zeros(i["n_reads"], np.uint32)
        off = np.zeros(i["n_reads"] + 1, np.uint64)
        prg = np.zeros(i["n_read_nodes"], np.uint32)
        ori = np.zeros(i["n_read_nodes"], np.uint8)
        _check(_lib.load().pmap_pangraph_reads(self._h, _p(rid, u32p), _p(off, u64p), _p(prg, u32p), _p(ori, u8p)))
        return {int(rid[j]): [(int(prg[q]), int(ori[q])) for q in range(int(off[j]), int(off[j + 1]))]
                for j in range(len(rid))}

    def hits(self):
        n = self.info()["n_hits"]
        a = {k: np.zeros(n, np.uint32) for k in ("read", "prg", "pos", "rec")}
        fwd = np.zeros(n, np.uint8)
        _check(_lib.load().pmap_pangraph_hits(self._h, _p(a["read"], u32p), _p(a["prg"], u32p), _p(a["pos"], u32p),
                                              _p(a["rec"], u32p), _p(fwd, u8p)))
        a["fwd"] = fwd
        return a

    def calls(self):
        n = self.info()["n_add_node_calls"]
        a = {k: np.zeros(n, np.uint32) for k in ("prg", "read", "size")}
        fwd = np.zeros(n, np.uint8)
        _check(_lib.load().pmap_pangraph_calls(self._h, _p(a["prg"], u32p), _p(a["read"], u32p), _p(a["size"], u32p),
                                               _p(fwd, u8p)))
        a["fwd"] = fwd
        return a

    def call_hits(self):
        """The cluster of every add_node call: (hit_off, pos, rec); call j owns [hit_off[j], hit_off[j+1])."""
        i = self.info()
        off = np.zeros(i["n_add_node_calls"] + 1, np.uint64)
        pos = np.zeros(i["n_hits"], np.uint32)
        rec = np.zeros(i["n_hits"], np.uint32)
        _check(_lib.load().pmap_pangraph_call_hits(self._h, _p(off, u64p), _p(pos, u32p), _p(rec, u32p)))
        return off, pos, rec

    def close(self):
        if self._h:
            _lib.load().pmap_pangraph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pangraph_from_read_file(filepath, index, w, k, max_diff, e_rate, min_cluster_size=10, genome_size=5000000,
                            illumina=False, clean=False, max_covg=300, batch_bases=0, keep_hits=True, coverage=False):
    """pangraph_from_read_file (include/utils.h:83-90): same arguments, same defaults.  keep_hits=False
    drops Read::hits; coverage=True also accumulates the kmer node coverage of add_hits_to_kmergraphs."""
    a = _lib.MapArgs(w, k, max_diff, e_rate, min_cluster_size, genome_size, int(illumina), int(clean), max_covg,
                     batch_bases, (0 if keep_hits else 1) | (2 if coverage else 0))
    h = C.c_void_p()
    covg = C.c_uint32()
    _check(_lib.load().pmap_pangraph_from_read_file(str(filepath).encode(), index._h, C.byref(a), C.byref(h),
                                                    C.byref(covg)))
    return Pangraph(h, covg.value, index.info()["n_recs"])
