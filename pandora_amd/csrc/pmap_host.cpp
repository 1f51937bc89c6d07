// Host stages around the device path: the FASTA/FASTQ reader
// (src/fastaq_handler.cpp:12-71), clusterComp ordering + filter_clusters
// (src/minihits.cpp:95-109, src/utils.cpp:251-288), the add_node sequence
// (src/utils.cpp:338-352, src/pangenome/pangraph.cpp:43-140) and the read loop of
// pangraph_from_read_file (src/utils.cpp:379-455).
#include <sys/mman.h>
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "pmap_internal.h"

// ---------------------------------------------------------------------------
// clusterComp + filter_clusters over cluster summaries

extern "C" int pmap_filter_clusters(uint64_t n, const uint32_t *read_id, const uint32_t *first_pos,
                                    const uint32_t *size, const uint32_t *prg_id, const uint32_t *first_rank,
                                    const uint8_t *is_forward, const uint32_t *last_pos, uint64_t *order,
                                    uint8_t *alive) {
    if (n && (!read_id || !first_pos || !size || !prg_id || !first_rank || !is_forward || !last_pos || !order || !alive))
        return pmap_fail(PMAP_EINVAL, "null argument");
    std::iota(order, order + n, 0ull);
    // clusterComp (src/minihits.cpp:95-109): read, first position, size descending, prg, first path,
    // reverse-strand cluster before forward one
    std::sort(order, order + n, [&](uint64_t a, uint64_t b) {
        if (read_id[a] != read_id[b]) return read_id[a] < read_id[b];
        if (first_pos[a] != first_pos[b]) return first_pos[a] < first_pos[b];
        if (size[a] != size[b]) return size[a] > size[b];
        if (prg_id[a] != prg_id[b]) return prg_id[a] < prg_id[b];
        if (first_rank[a] != first_rank[b]) return first_rank[a] < first_rank[b];
        if (is_forward[a] != is_forward[b]) return is_forward[a] < is_forward[b];
        return a < b;
    });
    for (uint64_t i = 0; i < n; ++i) alive[i] = 1;
    // filter_clusters (src/utils.cpp:256-286): compare each cluster with the closest surviving predecessor
    if (n > 0) {
        uint64_t prev = 0;
        for (uint64_t cur = 1; cur < n; ++cur) {
            const uint64_t c = order[cur], p = order[prev];
            if (read_id[c] == read_id[p] &&
                ((prg_id[c] == prg_id[p] && is_forward[c] != is_forward[p]) || last_pos[c] <= last_pos[p])) {
                if (size[p] >= size[c]) {
                    alive[cur] = 0;   // current erased, previous stays the comparison partner
                    continue;
                }
                alive[prev] = 0;
            }
            prev = cur;
        }
    }
    return PMAP_OK;
}

// ---------------------------------------------------------------------------
// FastaqHandler

namespace {

// std::istream semantics that FastaqHandler relies on, over a plain or gzip file read in chunks
// (gzread is transparent for uncompressed files)
struct LineStream {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eofbit = false, drained = false, io_error = false;
    bool open(const char *path) {
        f = gzopen(path, "rb");
        if (!f) return false;
        gzbuffer(f, 1 << 20);
        buf.resize(8 << 20);
        return true;
    }
    ~LineStream() { if (f) gzclose(f); }
    // moves the unread bytes to the front and reads more; false when nothing was added
    bool fill() {
        if (drained) return false;
        if (pos > 0) {
            memmove(buf.data(), buf.data() + pos, end - pos);
            end -= pos;
            pos = 0;
        }
        if (end == buf.size()) buf.resize(buf.size() * 2);   // one line longer than the buffer
        const int n = gzread(f, buf.data() + end, (unsigned) std::min<size_t>(buf.size() - end, 1u << 30));
        if (n < 0) io_error = true;
        if (n <= 0) { drained = true; return false; }
        end += (size_t) n;
        return true;
    }
    bool at_eof() {   // instream.peek() == EOF (src/fastaq_handler.cpp:38-41)
        if (eofbit) return true;
        if (pos < end || fill()) return false;
        eofbit = true;
        return true;
    }
    // getline(...).good(): false when the data ended before a newline was seen.  *line points into
    // the buffer and is valid until the next call.
    bool getline(const char **line, size_t *len) {
        *line = buf.data();
        *len = 0;
        if (eofbit) return false;
        if (pos >= end && !fill()) { eofbit = true; return false; }
        size_t scanned = 0;
        for (;;) {
            const char *nl = (const char *) memchr(buf.data() + pos + scanned, '\n', end - pos - scanned);
            if (nl) {
                *line = buf.data() + pos;
                *len = (size_t) (nl - (buf.data() + pos));
                pos += *len + 1;
                return true;
            }
            scanned = end - pos;
            if (!fill()) {   // unterminated last line
                *line = buf.data() + pos;
                *len = end - pos;
                pos = end;
                eofbit = true;
                return false;
            }
        }
    }
};

struct FastaqReader {
    LineStream in;
    // the last line read: only its first character and its emptiness matter to get_next
    bool line_empty = true;
    char line_first = 0;
    std::string pending_name;     // the header line that ended the record before
    std::string name, read;
    bool open(const char *path) { return in.open(path); }
    bool eof() { return in.at_eof(); }
    void get_next() {   // src/fastaq_handler.cpp:43-71
        if (!line_empty && (line_first == '>' || line_first == '@')) {
            name = pending_name;
            read.clear();
        }
        const char *l;
        size_t n;
        for (;;) {
            const bool good = in.getline(&l, &n);
            line_empty = n == 0;
            line_first = n ? l[0] : 0;
            if (!line_empty && (line_first == '>' || line_first == '@')) pending_name.assign(l + 1, n - 1);
            if (!good) return;
            if (!line_empty && line_first == '+') {
                in.getline(&l, &n);   // quality line
                line_empty = n == 0;
                line_first = n ? l[0] : 0;
                if (!line_empty && (line_first == '>' || line_first == '@')) pending_name.assign(l + 1, n - 1);
            } else if (line_empty || line_first == '>' || line_first == '@') {
                if (!read.empty() || line_empty) return;
                name = pending_name;
                read.clear();
            } else {
                read.append(l, n);
            }
        }
    }
};

}  // namespace

extern "C" int pmap_read_fastaq(const char *path, char **bases, uint64_t **offsets, uint32_t *n_reads,
                                char **names, uint64_t **name_offsets) {
    FastaqReader fh;
    if (!fh.open(path))
        return pmap_fail(PMAP_EIO, std::string("Unable to open fastaq file ") + path);   // src/fastaq_handler.cpp:18-21
    std::string all, allnames;
    std::vector<uint64_t> off{0}, noff{0};
    while (!fh.eof()) {
        fh.get_next();
        all += fh.read;
        off.push_back(all.size());
        allnames += fh.name;
        noff.push_back(allnames.size());
    }
    const uint32_t n = (uint32_t) (off.size() - 1);
    if (n_reads) *n_reads = n;
    if (bases) {
        *bases = (char *) malloc(all.size() + 1);
        memcpy(*bases, all.data(), all.size());
        (*bases)[all.size()] = 0;
    }
    if (offsets) {
        *offsets = (uint64_t *) malloc(sizeof(uint64_t) * off.size());
        memcpy(*offsets, off.data(), sizeof(uint64_t) * off.size());
    }
    if (names) {
        *names = (char *) malloc(allnames.size() + 1);
        memcpy(*names, allnames.data(), allnames.size());
        (*names)[allnames.size()] = 0;
    }
    if (name_offsets) {
        *name_offsets = (uint64_t *) malloc(sizeof(uint64_t) * noff.size());
        memcpy(*name_offsets, noff.data(), sizeof(uint64_t) * noff.size());
    }
    return PMAP_OK;
}

extern "C" void pmap_free(void *p) { free(p); }

// ---------------------------------------------------------------------------
// pangenome::Graph summary

struct PgHit {
    uint32_t read, prg, pos, rec, rank;
    uint8_t fwd;
};

// Growing array of hits: realloc moves big blocks by remapping their pages instead of copying them
// (a std::vector of several hundred MB copies itself at every growth step), and the block asks for
// huge pages so that filling it does not take a page fault per 4 KiB.
struct HitBuf {
    PgHit *p = nullptr;
    size_t n = 0, cap = 0;
    HitBuf() = default;
    HitBuf(const HitBuf &) = delete;
    HitBuf &operator=(const HitBuf &) = delete;
    ~HitBuf() { free(p); }
    size_t size() const { return n; }
    PgHit *begin() { return p; }
    PgHit *end() { return p + n; }
    const PgHit &operator[](size_t i) const { return p[i]; }
    bool reserve_more(size_t extra) {
        if (n + extra <= cap) return true;
        size_t ncap = std::max<size_t>(n + extra, cap + cap / 2 + 4096);
        PgHit *q = (PgHit *) realloc(p, ncap * sizeof(PgHit));
        if (!q) return false;
        p = q;
        cap = ncap;
#ifdef MADV_HUGEPAGE
        if (ncap * sizeof(PgHit) >= (size_t) 4 << 20) {
            const uintptr_t a = ((uintptr_t) p + 4095) & ~(uintptr_t) 4095;
            const uintptr_t e = ((uintptr_t) p + ncap * sizeof(PgHit)) & ~(uintptr_t) 4095;
            if (e > a) (void) madvise((void *) a, e - a, MADV_HUGEPAGE);
        }
#endif
        return true;
    }
    void push_unchecked(const PgHit &h) { p[n++] = h; }
};

struct PgRead {
    std::vector<uint32_t> nodes;
    std::vector<uint8_t> orient;
};

struct pmap_pangraph {
    std::map<uint32_t, uint32_t> node_covg;   // node id (= prg id) -> covg
    std::map<uint32_t, PgRead> reads;
    HitBuf hits;                      // in add_node call order, each call's cluster in pComp order
    std::vector<uint32_t> call_prg, call_read, call_size;
    std::vector<uint8_t> call_fwd;
    bool keep_hits = true;
    std::vector<uint64_t> by_path;    // hits in Read::hits order (indices into `hits`), built on demand
    bool by_path_valid = false;
    pmap_coverage_t *cov = nullptr;   // device coverage counters (PMAP_PG_COVERAGE)
};

// pangenome::Graph::add_node (src/pangenome/pangraph.cpp:126-140) with get_node (:59-76) and
// record_read_info (:108-122)
static void pg_add_node(pmap_pangraph *pg, uint32_t prg, uint32_t read, uint32_t size, bool first_fwd) {
    auto it = pg->node_covg.find(prg);
    if (it == pg->node_covg.end()) pg->node_covg[prg] = 1; else it->second += 1;
    PgRead &r = pg->reads[read];
    const uint8_t orientation = size > 0 && first_fwd;
    if (r.nodes.empty() || r.nodes.back() != prg || r.orient.back() != orientation) {
        r.nodes.push_back(prg);
        r.orient.push_back(orientation);
    }
    pg->call_prg.push_back(prg);
    pg->call_read.push_back(read);
    pg->call_size.push_back(size);
    pg->call_fwd.push_back(first_fwd);
}

static void pg_sort_hits(pmap_pangraph *pg) {
    if (pg->by_path_valid) return;
    // per (read, prg): pComp_path (src/minihits.cpp:77-93): path, forward first, position.  The hits themselves
    // stay in call order (pmap_pangraph_call_hits); this is an index into them.
    pg->by_path.resize(pg->hits.size());
    std::iota(pg->by_path.begin(), pg->by_path.end(), 0ull);
    const PgHit *h = pg->hits.p;
    std::sort(pg->by_path.begin(), pg->by_path.end(), [h](uint64_t x, uint64_t y) {
        const PgHit &a = h[x], &b = h[y];
        if (a.read != b.read) return a.read < b.read;
        if (a.prg != b.prg) return a.prg < b.prg;
        if (a.rank != b.rank) return a.rank < b.rank;
        if (a.fwd != b.fwd) return a.fwd > b.fwd;
        if (a.pos != b.pos) return a.pos < b.pos;
        return x < y;
    });
    pg->by_path_valid = true;
}

// ---------------------------------------------------------------------------
// pangraph_from_read_file as a three-stage pipeline: a reader thread parses the file into batches
// (pinned host memory), the calling thread uploads batch i+1 and enqueues its kernels while the GPU
// maps batch i, then finishes batch i-1 on the host (filter_clusters, add_node, coverage).

namespace {

struct Batch {
    char *data = nullptr;          // pinned
    size_t size = 0, cap = 0;
    std::vector<uint64_t> offsets{0};
    uint32_t first_id = 0;
    pmap_params_t prm;
    void reset(uint32_t first) {
        size = 0;
        offsets.assign(1, 0);
        first_id = first;
    }
    bool reserve(size_t want) {
        if (want <= cap) return true;
        size_t ncap = std::max(want, cap + cap / 2);
        char *p = (char *) pmap_pinned_alloc(ncap);
        if (!p) return false;
        if (size) memcpy(p, data, size);
        if (data) pmap_pinned_free(data);
        data = p;
        cap = ncap;
        return true;
    }
    void append(const std::string &seq) {
        memcpy(data + size, seq.data(), seq.size());
        size += seq.size();
        offsets.push_back(size);
    }
};

// bounded hand-over of batches between the reader thread and the mapping thread
struct BatchQueue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<Batch *> ready, free_list;
    bool done = false;      // the reader has pushed its last batch
    bool abort = false;     // the consumer gave up: the reader must stop
    double reader_waited = 0, mapper_waited = 0;    // seconds each side spent waiting for the other (PMAP_TRACE_PIPELINE)
    static double now() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    Batch *get_free() {
        std::unique_lock<std::mutex> l(m);
        const double t0 = now();
        cv.wait(l, [&] { return abort || !free_list.empty(); });
        reader_waited += now() - t0;
        if (abort) return nullptr;
        Batch *b = free_list.front();
        free_list.pop_front();
        return b;
    }
    void put_free(Batch *b) {
        { std::lock_guard<std::mutex> l(m); free_list.push_back(b); }
        cv.notify_all();
    }
    void put_ready(Batch *b) {
        { std::lock_guard<std::mutex> l(m); ready.push_back(b); }
        cv.notify_all();
    }
    Batch *get_ready() {    // nullptr: no more batches
        std::unique_lock<std::mutex> l(m);
        const double t0 = now();
        cv.wait(l, [&] { return done || !ready.empty(); });
        mapper_waited += now() - t0;
        if (ready.empty()) return nullptr;
        Batch *b = ready.front();
        ready.pop_front();
        return b;
    }
    void finish() {
        { std::lock_guard<std::mutex> l(m); done = true; }
        cv.notify_all();
    }
    void stop() {
        { std::lock_guard<std::mutex> l(m); abort = true; }
        cv.notify_all();
    }
};

struct ReaderResult {
    uint64_t covg = 0;
    int rc = PMAP_OK;
    std::string error;
};

// true iff every letter is one that nt4 (src/inthash.cpp:42-63) maps below 4: A, C, G, T in either case or
// a raw code 0..3.  This is the reader's hot loop: 16 letters per step where SSE2 is there.
bool all_acgt(const unsigned char *p, size_t n) {
    size_t i = 0;
    bool bad = false;
#if defined(__SSE2__)
    __m128i acc = _mm_setzero_si128();
    const __m128i up = _mm_set1_epi8((char) 0xDF), A = _mm_set1_epi8('A'), C = _mm_set1_epi8('C'),
                  G = _mm_set1_epi8('G'), T = _mm_set1_epi8('T'), hi = _mm_set1_epi8((char) 0xFC),
                  zero = _mm_setzero_si128(), ones = _mm_set1_epi8((char) 0xFF);
    for (; i + 16 <= n; i += 16) {
        const __m128i c = _mm_loadu_si128((const __m128i *) (p + i)), u = _mm_and_si128(c, up);
        __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(u, A), _mm_cmpeq_epi8(u, C)),
                                  _mm_or_si128(_mm_cmpeq_epi8(u, G), _mm_cmpeq_epi8(u, T)));
        ok = _mm_or_si128(ok, _mm_cmpeq_epi8(_mm_and_si128(c, hi), zero));   // c < 4
        acc = _mm_or_si128(acc, _mm_andnot_si128(ok, ones));
    }
    bad = _mm_movemask_epi8(acc) != 0;
#endif
    for (; i < n; ++i) {
        const unsigned char c = p[i], u = c & 0xDFu;
        bad |= !((u == 'A') | (u == 'C') | (u == 'G') | (u == 'T') | (c < 4));
    }
    return !bad;
}

// The read loop of pangraph_from_read_file (src/utils.cpp:397-440) up to the point where a read is
// handed to the sketch: ids, coverage stop, 10M-read stop, Illumina expected k-mer count.
void reader_thread(FastaqReader *fh, const pmap_map_args_t *a, pmap_params_t prm, uint64_t batch_bases,
                   BatchQueue *q, ReaderResult *out) {
    uint64_t covg = 0;
    uint32_t id = 0;
    Batch *b = q->get_free();
    if (b) b->reset(0);
    auto flush = [&]() -> bool {      // hands the batch over and takes an empty one
        b->prm = prm;
        q->put_ready(b);
        b = q->get_free();
        if (!b) return false;
        b->reset(id);
        return true;
    };
    while (b && !fh->eof()) {
        fh->get_next();
        const std::string &seq = fh->read;
        // Seq::initialize -> the sketch is non-empty iff the read has a full window and only ACGT letters
        const bool sketched = seq.size() + 1 >= (uint64_t) a->w + a->k &&
                              all_acgt((const unsigned char *) seq.data(), seq.size());
        if (sketched) {
            covg += seq.size();                                         // :404
            if (covg / a->genome_size > a->max_covg) break;             // :405-408
            if (a->illumina && prm.expected_kmers == std::numeric_limits<uint32_t>::max()) {
                // the threshold changes from here on: reads before this one had no sketch, so no hits
                prm.expected_kmers = (uint32_t) (seq.size() * 2 / a->w);    // :413-416
            }
        }
        // a read without a sketch consumes an id too (:409-412): it travels in the batch so that ids stay
        // consecutive, and yields nothing
        if (b->size + seq.size() > b->cap && b->size > 0 && !flush()) break;
        if (!b->reserve(b->size + seq.size() + 64)) {
            out->rc = PMAP_EHIP;
            out->error = "out of pinned host memory for a read batch";
            break;
        }
        b->append(seq);
        id++;
        if (sketched && id > 10000000) break;                           // :420-423
        if (b->size >= batch_bases && !flush()) break;
    }
    if (fh->in.io_error && out->rc == PMAP_OK) {
        out->rc = PMAP_EIO;
        out->error = "error while reading the fastaq file";
    }
    if (b) {
        b->prm = prm;
        q->put_ready(b);
    }
    out->covg = covg;
    q->finish();
}

// host side of one mapped batch: filter_clusters -> add_node calls (+ Read::hits, + coverage)
int finish_batch(const pmap_index_t *idx, pmap_result_t *res, pmap_pangraph *pg, bool keep_hits,
                 pmap_coverage_t *cov) {
    pmap_counts_t cn;
    int rc = pmap_result_counts(res, &cn);   // waits for the batch
    if (rc) return rc;
    const size_t nc = cn.n_clusters, nh = cn.n_hits;
    if (nc == 0) return PMAP_OK;
    std::vector<uint64_t> begin(nc), order(nc);
    std::vector<uint32_t> size(nc), rid(nc), prg(nc), fpos(nc), lpos(nc), frank(nc);
    std::vector<uint8_t> fwd(nc), alive(nc);
    if ((rc = pmap_result_clusters(res, begin.data(), size.data(), rid.data(), prg.data(), fwd.data(), fpos.data(),
                                   lpos.data(), frank.data())))
        return rc;
    if ((rc = pmap_filter_clusters(nc, rid.data(), fpos.data(), size.data(), prg.data(), frank.data(), fwd.data(),
                                   lpos.data(), order.data(), alive.data())))
        return rc;
    std::vector<uint32_t> hpos, hrec;
    if (keep_hits) {
        hpos.resize(nh);
        hrec.resize(nh);
        if ((rc = pmap_result_hits(res, 1, nullptr, hpos.data(), nullptr, hrec.data(), nullptr))) return rc;
    }
    for (size_t j = 0; j < nc; ++j) {
        if (!alive[j]) continue;
        const uint64_t c = order[j];
        pg_add_node(pg, prg[c], rid[c], size[c], fwd[c] != 0);
        if (keep_hits) {
            if (!pg->hits.reserve_more(size[c])) return pmap_fail(PMAP_ELIMIT, "out of host memory for the hits of the reads");
            for (uint64_t i = begin[c]; i < begin[c] + size[c]; ++i)
                pg->hits.push_unchecked(PgHit{rid[c], prg[c], hpos[i], hrec[i], idx->path_rank[hrec[i]], fwd[c]});
        }
    }
    pg->by_path_valid = false;
    if (cov) {
        // add_hits_to_kmergraphs (src/pangenome/pangraph.cpp:347-383) walks Node::reads, a multiset that
        // holds a read once per add_node call, and for each entry every hit of the read on that node: a
        // hit counts as many times as its (read, prg) has surviving clusters
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> calls;
        for (size_t j = 0; j < nc; ++j)
            if (alive[j]) calls[{rid[order[j]], prg[order[j]]}] += 1;
        std::vector<uint32_t> weight(nc, 0);
        for (size_t j = 0; j < nc; ++j)
            if (alive[j]) weight[order[j]] = calls[{rid[order[j]], prg[order[j]]}];
        if ((rc = pmap_coverage_add(cov, res, weight.data()))) return rc;
    }
    return PMAP_OK;
}

}  // namespace

extern "C" int pmap_pangraph_from_read_file(const char *reads_path, const pmap_index_t *idx,
                                            const pmap_map_args_t *a, pmap_pangraph_t **out, uint32_t *covg_out) {
    if (!reads_path || !idx || !a || !out) return pmap_fail(PMAP_EINVAL, "null argument");
    if (a->clean) return pmap_fail(PMAP_EINVAL, "clean=true (de Bruijn cleaning of the pangraph) is outside this library");
    if (a->w == 0 || a->genome_size == 0) return pmap_fail(PMAP_EINVAL, "w and genome_size must be positive");
    int rc = pmap_init(-1);
    if (rc) return rc;
    FastaqReader fh;
    if (!fh.open(reads_path))
        return pmap_fail(PMAP_EIO, std::string("Unable to open fastaq file ") + reads_path);

    pmap_params_t prm;
    memset(&prm, 0, sizeof(prm));
    prm.w = a->w;
    prm.k = a->k;
    prm.max_diff = a->max_diff;
    prm.fraction = pmap_fraction(a->e_rate, a->k);                 // src/utils.cpp:394
    prm.min_cluster_size = a->min_cluster_size;
    prm.expected_kmers = std::numeric_limits<uint32_t>::max();     // :395
    // 32 Mbases per batch: the reader's pinned buffers stay cache- and TLB-friendly on the host (a 2 GB FASTQ maps at
    // 2.9 Gbases/s with 32-Mbase batches and at 1.6 with 256-Mbase ones) and the GPU is far from busy either way
    const uint64_t batch_bases = a->batch_bases ? a->batch_bases : (32ull << 20);
    const bool keep_hits = !(a->flags & PMAP_PG_NO_HITS);

    pmap_pangraph *pg = new pmap_pangraph();
    pg->keep_hits = keep_hits;
    if ((a->flags & PMAP_PG_COVERAGE) && (rc = pmap_coverage_create(idx, &pg->cov))) {
        delete pg;
        return rc;
    }
    Batch batches[3];
    BatchQueue q;
    for (Batch &b : batches) {
        if (!b.reserve(batch_bases + (1u << 20))) {
            for (Batch &x : batches) if (x.data) pmap_pinned_free(x.data);
            delete pg;
            return pmap_fail(PMAP_EHIP, "out of pinned host memory for the read batches");
        }
        q.free_list.push_back(&b);
    }
    ReaderResult rr;
    std::thread reader(reader_thread, &fh, a, prm, batch_bases, &q, &rr);

    pmap_reads_t *rds[2] = {nullptr, nullptr};
    pmap_result_t *ress[2] = {nullptr, nullptr};
    int slot = 0, pending = -1;
    while (Batch *b = q.get_ready()) {
        const uint32_t n = (uint32_t) (b->offsets.size() - 1);
        if (n == 0) { q.put_free(b); continue; }
        if (!rds[slot]) rc = pmap_reads_create(b->data, b->offsets.data(), n, b->first_id, &rds[slot]);
        else rc = pmap_reads_update(rds[slot], b->data, b->offsets.data(), n, b->first_id);
        const pmap_params_t bprm = b->prm;
        q.put_free(b);                                             // the upload has finished
        if (rc == PMAP_OK) rc = pmap_map_async(idx, rds[slot], &bprm, &ress[slot]);
        if (rc == PMAP_OK && pending >= 0) rc = finish_batch(idx, ress[pending], pg, keep_hits, pg->cov);
        if (rc) break;
        pending = slot;
        slot ^= 1;
    }
    if (rc == PMAP_OK && pending >= 0) rc = finish_batch(idx, ress[pending], pg, keep_hits, pg->cov);
    q.stop();
    reader.join();
    if (getenv("PMAP_TRACE_PIPELINE"))
        fprintf(stderr, "[pmap] read-file pipeline: the mapping thread waited %.3f s for batches, the reader %.3f s for free buffers\n",
                q.mapper_waited, q.reader_waited);
    if (rc == PMAP_OK && rr.rc) rc = pmap_fail(rr.rc, rr.error);
    for (int i = 0; i < 2; ++i) {
        pmap_result_destroy(ress[i]);
        pmap_reads_destroy(rds[i]);
    }
    for (Batch &b : batches) if (b.data) pmap_pinned_free(b.data);
    if (rc) {
        pmap_pangraph_destroy(pg);
        return rc;
    }
    if (covg_out) *covg_out = (uint32_t) (rr.covg / a->genome_size);      // :442, 454
    *out = pg;
    return PMAP_OK;
}

extern "C" int pmap_pangraph_info(const pmap_pangraph_t *pg, pmap_pangraph_info_t *info) {
    if (!pg || !info) return pmap_fail(PMAP_EINVAL, "null argument");
    info->n_nodes = (uint32_t) pg->node_covg.size();
    info->n_reads = (uint32_t) pg->reads.size();
    info->n_read_nodes = 0;
    for (auto &r : pg->reads) info->n_read_nodes += r.second.nodes.size();
    info->n_hits = pg->hits.size();
    info->n_add_node_calls = pg->call_prg.size();
    return PMAP_OK;
}

extern "C" int pmap_pangraph_nodes(const pmap_pangraph_t *pg, uint32_t *prg_id, uint32_t *covg) {
    if (!pg) return pmap_fail(PMAP_EINVAL, "null argument");
    size_t i = 0;
    for (auto &n : pg->node_covg) {
        if (prg_id) prg_id[i] = n.first;
        if (covg) covg[i] = n.second;
        ++i;
    }
    return PMAP_OK;
}

extern "C" int pmap_pangraph_reads(const pmap_pangraph_t *pg, uint32_t *read_id, uint64_t *node_off,
                                   uint32_t *node_prg, uint8_t *node_orientation) {
    if (!pg) return pmap_fail(PMAP_EINVAL, "null argument");
    size_t i = 0;
    uint64_t o = 0;
    for (auto &r : pg->reads) {
        if (read_id) read_id[i] = r.first;
        if (node_off) node_off[i] = o;
        for (size_t j = 0; j < r.second.nodes.size(); ++j) {
            if (node_prg) node_prg[o] = r.second.nodes[j];
            if (node_orientation) node_orientation[o] = r.second.orient[j];
            ++o;
        }
        ++i;
    }
    if (node_off) node_off[i] = o;
    return PMAP_OK;
}

extern "C" int pmap_pangraph_hits(const pmap_pangraph_t *pgc, uint32_t *read_id, uint32_t *prg_id, uint32_t *pos,
                                  uint32_t *rec, uint8_t *is_forward) {
    if (!pgc) return pmap_fail(PMAP_EINVAL, "null argument");
    pmap_pangraph *pg = const_cast<pmap_pangraph *>(pgc);
    pg_sort_hits(pg);
    for (size_t i = 0; i < pg->hits.size(); ++i) {
        const PgHit &h = pg->hits[pg->by_path[i]];
        if (read_id) read_id[i] = h.read;
        if (prg_id) prg_id[i] = h.prg;
        if (pos) pos[i] = h.pos;
        if (rec) rec[i] = h.rec;
        if (is_forward) is_forward[i] = h.fwd;
    }
    return PMAP_OK;
}

extern "C" int pmap_pangraph_calls(const pmap_pangraph_t *pg, uint32_t *prg_id, uint32_t *read_id, uint32_t *size,
                                   uint8_t *is_forward) {
    if (!pg) return pmap_fail(PMAP_EINVAL, "null argument");
    const size_t n = pg->call_prg.size();
    if (prg_id) memcpy(prg_id, pg->call_prg.data(), n * 4);
    if (read_id) memcpy(read_id, pg->call_read.data(), n * 4);
    if (size) memcpy(size, pg->call_size.data(), n * 4);
    if (is_forward) memcpy(is_forward, pg->call_fwd.data(), n);
    return PMAP_OK;
}

// the cluster of every add_node call (src/utils.cpp:346-351): call j owns hits [hit_off[j], hit_off[j+1])
extern "C" int pmap_pangraph_call_hits(const pmap_pangraph_t *pg, uint64_t *hit_off, uint32_t *pos, uint32_t *rec) {
    if (!pg) return pmap_fail(PMAP_EINVAL, "null argument");
    if (!pg->keep_hits) return pmap_fail(PMAP_EINVAL, "the pangraph was built with PMAP_PG_NO_HITS");
    const size_t n = pg->call_prg.size();
    if (hit_off) {
        uint64_t o = 0;
        for (size_t j = 0; j < n; ++j) {
            hit_off[j] = o;
            o += pg->call_size[j];
        }
        hit_off[n] = o;
        if (o != pg->hits.size()) return pmap_fail(PMAP_EINVAL, "hit lists and add_node calls out of step");
    }
    for (size_t i = 0; i < pg->hits.size(); ++i) {
        if (pos) pos[i] = pg->hits[i].pos;
        if (rec) rec[i] = pg->hits[i].rec;
    }
    return PMAP_OK;
}

// KmerNode coverage after add_hits_to_kmergraphs (src/pangenome/pangraph.cpp:347-383), per index record
extern "C" int pmap_pangraph_coverage(const pmap_pangraph_t *pg, uint32_t *forward, uint32_t *reverse) {
    if (!pg) return pmap_fail(PMAP_EINVAL, "null argument");
    if (!pg->cov) return pmap_fail(PMAP_EINVAL, "the pangraph was built without PMAP_PG_COVERAGE");
    return pmap_coverage_fetch(pg->cov, forward, reverse);
}

extern "C" void pmap_pangraph_destroy(pmap_pangraph_t *pg) {
    if (!pg) return;
    pmap_coverage_destroy(pg->cov);
    delete pg;
}
