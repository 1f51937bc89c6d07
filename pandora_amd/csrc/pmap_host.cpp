// Host stages around the device path: the FASTA/FASTQ reader
// (src/fastaq_handler.cpp:12-71), clusterComp ordering + filter_clusters
// (src/minihits.cpp:95-109, src/utils.cpp:251-288), the add_node sequence
// (src/utils.cpp:338-352, src/pangenome/pangraph.cpp:43-140) and the read loop of
// pangraph_from_read_file (src/utils.cpp:379-455).
#include <fcntl.h>
#include <sys/mman.h>
#include <atomic>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#include <tmmintrin.h>
#include <immintrin.h>
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
    auto less = [&](uint64_t a, uint64_t b) {
        if (read_id[a] != read_id[b]) return read_id[a] < read_id[b];
        if (first_pos[a] != first_pos[b]) return first_pos[a] < first_pos[b];
        if (size[a] != size[b]) return size[a] > size[b];
        if (prg_id[a] != prg_id[b]) return prg_id[a] < prg_id[b];
        if (first_rank[a] != first_rank[b]) return first_rank[a] < first_rank[b];
        if (is_forward[a] != is_forward[b]) return is_forward[a] < is_forward[b];
        return a < b;
    };
    // clusters that come grouped by read in ascending order, as a mapped batch delivers them, need sorting only
    // inside each read's handful
    bool by_read = true;
    for (uint64_t i = 1; i < n && by_read; ++i) by_read = read_id[i - 1] <= read_id[i];
    if (by_read) {
        for (uint64_t a = 0; a < n;) {
            uint64_t b = a + 1;
            while (b < n && read_id[b] == read_id[a]) ++b;
            if (b - a > 1) std::sort(order + a, order + b, less);
            a = b;
        }
    } else {
        std::sort(order, order + n, less);
    }
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
    // a stream has no seams: see MemLines
    bool at_limit() const { return false; }
    bool quality_line_beyond_piece() const { return false; }
};

// The same over a piece [begin, piece_end) of a file held in memory (a mapped plain file), for parsing the
// pieces of one file on several threads.  A piece that is not the file's last ends at a seam: the start of
// a line that begins with '>' or '@'.  The piece's parser reads that one line too (`limit` is its end),
// because whether the record before the seam is complete shows only there, and then stops.
struct MemLines {
    const char *p = nullptr, *piece_end = nullptr, *limit = nullptr;
    bool last = true;        // the piece runs to the end of the file: end-of-file semantics apply at `limit`
    bool eofbit = false;
    bool at_eof() {
        if (eofbit) return true;
        if (p < limit) return false;
        if (last) eofbit = true;
        return last;
    }
    bool getline(const char **line, size_t *len) {
        *line = p;
        *len = 0;
        if (eofbit || p >= limit) {
            eofbit = true;
            return false;
        }
        const char *nl = (const char *) memchr(p, '\n', (size_t) (limit - p));
        if (nl) {
            *line = p;
            *len = (size_t) (nl - p);
            p = nl + 1;
            return true;
        }
        *line = p;
        *len = (size_t) (limit - p);
        p = limit;
        // the seam line of a piece may be the file's last line and lack its newline: what the end of the
        // file does to the reader is the last piece's business
        if (!last) return true;
        eofbit = true;
        return false;
    }
    bool at_limit() const { return !last && p >= limit; }
    // after a '+' line: the quality line it announces would be the seam line or lie beyond it, so the
    // seam is not where a record starts
    bool quality_line_beyond_piece() const { return !last && p >= piece_end; }
};

enum { NEXT_RECORD = 0, NEXT_AT_SEAM = 1, NEXT_BAD_SEAM = 2, NEXT_NO_MEMORY = 3 };

// FastaqHandler's `read` member as a plain string
struct StringAcc {
    std::string s;
    bool empty() const { return s.empty(); }
    void clear() { s.clear(); }
    void append(const char *p, size_t n) { s.append(p, n); }
};

// FastaqHandler::get_next (src/fastaq_handler.cpp:43-71) over either source; Acc holds the `read` member
template <typename Src, typename Acc = StringAcc>
struct FastaqReader {
    Src in;
    // the last line read: only its first character and its emptiness matter to get_next
    bool line_empty = true;
    char line_first = 0;
    std::string pending_name;     // the header line that ended the record before
    std::string name;
    Acc read;
    bool eof() { return in.at_eof(); }
    int get_next() {
        if (!line_empty && (line_first == '>' || line_first == '@')) {
            name = pending_name;
            read.clear();
        }
        const char *l;
        size_t n;
        for (;;) {
            if (in.at_limit()) return NEXT_AT_SEAM;
            const bool good = in.getline(&l, &n);
            line_empty = n == 0;
            line_first = n ? l[0] : 0;
            if (!line_empty && (line_first == '>' || line_first == '@')) pending_name.assign(l + 1, n - 1);
            if (!good) return NEXT_RECORD;
            if (!line_empty && line_first == '+') {
                if (in.quality_line_beyond_piece()) return NEXT_BAD_SEAM;
                in.getline(&l, &n);   // quality line
                line_empty = n == 0;
                line_first = n ? l[0] : 0;
                if (!line_empty && (line_first == '>' || line_first == '@')) pending_name.assign(l + 1, n - 1);
            } else if (line_empty || line_first == '>' || line_first == '@') {
                if (!read.empty() || line_empty) return NEXT_RECORD;
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
    FastaqReader<LineStream> fh;
    if (!fh.in.open(path))
        return pmap_fail(PMAP_EIO, std::string("Unable to open fastaq file ") + path);   // src/fastaq_handler.cpp:18-21
    std::string all, allnames;
    std::vector<uint64_t> off{0}, noff{0};
    while (!fh.eof()) {
        fh.get_next();
        all += fh.read.s;
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


static int read_fastaq_pieces(const char *path, uint64_t piece_bytes, char **bases, uint64_t **offsets, uint32_t *n_reads);

extern "C" int pmap_read_fastaq_pieces(const char *path, uint64_t piece_bytes, char **bases, uint64_t **offsets,
                                       uint32_t *n_reads) {
    if (!path || piece_bytes == 0) return pmap_fail(PMAP_EINVAL, "null argument");
    return read_fastaq_pieces(path, piece_bytes, bases, offsets, n_reads);
}

// ---------------------------------------------------------------------------
// pangenome::Graph summary

// One hit of Read::hits: position in the read, record, path rank of the record.  Read, prg and strand are those of
// the add_node call the hit came in with (all hits of a cluster share them): the call arrays hold them once.
struct PgHit {
    uint32_t pos, rec, rank;
};

// The hits of a batch are written once, every element, right after the array is made: the array is not
// zeroed first (a second pass over memory that is about to be overwritten), and its pages are mapped in one call
// (madvise MADV_POPULATE_WRITE; without it they come in by a fault per page while the records are written).
template <typename T>
struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <typename U> void construct(U *p) { ::new ((void *) p) U; }
    template <typename U, typename... Args> void construct(U *p, Args &&...args) { ::new ((void *) p) U(std::forward<Args>(args)...); }
};
typedef std::vector<PgHit, NoInitAlloc<PgHit>> PgHitVec;

static void populate_for_write(void *p, size_t bytes) {
    static std::atomic<bool> works{true};       // (kernels without it answer EINVAL once)
    const uintptr_t page = 4096, lo = ((uintptr_t) p + page - 1) & ~(page - 1), hi = ((uintptr_t) p + bytes) & ~(page - 1);
    if (works.load(std::memory_order_relaxed) && hi > lo && bytes >= (1u << 20) &&
        madvise((void *) lo, hi - lo, MADV_POPULATE_WRITE) != 0)
        works.store(false, std::memory_order_relaxed);
}

// A run of add_node calls and what they leave in pangenome::Graph::reads.  The read loop numbers the reads in
// file order and a read's calls come together (clusterComp orders by read first), so the reads are kept in rising
// id order: read i has the nodes rn_prg / rn_orient [read_start[i], read_start[i + 1]), the last one up to the end.
struct PgCalls {
    std::vector<uint32_t> call_prg, call_read, call_size;
    std::vector<uint8_t> call_fwd;
    std::vector<uint32_t> read_ids;
    std::vector<uint64_t> read_start;
    std::vector<uint32_t> rn_prg;
    std::vector<uint8_t> rn_orient;

    // pangenome::Graph::add_node (src/pangenome/pangraph.cpp:126-140) with record_read_info (:108-122); the
    // node's own part, get_node (:59-76) and covg, is counted from the calls when it is asked for
    void add(uint32_t prg, uint32_t read, uint32_t size, bool first_fwd) {
        if (read_ids.empty() || read_ids.back() != read) {
            read_ids.push_back(read);
            read_start.push_back(rn_prg.size());
        }
        const uint8_t orientation = size > 0 && first_fwd;
        if (rn_prg.size() == read_start.back() || rn_prg.back() != prg || rn_orient.back() != orientation) {
            rn_prg.push_back(prg);
            rn_orient.push_back(orientation);
        }
        call_prg.push_back(prg);
        call_read.push_back(read);
        call_size.push_back(size);
        call_fwd.push_back(first_fwd);
    }
    // the calls of a later stretch of the file
    void append(const PgCalls &o) {
        size_t c0 = 0, r0 = 0, n0 = 0;      // calls, reads and read nodes of `o` that go through add()
        if (!o.read_ids.empty() && !read_ids.empty() && read_ids.back() == o.read_ids.front()) {
            // (a read never spans two batches; if one did, its calls would go on as add() has them)
            for (; c0 < o.call_read.size() && o.call_read[c0] == o.read_ids.front(); ++c0)
                add(o.call_prg[c0], o.call_read[c0], o.call_size[c0], o.call_fwd[c0] != 0);
            r0 = 1;
            n0 = o.read_start.size() > 1 ? (size_t) o.read_start[1] : o.rn_prg.size();
        }
        const uint64_t shift = (uint64_t) rn_prg.size() - n0;
        read_ids.insert(read_ids.end(), o.read_ids.begin() + r0, o.read_ids.end());
        for (size_t i = r0; i < o.read_start.size(); ++i) read_start.push_back(o.read_start[i] + shift);
        rn_prg.insert(rn_prg.end(), o.rn_prg.begin() + n0, o.rn_prg.end());
        rn_orient.insert(rn_orient.end(), o.rn_orient.begin() + n0, o.rn_orient.end());
        call_prg.insert(call_prg.end(), o.call_prg.begin() + c0, o.call_prg.end());
        call_read.insert(call_read.end(), o.call_read.begin() + c0, o.call_read.end());
        call_size.insert(call_size.end(), o.call_size.begin() + c0, o.call_size.end());
        call_fwd.insert(call_fwd.end(), o.call_fwd.begin() + c0, o.call_fwd.end());
    }
};

struct pmap_pangraph : PgCalls {
    // pangenome::Graph::nodes: node id (= prg id) -> covg, the number of calls that named it; counted on demand
    std::map<uint32_t, uint32_t> node_covg;
    bool nodes_valid = false;
    // Read::hits in add_node call order, each call's cluster in pComp order: one segment per mapped batch, kept as
    // the mapper thread made it (copying 10^7 records once more on the retiring thread made that thread the slow
    // one); `flat` = the segments in one array, made when hits are asked for by global index
    std::vector<PgHitVec> hit_seg;
    size_t n_hits = 0;
    PgHitVec flat;
    bool flat_valid = false;
    bool keep_hits = true;
    std::vector<uint64_t> by_path;    // hits in Read::hits order (indices into `hits`), built on demand
    std::vector<uint32_t> hit_call;   // the call of every hit, built with by_path
    bool by_path_valid = false;
    bool has_cov = false;             // PMAP_PG_COVERAGE: kmer node coverage, summed over the mapping contexts
    std::vector<uint32_t> cov_fwd, cov_rev;
};

static void pg_count_nodes(pmap_pangraph *pg) {
    if (pg->nodes_valid) return;
    pg->node_covg.clear();
    uint32_t top = 0;
    for (uint32_t p : pg->call_prg) top = std::max(top, p);
    std::vector<uint32_t> n(pg->call_prg.empty() ? 0 : (size_t) top + 1, 0);
    for (uint32_t p : pg->call_prg) n[p]++;
    for (size_t p = 0; p < n.size(); ++p)
        if (n[p]) pg->node_covg.emplace_hint(pg->node_covg.end(), (uint32_t) p, n[p]);
    pg->nodes_valid = true;
}

static void pg_flatten_hits(pmap_pangraph *pg) {
    if (pg->flat_valid) return;
    pg->flat.clear();
    pg->flat.reserve(pg->n_hits);
    for (const auto &seg : pg->hit_seg) pg->flat.insert(pg->flat.end(), seg.begin(), seg.end());
    pg->flat_valid = true;
}

static void pg_sort_hits(pmap_pangraph *pg) {
    if (pg->by_path_valid) return;
    pg_flatten_hits(pg);
    // the call every hit belongs to: call j owns the next call_size[j] hits
    pg->hit_call.resize(pg->n_hits);
    {
        size_t o = 0;
        for (size_t j = 0; j < pg->call_size.size() && o < pg->hit_call.size(); ++j) {
            const size_t e = std::min(pg->hit_call.size(), o + pg->call_size[j]);
            std::fill(pg->hit_call.begin() + o, pg->hit_call.begin() + e, (uint32_t) j);
            o = e;
        }
    }
    // per (read, prg): pComp_path (src/minihits.cpp:77-93): path, forward first, position.  The hits themselves
    // stay in call order (pmap_pangraph_call_hits); this is an index into them.
    pg->by_path.resize(pg->n_hits);
    std::iota(pg->by_path.begin(), pg->by_path.end(), 0ull);
    const PgHit *h = pg->flat.data();
    const uint32_t *hc = pg->hit_call.data(), *cread = pg->call_read.data(), *cprg = pg->call_prg.data();
    const uint8_t *cfwd = pg->call_fwd.data();
    std::sort(pg->by_path.begin(), pg->by_path.end(), [=](uint64_t x, uint64_t y) {
        const PgHit &a = h[x], &b = h[y];
        const uint32_t ca = hc[x], cb = hc[y];
        if (cread[ca] != cread[cb]) return cread[ca] < cread[cb];
        if (cprg[ca] != cprg[cb]) return cprg[ca] < cprg[cb];
        if (a.rank != b.rank) return a.rank < b.rank;
        if (cfwd[ca] != cfwd[cb]) return cfwd[ca] > cfwd[cb];
        if (a.pos != b.pos) return a.pos < b.pos;
        return x < y;
    });
    pg->by_path_valid = true;
}

// ---------------------------------------------------------------------------
// pangraph_from_read_file as a pipeline of threads:
//   parse     plain file: the file is mapped into memory and cut into pieces at seams (line starts that
//             look like a record header); several threads run the FastaqHandler state machine over one
//             piece each, check every read for letters outside ACGT and copy the reads into a pinned
//             buffer - a piece becomes one device batch.  gzip file: one thread does the same over the
//             inflated stream.
//   sequence  one thread takes the pieces in file order, checks that every seam really was a record
//             boundary (else the two pieces are parsed again as one), and applies the read loop's rules
//             (read ids, coverage stop, 10M-read stop, Illumina k-mer count; src/utils.cpp:397-440).
//   map       one thread per mapping context (pmap_init_devices; by default three on the caller's GPU):
//             upload, kernels, results back, filter_clusters, coverage - whole batches, in any order.
//   retire    the calling thread performs the add_node calls batch by batch in file order.

namespace {

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Pinned buffers are expensive to make (the driver locks and maps every page): the ones a call is done with
// are kept for the next call, up to 1 GiB of them (PMAP_PINNED_CACHE_BYTES: another bound, 0 keeps none; a call
// of pmap_pangraph_from_read_file on a plain file uses about 20 buffers of 8 MiB).  pmap_release_host_cache()
// gives them back.
struct PinnedCache {
    std::mutex m;
    std::vector<std::pair<char *, size_t>> free_list;
    size_t held = 0;
    size_t cap = getenv("PMAP_PINNED_CACHE_BYTES") ? (size_t) strtoull(getenv("PMAP_PINNED_CACHE_BYTES"), nullptr, 10) : (size_t) 1 << 30;
    char *take(size_t want, size_t *got) {
        std::lock_guard<std::mutex> l(m);
        size_t best = free_list.size();
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].second >= want && (best == free_list.size() || free_list[i].second < free_list[best].second))
                best = i;
        if (best == free_list.size()) return nullptr;
        char *p = free_list[best].first;
        *got = free_list[best].second;
        held -= *got;
        free_list.erase(free_list.begin() + best);
        return p;
    }
    bool give(char *p, size_t bytes) {       // false: not kept, the caller frees it
        std::lock_guard<std::mutex> l(m);
        if (bytes < (1u << 20) || held + bytes > cap) return false;
        free_list.push_back({p, bytes});
        held += bytes;
        return true;
    }
    size_t release_all() {
        std::vector<std::pair<char *, size_t>> mine;
        {
            std::lock_guard<std::mutex> l(m);
            mine.swap(free_list);
            held = 0;
        }
        size_t bytes = 0;
        for (auto &b : mine) {
            pmap_pinned_free(b.first);
            bytes += b.second;
        }
        return bytes;
    }
};
PinnedCache g_pinned_cache;

bool all_acgt(const unsigned char *p, size_t n);
bool pack_acgt(const unsigned char *p, size_t n, uint32_t *out);

// Reads leave the host 2-bit packed (a quarter of the bytes cross the bus; at one byte per base the copies of
// three mapper threads held the whole pipeline at what the link carries, 23 GB/s): the parser packs a read as
// soon as it is complete, its letters still in the cache.  PMAP_HOST_PACK=0: letters travel, the GPU packs.
bool host_pack_default() {
    const char *e = getenv("PMAP_HOST_PACK");
    return !e || atoi(e) != 0;
}

struct Batch {
    char *data = nullptr;          // pinned (plain memory when `pinned` is off: parsing without a GPU, or host_pack)
    bool pinned = true;
    bool host_pack = false;        // `words` travels instead of `data`
    bool zero_copy = false;        // host_pack, parsed from a file in memory: reads are packed from where they lie (PackAcc),
                                   // `data` is not used and `size` only counts letters
    uint32_t *words = nullptr;     // pinned: ceil(length / 16) words per read
    size_t n_words = 0, words_cap = 0;
    size_t size = 0, cap = 0;
    std::vector<uint64_t> offsets{0};
    std::vector<uint8_t> sketchable;      // per read: long enough for one window and only ACGT letters
    std::vector<uint8_t> acgt;            // per read: only ACGT letters
    uint32_t first_id = 0;
    uint64_t seq = 0;
    pmap_params_t prm;
    // a piece of a mapped file
    const char *piece_begin = nullptr, *piece_end = nullptr;
    bool piece_last = false;
    int exit_code = NEXT_RECORD;
    void reset() {
        size = 0;
        n_words = 0;
        offsets.assign(1, 0);
        sketchable.clear();
        acgt.clear();
        exit_code = NEXT_RECORD;
    }
    bool reserve_words(size_t want) {
        if (want <= words_cap) return true;
        size_t ncap = std::max(want, words_cap + words_cap / 2), got = 0;
        char *q = g_pinned_cache.take(4 * ncap, &got);
        if (q) ncap = got / 4;
        else q = (char *) pmap_pinned_alloc(4 * ncap);
        if (!q) return false;
        if (n_words) memcpy(q, words, 4 * n_words);
        release_words();
        words = (uint32_t *) q;
        words_cap = ncap;
        return true;
    }
    void release_words() {
        if (words && !g_pinned_cache.give((char *) words, 4 * words_cap)) pmap_pinned_free(words);
        words = nullptr;
        words_cap = 0;
    }
    // the read [start, start + len) of `data` is complete: packs it (host_pack) and notes whether it has a sketch
    bool finish_read(size_t start, size_t len, uint32_t w, uint32_t k) {
        return finish_read_at((const unsigned char *) data + start, len, w, k);
    }
    bool finish_read_at(const unsigned char *letters, size_t len, uint32_t w, uint32_t k) {
        bool ok;
        if (host_pack) {
            const size_t nw = (len + 15) / 16;
            if (n_words + nw > words_cap && !reserve_words(n_words + nw + (1u << 16))) return false;
            ok = pack_acgt(letters, len, words + n_words);
            n_words += nw;
        } else {
            ok = all_acgt(letters, len);
        }
        acgt.push_back(ok);
        sketchable.push_back(len + 1 >= (uint64_t) w + k && ok);
        return true;
    }
    // `live`: bytes from the start of the buffer that must survive a move (a read under construction lies
    // behind `size`)
    bool reserve(size_t want, size_t live = 0) {
        if (want <= cap) return true;
        if (host_pack && zero_copy) return reserve_words(n_words + want / 16 + 1024);   // the letters stay in the file
        size_t ncap = std::max(want, cap + cap / 2);
        char *q = pinned ? g_pinned_cache.take(want, &ncap) : nullptr;
        if (!q) q = (char *) (pinned ? pmap_pinned_alloc(ncap) : malloc(ncap));
        if (!q) return false;
        live = std::min(std::max(live, size), cap);
        if (live) memcpy(q, data, live);
        release();
        data = q;
        cap = ncap;
        return true;
    }
    void release() {
        if (data && pinned) {
            if (!g_pinned_cache.give(data, cap)) pmap_pinned_free(data);
        } else if (data) {
            free(data);
        }
        data = nullptr;
        cap = 0;
        release_words();
    }
    bool append(const std::string &seq_, uint32_t w, uint32_t k) {      // false: out of memory
        if (size + seq_.size() > cap && !reserve(size + seq_.size() + 64)) return false;
        memcpy(data + size, seq_.data(), seq_.size());
        const size_t start = size;
        size += seq_.size();
        offsets.push_back(size);
        return finish_read(start, seq_.size(), w, k);
    }
    size_t n_reads() const { return offsets.size() - 1; }
    void truncate(size_t n) {
        if (host_pack)
            for (size_t i = n; i < n_reads(); ++i) n_words -= (size_t) (offsets[i + 1] - offsets[i] + 15) / 16;
        offsets.resize(n + 1);
        sketchable.resize(n);
        acgt.resize(n);
        size = offsets[n];
    }
};

// true iff every letter is one that nt4 (src/inthash.cpp:42-63) maps below 4: A, C, G, T in either case or
// a raw code 0..3.  This is the parsers' hot loop: 16 letters per step where SSE2 is there.
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

// nt4 (src/inthash.cpp:42-63) of n letters into ceil(n / 16) words of 16 bases, the first base in the top two
// bits, unused low bits zero; false iff some letter maps to 4 (the words are then of no use).  Letters:
// code = ((c >> 1) ^ (c >> 2)) & 3 for A, C, G, T in either case; the bytes 0..3 stand for themselves.
__attribute__((target("ssse3"))) bool pack_acgt_ssse3(const unsigned char *p, size_t n, uint32_t *out) {
    const __m128i up = _mm_set1_epi8((char) 0xDF), A = _mm_set1_epi8('A'), C = _mm_set1_epi8('C'),
                  G = _mm_set1_epi8('G'), T = _mm_set1_epi8('T'), hi = _mm_set1_epi8((char) 0xFC),
                  zero = _mm_setzero_si128(), ones = _mm_set1_epi8((char) 0xFF), three = _mm_set1_epi8(3),
                  m41 = _mm_set1_epi16(0x0104), m161 = _mm_set1_epi32(0x00010010);
    __m128i acc = zero;
    size_t i = 0, w = 0;
    for (; i + 16 <= n; i += 16, ++w) {
        const __m128i c = _mm_loadu_si128((const __m128i *) (p + i)), u = _mm_and_si128(c, up);
        const __m128i raw = _mm_cmpeq_epi8(_mm_and_si128(c, hi), zero);           // c < 4
        __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(u, A), _mm_cmpeq_epi8(u, C)),
                                  _mm_or_si128(_mm_cmpeq_epi8(u, G), _mm_cmpeq_epi8(u, T)));
        acc = _mm_or_si128(acc, _mm_andnot_si128(_mm_or_si128(ok, raw), ones));
        __m128i code = _mm_and_si128(_mm_xor_si128(_mm_srli_epi16(c, 1), _mm_srli_epi16(c, 2)), three);
        code = _mm_or_si128(_mm_andnot_si128(raw, code), _mm_and_si128(raw, c));
        // pairs of letters -> 4 bits (first letter * 4 + second), pairs of those -> 8 bits, one byte per 32-bit lane
        const __m128i nib = _mm_maddubs_epi16(code, m41);        // bytes (lo, hi) * (4, 1): lo is the earlier letter
        const __m128i byt = _mm_madd_epi16(nib, m161);           // words (lo, hi) * (16, 1)
        const __m128i b16 = _mm_packs_epi32(byt, byt), b8 = _mm_packus_epi16(b16, b16);
        out[w] = __builtin_bswap32((uint32_t) _mm_cvtsi128_si32(b8));              // first four letters on top
    }
    bool bad = _mm_movemask_epi8(acc) != 0;
    if (i < n) {
        uint32_t word = 0;
        for (size_t j = 0; i + j < n; ++j) {
            const unsigned char c = p[i + j], u = c & 0xDFu;
            const uint32_t code = c < 4 ? c : u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u;
            bad |= code > 3;
            word |= (code & 3u) << (30 - 2 * j);
        }
        out[w] = word;
    }
    return !bad;
}

// the same, 32 letters (two words) per step
__attribute__((target("avx2"))) bool pack_acgt_avx2(const unsigned char *p, size_t n, uint32_t *out) {
    const __m256i up = _mm256_set1_epi8((char) 0xDF), A = _mm256_set1_epi8('A'), C = _mm256_set1_epi8('C'),
                  G = _mm256_set1_epi8('G'), T = _mm256_set1_epi8('T'), hi = _mm256_set1_epi8((char) 0xFC),
                  zero = _mm256_setzero_si256(), ones = _mm256_set1_epi8((char) 0xFF), three = _mm256_set1_epi8(3),
                  m41 = _mm256_set1_epi16(0x0104), m161 = _mm256_set1_epi32(0x00010010);
    __m256i acc = zero;
    size_t i = 0, w = 0;
    for (; i + 32 <= n; i += 32, w += 2) {
        const __m256i c = _mm256_loadu_si256((const __m256i *) (p + i)), u = _mm256_and_si256(c, up);
        const __m256i raw = _mm256_cmpeq_epi8(_mm256_and_si256(c, hi), zero);           // c < 4
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, A), _mm256_cmpeq_epi8(u, C)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(u, G), _mm256_cmpeq_epi8(u, T)));
        acc = _mm256_or_si256(acc, _mm256_andnot_si256(_mm256_or_si256(ok, raw), ones));
        __m256i code = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(c, 1), _mm256_srli_epi16(c, 2)), three);
        code = _mm256_or_si256(_mm256_andnot_si256(raw, code), _mm256_and_si256(raw, c));
        // as in the 16-letter form, in each half of the register: its four bytes end up in the half's low word
        const __m256i nib = _mm256_maddubs_epi16(code, m41);
        const __m256i byt = _mm256_madd_epi16(nib, m161);
        const __m256i b16 = _mm256_packs_epi32(byt, byt), b8 = _mm256_packus_epi16(b16, b16);
        out[w] = __builtin_bswap32((uint32_t) _mm256_extract_epi32(b8, 0));
        out[w + 1] = __builtin_bswap32((uint32_t) _mm256_extract_epi32(b8, 4));
    }
    const bool bad = _mm256_movemask_epi8(acc) != 0;
    // what is left, fewer than 32 letters, by the 16-letter form (whole words: i is a multiple of 16)
    return pack_acgt_ssse3(p + i, n - i, out + w) && !bad;
}

bool pack_acgt_plain(const unsigned char *p, size_t n, uint32_t *out) {
    bool bad = false;
    for (size_t i = 0, w = 0; i < n; i += 16, ++w) {
        uint32_t word = 0;
        for (size_t j = 0; j < 16 && i + j < n; ++j) {
            const unsigned char c = p[i + j], u = c & 0xDFu;
            const uint32_t code = c < 4 ? c : u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u;
            bad |= code > 3;
            word |= (code & 3u) << (30 - 2 * j);
        }
        out[w] = word;
    }
    return !bad;
}

bool pack_acgt(const unsigned char *p, size_t n, uint32_t *out) {
    static const bool avx2 = __builtin_cpu_supports("avx2"), ssse3 = __builtin_cpu_supports("ssse3");
    return avx2 ? pack_acgt_avx2(p, n, out) : ssse3 ? pack_acgt_ssse3(p, n, out) : pack_acgt_plain(p, n, out);
}

// FastaqHandler's `read` member kept inside a batch's buffer: the lines of a read are copied once, to where
// the batch wants them.  The member outlives the call that returned it (a blank line makes the next call
// return it again, or append to it): a returned read stays readable where it is, and is copied to the end
// of the buffer only if the reader touches it again.
struct BatchAcc {
    Batch *b = nullptr;
    size_t start = 0, len = 0;
    bool returned = false;      // [start, start + len) is already a read of the batch
    bool oom = false;
    bool empty() const { return len == 0; }
    void clear() {
        start = b->size;
        len = 0;
        returned = false;
    }
    void unshare() {
        if (!b->reserve(b->size + len + 64, start + len)) { oom = true; return; }
        memmove(b->data + b->size, b->data + start, len);
        start = b->size;
        returned = false;
    }
    void append(const char *p, size_t n) {
        if (returned) unshare();
        if (oom) return;
        if (start + len + n > b->cap && !b->reserve(start + len + n + 64, start + len)) { oom = true; return; }
        memcpy(b->data + start + len, p, n);
        len += n;
    }
    // the reader returned: the member's content is the batch's next read
    bool emit(uint32_t w, uint32_t k) {
        if (returned) unshare();
        if (oom) return false;
        b->size = start + len;
        b->offsets.push_back(b->size);
        if (!b->finish_read(start, len, w, k)) { oom = true; return false; }
        returned = true;
        return true;
    }
};

// The same member for batches that travel packed (Batch::zero_copy): a read that is one line of the file - nearly
// every read of a FASTQ file - is packed from where it lies in the mapped file and its letters are never copied
// (the parser threads of a 16-core host move 4 bytes of memory traffic per base otherwise, and that, not
// their instructions, is what bounds them); a read of several lines is put together in `own` first.
struct PackAcc {
    Batch *b = nullptr;
    const char *ext = nullptr;
    std::string own;
    size_t len = 0;
    bool pieced = false;        // the read's letters are in `own`
    bool oom = false;
    bool empty() const { return len == 0; }
    void clear() {
        ext = nullptr;
        pieced = false;
        own.clear();
        len = 0;
    }
    void append(const char *p, size_t n) {
        if (len == 0 && !pieced) {
            ext = p;
            len = n;
            return;
        }
        if (!pieced) {
            own.assign(ext, len);
            pieced = true;
        }
        own.append(p, n);
        len = own.size();
    }
    bool emit(uint32_t w, uint32_t k) {
        b->size += len;
        b->offsets.push_back(b->size);
        if (!b->finish_read_at((const unsigned char *) (pieced ? own.data() : ext), len, w, k)) {
            oom = true;
            return false;
        }
        return true;
    }
};

// The read loop of pangraph_from_read_file (src/utils.cpp:397-440) up to the point where a read is handed
// to the sketch: ids, coverage stop, 10M-read stop, Illumina expected k-mer count.
struct ReadLoop {
    const pmap_map_args_t *a;
    pmap_params_t prm;
    uint64_t covg = 0;
    uint32_t id = 0;
    bool stopped = false;
    // false: the loop ends in front of this read.  true: the read takes the next id and travels in its batch
    // (a read without a sketch consumes an id too, :409-412, and yields nothing); `stopped` may be set behind it.
    bool take(size_t len, bool sketched) {
        if (sketched) {
            covg += len;                                                // :404
            if (covg / a->genome_size > a->max_covg) {                  // :405-408
                stopped = true;
                return false;
            }
            // the threshold changes from here on: reads before this one had no sketch, so no hits
            if (a->illumina && prm.expected_kmers == std::numeric_limits<uint32_t>::max())
                prm.expected_kmers = (uint32_t) (len * 2 / a->w);       // :413-416
        }
        id++;
        if (sketched && id > 10000000) stopped = true;                  // :420-423
        return true;
    }
};

struct Outcome : PgCalls {  // what one mapped batch contributes to the pangraph
    PgHitVec hits;                  // call order, each call's cluster in pComp order
};

struct Pipeline {
    const pmap_index_t *idx = nullptr;
    const pmap_map_args_t *a = nullptr;
    bool keep_hits = true, want_cov = false;
    bool host_pack = host_pack_default();
    bool pieces = false;            // the batches are pieces of a file in memory
    uint64_t piece_bytes = 0, batch_bases = 0;

    std::mutex m;
    std::condition_variable cv;
    bool abort = false;             // an error somewhere: everybody winds down
    int rc = PMAP_OK;
    std::string error;
    // pinned buffers
    std::vector<Batch *> all_buffers, free_buffers;
    size_t max_buffers = 4;
    // parsed pieces waiting for the sequencer, by piece number
    std::map<uint64_t, Batch *> parsed;
    // sequenced batches waiting for a mapper
    std::deque<Batch *> ready;
    bool sequenced_all = false;
    uint64_t n_batches = 0;         // valid once sequenced_all
    // mapped batches waiting to be retired, by batch number
    std::map<uint64_t, Outcome *> done;
    uint64_t covg = 0;
    double waited_parse = 0, waited_map = 0, waited_retire = 0;
    // (PMAP_TRACE_PIPELINE) parser threads: waiting for a buffer, parsing; mapper threads: upload, map, what follows
    double t_wait_buffer = 0, t_parse = 0, t_upload = 0, t_map = 0, t_finish = 0;
    void add_time(double *slot, double dt) {
        std::lock_guard<std::mutex> l(m);
        *slot += dt;
    }

    void fail(int code, const std::string &msg) {
        std::lock_guard<std::mutex> l(m);
        if (rc == PMAP_OK) {
            rc = code;
            error = msg;
        }
        abort = true;
        cv.notify_all();
    }
    Batch *get_buffer(size_t want) {        // nullptr: aborted, or out of pinned memory (fail() has been called)
        Batch *b = nullptr;
        {
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [&] { return abort || !free_buffers.empty() || all_buffers.size() < max_buffers; });
            if (abort) return nullptr;
            if (!free_buffers.empty()) {
                b = free_buffers.back();
                free_buffers.pop_back();
            } else {
                b = new Batch();
                b->host_pack = host_pack;
                b->pinned = !host_pack;         // the letters stay on the host
                b->zero_copy = host_pack && pieces;
                all_buffers.push_back(b);
            }
        }
        b->reset();
        if (!b->reserve(want)) {
            fail(PMAP_EHIP, "out of pinned host memory for a read batch");
            return nullptr;
        }
        return b;
    }
    void put_buffer(Batch *b) {
        { std::lock_guard<std::mutex> l(m); free_buffers.push_back(b); }
        cv.notify_all();
    }
    void push_ready(Batch *b) {
        { std::lock_guard<std::mutex> l(m); ready.push_back(b); }
        cv.notify_all();
    }
    void finish_sequencing(uint64_t batches, uint64_t covg_) {
        { std::lock_guard<std::mutex> l(m); sequenced_all = true; n_batches = batches; covg = covg_; }
        cv.notify_all();
    }
    ~Pipeline() {
        for (Batch *b : all_buffers) {
            b->release();
            delete b;
        }
        for (auto &kv : done) delete kv.second;
    }
};

// ---- source 1: a gzip (or unmappable) file, parsed and sequenced by one thread

void stream_source(Pipeline *pl, FastaqReader<LineStream> *fh, pmap_params_t prm0) {
    ReadLoop loop{pl->a, prm0};
    uint64_t seq = 0;
    Batch *b = pl->get_buffer(pl->batch_bases + (1u << 20));
    auto flush = [&]() -> bool {      // hands the batch over and takes an empty one
        b->prm = loop.prm;
        b->seq = seq++;
        pl->push_ready(b);
        b = pl->get_buffer(pl->batch_bases + (1u << 20));
        if (b) b->first_id = loop.id;
        return b != nullptr;
    };
    if (b) b->first_id = 0;
    while (b && !fh->eof()) {
        fh->get_next();
        const std::string &seq_ = fh->read.s;
        // Seq::initialize -> the sketch is non-empty iff the read has a full window and only ACGT letters
        const bool sketched = seq_.size() + 1 >= (uint64_t) pl->a->w + pl->a->k &&
                              all_acgt((const unsigned char *) seq_.data(), seq_.size());
        if (!loop.take(seq_.size(), sketched)) break;
        if (b->size + seq_.size() > b->cap && b->size > 0) {
            loop.id--;                   // the batch in hand ends before this read
            const bool ok = flush();
            loop.id++;
            if (!ok) break;
        }
        if (!b->append(seq_, pl->a->w, pl->a->k)) {
            pl->fail(PMAP_EHIP, "out of pinned host memory for a read batch");
            break;
        }
        if (loop.stopped) break;
        if (b->size >= pl->batch_bases && !flush()) break;
    }
    if (fh->in.io_error) pl->fail(PMAP_EIO, "error while reading the fastaq file");
    if (b) {
        if (b->n_reads() > 0) {
            b->prm = loop.prm;
            b->seq = seq++;
            pl->push_ready(b);
        } else {
            pl->put_buffer(b);
        }
    }
    pl->finish_sequencing(seq, loop.covg);
}

// ---- source 2: a plain file mapped into memory, parsed in pieces by several threads

struct MappedFile {
    const char *base = nullptr;
    size_t size = 0;
    int fd = -1;
    bool open(const char *path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return false;
        void *p = mmap(nullptr, (size_t) st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) return false;
        base = (const char *) p;
        size = (size_t) st.st_size;
        (void) madvise(p, size, MADV_SEQUENTIAL);
        return true;
    }
    ~MappedFile() {
        if (base) munmap((void *) base, size);
        if (fd >= 0) close(fd);
    }
};

// The first seam at or behind `from`: a line start whose line begins with '>' or '@' and is not the file's
// last line; `end` if there is none.  (Behind a seam line that ends the file the reader's next step depends
// on whether a record was open in front of it, which the parser of the piece behind the seam cannot know;
// with at least one more line behind the seam line it does not.)
const char *find_seam(const char *from, const char *end) {
    const char *p = from;
    while (p < end) {
        const char *nl = (const char *) memchr(p, '\n', (size_t) (end - p));
        if (!nl || nl + 1 >= end) return end;
        if (nl[1] == '>' || nl[1] == '@') {
            const char *nl2 = (const char *) memchr(nl + 1, '\n', (size_t) (end - (nl + 1)));
            if (!nl2 || nl2 + 1 >= end) return end;
            return nl + 1;
        }
        p = nl + 1;
    }
    return end;
}

// runs the state machine over the piece [begin, end) (+ the seam line behind it) into the batch
template <typename Acc>
void parse_piece_with(Batch *b, const char *begin, const char *end, const char *file_end, uint32_t w, uint32_t k) {
    b->piece_begin = begin;
    b->piece_end = end;
    b->piece_last = end >= file_end;
    FastaqReader<MemLines, Acc> fh;
    fh.read.b = b;
    fh.read.clear();
    fh.in.p = begin;
    fh.in.piece_end = end;
    fh.in.last = b->piece_last;
    if (b->piece_last) {
        fh.in.limit = file_end;
    } else {
        const char *nl = (const char *) memchr(end, '\n', (size_t) (file_end - end));
        fh.in.limit = nl ? nl + 1 : file_end;
    }
    b->exit_code = NEXT_RECORD;
    // every get_next of the reference's loop `while (!fh.eof()) { fh.get_next(); ... }` that completes inside
    // the piece or on its seam line; a call that is still under way at the seam goes on in the next piece,
    // whose parser starts from the state the seam line leaves behind whatever came before it
    while (fh.in.last ? !fh.eof() : !fh.in.at_limit()) {
        const int code = fh.get_next();
        if (code != NEXT_RECORD) {
            b->exit_code = code;
            break;
        }
        // (blank lines make the reader return a read twice: a piece's reads can outgrow the piece)
        if (!fh.read.emit(w, k)) {
            b->exit_code = NEXT_NO_MEMORY;
            break;
        }
    }
    if (fh.read.oom) b->exit_code = NEXT_NO_MEMORY;
}

void parse_piece(Batch *b, const char *begin, const char *end, const char *file_end, uint32_t w, uint32_t k) {
    if (b->zero_copy) parse_piece_with<PackAcc>(b, begin, end, file_end, w, k);
    else parse_piece_with<BatchAcc>(b, begin, end, file_end, w, k);
}

// The pieces of a mapped file one after the other on the calling thread, with the same seam rule as the
// threaded pipeline: emit(batch) for every stretch of the file that was parsed from a true record boundary.
template <typename Emit>
bool parse_in_pieces(const char *base, size_t size, uint64_t piece_bytes, uint32_t w, uint32_t k, Batch *b, Emit emit) {
    const char *cur = base, *file_end = base + size;
    while (cur < file_end) {
        const char *begin = cur;
        const char *end = begin + piece_bytes >= file_end ? file_end : find_seam(begin + piece_bytes, file_end);
        for (;;) {
            b->reset();
            if (!b->reserve((size_t) (end - begin) + 64)) return false;
            parse_piece(b, begin, end, file_end, w, k);
            if (b->exit_code == NEXT_NO_MEMORY) return false;
            if (b->exit_code != NEXT_BAD_SEAM) break;
            // the seam lay inside a record: take the next piece in as well
            end = end + piece_bytes >= file_end ? file_end : find_seam(end + piece_bytes, file_end);
        }
        emit(*b);
        cur = end;
    }
    return true;
}

struct PieceSource {
    Pipeline *pl;
    MappedFile *file;
    std::mutex m;                  // hands out the pieces
    const char *cur = nullptr;
    uint64_t next_piece = 0;
    bool stop = false;             // the sequencer has seen the end of the read loop

    void worker() {
        const char *file_end = file->base + file->size;
        bool populate = !(getenv("PMAP_POPULATE") && atoi(getenv("PMAP_POPULATE")) == 0);
        for (;;) {
            // the buffer first, then the piece: whoever holds the lowest unparsed piece always has a buffer
            const double tw = now_s();
            Batch *b = pl->get_buffer(pl->piece_bytes + (1u << 16));
            if (!b) return;
            const double tp = now_s();
            pl->add_time(&pl->t_wait_buffer, tp - tw);
            const char *begin, *end;
            {
                std::lock_guard<std::mutex> l(m);
                if (stop || cur >= file_end) {
                    pl->put_buffer(b);
                    return;
                }
                begin = cur;
                end = begin + pl->piece_bytes >= file_end ? file_end : find_seam(begin + pl->piece_bytes, file_end);
                cur = end;
                b->seq = next_piece++;
            }
            // the piece's pages into this process' page table in one go (one call instead of a fault per 64 KB;
            // kernels without it answer EINVAL and the pages come in by faults as before).  PMAP_POPULATE=0: never
            if (populate) {
                const uintptr_t page = 4096, lo = (uintptr_t) begin & ~(page - 1);
                if (madvise((void *) lo, (size_t) ((uintptr_t) end - lo), MADV_POPULATE_READ) != 0) populate = false;
            }
            // a read is never longer than its piece
            if (!b->reserve((size_t) (end - begin) + 64)) {
                pl->fail(PMAP_EHIP, "out of pinned host memory for a read batch");
                return;
            }
            parse_piece(b, begin, end, file_end, pl->a->w, pl->a->k);
            if (b->exit_code == NEXT_NO_MEMORY) {
                pl->fail(PMAP_EHIP, "out of pinned host memory for a read batch");
                return;
            }
            {
                std::lock_guard<std::mutex> l(pl->m);
                pl->parsed[b->seq] = b;
                pl->t_parse += now_s() - tp;
            }
            pl->cv.notify_all();
        }
    }

    // all pieces have been handed out and none with this number will ever arrive
    bool no_such_piece(uint64_t i) {
        std::lock_guard<std::mutex> l(m);
        return (stop || cur >= file->base + file->size) && i >= next_piece;
    }

    Batch *wait_piece(uint64_t i) {      // nullptr: there is no piece i (end of file), or abort
        std::unique_lock<std::mutex> l(pl->m);
        const double t0 = now_s();
        for (;;) {
            auto it = pl->parsed.find(i);
            if (it != pl->parsed.end()) {
                Batch *b = it->second;
                pl->parsed.erase(it);
                pl->waited_parse += now_s() - t0;
                return b;
            }
            if (pl->abort) return nullptr;
            l.unlock();
            const bool none = no_such_piece(i);
            l.lock();
            if (none && pl->parsed.find(i) == pl->parsed.end()) return nullptr;
            pl->cv.wait_for(l, std::chrono::milliseconds(2));
        }
    }

    void sequencer(pmap_params_t prm0) {
        ReadLoop loop{pl->a, prm0};
        const char *file_end = file->base + file->size;
        uint64_t piece = 0, seq = 0;
        while (!loop.stopped) {
            Batch *b = wait_piece(piece++);
            if (!b) break;
            // a seam inside a record (the line behind a '+' line starts with '@' or '>'): the piece behind it
            // was parsed from a wrong state.  Both are parsed again as one piece, here.
            while (b->exit_code == NEXT_BAD_SEAM) {
                Batch *nb = wait_piece(piece++);
                if (!nb) {
                    pl->fail(PMAP_EIO, "internal: a piece of the read file has no successor");
                    pl->put_buffer(b);
                    b = nullptr;
                    break;
                }
                const char *begin = b->piece_begin, *end = nb->piece_end;
                pl->put_buffer(nb);
                b->reset();
                if (!b->reserve((size_t) (end - begin) + 64)) {
                    pl->fail(PMAP_EHIP, "out of pinned host memory for a read batch");
                    pl->put_buffer(b);
                    b = nullptr;
                    break;
                }
                parse_piece(b, begin, end, file_end, pl->a->w, pl->a->k);
                if (b->exit_code == NEXT_NO_MEMORY) {
                    pl->fail(PMAP_EHIP, "out of pinned host memory for a read batch");
                    pl->put_buffer(b);
                    b = nullptr;
                    break;
                }
            }
            if (!b) break;
            b->first_id = loop.id;
            size_t n = 0;
            for (; n < b->n_reads(); ++n) {
                if (!loop.take((size_t) (b->offsets[n + 1] - b->offsets[n]), b->sketchable[n] != 0)) break;
                if (loop.stopped) { ++n; break; }
            }
            b->truncate(n);
            if (n == 0) {
                pl->put_buffer(b);
                continue;
            }
            b->prm = loop.prm;
            b->seq = seq++;
            pl->push_ready(b);
        }
        {
            std::lock_guard<std::mutex> l(m);
            stop = true;
        }
        pl->finish_sequencing(seq, loop.covg);
        // pieces that were parsed beyond the stop
        std::lock_guard<std::mutex> l(pl->m);
        for (auto &kv : pl->parsed) pl->free_buffers.push_back(kv.second);
        pl->parsed.clear();
        pl->cv.notify_all();
    }
};

// host side of one mapped batch: filter_clusters -> the add_node calls it will make (+ Read::hits, + coverage)
// page-locked room for the two hit columns a mapper thread fetches per batch (the copy from the device goes
// straight into it; into pageable memory it is staged by the runtime at a quarter of the speed), kept by the thread
struct HitStage {
    uint32_t *pos = nullptr, *rec = nullptr, *rank = nullptr;
    size_t cap = 0;
    bool reserve(size_t n) {
        if (n <= cap) return true;
        release();
        const size_t want = n + n / 4 + 1024;
        pos = (uint32_t *) pmap_pinned_alloc(sizeof(uint32_t) * want);
        rec = (uint32_t *) pmap_pinned_alloc(sizeof(uint32_t) * want);
        rank = (uint32_t *) pmap_pinned_alloc(sizeof(uint32_t) * want);
        if (!pos || !rec || !rank) {
            release();
            return false;
        }
        cap = want;
        return true;
    }
    void release() {
        if (pos) pmap_pinned_free(pos);
        if (rec) pmap_pinned_free(rec);
        if (rank) pmap_pinned_free(rank);
        pos = rec = rank = nullptr;
        cap = 0;
    }
};

int finish_batch(const pmap_index_t *idx, pmap_result_t *res, Outcome *out, bool keep_hits, pmap_coverage_t *cov,
                 HitStage *stage) {
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
    const uint32_t *hpos = nullptr, *hrec = nullptr, *hrank = nullptr;
    size_t kept_hits = 0;
    if (keep_hits) {
        if (!stage->reserve(nh)) return pmap_fail(PMAP_EHIP, "out of pinned host memory for a batch's hits");
        if ((rc = pmap_result_hits(res, 1, nullptr, stage->pos, nullptr, stage->rec, nullptr)) ||
            (rc = pmap_result_hit_ranks(res, stage->rank)))
            return rc;
        hpos = stage->pos;
        hrec = stage->rec;
        hrank = stage->rank;
        for (size_t j = 0; j < nc; ++j)
            if (alive[j]) kept_hits += size[order[j]];
        out->hits.resize(kept_hits);
        populate_for_write(out->hits.data(), sizeof(PgHit) * kept_hits);
    }
    size_t o = 0;
    for (size_t j = 0; j < nc; ++j) {
        if (!alive[j]) continue;
        const uint64_t c = order[j];
        out->add(prg[c], rid[c], size[c], fwd[c] != 0);       // src/utils.cpp:346-351
        if (keep_hits) {
            // (a cluster never reaches beyond the hits: the rows come from the same batch)
            const uint64_t e = std::min<uint64_t>(begin[c] + size[c], nh);
            for (uint64_t i = begin[c]; i < e; ++i) out->hits[o++] = PgHit{hpos[i], hrec[i], hrank[i]};
        }
    }
    if (keep_hits) out->hits.resize(o);
    if (cov) {
        // add_hits_to_kmergraphs (src/pangenome/pangraph.cpp:347-383) walks Node::reads, a multiset that
        // holds a read once per add_node call, and for each entry every hit of the read on that node: a
        // hit counts as many times as its (read, prg) has surviving clusters
        // (a read's clusters are consecutive in clusterComp order and few: counted pair against pair)
        std::vector<uint32_t> weight(nc, 0);
        for (size_t a = 0; a < nc;) {
            size_t b = a + 1;
            while (b < nc && rid[order[b]] == rid[order[a]]) ++b;
            for (size_t j = a; j < b; ++j) {
                if (!alive[j]) continue;
                uint32_t same = 0;
                for (size_t q = a; q < b; ++q) same += alive[q] && prg[order[q]] == prg[order[j]];
                weight[order[j]] = same;
            }
            a = b;
        }
        if ((rc = pmap_coverage_add(cov, res, weight.data()))) return rc;
    }
    return PMAP_OK;
}

// one mapping context's thread: whole batches, one after the other
void mapper_thread(Pipeline *pl, void *ctx, std::vector<uint32_t> *cov_fwd, std::vector<uint32_t> *cov_rev) {
    pmap_reads_t *rd = nullptr;
    pmap_result_t *res = nullptr;
    pmap_coverage_t *cov = nullptr;
    HitStage stage;
    int rc = pmap_bind_context(ctx);
    if (rc == PMAP_OK && pl->want_cov) rc = pmap_coverage_create(pl->idx, &cov);
    while (rc == PMAP_OK) {
        Batch *b = nullptr;
        {
            std::unique_lock<std::mutex> l(pl->m);
            const double t0 = now_s();
            // not too far ahead of the thread that retires the batches
            pl->cv.wait(l, [&] { return pl->abort || ((!pl->ready.empty() && pl->done.size() < 8) || (pl->ready.empty() && pl->sequenced_all)); });
            pl->waited_map += now_s() - t0;
            if (pl->abort || pl->ready.empty()) break;
            b = pl->ready.front();
            pl->ready.pop_front();
        }
        const uint32_t n = (uint32_t) b->n_reads();
        const double t0 = now_s();
        if (b->host_pack) {
            static const uint32_t no_words = 0;
            const uint32_t *wp = b->words ? b->words : &no_words;
            if (!rd) rc = pmap_reads_create_packed(wp, b->offsets.data(), b->acgt.data(), n, b->first_id, &rd);
            else rc = pmap_reads_update_packed(rd, wp, b->offsets.data(), b->acgt.data(), n, b->first_id);
        } else if (!rd) rc = pmap_reads_create(b->data, b->offsets.data(), n, b->first_id, &rd);
        else rc = pmap_reads_update(rd, b->data, b->offsets.data(), n, b->first_id);
        const pmap_params_t prm = b->prm;
        const uint64_t seq = b->seq;
        pl->put_buffer(b);                                          // the upload has finished
        Outcome *out = new Outcome();
        const double t1 = now_s();
        if (rc == PMAP_OK) rc = pmap_map(pl->idx, rd, &prm, &res);
        const double t2 = now_s();
        if (rc == PMAP_OK) rc = finish_batch(pl->idx, res, out, pl->keep_hits, cov, &stage);
        {
            std::lock_guard<std::mutex> l(pl->m);
            pl->t_upload += t1 - t0;
            pl->t_map += t2 - t1;
            pl->t_finish += now_s() - t2;
        }
        if (rc != PMAP_OK) {
            delete out;
            break;
        }
        {
            std::lock_guard<std::mutex> l(pl->m);
            pl->done[seq] = out;
        }
        pl->cv.notify_all();
    }
    if (rc == PMAP_OK && cov) {
        cov_fwd->assign(pl->idx->n_recs, 0);
        cov_rev->assign(pl->idx->n_recs, 0);
        rc = pmap_coverage_fetch(cov, cov_fwd->data(), cov_rev->data());
    }
    if (rc != PMAP_OK) pl->fail(rc, pmap_last_error());
    stage.release();
    pmap_coverage_destroy(cov);
    pmap_result_destroy(res);
    pmap_reads_destroy(rd);
}

unsigned parse_threads(size_t n_contexts) {
    if (const char *e = getenv("PMAP_PARSE_THREADS")) return (unsigned) std::max(0, atoi(e));
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 4;
    if (const char *e = getenv("OMP_NUM_THREADS")) {       // this process' share of a big host
        const int v = atoi(e);
        if (v > 0) hw = std::min<unsigned>(hw, (unsigned) v);
    }
    // the cores the mapper threads (one per context), the sequencer and the retiring thread leave; those threads
    // are light - reads travel packed, a batch's clusters are sorted read by read, the add_node calls are made
    // batch by batch - so with many contexts the parsers still get five cores in eight (16 cores: 3 contexts 11
    // parsers 48 Gbases/s, 14 parsers 41; 2 contexts 8 parsers 33, 14 parsers 41; 8 contexts 6 parsers 31, 12 parsers 34)
    const unsigned others = (unsigned) n_contexts + 2;
    const unsigned want = std::max(hw > others ? hw - others : 0u, hw * 5 / 8);
    return std::max(1u, std::min(want, 32u));
}

bool looks_gzipped(const char *path) {
    unsigned char magic[2] = {0, 0};
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    const size_t n = fread(magic, 1, 2, f);
    fclose(f);
    return n == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
}

}  // namespace

extern "C" uint64_t pmap_release_host_cache(void) { return g_pinned_cache.release_all(); }

extern "C" int pmap_pack_reads(const char *bases, const uint64_t *offsets, uint32_t n_reads, uint32_t *packed,
                               uint8_t *acgt_only) {
    if (!offsets || (n_reads && (!packed || !acgt_only))) return pmap_fail(PMAP_EINVAL, "null argument");
    size_t w = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        if (offsets[r + 1] < offsets[r]) return pmap_fail(PMAP_EINVAL, "read offsets not monotone");
        const size_t len = (size_t) (offsets[r + 1] - offsets[r]);
        if (len && !bases) return pmap_fail(PMAP_EINVAL, "null argument");
        acgt_only[r] = pack_acgt((const unsigned char *) bases + offsets[r], len, packed + w) ? 1 : 0;
        w += (len + 15) / 16;
    }
    return PMAP_OK;
}

static int read_fastaq_pieces(const char *path, uint64_t piece_bytes, char **bases, uint64_t **offsets, uint32_t *n_reads) {
    MappedFile mapped;
    std::string all;
    std::vector<uint64_t> off{0};
    if (!mapped.open(path)) {
        // empty or unmappable: nothing to cut into pieces, but the file must exist
        FILE *f = fopen(path, "rb");
        if (!f) return pmap_fail(PMAP_EIO, std::string("Unable to open fastaq file ") + path);
        fclose(f);
    } else {
        Batch b;
        b.pinned = false;
        const bool ok = parse_in_pieces(mapped.base, mapped.size, piece_bytes, 1, 1, &b, [&](const Batch &p) {
            const uint64_t base0 = all.size();
            all.append(p.data, p.size);
            for (size_t i = 1; i < p.offsets.size(); ++i) off.push_back(base0 + p.offsets[i]);
        });
        b.release();
        if (!ok) return pmap_fail(PMAP_ELIMIT, "out of memory while reading the fastaq file");
    }
    if (n_reads) *n_reads = (uint32_t) (off.size() - 1);
    if (bases) {
        *bases = (char *) malloc(all.size() + 1);
        memcpy(*bases, all.data(), all.size());
        (*bases)[all.size()] = 0;
    }
    if (offsets) {
        *offsets = (uint64_t *) malloc(sizeof(uint64_t) * off.size());
        memcpy(*offsets, off.data(), sizeof(uint64_t) * off.size());
    }
    return PMAP_OK;
}

extern "C" int pmap_pangraph_from_read_file(const char *reads_path, const pmap_index_t *idx,
                                            const pmap_map_args_t *a, pmap_pangraph_t **out, uint32_t *covg_out) {
    if (!reads_path || !idx || !a || !out) return pmap_fail(PMAP_EINVAL, "null argument");
    if (a->clean) return pmap_fail(PMAP_EINVAL, "clean=true (de Bruijn cleaning of the pangraph) is outside this library");
    if (a->w == 0 || a->genome_size == 0) return pmap_fail(PMAP_EINVAL, "w and genome_size must be positive");
    std::vector<void *> contexts;
    int rc = pmap_contexts_for_file(&contexts);
    if (rc) return rc;

    pmap_params_t prm;
    memset(&prm, 0, sizeof(prm));
    prm.w = a->w;
    prm.k = a->k;
    prm.max_diff = a->max_diff;
    prm.fraction = pmap_fraction(a->e_rate, a->k);                 // src/utils.cpp:394
    prm.min_cluster_size = a->min_cluster_size;
    prm.expected_kmers = std::numeric_limits<uint32_t>::max();     // :395

    Pipeline pl;
    pl.idx = idx;
    pl.a = a;
    pl.keep_hits = !(a->flags & PMAP_PG_NO_HITS);
    pl.want_cov = (a->flags & PMAP_PG_COVERAGE) != 0;
    // A piece of a mapped file becomes one batch (with qualities and headers about two bytes per base).  Big pieces
    // cost the mapper threads less per base (8 GB of FASTQ on 16 cores: 64 MB pieces 42-49 Gbases/s, 96-128 MB 52-56,
    // 192-256 MB 45-51), small files want enough pieces for every parser thread: a quarter of a thread's share of
    // the file, between 16 and 128 MB (set below, once the file is open).  The sequential reader cuts its batches
    // at 32 Mbases.
    pl.batch_bases = a->batch_bases ? a->batch_bases : (32ull << 20);
    pl.piece_bytes = a->batch_bases ? a->batch_bases : (64ull << 20);

    // the file must open (src/fastaq_handler.cpp:18-21) before any thread starts
    FastaqReader<LineStream> fh;
    MappedFile mapped;
    const unsigned n_parse = parse_threads(contexts.size());
    const bool parallel = n_parse > 0 && !looks_gzipped(reads_path) && mapped.open(reads_path);
    if (!parallel && !fh.in.open(reads_path))
        return pmap_fail(PMAP_EIO, std::string("Unable to open fastaq file ") + reads_path);

    pmap_pangraph *pg = new pmap_pangraph();
    pg->keep_hits = pl.keep_hits;
    pg->has_cov = pl.want_cov;
    const double t_start = now_s();

    std::vector<std::thread> threads;
    PieceSource src{&pl, &mapped};
    pl.pieces = parallel;
    if (parallel) {
        src.cur = mapped.base;
        if (!a->batch_bases)
            pl.piece_bytes = std::min<uint64_t>(128ull << 20, std::max<uint64_t>(16ull << 20, mapped.size / (4ull * std::max(1u, n_parse))));
        const unsigned workers = (unsigned) std::max<uint64_t>(1, std::min<uint64_t>(n_parse, mapped.size / std::max<uint64_t>(pl.piece_bytes, 1) + 1));
        pl.max_buffers = workers + 2 * contexts.size() + 2;
        for (unsigned i = 0; i < workers; ++i) threads.emplace_back(&PieceSource::worker, &src);
        threads.emplace_back(&PieceSource::sequencer, &src, prm);
    } else {
        pl.max_buffers = 2 * contexts.size() + 2;
        threads.emplace_back(stream_source, &pl, &fh, prm);
    }
    std::vector<std::vector<uint32_t>> cov_fwd(contexts.size()), cov_rev(contexts.size());
    for (size_t i = 0; i < contexts.size(); ++i)
        threads.emplace_back(mapper_thread, &pl, contexts[i], &cov_fwd[i], &cov_rev[i]);

    // retire: the add_node calls (src/utils.cpp:346-351) of every batch, in file order
    for (uint64_t next = 0;;) {
        Outcome *o = nullptr;
        {
            std::unique_lock<std::mutex> l(pl.m);
            const double t0 = now_s();
            pl.cv.wait(l, [&] { return pl.abort || pl.done.count(next) || (pl.sequenced_all && next >= pl.n_batches); });
            pl.waited_retire += now_s() - t0;
            if (pl.abort || !pl.done.count(next)) break;
            o = pl.done[next];
            pl.done.erase(next);
        }
        pl.cv.notify_all();
        pg->append(*o);     // (the mapper thread has made the calls: here they only join those of the batches before)
        if (!o->hits.empty()) {
            pg->n_hits += o->hits.size();
            pg->hit_seg.push_back(std::move(o->hits));
        }
        delete o;
        ++next;
    }
    {
        // whatever ended the loop, every thread must be able to leave: buffers and queues are released by `abort`
        std::lock_guard<std::mutex> l(pl.m);
        if (pl.rc != PMAP_OK) pl.abort = true;
    }
    pl.cv.notify_all();
    for (auto &t : threads) t.join();
    pg->by_path_valid = false;
    pg->flat_valid = false;
    pg->nodes_valid = false;
    if (getenv("PMAP_TRACE_PIPELINE"))
        fprintf(stderr, "[pmap] read-file pipeline: %.3f s, %s, %zu mapping contexts; waited: sequencer for pieces %.3f s, "
                        "mappers for batches %.3f s (all of them together), retiring thread for mapped batches %.3f s; "
                        "parser threads together: %.3f s waiting for a buffer, %.3f s parsing; mapper threads together: "
                        "%.3f s upload, %.3f s map, %.3f s filter + coverage\n",
                now_s() - t_start, parallel ? "mapped file parsed in pieces" : "one parsing thread", contexts.size(),
                pl.waited_parse, pl.waited_map, pl.waited_retire, pl.t_wait_buffer, pl.t_parse, pl.t_upload, pl.t_map,
                pl.t_finish);
    if (pl.rc != PMAP_OK) {
        const int code = pl.rc;
        const std::string msg = pl.error;
        pmap_pangraph_destroy(pg);
        return pmap_fail(code, msg);
    }
    if (pl.want_cov) {
        pg->cov_fwd.assign(idx->n_recs, 0);
        pg->cov_rev.assign(idx->n_recs, 0);
        for (size_t i = 0; i < contexts.size(); ++i)
            for (size_t j = 0; j < cov_fwd[i].size(); ++j) {
                pg->cov_fwd[j] += cov_fwd[i][j];
                pg->cov_rev[j] += cov_rev[i][j];
            }
    }
    if (covg_out) *covg_out = (uint32_t) (pl.covg / a->genome_size);      // :442, 454
    *out = pg;
    return PMAP_OK;
}

extern "C" int pmap_pangraph_info(const pmap_pangraph_t *pg, pmap_pangraph_info_t *info) {
    if (!pg || !info) return pmap_fail(PMAP_EINVAL, "null argument");
    pg_count_nodes(const_cast<pmap_pangraph *>(pg));
    info->n_nodes = (uint32_t) pg->node_covg.size();
    info->n_reads = (uint32_t) pg->read_ids.size();
    info->n_read_nodes = pg->rn_prg.size();
    info->n_hits = pg->n_hits;
    info->n_add_node_calls = pg->call_prg.size();
    return PMAP_OK;
}

extern "C" int pmap_pangraph_nodes(const pmap_pangraph_t *pg, uint32_t *prg_id, uint32_t *covg) {
    if (!pg) return pmap_fail(PMAP_EINVAL, "null argument");
    pg_count_nodes(const_cast<pmap_pangraph *>(pg));
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
    const size_t i = pg->read_ids.size(), o = pg->rn_prg.size();
    if (read_id && i) memcpy(read_id, pg->read_ids.data(), sizeof(uint32_t) * i);
    if (node_off && i) memcpy(node_off, pg->read_start.data(), sizeof(uint64_t) * i);
    if (node_prg && o) memcpy(node_prg, pg->rn_prg.data(), sizeof(uint32_t) * o);
    if (node_orientation && o) memcpy(node_orientation, pg->rn_orient.data(), o);
    if (node_off) node_off[i] = o;
    return PMAP_OK;
}

extern "C" int pmap_pangraph_hits(const pmap_pangraph_t *pgc, uint32_t *read_id, uint32_t *prg_id, uint32_t *pos,
                                  uint32_t *rec, uint8_t *is_forward) {
    if (!pgc) return pmap_fail(PMAP_EINVAL, "null argument");
    pmap_pangraph *pg = const_cast<pmap_pangraph *>(pgc);
    pg_sort_hits(pg);
    for (size_t i = 0; i < pg->n_hits; ++i) {
        const PgHit &h = pg->flat[pg->by_path[i]];
        const uint32_t call = pg->hit_call[pg->by_path[i]];
        if (read_id) read_id[i] = pg->call_read[call];
        if (prg_id) prg_id[i] = pg->call_prg[call];
        if (pos) pos[i] = h.pos;
        if (rec) rec[i] = h.rec;
        if (is_forward) is_forward[i] = pg->call_fwd[call];
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
        if (o != pg->n_hits) return pmap_fail(PMAP_EINVAL, "hit lists and add_node calls out of step");
    }
    size_t i = 0;
    for (const auto &seg : pg->hit_seg)
        for (const PgHit &h : seg) {
            if (pos) pos[i] = h.pos;
            if (rec) rec[i] = h.rec;
            ++i;
        }
    return PMAP_OK;
}

// KmerNode coverage after add_hits_to_kmergraphs (src/pangenome/pangraph.cpp:347-383), per index record
extern "C" int pmap_pangraph_coverage(const pmap_pangraph_t *pg, uint32_t *forward, uint32_t *reverse) {
    if (!pg) return pmap_fail(PMAP_EINVAL, "null argument");
    if (!pg->has_cov) return pmap_fail(PMAP_EINVAL, "the pangraph was built without PMAP_PG_COVERAGE");
    if (forward) memcpy(forward, pg->cov_fwd.data(), pg->cov_fwd.size() * sizeof(uint32_t));
    if (reverse) memcpy(reverse, pg->cov_rev.data(), pg->cov_rev.size() * sizeof(uint32_t));
    return PMAP_OK;
}

extern "C" void pmap_pangraph_destroy(pmap_pangraph_t *pg) {
    if (!pg) return;
    delete pg;
}
