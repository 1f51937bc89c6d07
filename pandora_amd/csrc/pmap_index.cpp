// Host side of the flat minimizer index: replaces Index::minhash
// (include/index.h:16), Index::add_record / Index::load (src/index.cpp:20-37,
// 75-124) and the per-PRG KmerGraph::min_path_length value
// (src/kmergraph.cpp:664-684, 716-820).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>

#include <sys/stat.h>

#include <chrono>
#include <thread>

#include "pmap_internal.h"

static thread_local std::string g_error;

// PMAP_TRACE_INDEX=1 prints the time of every stage of an index load / build to stderr
struct StageClock {
    bool on = getenv("PMAP_TRACE_INDEX") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pmap index] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

static unsigned worker_count(size_t items, size_t min_per_worker) {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    hw = std::min(hw, 16u);
    return (unsigned) std::max<size_t>(1, std::min<size_t>(hw, items / std::max<size_t>(min_per_worker, 1)));
}

// std::sort over worker threads: sorted runs, then pairwise merges
template <typename T, typename Less>
static void parallel_sort(std::vector<T> &v, Less less) {
    const unsigned nw = worker_count(v.size(), 1u << 16);
    if (nw <= 1) {
        std::sort(v.begin(), v.end(), less);
        return;
    }
    std::vector<size_t> cut(nw + 1);
    for (unsigned i = 0; i <= nw; ++i) cut[i] = v.size() * i / nw;
    {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < nw; ++i)
            th.emplace_back([&, i] { std::sort(v.begin() + cut[i], v.begin() + cut[i + 1], less); });
        for (auto &t : th) t.join();
    }
    for (unsigned width = 1; width < nw; width *= 2) {
        std::vector<std::thread> th;
        for (unsigned i = 0; i + width < nw; i += 2 * width) {
            const size_t a = cut[i], m = cut[i + width], b = cut[std::min(i + 2 * width, nw)];
            th.emplace_back([&, a, m, b] { std::inplace_merge(v.begin() + a, v.begin() + m, v.begin() + b, less); });
        }
        for (auto &t : th) t.join();
    }
}

void pmap_set_error(const std::string &msg) { g_error = msg; }

int pmap_fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

extern "C" const char *pmap_last_error(void) { return g_error.c_str(); }

// prg::Path::operator< (src/prg/path.cpp:166-183) with Interval::operator<
// (src/interval.cpp:47-53) as a three-way comparison over flat intervals.
static int path_cmp(const uint32_t *as, const uint32_t *al, uint64_t an, const uint32_t *bs, const uint32_t *bl,
                    uint64_t bn) {
    uint64_t n = an < bn ? an : bn;
    for (uint64_t i = 0; i < n; ++i) {
        if (as[i] != bs[i]) return as[i] < bs[i] ? -1 : 1;
        if (al[i] != bl[i]) return al[i] < bl[i] ? -1 : 1;
    }
    if (an != bn) return an < bn ? -1 : 1;   // a strict prefix is the smaller path
    return 0;
}

int pmap_index_build_host(const pmap_index_desc_t *d, pmap_index *idx) {
    if (d->k < 1 || d->k > 31)
        return pmap_fail(PMAP_EINVAL, "k must be in [1,31]: the 2k-bit hash mask overflows at k=32 (src/seq.cpp:97)");
    if (d->n_recs >= 0xFFFFFFFFull) return pmap_fail(PMAP_ELIMIT, "more than 2^32-2 index records");
    if (d->n_prg >= 0x7FFFFFFFu) return pmap_fail(PMAP_ELIMIT, "more than 2^31-2 PRGs");
    if (d->n_keys && d->rec_off[d->n_keys] != d->n_recs)
        return pmap_fail(PMAP_EINVAL, "rec_off[n_keys] != n_recs");
    const uint64_t n_iv = d->n_recs ? d->path_off[d->n_recs] : 0;
    const uint64_t key_limit = 1ULL << (2 * d->k);
    StageClock clk;

    idx->k = d->k;
    idx->n_prg = d->n_prg;
    idx->n_recs = d->n_recs;
    idx->rec_prg.assign(d->rec_prg, d->rec_prg + d->n_recs);
    idx->rec_knode.assign(d->rec_knode, d->rec_knode + d->n_recs);
    idx->rec_strand.assign(d->rec_strand, d->rec_strand + d->n_recs);
    idx->path_off.assign(d->path_off, d->path_off + d->n_recs + (d->n_recs ? 1 : 0));
    if (idx->path_off.empty()) idx->path_off.push_back(0);
    idx->iv_start.assign(d->iv_start, d->iv_start + n_iv);
    idx->iv_len.assign(d->iv_len, d->iv_len + n_iv);
    idx->min_path_length.assign(d->min_path_length, d->min_path_length + d->n_prg);
    idx->key_of_rec.assign(d->n_recs, 0);
    idx->path_rank.assign(d->n_recs, 0);

    for (uint64_t i = 0; i < d->n_keys; ++i) {
        if (d->keys[i] >= key_limit) return pmap_fail(PMAP_EINVAL, "index key does not fit in 2k bits");
        if (d->rec_off[i] > d->rec_off[i + 1]) return pmap_fail(PMAP_EINVAL, "rec_off not monotone");
        for (uint64_t j = d->rec_off[i]; j < d->rec_off[i + 1]; ++j) idx->key_of_rec[j] = d->keys[i];
    }
    for (uint64_t j = 0; j < d->n_recs; ++j) {
        if (d->rec_prg[j] >= d->n_prg) return pmap_fail(PMAP_EINVAL, "record prg_id >= n_prg");
        if (d->rec_knode[j] >= 0x7FFFFFFFu) return pmap_fail(PMAP_ELIMIT, "kmer node id >= 2^31-1");
        if (d->path_off[j] > d->path_off[j + 1]) return pmap_fail(PMAP_EINVAL, "path_off not monotone");
        uint64_t len = 0;
        for (uint64_t q = d->path_off[j]; q < d->path_off[j + 1]; ++q) len += d->iv_len[q];
        if (len != d->k)   // the reference asserts this when the record is hit (src/minihit.cpp:17)
            return pmap_fail(PMAP_EINVAL, "a record's path length differs from k");
    }

    clk.lap("copy + validate");
    // rank of every (prg, path): order of (prg_id, prg::Path::operator<)
    auto pcmp = [&](uint32_t a, uint32_t b) {
        if (d->rec_prg[a] != d->rec_prg[b]) return d->rec_prg[a] < d->rec_prg[b] ? -1 : 1;
        uint64_t a0 = d->path_off[a], a1 = d->path_off[a + 1], b0 = d->path_off[b], b1 = d->path_off[b + 1];
        return path_cmp(d->iv_start + a0, d->iv_len + a0, a1 - a0, d->iv_start + b0, d->iv_len + b0, b1 - b0);
    };
    // sorted on a 64-bit prefix (prg, start of the first interval) that sits next to the record number;
    // the full comparison only breaks ties
    struct PathKey {
        uint64_t prefix;
        uint32_t rec;
    };
    std::vector<PathKey> by_path(d->n_recs);
    for (uint64_t j = 0; j < d->n_recs; ++j) {
        const uint64_t first = d->path_off[j] < d->path_off[j + 1] ? d->iv_start[d->path_off[j]] : 0;
        by_path[j] = PathKey{(uint64_t) d->rec_prg[j] << 32 | first, (uint32_t) j};
    }
    parallel_sort(by_path, [&](const PathKey &a, const PathKey &b) {
        if (a.prefix != b.prefix) return a.prefix < b.prefix;
        return pcmp(a.rec, b.rec) < 0;
    });
    uint32_t rank = 0;
    for (uint64_t i = 0; i < d->n_recs; ++i) {
        if (i > 0 && pcmp(by_path[i - 1].rec, by_path[i].rec) != 0) ++rank;
        idx->path_rank[by_path[i].rec] = rank;
    }
    idx->n_paths = d->n_recs ? rank + 1 : 0;

    clk.lap("path ranks");
    // group equal keys (Index::load appends the records of a repeated key line, src/index.cpp:97-110)
    std::vector<std::pair<uint64_t, uint64_t>> kpairs(d->n_keys);   // (key, position in the caller's list)
    for (uint64_t i = 0; i < d->n_keys; ++i) kpairs[i] = {d->keys[i], i};
    parallel_sort(kpairs, [](const std::pair<uint64_t, uint64_t> &a, const std::pair<uint64_t, uint64_t> &b) {
        return a < b;
    });
    std::vector<uint64_t> korder(d->n_keys);
    for (uint64_t i = 0; i < d->n_keys; ++i) korder[i] = kpairs[i].second;
    kpairs.clear();
    kpairs.shrink_to_fit();
    idx->recs.clear();
    idx->recs.reserve(d->n_recs);
    std::vector<uint64_t> ukeys;
    std::vector<uint32_t> uoff, ucnt;
    std::vector<uint32_t> tmp;
    for (uint64_t g = 0; g < d->n_keys;) {
        uint64_t key = d->keys[korder[g]];
        tmp.clear();
        while (g < d->n_keys && d->keys[korder[g]] == key) {
            uint64_t ki = korder[g++];
            for (uint64_t j = d->rec_off[ki]; j < d->rec_off[ki + 1]; ++j) tmp.push_back((uint32_t) j);
        }
        if (tmp.empty()) continue;
        // records of one key sorted by (prg, path rank, strand): hits emitted in this order are
        // already in MinimizerHit::operator< order within one (read, position, prg, strand)
        std::sort(tmp.begin(), tmp.end(), [&](uint32_t a, uint32_t b) {
            if (d->rec_prg[a] != d->rec_prg[b]) return d->rec_prg[a] < d->rec_prg[b];
            if (idx->path_rank[a] != idx->path_rank[b]) return idx->path_rank[a] < idx->path_rank[b];
            if (d->rec_strand[a] != d->rec_strand[b]) return d->rec_strand[a] < d->rec_strand[b];
            return a < b;
        });
        for (size_t i = 1; i < tmp.size(); ++i) {
            uint32_t a = tmp[i - 1], b = tmp[i];
            if (d->rec_prg[a] == d->rec_prg[b] && idx->path_rank[a] == idx->path_rank[b] &&
                (d->rec_strand[a] != 0) == (d->rec_strand[b] != 0)) {
                std::ostringstream os;
                os << "duplicate index record (key " << key << ", prg " << d->rec_prg[a]
                   << "): MinimizerHits::sort would abort on it";
                return pmap_fail(PMAP_EDUP, os.str());
            }
        }
        ukeys.push_back(key);
        uoff.push_back((uint32_t) idx->recs.size());
        ucnt.push_back((uint32_t) tmp.size());
        idx->max_cnt = std::max(idx->max_cnt, (uint32_t) tmp.size());
        for (uint32_t j : tmp) {
            PmapRec r;
            r.prg = d->rec_prg[j];
            r.knode_strand = d->rec_knode[j] << 1 | (d->rec_strand[j] ? 1u : 0u);
            r.rank = idx->path_rank[j];
            r.orig = j;
            idx->recs.push_back(r);
        }
    }
    idx->n_keys = ukeys.size();
    clk.lap("group keys, order records");

    uint32_t log2cap = 6;
    while ((1ull << log2cap) < 2 * idx->n_keys) ++log2cap;
    if (log2cap > 31) return pmap_fail(PMAP_ELIMIT, "minimizer table would exceed 2^31 slots");
    idx->log2cap = log2cap;
    const uint32_t cap = 1u << log2cap, capmask = cap - 1;
    PmapSlot empty{PMAP_EMPTY_KEY, 0, 0, PmapRec{0, 0, 0, 0}};
    idx->slots.assign(cap, empty);
    for (size_t i = 0; i < ukeys.size(); ++i) {
        uint32_t s = pmap_slot_of(ukeys[i], log2cap);
        while (idx->slots[s].key != PMAP_EMPTY_KEY) s = (s + 1) & capmask;
        idx->slots[s] = PmapSlot{ukeys[i], uoff[i], ucnt[i], idx->recs[uoff[i]]};
    }
    // presence filter: 8 to 16 bits per key, two of them set per key, while that takes at most 2^24 bits (2 MiB);
    // 2^23 bits (1 MiB) when that still leaves 6 bits per key: the filter shares the 4 MiB L2 of an XCD with the
    // table lines and tile entries that stream through it, and a filter line that has left the L2 costs as much
    // as the table read it might have saved (measured, 1.1 M keys and every k-mer a lookup: 2^24 bits 33.9,
    // 2^23 bits 35.4 Gbases/s; 158 000 keys: 2^22 bits 521, 2^21 536, 2^20 515 Gbases/s); one bit per key while
    // one bit per key fills at most 1/16 of the filter; pointless once the keys outnumber a quarter of the bits
    uint32_t fbits = 6;
    while ((1ull << fbits) < 8 * idx->n_keys && fbits < 24) ++fbits;
    if (fbits == 24 && idx->n_keys * 6 <= (1ull << 23)) fbits = 23;
    if (const char *cap_env = getenv("PMAP_FILTER_MAX_BITS")) {      // tests, sweeps: 8 bits per slot, at most so many
        const int v = atoi(cap_env);
        if (v >= 6 && v <= 30) fbits = std::min<uint32_t>(log2cap + 3, (uint32_t) v);
    }
    if (idx->n_keys * 4 > (1ull << fbits)) fbits = 0;
    else if (idx->n_keys * 16 > (1ull << fbits)) fbits |= PMAP_FILTER_TWO;
    // Too many keys for a filter of that size: a bitmap over the whole key range when that has at most 2^30
    // values (k <= 15) and the keys crowd its low end, as the minimizers of windows of several k-mers do
    // (lookups then fall where the keys are, and the busy part of the bitmap stays cached; measured on a
    // 25 M-key index with w = 14: lookup phase 0.71 -> 0.50 ms).  Keys spread evenly over the range (w = 1:
    // every k-mer) gain nothing from it: half of the lookups find their key, and the bitmap is one more
    // request for each.  PMAP_FILTER_DIRECT=0 / 1: never / for every index with k <= 15 (tests).
    int direct = 0;
    if (fbits == 0 && idx->k <= 15 && !ukeys.empty()) {
        const uint64_t low_end = (1ull << (2 * idx->k)) >> 3;
        uint64_t low = 0;
        for (uint64_t key : ukeys) low += key < low_end;
        direct = 2 * low >= ukeys.size();
    }
    if (const char *dir_env = getenv("PMAP_FILTER_DIRECT")) direct = atoi(dir_env) != 0;
    if (direct && idx->k >= 3 && idx->k <= 15) fbits = 2 * idx->k | PMAP_FILTER_DIRECT;
    idx->filter_bits = fbits;
    idx->filter.clear();
    if (fbits) {
        idx->filter.assign((size_t) 1 << ((fbits & 0xFFu) - 5), 0u);
        for (size_t i = 0; i < ukeys.size(); ++i) {
            const uint32_t hm = pmap_key_mix(ukeys[i]);
            idx->filter[pmap_filter_word(ukeys[i], hm, fbits)] |= pmap_filter_mask(ukeys[i], hm, fbits);
        }
    }
    clk.lap("table + filter");
    return PMAP_OK;
}

// ---------------------------------------------------------------------------
// text formats

static inline bool is_digit(char c) { return (unsigned) (c - '0') < 10u; }

struct Cursor {
    const char *p, *end;
    bool more() const { return p < end; }
    bool read_uint(uint64_t *v) {   // skips anything that is not a digit first
        while (p < end && !is_digit(*p)) ++p;
        if (p >= end) return false;
        uint64_t x = 0;
        while (p < end && is_digit(*p)) x = x * 10 + (uint64_t) (*p++ - '0');
        *v = x;
        return true;
    }
};

static bool slurp(const char *path, std::string *out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    bool ok = fseek(f, 0, SEEK_END) == 0;
    const long size = ok ? ftell(f) : -1;
    ok = ok && size >= 0 && fseek(f, 0, SEEK_SET) == 0;
    if (ok) {
        out->resize((size_t) size);
        ok = size == 0 || fread(&(*out)[0], 1, (size_t) size, f) == (size_t) size;
    }
    fclose(f);
    return ok;
}

// MiniRecord text `(prg, N{[s, e)[s, e)...}, knode, strand)` (src/minirecord.cpp:23-38,
// src/prg/path.cpp:228-249, src/interval.cpp:20-37): the integers in order.
static bool parse_record(Cursor &c, uint32_t *prg, std::vector<uint32_t> *ivs, std::vector<uint32_t> *ivl,
                         uint32_t *knode, uint8_t *strand) {
    uint64_t v, n;
    if (!c.read_uint(&v)) return false;
    *prg = (uint32_t) v;
    if (!c.read_uint(&n)) return false;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t s, e;
        if (!c.read_uint(&s) || !c.read_uint(&e) || e < s) return false;
        ivs->push_back((uint32_t) s);
        ivl->push_back((uint32_t) (e - s));
    }
    if (!c.read_uint(&v)) return false;
    *knode = (uint32_t) v;
    if (!c.read_uint(&v)) return false;
    *strand = v != 0;
    while (c.more() && *c.p != ')') ++c.p;
    if (c.more()) ++c.p;
    return true;
}

extern "C" int pmap_index_load(const char *idx_path, uint32_t k, uint32_t n_prg, const uint32_t *mpl,
                               pmap_index_t **out) {
    std::string text;
    StageClock clk0;
    if (!slurp(idx_path, &text))
        return pmap_fail(PMAP_EIO, std::string("Unable to open index file ") + idx_path +
                                       ". Does it exist? Have you run pandora index?");   // src/index.cpp:115
    clk0.lap("read file");
    StageClock clk;
    // One line of the file is one key with its records (Index::save, src/index.cpp:51-68), so pieces cut
    // at line starts ("\n" + digit, after a record's closing bracket or the count line) parse independently.
    struct Piece {
        const char *begin, *end;
        std::vector<uint64_t> keys, rec_off, path_off;    // offsets relative to the piece
        std::vector<uint32_t> rec_prg, rec_knode, ivs, ivl;
        std::vector<uint8_t> rec_strand;
        const char *error = nullptr;
    };
    const char *t0 = text.data(), *t1 = text.data() + text.size();
    // Index::load (src/index.cpp:86-113): the first token that starts with a digit is the entry count
    const char *body = t0;
    while (body < t1 && !is_digit(*body)) {
        if (*body == '(') return pmap_fail(PMAP_EIO, "index file: record before any key");
        ++body;
    }
    while (body < t1 && is_digit(*body)) ++body;
    const unsigned n_pieces = worker_count((size_t) (t1 - body), 4u << 20);
    std::vector<Piece> pieces(n_pieces);
    {
        const char *cur = body;
        for (unsigned i = 0; i < n_pieces; ++i) {
            const char *stop = t1;
            if (i + 1 < n_pieces) {
                stop = body + (size_t) (t1 - body) * (i + 1) / n_pieces;
                if (stop < cur) stop = cur;
                while (stop < t1 && !(stop > t0 + 1 && stop[-1] == '\n' && is_digit(*stop) &&
                                      (stop[-2] == ')' || stop[-2] == '\t' || is_digit(stop[-2]))))
                    ++stop;
            }
            pieces[i].begin = cur;
            pieces[i].end = stop;
            cur = stop;
        }
    }
    auto parse_piece = [](Piece *pc) {
        Cursor c{pc->begin, pc->end};
        // a token starting with a digit is a key followed by its record count; '(' starts a record
        while (c.more()) {
            char ch = *c.p;
            if (is_digit(ch)) {
                uint64_t v, cnt;
                c.read_uint(&v);
                if (!c.read_uint(&cnt)) { pc->error = "index file: key without record count"; return; }
                if (!pc->keys.empty()) pc->rec_off.push_back(pc->rec_prg.size());
                pc->keys.push_back(v);
            } else if (ch == '(') {
                if (pc->keys.empty()) { pc->error = "index file: record before any key"; return; }
                uint32_t prg, knode;
                uint8_t strand;
                if (!parse_record(c, &prg, &pc->ivs, &pc->ivl, &knode, &strand)) {
                    pc->error = "index file: malformed record";
                    return;
                }
                pc->rec_prg.push_back(prg);
                pc->rec_knode.push_back(knode);
                pc->rec_strand.push_back(strand);
                pc->path_off.push_back(pc->ivs.size());
            } else {
                ++c.p;
            }
        }
        if (!pc->keys.empty()) pc->rec_off.push_back(pc->rec_prg.size());
    };
    {
        std::vector<std::thread> th;
        for (unsigned i = 1; i < n_pieces; ++i) th.emplace_back(parse_piece, &pieces[i]);
        parse_piece(&pieces[0]);
        for (auto &t : th) t.join();
    }
    std::vector<uint64_t> keys, rec_off{0}, path_off{0};
    std::vector<uint32_t> rec_prg, rec_knode, ivs, ivl;
    std::vector<uint8_t> rec_strand;
    for (Piece &pc : pieces) {
        if (pc.error) return pmap_fail(PMAP_EIO, pc.error);
        const uint64_t r0 = rec_prg.size(), i0 = ivs.size();
        keys.insert(keys.end(), pc.keys.begin(), pc.keys.end());
        for (uint64_t o : pc.rec_off) rec_off.push_back(r0 + o);
        for (uint64_t o : pc.path_off) path_off.push_back(i0 + o);
        rec_prg.insert(rec_prg.end(), pc.rec_prg.begin(), pc.rec_prg.end());
        rec_knode.insert(rec_knode.end(), pc.rec_knode.begin(), pc.rec_knode.end());
        rec_strand.insert(rec_strand.end(), pc.rec_strand.begin(), pc.rec_strand.end());
        ivs.insert(ivs.end(), pc.ivs.begin(), pc.ivs.end());
        ivl.insert(ivl.end(), pc.ivl.begin(), pc.ivl.end());
        pc = Piece();
    }
    clk.lap("parse text");
    uint32_t need_prg = 0;
    for (uint32_t p : rec_prg) need_prg = std::max(need_prg, p + 1);
    std::vector<uint32_t> mplv(std::max(n_prg, need_prg), 0);
    if (mpl) std::copy(mpl, mpl + n_prg, mplv.begin());
    pmap_index_desc_t d;
    memset(&d, 0, sizeof(d));
    d.k = k;
    d.n_keys = keys.size();
    d.keys = keys.data();
    d.rec_off = rec_off.data();
    d.n_recs = rec_prg.size();
    d.rec_prg = rec_prg.data();
    d.rec_knode = rec_knode.data();
    d.rec_strand = rec_strand.data();
    d.path_off = path_off.data();
    d.iv_start = ivs.data();
    d.iv_len = ivl.data();
    d.n_prg = (uint32_t) mplv.size();
    d.min_path_length = mplv.data();
    return pmap_index_create(&d, out);
}

// ---------------------------------------------------------------------------
// Binary cache of a built index: the text format costs seconds per million records to parse and
// rank; the cache is read back at file speed.  Layout: CacheHeader, then the arrays in the order of
// cache_arrays(), each padded to 64 bytes, then the magic once more.

namespace {

struct CacheHeader {
    char magic[8];               // "PMAPIDX1"
    uint32_t version, k, n_prg, log2cap, filter_bits, pad;
    uint64_t n_keys, n_recs, n_paths, n_iv, n_slots, n_filter_words;
    uint64_t src_size, src_mtime_ns, mpl_hash;   // what the cache was built from
};
const char CACHE_MAGIC[8] = {'P', 'M', 'A', 'P', 'I', 'D', 'X', '1'};
const uint32_t CACHE_VERSION = 3;

uint64_t fnv1a(const void *data, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char *p = (const unsigned char *) data;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

struct ArrayRef {
    void *data;
    size_t bytes;
};

// every array of the index, sized from the header (resizes the vectors when `size_them` is set)
std::vector<ArrayRef> cache_arrays(pmap_index *idx, const CacheHeader &h, bool size_them) {
    if (size_them) {
        idx->key_of_rec.resize(h.n_recs);
        idx->rec_prg.resize(h.n_recs);
        idx->rec_knode.resize(h.n_recs);
        idx->path_rank.resize(h.n_recs);
        idx->rec_strand.resize(h.n_recs);
        idx->path_off.resize(h.n_recs + 1);
        idx->iv_start.resize(h.n_iv);
        idx->iv_len.resize(h.n_iv);
        idx->min_path_length.resize(h.n_prg);
        idx->slots.resize(h.n_slots);
        idx->recs.resize(h.n_recs);
        idx->filter.resize(h.n_filter_words);
    }
    return {{idx->key_of_rec.data(), h.n_recs * sizeof(uint64_t)},
            {idx->rec_prg.data(), h.n_recs * sizeof(uint32_t)},
            {idx->rec_knode.data(), h.n_recs * sizeof(uint32_t)},
            {idx->path_rank.data(), h.n_recs * sizeof(uint32_t)},
            {idx->rec_strand.data(), h.n_recs},
            {idx->path_off.data(), (h.n_recs + 1) * sizeof(uint64_t)},
            {idx->iv_start.data(), h.n_iv * sizeof(uint32_t)},
            {idx->iv_len.data(), h.n_iv * sizeof(uint32_t)},
            {idx->min_path_length.data(), h.n_prg * sizeof(uint32_t)},
            {idx->slots.data(), h.n_slots * sizeof(PmapSlot)},
            {idx->recs.data(), h.n_recs * sizeof(PmapRec)},
            {idx->filter.data(), h.n_filter_words * sizeof(uint32_t)}};
}

bool source_stamp(const char *path, uint64_t *size, uint64_t *mtime_ns) {
    struct stat st;
    if (!path || stat(path, &st) != 0) return false;
    *size = (uint64_t) st.st_size;
    *mtime_ns = (uint64_t) st.st_mtim.tv_sec * 1000000000ull + (uint64_t) st.st_mtim.tv_nsec;
    return true;
}

}  // namespace

extern "C" int pmap_index_save_cache(const pmap_index_t *cidx, const char *cache_path, const char *source_path) {
    if (!cidx || !cache_path) return pmap_fail(PMAP_EINVAL, "null argument");
    pmap_index *idx = const_cast<pmap_index *>(cidx);
    CacheHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, CACHE_MAGIC, 8);
    h.version = CACHE_VERSION;
    h.k = idx->k;
    h.n_prg = idx->n_prg;
    h.log2cap = idx->log2cap;
    h.filter_bits = idx->filter_bits;
    h.n_keys = idx->n_keys;
    h.n_recs = idx->n_recs;
    h.n_paths = idx->n_paths;
    h.n_iv = idx->iv_start.size();
    h.n_slots = idx->slots.size();
    h.n_filter_words = idx->filter.size();
    if (source_path && !source_stamp(source_path, &h.src_size, &h.src_mtime_ns))
        return pmap_fail(PMAP_EIO, std::string("cannot stat ") + source_path);
    h.mpl_hash = fnv1a(idx->min_path_length.data(), idx->min_path_length.size() * sizeof(uint32_t));
    // written under a temporary name and renamed: a reader never sees half a cache
    const std::string tmp = std::string(cache_path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return pmap_fail(PMAP_EIO, std::string("cannot write ") + tmp);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    static const char zeros[64] = {0};
    for (const ArrayRef &a : cache_arrays(idx, h, false)) {
        if (ok && a.bytes) ok = fwrite(a.data, 1, a.bytes, f) == a.bytes;
        const size_t padn = (64 - a.bytes % 64) % 64;
        if (ok && padn) ok = fwrite(zeros, 1, padn, f) == padn;
    }
    ok = ok && fwrite(CACHE_MAGIC, 8, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), cache_path) != 0) {
        remove(tmp.c_str());
        return pmap_fail(PMAP_EIO, std::string("cannot write ") + cache_path);
    }
    return PMAP_OK;
}

extern "C" int pmap_index_load_cache(const char *cache_path, pmap_index_t **out) {
    if (!cache_path || !out) return pmap_fail(PMAP_EINVAL, "null argument");
    StageClock clk;
    FILE *f = fopen(cache_path, "rb");
    if (!f) return pmap_fail(PMAP_EIO, std::string("Unable to open index cache ") + cache_path);
    CacheHeader h;
    auto bad = [&](const char *why) {
        fclose(f);
        return pmap_fail(PMAP_EIO, std::string("index cache ") + cache_path + ": " + why);
    };
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, CACHE_MAGIC, 8) != 0) return bad("not an index cache");
    if (h.version != CACHE_VERSION) return bad("written by another version");
    const uint32_t fb = h.filter_bits & 0xFFu;     // number of filter bits; PMAP_FILTER_TWO rides above it
    if (h.k < 1 || h.k > 31 || h.log2cap > 31 || h.n_slots != (1ull << h.log2cap) || h.n_recs >= 0xFFFFFFFFull ||
        h.n_keys > h.n_recs || (h.filter_bits & ~(0xFFu | PMAP_FILTER_TWO | PMAP_FILTER_DIRECT)) || fb > 30 || (fb && fb < 6) ||
        (!fb && h.filter_bits) || ((h.filter_bits & PMAP_FILTER_DIRECT) && (fb != 2 * h.k || (h.filter_bits & PMAP_FILTER_TWO))) ||
        h.n_filter_words != (fb ? (1ull << (fb - 5)) : 0))
        return bad("inconsistent header");
    // the file must be exactly as long as the header says, before anything is allocated from it
    uint64_t expect = sizeof(h) + 8;
    const uint64_t sizes[] = {h.n_recs * 8, h.n_recs * 4, h.n_recs * 4, h.n_recs * 4, h.n_recs, (h.n_recs + 1) * 8,
                              h.n_iv * 4, h.n_iv * 4, (uint64_t) h.n_prg * 4, h.n_slots * sizeof(PmapSlot), h.n_recs * 16,
                              h.n_filter_words * 4};
    for (uint64_t b : sizes) expect += (b + 63) / 64 * 64;
    struct stat st;
    if (fstat(fileno(f), &st) != 0 || (uint64_t) st.st_size != expect) return bad("truncated or damaged");
    pmap_index *idx = new pmap_index();
    idx->k = h.k;
    idx->n_prg = h.n_prg;
    idx->log2cap = h.log2cap;
    idx->filter_bits = h.filter_bits;
    idx->n_keys = h.n_keys;
    idx->n_recs = h.n_recs;
    idx->n_paths = h.n_paths;
    bool ok = true;
    char skip[64];
    for (const ArrayRef &a : cache_arrays(idx, h, true)) {
        if (ok && a.bytes) ok = fread(a.data, 1, a.bytes, f) == a.bytes;
        const size_t padn = (64 - a.bytes % 64) % 64;
        if (ok && padn) ok = fread(skip, 1, padn, f) == padn;
    }
    char tail[8];
    ok = ok && fread(tail, 8, 1, f) == 1 && memcmp(tail, CACHE_MAGIC, 8) == 0;
    // cheap structural checks: everything the kernels index with stays in range
    if (ok) {
        ok = idx->path_off[h.n_recs] == h.n_iv;
        uint64_t used = 0;
        for (const PmapSlot &sl : idx->slots)
            if (sl.key != PMAP_EMPTY_KEY) {
                ++used;
                idx->max_cnt = std::max(idx->max_cnt, sl.cnt);
                if ((uint64_t) sl.off + sl.cnt > h.n_recs || sl.cnt == 0) ok = false;
                else if (memcmp(&sl.first, &idx->recs[sl.off], sizeof(PmapRec)) != 0) ok = false;
            }
        if (used != h.n_keys) ok = false;
        for (const PmapRec &r : idx->recs)
            if (r.prg >= h.n_prg || r.orig >= h.n_recs) ok = false;
    }
    if (!ok) {
        delete idx;
        return bad("truncated or damaged");
    }
    fclose(f);
    clk.lap("load cache");
    *out = idx;
    return PMAP_OK;
}

extern "C" int pmap_index_load_cached(const char *idx_path, uint32_t k, uint32_t n_prg, const uint32_t *mpl,
                                      const char *cache_path, pmap_index_t **out, int *from_cache) {
    if (!idx_path || !cache_path || !out) return pmap_fail(PMAP_EINVAL, "null argument");
    if (from_cache) *from_cache = 0;
    uint64_t size = 0, mtime = 0;
    if (!source_stamp(idx_path, &size, &mtime))
        return pmap_fail(PMAP_EIO, std::string("Unable to open index file ") + idx_path +
                                       ". Does it exist? Have you run pandora index?");   // src/index.cpp:115
    FILE *f = fopen(cache_path, "rb");
    if (f) {
        CacheHeader h;
        const bool have = fread(&h, sizeof(h), 1, f) == 1;
        fclose(f);
        // usable only if built from this very file with these parameters
        if (have && memcmp(h.magic, CACHE_MAGIC, 8) == 0 && h.version == CACHE_VERSION && h.k == k &&
            h.src_size == size && h.src_mtime_ns == mtime) {
            pmap_index_t *idx = nullptr;
            if (pmap_index_load_cache(cache_path, &idx) == PMAP_OK) {
                // min_path_length shorter than the cache's table was padded with zeros when it was built
                std::vector<uint32_t> mplv(idx->n_prg, 0);
                if (mpl) std::copy(mpl, mpl + std::min<uint32_t>(n_prg, idx->n_prg), mplv.begin());
                if (n_prg <= idx->n_prg && fnv1a(mplv.data(), mplv.size() * sizeof(uint32_t)) == h.mpl_hash) {
                    if (from_cache) *from_cache = 1;
                    *out = idx;
                    return PMAP_OK;
                }
                pmap_index_destroy(idx);
            }
        }
    }
    int rc = pmap_index_load(idx_path, k, n_prg, mpl, out);
    if (rc) return rc;
    // a cache that cannot be written is not an error of the load
    if (pmap_index_save_cache(*out, cache_path, idx_path) != PMAP_OK) pmap_set_error("");
    return PMAP_OK;
}

extern "C" int pmap_index_create(const pmap_index_desc_t *desc, pmap_index_t **out) {
    if (!desc || !out) return pmap_fail(PMAP_EINVAL, "null argument");
    pmap_index *idx = new pmap_index();
    // host structures only: the table and records go to HBM on the first pmap_map call
    int rc = pmap_index_build_host(desc, idx);
    if (rc != PMAP_OK) {
        pmap_index_release_device(idx);
        delete idx;
        return rc;
    }
    *out = idx;
    return PMAP_OK;
}

extern "C" void pmap_index_destroy(pmap_index_t *idx) {
    if (!idx) return;
    pmap_index_release_device(idx);
    delete idx;
}

extern "C" int pmap_index_info(const pmap_index_t *idx, pmap_index_info_t *info) {
    if (!idx || !info) return pmap_fail(PMAP_EINVAL, "null argument");
    info->k = idx->k;
    info->n_keys = idx->n_keys;
    info->n_recs = idx->n_recs;
    info->n_paths = idx->n_paths;
    info->table_slots = idx->slots.size();
    info->device_bytes = idx->device_bytes;
    info->n_prg = idx->n_prg;
    info->filter_bits = idx->filter_bits & 0xFFu;
    info->filter_bits_per_key = !idx->filter_bits ? 0u : (idx->filter_bits & PMAP_FILTER_TWO) ? 2u : 1u;
    info->filter_direct = (idx->filter_bits & PMAP_FILTER_DIRECT) ? 1u : 0u;
    return PMAP_OK;
}

extern "C" int pmap_index_records(const pmap_index_t *idx, uint64_t *key_of_rec, uint32_t *rec_prg,
                                  uint32_t *rec_knode, uint8_t *rec_strand, uint32_t *path_rank,
                                  uint64_t *path_off) {
    if (!idx) return pmap_fail(PMAP_EINVAL, "null argument");
    size_t n = idx->n_recs;
    if (key_of_rec) memcpy(key_of_rec, idx->key_of_rec.data(), n * sizeof(uint64_t));
    if (rec_prg) memcpy(rec_prg, idx->rec_prg.data(), n * sizeof(uint32_t));
    if (rec_knode) memcpy(rec_knode, idx->rec_knode.data(), n * sizeof(uint32_t));
    if (rec_strand) memcpy(rec_strand, idx->rec_strand.data(), n);
    if (path_rank) memcpy(path_rank, idx->path_rank.data(), n * sizeof(uint32_t));
    if (path_off) memcpy(path_off, idx->path_off.data(), (n + 1) * sizeof(uint64_t));
    return PMAP_OK;
}

extern "C" int pmap_index_intervals(const pmap_index_t *idx, uint32_t *iv_start, uint32_t *iv_len,
                                    uint64_t *n_intervals) {
    if (!idx) return pmap_fail(PMAP_EINVAL, "null argument");
    if (n_intervals) *n_intervals = idx->iv_start.size();
    if (iv_start) memcpy(iv_start, idx->iv_start.data(), idx->iv_start.size() * sizeof(uint32_t));
    if (iv_len) memcpy(iv_len, idx->iv_len.data(), idx->iv_len.size() * sizeof(uint32_t));
    return PMAP_OK;
}

// ---------------------------------------------------------------------------
// KmerGraph::load (src/kmergraph.cpp:716-820) reduced to what min_path_length
// (src/kmergraph.cpp:664-684) reads: node ids, node paths, out-edges in file order.
extern "C" int pmap_gfa_min_path_length(const char *gfa_path, uint32_t *out) {
    std::ifstream f(gfa_path);
    if (!f.is_open()) return pmap_fail(PMAP_EIO, std::string("Unable to open kmergraph file ") + gfa_path);
    struct Node {
        std::vector<uint32_t> ivs, ivl;
        std::vector<uint32_t> outs;
        bool present = false;
    };
    std::vector<Node> nodes;
    std::vector<std::pair<uint32_t, uint32_t>> edges;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        if (line[0] == 'S') {
            // S \t id \t N{[s, e)...} \t FC:i:n \t \t RC:i:n
            size_t t1 = line.find('\t'), t2 = line.find('\t', t1 + 1), t3 = line.find('\t', t2 + 1);
            if (t1 == std::string::npos || t2 == std::string::npos)
                return pmap_fail(PMAP_EIO, "kmergraph GFA: malformed S line");
            uint32_t id = (uint32_t) strtoul(line.c_str() + t1 + 1, nullptr, 10);
            std::string ptxt = line.substr(t2 + 1, t3 == std::string::npos ? std::string::npos : t3 - t2 - 1);
            if (ptxt.empty() || !isdigit((unsigned char) ptxt[0]))   // src/kmergraph.cpp:750
                return pmap_fail(PMAP_EIO, "kmergraph GFA does not label nodes with their PRG path");
            Cursor c{ptxt.data(), ptxt.data() + ptxt.size()};
            uint64_t n;
            c.read_uint(&n);
            if (id >= nodes.size()) nodes.resize(id + 1);
            nodes[id].present = true;
            for (uint64_t i = 0; i < n; ++i) {
                uint64_t s, e;
                if (!c.read_uint(&s) || !c.read_uint(&e) || e < s)
                    return pmap_fail(PMAP_EIO, "kmergraph GFA: malformed path");
                nodes[id].ivs.push_back((uint32_t) s);
                nodes[id].ivl.push_back((uint32_t) (e - s));
            }
        } else if (line[0] == 'L') {
            // L \t from \t + \t to \t + \t 0M
            std::vector<std::string> tok;
            std::stringstream ss(line);
            std::string t;
            while (std::getline(ss, t, '\t'))
                if (!t.empty()) tok.push_back(t);
            if (tok.size() < 5) return pmap_fail(PMAP_EIO, "kmergraph GFA: malformed L line");
            uint32_t from = (uint32_t) strtoul(tok[1].c_str(), nullptr, 10);
            uint32_t to = (uint32_t) strtoul(tok[3].c_str(), nullptr, 10);
            if (tok[2] != tok[4]) std::swap(from, to);   // src/kmergraph.cpp:803-810
            edges.push_back({from, to});
        }
    }
    for (auto &e : edges) {
        if (e.first >= nodes.size() || e.second >= nodes.size())
            return pmap_fail(PMAP_EIO, "kmergraph GFA: edge to unknown node");
        auto &o = nodes[e.first].outs;
        if (std::find(o.begin(), o.end(), e.second) == o.end()) o.push_back(e.second);   // add_edge, :177
    }
    if (nodes.empty()) { *out = 0; return PMAP_OK; }
    // sort_topologically (:220-225): by prg::Path::operator<
    std::vector<uint32_t> sorted(nodes.size());
    std::iota(sorted.begin(), sorted.end(), 0u);
    std::sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) {
        return path_cmp(nodes[a].ivs.data(), nodes[a].ivl.data(), nodes[a].ivs.size(), nodes[b].ivs.data(),
                        nodes[b].ivl.data(), nodes[b].ivs.size()) < 0;
    });
    // :674-683: len is written by sorted position and read by node id, as the reference does
    std::vector<uint32_t> len(sorted.size(), 0);
    for (uint32_t j = (uint32_t) sorted.size() - 1; j != 0; --j)
        for (uint32_t o : nodes[sorted[j - 1]].outs)
            if (len[o] + 1 > len[j - 1]) len[j - 1] = len[o] + 1;
    *out = len[0];
    return PMAP_OK;
}

extern "C" float pmap_fraction(float e_rate, uint32_t k) {
    float x = e_rate * k;
    return (float) (0.5 / expf(x));
}
