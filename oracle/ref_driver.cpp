// C entry points over the reference's own classes, for checking the oracle.
//
// TEST INFRASTRUCTURE ONLY.  This file is compiled together with the
// boost-free translation units of the reference, taken where they lie under
// /root/reference (src/inthash.cpp, interval.cpp, prg/path.cpp, minimizer.cpp,
// minirecord.cpp, minihit.cpp, minihits.cpp), into oracle/_ref/libpandora_ref.so
// by oracle/Makefile.  No reference source is copied; this file only calls the
// reference's public types and functions.
#include <cstdint>
#include <cstring>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>
#include <deque>

#include "inthash.h"
#include "interval.h"
#include "minihit.h"
#include "minihits.h"
#include "minimizer.h"
#include "minirecord.h"
#include "prg/path.h"

static prg::Path make_path(const uint32_t *s, const uint32_t *l, uint64_t n) {
    prg::Path p;
    std::deque<Interval> d;
    for (uint64_t i = 0; i < n; ++i) d.push_back(Interval(s[i], s[i] + l[i]));
    p.initialize(d);
    return p;
}

extern "C" {

uint32_t ref_nt4(uint8_t c) { return nt4((char) c); }

uint64_t ref_hash64(uint64_t key, uint64_t mask) { return hash64(key, mask); }

void ref_kmerhash(const char *s, uint32_t k, uint64_t *fwd, uint64_t *rc) {
    KmerHash h;
    auto r = h.kmerhash(std::string(s), k);
    *fwd = r.first;
    *rc = r.second;
}

// Path::operator< both ways: -1 if a<b, 1 if b<a, 0 if neither
int ref_path_cmp(const uint32_t *as, const uint32_t *al, uint64_t an, const uint32_t *bs,
                 const uint32_t *bl, uint64_t bn) {
    prg::Path a = make_path(as, al, an), b = make_path(bs, bl, bn);
    if (a < b) return -1;
    if (b < a) return 1;
    return 0;
}

// Minimizer::operator<: -1, 0, 1
int ref_minimizer_cmp(uint64_t ka, uint32_t sa, uint32_t ea, int stra, uint64_t kb, uint32_t sb, uint32_t eb,
                      int strb) {
    Minimizer a(ka, sa, ea, stra != 0), b(kb, sb, eb, strb != 0);
    if (a < b) return -1;
    if (b < a) return 1;
    return 0;
}

// MiniRecord text round trip (operator>> then operator<<): parses `text`,
// returns fields, writes the re-serialised form into out (cap bytes).
int ref_minirecord_parse(const char *text, uint32_t *prg_id, uint32_t *knode, int *strand, uint32_t *iv_start,
                         uint32_t *iv_len, uint32_t iv_cap, uint32_t *n_iv, char *out, uint32_t cap) {
    std::istringstream in{std::string(text)};
    MiniRecord mr;
    in >> mr;
    *prg_id = mr.prg_id;
    *knode = mr.knode_id;
    *strand = mr.strand;
    *n_iv = (uint32_t) mr.path.path.size();
    for (uint32_t i = 0; i < *n_iv && i < iv_cap; ++i) {
        iv_start[i] = mr.path.path[i].start;
        iv_len[i] = mr.path.path[i].length;
    }
    std::ostringstream os;
    os << mr;
    std::string s = os.str();
    if (s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return 0;
}

// Builds MinimizerHit objects from flat arrays (path of hit i = intervals
// [path_off[i], path_off[i+1])), inserts them in MinimizerHits::uhits and calls
// MinimizerHits::sort().  order[j] = input index of the j-th hit in the sorted
// std::set.  Returns the set size.
int64_t ref_sort_hits(uint64_t n, const uint32_t *read, const uint32_t *pos, const uint32_t *prg,
                      const uint8_t *fwd, const uint64_t *path_off, const uint32_t *iv_start,
                      const uint32_t *iv_len, uint32_t k, uint64_t *order) {
    MinimizerHits mh((uint32_t) n);
    std::vector<MinimizerHitPtr> ptrs;
    for (uint64_t i = 0; i < n; ++i) {
        prg::Path p = make_path(iv_start + path_off[i], iv_len + path_off[i], path_off[i + 1] - path_off[i]);
        auto h = std::make_shared<MinimizerHit>(read[i], Interval(pos[i], pos[i] + k), prg[i], p, (uint32_t) i,
                                                fwd[i] != 0);
        ptrs.push_back(h);
        mh.uhits.insert(h);
    }
    // MinimizerHits::sort asserts that nothing was deduplicated; do the same
    // insertion without the assert so duplicates can be reported to the caller
    mh.hits.insert(mh.uhits.begin(), mh.uhits.end());
    uint64_t j = 0;
    for (const auto &h : mh.hits) order[j++] = h->kmer_node_id;   // kmer_node_id carries the input index
    return (int64_t) j;
}

// Orders clusters with std::set<std::set<MinimizerHitPtr,pComp>, clusterComp>.
// Cluster c is hits [cl_off[c], cl_off[c+1]) of the flat hit arrays.  order[j] =
// input cluster index of the j-th cluster in set order.  Returns the set size.
int64_t ref_order_clusters(uint64_t n_cl, const uint64_t *cl_off, const uint32_t *read, const uint32_t *pos,
                           const uint32_t *prg, const uint8_t *fwd, const uint64_t *path_off,
                           const uint32_t *iv_start, const uint32_t *iv_len, uint32_t k, uint64_t *order) {
    std::set<std::set<MinimizerHitPtr, pComp>, clusterComp> clusters;
    std::vector<std::pair<MinimizerHitPtr, uint64_t>> firsts;
    for (uint64_t c = 0; c < n_cl; ++c) {
        std::set<MinimizerHitPtr, pComp> cl;
        for (uint64_t i = cl_off[c]; i < cl_off[c + 1]; ++i) {
            prg::Path p = make_path(iv_start + path_off[i], iv_len + path_off[i], path_off[i + 1] - path_off[i]);
            cl.insert(std::make_shared<MinimizerHit>(read[i], Interval(pos[i], pos[i] + k), prg[i], p,
                                                     (uint32_t) c, fwd[i] != 0));
        }
        clusters.insert(cl);
    }
    uint64_t j = 0;
    for (const auto &cl : clusters) order[j++] = (*cl.begin())->kmer_node_id;  // carries the cluster index
    return (int64_t) j;
}

}  // extern "C"
