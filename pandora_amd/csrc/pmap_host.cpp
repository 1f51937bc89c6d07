// Host stages around the device path: the FASTA/FASTQ reader
// (src/fastaq_handler.cpp:12-71), clusterComp ordering + filter_clusters
// (src/minihits.cpp:95-109, src/utils.cpp:251-288), the add_node sequence
// (src/utils.cpp:338-352, src/pangenome/pangraph.cpp:43-140) and the read loop of
// pangraph_from_read_file (src/utils.cpp:379-455).
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>
#include <string>
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

// std::istream semantics that FastaqHandler relies on, over a memory buffer
struct LineStream {
    const std::string &buf;
    size_t pos = 0;
    bool eofbit = false;
    explicit LineStream(const std::string &b) : buf(b) {}
    bool at_eof() {   // instream.peek() == EOF (src/fastaq_handler.cpp:38-41)
        if (eofbit || pos >= buf.size()) { eofbit = true; return true; }
        return false;
    }
    // getline(...).good(): false when the data ended before a newline was seen
    bool getline(std::string *line) {
        line->clear();
        if (eofbit || pos >= buf.size()) { eofbit = true; return false; }
        size_t nl = buf.find('\n', pos);
        if (nl == std::string::npos) {
            line->assign(buf, pos, std::string::npos);
            pos = buf.size();
            eofbit = true;
            return false;
        }
        line->assign(buf, pos, nl - pos);
        pos = nl + 1;
        return true;
    }
};

struct FastaqReader {
    std::string data;
    LineStream in;
    std::string line, name, read;
    explicit FastaqReader(std::string d) : data(std::move(d)), in(data) {}
    bool eof() { return in.at_eof(); }
    void get_next() {   // src/fastaq_handler.cpp:43-71
        if (!line.empty() && (line[0] == '>' || line[0] == '@')) {
            name = line.substr(1);
            read.clear();
        }
        while (in.getline(&line)) {
            if (!line.empty() && line[0] == '+') {
                in.getline(&line);   // quality line
            } else if (line.empty() || line[0] == '>' || line[0] == '@') {
                if (!read.empty() || line.empty()) return;
                name = line.substr(1);
                read.clear();
            } else {
                read += line;
            }
        }
    }
};

bool load_file(const char *path, std::string *out) {
    gzFile f = gzopen(path, "rb");   // transparent for uncompressed files
    if (!f) return false;
    std::vector<char> chunk(1 << 20);
    int n;
    while ((n = gzread(f, chunk.data(), (unsigned) chunk.size())) > 0) out->append(chunk.data(), (size_t) n);
    gzclose(f);
    return n == 0;
}

}  // namespace

extern "C" int pmap_read_fastaq(const char *path, char **bases, uint64_t **offsets, uint32_t *n_reads,
                                char **names, uint64_t **name_offsets) {
    std::string data;
    if (!load_file(path, &data))
        return pmap_fail(PMAP_EIO, std::string("Unable to open fastaq file ") + path);   // src/fastaq_handler.cpp:18-21
    FastaqReader fh(std::move(data));
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

struct PgRead {
    std::vector<uint32_t> nodes;
    std::vector<uint8_t> orient;
};

struct pmap_pangraph {
    std::map<uint32_t, uint32_t> node_covg;   // node id (= prg id) -> covg
    std::map<uint32_t, PgRead> reads;
    std::vector<PgHit> hits;
    std::vector<uint32_t> call_prg, call_read, call_size;
    std::vector<uint8_t> call_fwd;
    bool hits_sorted = false;
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
    if (pg->hits_sorted) return;
    // per (read, prg): pComp_path (src/minihits.cpp:77-93): path, forward first, position
    std::sort(pg->hits.begin(), pg->hits.end(), [](const PgHit &a, const PgHit &b) {
        if (a.read != b.read) return a.read < b.read;
        if (a.prg != b.prg) return a.prg < b.prg;
        if (a.rank != b.rank) return a.rank < b.rank;
        if (a.fwd != b.fwd) return a.fwd > b.fwd;
        return a.pos < b.pos;
    });
    pg->hits_sorted = true;
}

// one device batch -> filter_clusters -> add_node calls
static int map_batch(const pmap_index_t *idx, const pmap_params_t *prm, const std::string &bases,
                     const std::vector<uint64_t> &offsets, uint32_t first_id, pmap_result_t **res,
                     pmap_pangraph *pg) {
    const uint32_t n = (uint32_t) (offsets.size() - 1);
    if (n == 0) return PMAP_OK;
    pmap_reads_t *rd = nullptr;
    int rc = pmap_reads_create(bases.data(), offsets.data(), n, first_id, &rd);
    if (rc) return rc;
    rc = pmap_map(idx, rd, prm, res);
    pmap_reads_destroy(rd);
    if (rc) return rc;
    pmap_counts_t cn;
    pmap_result_counts(*res, &cn);
    const size_t nc = cn.n_clusters, nh = cn.n_hits;
    if (nc == 0) return PMAP_OK;
    std::vector<uint64_t> begin(nc), order(nc);
    std::vector<uint32_t> size(nc), rid(nc), prg(nc), fpos(nc), lpos(nc), frank(nc);
    std::vector<uint8_t> fwd(nc), alive(nc);
    if ((rc = pmap_result_clusters(*res, begin.data(), size.data(), rid.data(), prg.data(), fwd.data(), fpos.data(),
                                   lpos.data(), frank.data())))
        return rc;
    if ((rc = pmap_filter_clusters(nc, rid.data(), fpos.data(), size.data(), prg.data(), frank.data(), fwd.data(),
                                   lpos.data(), order.data(), alive.data())))
        return rc;
    std::vector<uint32_t> hpos(nh), hrec(nh);
    if ((rc = pmap_result_hits(*res, 1, nullptr, hpos.data(), nullptr, hrec.data(), nullptr))) return rc;
    for (size_t j = 0; j < nc; ++j) {
        if (!alive[j]) continue;
        const uint64_t c = order[j];
        pg_add_node(pg, prg[c], rid[c], size[c], fwd[c] != 0);
        for (uint64_t i = begin[c]; i < begin[c] + size[c]; ++i)
            pg->hits.push_back(PgHit{rid[c], prg[c], hpos[i], hrec[i], idx->path_rank[hrec[i]], fwd[c]});
    }
    pg->hits_sorted = false;
    return PMAP_OK;
}

extern "C" int pmap_pangraph_from_read_file(const char *reads_path, const pmap_index_t *idx,
                                            const pmap_map_args_t *a, pmap_pangraph_t **out, uint32_t *covg_out) {
    if (!reads_path || !idx || !a || !out) return pmap_fail(PMAP_EINVAL, "null argument");
    if (a->clean) return pmap_fail(PMAP_EINVAL, "clean=true (de Bruijn cleaning of the pangraph) is outside this library");
    if (a->w == 0 || a->genome_size == 0) return pmap_fail(PMAP_EINVAL, "w and genome_size must be positive");
    std::string data;
    if (!load_file(reads_path, &data))
        return pmap_fail(PMAP_EIO, std::string("Unable to open fastaq file ") + reads_path);
    FastaqReader fh(std::move(data));

    pmap_params_t prm;
    memset(&prm, 0, sizeof(prm));
    prm.w = a->w;
    prm.k = a->k;
    prm.max_diff = a->max_diff;
    prm.fraction = pmap_fraction(a->e_rate, a->k);                 // src/utils.cpp:394
    prm.min_cluster_size = a->min_cluster_size;
    prm.expected_kmers = std::numeric_limits<uint32_t>::max();     // :395
    const uint64_t batch_bases = a->batch_bases ? a->batch_bases : (256ull << 20);

    pmap_pangraph *pg = new pmap_pangraph();
    pmap_result_t *res = nullptr;
    uint64_t covg = 0;
    uint32_t id = 0;
    std::string bases;
    std::vector<uint64_t> offsets{0};
    uint32_t batch_first = 0;
    int rc = PMAP_OK;
    auto flush = [&]() {
        int r = map_batch(idx, &prm, bases, offsets, batch_first, &res, pg);
        bases.clear();
        offsets.assign(1, 0);
        batch_first = id;
        return r;
    };
    while (!fh.eof()) {
        fh.get_next();
        const std::string &seq = fh.read;
        // Seq::initialize -> the sketch is non-empty iff the read has a full window and only ACGT letters
        bool sketched = seq.size() + 1 >= (uint64_t) a->w + a->k;
        if (sketched)
            for (unsigned char ch : seq) {
                unsigned u = ch & 0xDFu;
                if (!(ch < 4 || u == 'A' || u == 'C' || u == 'G' || u == 'T')) { sketched = false; break; }
            }
        if (sketched) {
            covg += seq.size();                                         // :404
            if (covg / a->genome_size > a->max_covg) break;             // :405-408
        } else {
            // consumes an id (:409-412); travels in the batch so ids stay consecutive, yields nothing
            bases += seq;
            offsets.push_back(bases.size());
            id++;
            continue;
        }
        if (a->illumina && prm.expected_kmers == std::numeric_limits<uint32_t>::max()) {
            // the threshold changes from here on: reads before this one had no sketch, so no hits
            prm.expected_kmers = (uint32_t) (seq.size() * 2 / a->w);    // :413-416
        }
        bases += seq;
        offsets.push_back(bases.size());
        id++;
        if (id > 10000000) break;                                       // :420-423
        if (bases.size() >= batch_bases && (rc = flush())) break;
    }
    if (rc == PMAP_OK) rc = flush();
    pmap_result_destroy(res);
    if (rc) {
        delete pg;
        return rc;
    }
    if (covg_out) *covg_out = (uint32_t) (covg / a->genome_size);      // :442, 454
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
        const PgHit &h = pg->hits[i];
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

extern "C" void pmap_pangraph_destroy(pmap_pangraph_t *pg) { delete pg; }
