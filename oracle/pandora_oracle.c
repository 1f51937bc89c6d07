/*
 * CPU oracle for the `pandora map` read -> PRG minimizer path: plain-C
 * restatement of the reference algorithm.  TEST INFRASTRUCTURE ONLY (see
 * pandora_oracle.h for the parity status and for who may use it).
 */
#include "pandora_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* src/inthash.cpp:42-59: 0,1,2,3 and A/a C/c G/g T/t map to 0..3, all else 4 */
uint32_t orc_nt4(uint8_t c) {
    switch (c) {
        case 0: case 'A': case 'a': return 0;
        case 1: case 'C': case 'c': return 1;
        case 2: case 'G': case 'g': return 2;
        case 3: case 'T': case 't': return 3;
        default: return 4;
    }
}

/* src/inthash.cpp:115-125 */
uint64_t orc_hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

/* src/inthash.cpp:129-162: ambiguous letters are skipped, not rejected */
void orc_kmerhash(const char *s, uint32_t k, uint64_t *fwd, uint64_t *rc) {
    uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1, kmer[2] = {0, 0};
    for (const char *p = s; *p; ++p) {
        uint32_t c = orc_nt4((uint8_t) *p);
        if (c < 4) {
            kmer[0] = (kmer[0] << 2 | c) & mask;
            kmer[1] = (kmer[1] >> 2) | (3ULL ^ c) << shift1;
        }
    }
    *fwd = orc_hash64(kmer[0], mask);
    *rc = orc_hash64(kmer[1], mask);
}

/* ------------------------------------------------------------------ */
typedef struct {
    uint64_t kmer;
    uint32_t start;
    uint8_t strand;
} mini_t;

/* Seq::minimizer_sketch (src/seq.cpp:91-122) with its helpers (:35-89).
 * mark[j] is set for every k-mer start j inserted into the sketch set; hv/sv
 * receive its hash and strand.  Returns 0 when the sketch is empty (too short,
 * or a non-ACGT letter cleared it), else 1. */
static int sketch_mark(const char *seq, uint64_t len, uint32_t w, uint32_t k, uint8_t *mark,
                       uint64_t *hv, uint8_t *sv) {
    if (len + 1 < (uint64_t) w + k) return 0;                      /* :92-94 */
    uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1, smallest = UINT64_MAX;
    uint64_t kmer[2] = {0, 0}, kh[2] = {0, 0};
    uint32_t buff = 0;
    mini_t *window = (mini_t *) malloc(sizeof(mini_t) * (w + 1));
    uint32_t wsize = 0;
    int any = 0;
    for (uint64_t i = 0; i < len; ++i) {
        uint32_t c = orc_nt4((uint8_t) seq[i]);                    /* :41 */
        if (c >= 4) {                                              /* :49-53, :106-107 */
            free(window);
            return -1;
        }
        kmer[0] = (kmer[0] << 2 | c) & mask;                       /* :43 */
        kmer[1] = (kmer[1] >> 2) | (3ULL ^ c) << shift1;           /* :44 */
        kh[0] = orc_hash64(kmer[0], mask);
        kh[1] = orc_hash64(kmer[1], mask);
        buff++;
        if (buff >= k) {                                           /* :109-111 */
            window[wsize].kmer = kh[0] < kh[1] ? kh[0] : kh[1];
            window[wsize].start = buff - k;
            window[wsize].strand = (kh[0] <= kh[1]);
            wsize++;
        }
        if (wsize == w) {                                          /* :113-114 minimize_window */
            uint32_t pos_of_smallest = 0;
            uint64_t s = UINT64_MAX;
            for (uint32_t j = 0; j < wsize; ++j)                   /* :56-67 `<=` keeps the last argmin */
                if (window[j].kmer <= s) { s = window[j].kmer; pos_of_smallest = j; }
            smallest = s;
            for (uint32_t j = 0; j < wsize; ++j)                   /* :69-76 every tie goes in */
                if (window[j].kmer == s) {
                    mark[window[j].start] = 1;
                    hv[window[j].start] = window[j].kmer;
                    sv[window[j].start] = window[j].strand;
                    any = 1;
                }
            uint32_t keep = wsize - (pos_of_smallest + 1);         /* :82 erase through the argmin */
            memmove(window, window + pos_of_smallest + 1, sizeof(mini_t) * keep);
            wsize = keep;
        } else if (buff >= w + k && wsize > 0 && window[wsize - 1].kmer <= smallest) { /* :115-116 */
            mini_t *b = &window[wsize - 1];
            mark[b->start] = 1;
            hv[b->start] = b->kmer;
            sv[b->start] = b->strand;
            any = 1;
            smallest = b->kmer;                                    /* :85-89 */
            wsize = 0;
        }
    }
    free(window);
    return any;
}

static int mini_set_cmp(const void *a, const void *b) {           /* src/minimizer.cpp:18-34 */
    const mini_t *x = (const mini_t *) a, *y = (const mini_t *) b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    if (x->strand != y->strand) return x->strand > y->strand ? -1 : 1;
    return 0;
}

int64_t orc_sketch(const char *seq, uint64_t len, uint32_t w, uint32_t k, uint64_t *kmer,
                   uint32_t *start, uint8_t *strand, uint64_t cap) {
    uint64_t n = len ? len : 1;
    uint8_t *mark = (uint8_t *) calloc(n, 1);
    uint64_t *hv = (uint64_t *) malloc(n * sizeof(uint64_t));
    uint8_t *sv = (uint8_t *) malloc(n);
    int64_t cnt = 0;
    if (sketch_mark(seq, len, w, k, mark, hv, sv) > 0) {
        for (uint64_t j = 0; j < len; ++j) cnt += mark[j];
        if ((uint64_t) cnt > cap) {
            cnt = -1;
        } else {
            mini_t *m = (mini_t *) malloc(sizeof(mini_t) * (size_t) (cnt ? cnt : 1));
            uint64_t o = 0;
            for (uint64_t j = 0; j < len; ++j)
                if (mark[j]) { m[o].kmer = hv[j]; m[o].start = (uint32_t) j; m[o].strand = sv[j]; o++; }
            qsort(m, (size_t) cnt, sizeof(mini_t), mini_set_cmp);
            for (int64_t i = 0; i < cnt; ++i) { kmer[i] = m[i].kmer; start[i] = m[i].start; strand[i] = m[i].strand; }
            free(m);
        }
    }
    free(mark); free(hv); free(sv);
    return cnt;
}

/* ------------------------------------------------------------------ */
/* Interval::operator< (src/interval.cpp:47-53) inside Path::operator<
 * (src/prg/path.cpp:166-183). */
int orc_path_cmp(const uint32_t *as, const uint32_t *al, uint64_t an, const uint32_t *bs,
                 const uint32_t *bl, uint64_t bn) {
    uint64_t i = 0;
    while (i < an && i < bn) {
        if (as[i] != bs[i] || al[i] != bl[i]) {
            if (as[i] != bs[i]) return as[i] < bs[i] ? -1 : 1;
            return al[i] < bl[i] ? -1 : 1;
        }
        ++i;
    }
    if (i == an && i != bn) return -1;
    if (i != an && i == bn) return 1;
    return 0;
}

float orc_fraction(float e_rate, uint32_t k) {
    float x = e_rate * k;                 /* float * uint32 -> float */
    float f = (float) (0.5 / expf(x));    /* 0.5 is double: float result widened, divided, narrowed */
    return f;
}

uint32_t orc_threshold(uint32_t mpl, uint32_t expected, float fraction, uint32_t min_cluster_size) {
    uint32_t m = mpl < expected ? mpl : expected;
    uint32_t length_based = (uint32_t) ((float) m * fraction);
    return length_based > min_cluster_size ? length_based : min_cluster_size;
}

/* ------------------------------------------------------------------ */
typedef struct {
    uint32_t read, pos, prg, rec;
    uint8_t fwd;
} hit_t;

typedef struct {
    uint64_t begin;
    uint32_t size, read, prg, first_pos, last_pos, first_rec;
    uint8_t fwd;
} cl_t;

static __thread const orc_index_t *g_idx; /* comparator context (qsort has no user pointer); per thread: callers may run in parallel */

static int rec_path_cmp(uint32_t ra, uint32_t rb) {
    const orc_index_t *x = g_idx;
    uint64_t a0 = x->path_off[ra], a1 = x->path_off[ra + 1], b0 = x->path_off[rb], b1 = x->path_off[rb + 1];
    return orc_path_cmp(x->iv_start + a0, x->iv_len + a0, a1 - a0, x->iv_start + b0, x->iv_len + b0, b1 - b0);
}

/* MinimizerHit::operator< (src/minihit.cpp:44-66) */
static int hit_cmp(const void *a, const void *b) {
    const hit_t *x = (const hit_t *) a, *y = (const hit_t *) b;
    if (x->read != y->read) return x->read < y->read ? -1 : 1;
    if (x->prg != y->prg) return x->prg < y->prg ? -1 : 1;
    if (x->fwd != y->fwd) return x->fwd > y->fwd ? -1 : 1;          /* forward first */
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return rec_path_cmp(x->rec, y->rec);
}

/* clusterComp (src/minihits.cpp:95-109) */
static int cl_cmp(const void *a, const void *b) {
    const cl_t *x = (const cl_t *) a, *y = (const cl_t *) b;
    if (x->read != y->read) return x->read < y->read ? -1 : 1;
    if (x->first_pos != y->first_pos) return x->first_pos < y->first_pos ? -1 : 1;
    if (x->size != y->size) return x->size > y->size ? -1 : 1;      /* bigger first */
    if (x->prg != y->prg) return x->prg < y->prg ? -1 : 1;
    int c = rec_path_cmp(x->first_rec, y->first_rec);
    if (c) return c;
    if (x->fwd != y->fwd) return x->fwd < y->fwd ? -1 : 1;
    return 0;
}

static int64_t find_key(const orc_index_t *idx, const uint64_t *sorted_keys, const uint64_t *perm,
                        uint64_t key) {
    uint64_t lo = 0, hi = idx->n_keys;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo < idx->n_keys && sorted_keys[lo] == key) return (int64_t) perm[lo];
    return -1;
}

typedef struct { uint64_t key, idx; } kp_t;
static int kp_cmp(const void *a, const void *b) {
    const kp_t *x = (const kp_t *) a, *y = (const kp_t *) b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

#define GROW(ptr, n, cap, type)                                             \
    do {                                                                    \
        if ((n) >= (cap)) {                                                 \
            (cap) = (cap) ? (cap) * 2 : 1024;                               \
            (ptr) = (type *) realloc((ptr), sizeof(type) * (size_t) (cap)); \
        }                                                                   \
    } while (0)

int orc_map(const orc_index_t *idx, const orc_params_t *prm, const char *bases, const uint64_t *offsets,
            const uint32_t *read_ids, uint32_t n_reads, orc_result_t *out) {
    memset(out, 0, sizeof(*out));
    out->n_reads = n_reads;
    const uint32_t w = prm->w, k = prm->k;

    /* key lookup structure standing in for unordered_map::find (src/utils.cpp:175) */
    kp_t *kp = (kp_t *) malloc(sizeof(kp_t) * (size_t) (idx->n_keys ? idx->n_keys : 1));
    for (uint64_t i = 0; i < idx->n_keys; ++i) { kp[i].key = idx->keys[i]; kp[i].idx = i; }
    qsort(kp, (size_t) idx->n_keys, sizeof(kp_t), kp_cmp);
    uint64_t *skeys = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (idx->n_keys ? idx->n_keys : 1));
    uint64_t *perm = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (idx->n_keys ? idx->n_keys : 1));
    for (uint64_t i = 0; i < idx->n_keys; ++i) { skeys[i] = kp[i].key; perm[i] = kp[i].idx; }
    free(kp);

    uint64_t maxlen = 1;
    for (uint32_t r = 0; r < n_reads; ++r)
        if (offsets[r + 1] - offsets[r] > maxlen) maxlen = offsets[r + 1] - offsets[r];
    uint8_t *mark = (uint8_t *) malloc((size_t) maxlen);
    uint64_t *hv = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) maxlen);
    uint8_t *sv = (uint8_t *) malloc((size_t) maxlen);

    out->sketch_off = (uint64_t *) calloc((size_t) n_reads + 1, sizeof(uint64_t));
    uint64_t mcap = 0, hcap = 0, nm = 0, nh = 0;
    hit_t *hits = NULL;

    for (uint32_t r = 0; r < n_reads; ++r) {
        const char *seq = bases + offsets[r];
        uint64_t len = offsets[r + 1] - offsets[r];
        out->sketch_off[r] = nm;
        memset(mark, 0, (size_t) (len ? len : 1));
        if (sketch_mark(seq, len, w, k, mark, hv, sv) <= 0) continue;
        for (uint64_t j = 0; j < len; ++j) {
            if (!mark[j]) continue;
            if (nm >= mcap) {
                mcap = mcap ? mcap * 2 : 4096;
                out->mini_kmer = (uint64_t *) realloc(out->mini_kmer, sizeof(uint64_t) * (size_t) mcap);
                out->mini_pos = (uint32_t *) realloc(out->mini_pos, sizeof(uint32_t) * (size_t) mcap);
                out->mini_strand = (uint8_t *) realloc(out->mini_strand, (size_t) mcap);
            }
            out->mini_kmer[nm] = hv[j];
            out->mini_pos[nm] = (uint32_t) j;
            out->mini_strand[nm] = sv[j];
            nm++;
            /* add_read_hits (src/utils.cpp:174-183) + MinimizerHit ctor (src/minihit.cpp:14-22) */
            int64_t ki = find_key(idx, skeys, perm, hv[j]);
            if (ki < 0) continue;
            for (uint64_t q = idx->rec_off[ki]; q < idx->rec_off[ki + 1]; ++q) {
                GROW(hits, nh, hcap, hit_t);
                hits[nh].read = read_ids[r];
                hits[nh].pos = (uint32_t) j;
                hits[nh].prg = idx->rec_prg[q];
                hits[nh].rec = (uint32_t) q;
                hits[nh].fwd = (sv[j] == idx->rec_strand[q]);
                nh++;
            }
        }
    }
    out->sketch_off[n_reads] = nm;
    out->n_mini = nm;
    free(mark); free(hv); free(sv); free(skeys); free(perm);

    /* MinimizerHits::sort (src/minihits.cpp:42-53): std::set<.., pComp> = sorted + unique */
    g_idx = idx;
    qsort(hits, (size_t) nh, sizeof(hit_t), hit_cmp);
    uint64_t u = 0;
    for (uint64_t i = 0; i < nh; ++i) {
        if (u > 0 && hit_cmp(&hits[u - 1], &hits[i]) == 0) { out->n_dup_hits++; continue; }
        hits[u++] = hits[i];
    }
    nh = u;
    out->n_hits = nh;
    out->hit_read = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) (nh ? nh : 1));
    out->hit_pos = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) (nh ? nh : 1));
    out->hit_prg = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) (nh ? nh : 1));
    out->hit_rec = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) (nh ? nh : 1));
    out->hit_fwd = (uint8_t *) malloc((size_t) (nh ? nh : 1));
    for (uint64_t i = 0; i < nh; ++i) {
        out->hit_read[i] = hits[i].read; out->hit_pos[i] = hits[i].pos; out->hit_prg[i] = hits[i].prg;
        out->hit_rec[i] = hits[i].rec; out->hit_fwd[i] = hits[i].fwd;
    }

    /* define_clusters (src/utils.cpp:189-249) */
    cl_t *cl = NULL;
    uint64_t ncl = 0, ccap = 0, nall = 0;
    if (nh > 0) {
        uint64_t begin = 0;
        for (uint64_t i = 1; i <= nh; ++i) {
            int cut = (i == nh);
            if (!cut) {
                const hit_t *c = &hits[i], *p = &hits[i - 1];
                cut = c->read != p->read || c->prg != p->prg || c->fwd != p->fwd ||
                      abs((int) c->pos - (int) p->pos) > prm->max_diff;      /* :206-209 */
            }
            if (!cut) continue;
            const hit_t *p = &hits[i - 1];
            uint32_t mpl = prm->min_path_length[p->prg];
            uint32_t thr = orc_threshold(mpl, prm->expected_kmers, prm->fraction, prm->min_cluster_size);
            uint64_t size = i - begin;
            nall++;
            if (size > thr) {                                                 /* :219-220 strict */
                GROW(cl, ncl, ccap, cl_t);
                cl[ncl].begin = begin; cl[ncl].size = (uint32_t) size; cl[ncl].read = p->read;
                cl[ncl].prg = p->prg; cl[ncl].fwd = p->fwd; cl[ncl].first_pos = hits[begin].pos;
                cl[ncl].last_pos = p->pos; cl[ncl].first_rec = hits[begin].rec;
                ncl++;
            }
            begin = i;
        }
    }
    qsort(cl, (size_t) ncl, sizeof(cl_t), cl_cmp);      /* std::set<..., clusterComp> iteration order */
    out->n_all_clusters = nall;
    out->n_clusters = ncl;
    size_t na = (size_t) (ncl ? ncl : 1);
    out->cl_begin = (uint64_t *) malloc(sizeof(uint64_t) * na);
    out->cl_size = (uint32_t *) malloc(sizeof(uint32_t) * na);
    out->cl_read = (uint32_t *) malloc(sizeof(uint32_t) * na);
    out->cl_prg = (uint32_t *) malloc(sizeof(uint32_t) * na);
    out->cl_first_pos = (uint32_t *) malloc(sizeof(uint32_t) * na);
    out->cl_last_pos = (uint32_t *) malloc(sizeof(uint32_t) * na);
    out->cl_fwd = (uint8_t *) malloc(na);
    out->cl_alive = (uint8_t *) malloc(na);
    for (uint64_t i = 0; i < ncl; ++i) {
        out->cl_begin[i] = cl[i].begin; out->cl_size[i] = cl[i].size; out->cl_read[i] = cl[i].read;
        out->cl_prg[i] = cl[i].prg; out->cl_first_pos[i] = cl[i].first_pos; out->cl_last_pos[i] = cl[i].last_pos;
        out->cl_fwd[i] = cl[i].fwd; out->cl_alive[i] = 1;
    }

    /* filter_clusters (src/utils.cpp:251-288): walk consecutive survivors */
    if (ncl > 0) {
        uint64_t prev = 0;
        for (uint64_t cur = 1; cur < ncl; ++cur) {
            if (cl[cur].read == cl[prev].read &&
                ((cl[cur].prg == cl[prev].prg && cl[cur].fwd != cl[prev].fwd) ||
                 cl[cur].last_pos <= cl[prev].last_pos)) {
                if (cl[prev].size >= cl[cur].size) {
                    out->cl_alive[cur] = 0;     /* erase current, previous stays (:276-279) */
                    continue;
                }
                out->cl_alive[prev] = 0;        /* erase previous (:280-283) */
            }
            prev = cur;
        }
    }
    free(cl);
    free(hits);
    return 0;
}

void orc_result_free(orc_result_t *r) {
    free(r->sketch_off); free(r->mini_kmer); free(r->mini_pos); free(r->mini_strand);
    free(r->hit_read); free(r->hit_pos); free(r->hit_prg); free(r->hit_rec); free(r->hit_fwd);
    free(r->cl_begin); free(r->cl_size); free(r->cl_read); free(r->cl_prg); free(r->cl_first_pos);
    free(r->cl_last_pos); free(r->cl_fwd); free(r->cl_alive);
    memset(r, 0, sizeof(*r));
}

/* ------------------------------------------------------------------ */
uint32_t orc_read_loop(const char *bases, const uint64_t *offsets, uint32_t n_reads, uint32_t w, uint32_t k,
                       uint32_t genome_size, uint32_t max_covg, int illumina, uint8_t *take, uint32_t *id_out,
                       uint32_t *covg_out, uint32_t *expected_kmers) {
    uint64_t covg = 0;
    uint32_t expected = UINT32_MAX, id = 0, i = 0;
    uint64_t maxlen = 1;
    for (uint32_t r = 0; r < n_reads; ++r)
        if (offsets[r + 1] - offsets[r] > maxlen) maxlen = offsets[r + 1] - offsets[r];
    uint8_t *mark = (uint8_t *) malloc((size_t) maxlen);
    uint64_t *hv = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) maxlen);
    uint8_t *sv = (uint8_t *) malloc((size_t) maxlen);
    for (i = 0; i < n_reads; ++i) {
        uint64_t len = offsets[i + 1] - offsets[i];
        take[i] = 0;
        id_out[i] = id;
        memset(mark, 0, (size_t) (len ? len : 1));
        int nonempty = sketch_mark(bases + offsets[i], len, w, k, mark, hv, sv) > 0;  /* :402-403 */
        if (nonempty) {
            covg += len;                                                              /* :404 */
            if (covg / genome_size > max_covg) { ++i; break; }                        /* :405-408 */
        } else {
            id++;                                                                     /* :409-412 */
            continue;
        }
        if (illumina && expected == UINT32_MAX) expected = (uint32_t) (len * 2 / w);  /* :413-416 */
        take[i] = 1;                                                                  /* :418 */
        id++;
        if (id > 10000000) { ++i; break; }                                            /* :420-423 */
    }
    for (uint32_t j = i; j < n_reads; ++j) { take[j] = 0; id_out[j] = id; }
    free(mark); free(hv); free(sv);
    *covg_out = (uint32_t) (covg / genome_size);                                      /* :442,454 */
    *expected_kmers = expected;
    return i;
}

/* ------------------------------------------------------------------ */
typedef struct {
    const char *buf;
    size_t n, p;
    int eofbit;
} mstream_t;

/* std::getline(...).good(): 0 if the extraction hit end-of-data before a newline */
static int ms_getline(mstream_t *s, const char **line, size_t *len) {
    if (s->eofbit || s->p >= s->n) { s->eofbit = 1; *line = s->buf + s->n; *len = 0; return 0; }
    size_t b = s->p;
    while (s->p < s->n && s->buf[s->p] != '\n') s->p++;
    *line = s->buf + b;
    *len = s->p - b;
    if (s->p >= s->n) { s->eofbit = 1; return 0; }
    s->p++;
    return 1;
}

void orc_fastaq_iterate(const char *buf, size_t n, orc_read_cb cb, void *user) {
    mstream_t s = {buf, n, 0, 0};
    const char *line = buf; size_t line_len = 0;
    const char *name = buf; size_t name_len = 0;
    size_t rcap = 1024, rlen = 0;
    char *read = (char *) malloc(rcap);
    while (!(s.eofbit || s.p >= s.n)) {                                   /* FastaqHandler::eof :38-41 */
        /* get_next :43-71 */
        if (line_len > 0 && (line[0] == '>' || line[0] == '@')) { name = line + 1; name_len = line_len - 1; rlen = 0; }
        while (ms_getline(&s, &line, &line_len)) {
            if (line_len > 0 && line[0] == '+') {
                ms_getline(&s, &line, &line_len);                         /* quality line skipped */
            } else if (line_len == 0 || line[0] == '>' || line[0] == '@') {
                if (rlen > 0 || line_len == 0) break;
                name = line + 1; name_len = line_len - 1; rlen = 0;
            } else {
                if (rlen + line_len > rcap) { while (rlen + line_len > rcap) rcap *= 2; read = (char *) realloc(read, rcap); }
                memcpy(read + rlen, line, line_len);
                rlen += line_len;
            }
        }
        cb(name, name_len, read, rlen, user);
    }
    free(read);
}
