/*
 * CPU oracle for the `pandora map` read -> PRG minimizer path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * (leoisl/pandora) algorithm, used by tests/, by __graft_entry__.smoke() and by
 * bench.py's cpu_baseline leg as the checker.  Nothing in the product path
 * (pandora_amd/) links, loads or calls it.
 *
 * Parity status: PINNED.  The restatement is checked against
 *   - the known answers held by the reference's own unit tests
 *     (test/inthash_test.cpp, test/seq_test.cpp, test/minimizer_test.cpp,
 *      test/minihit_test.cpp, test/minihits_test.cpp, test/utils_test.cpp,
 *      test/path_test.cpp, test/minirecord_test.cpp, test/index_test.cpp),
 *   - the reference's data fixtures (test/test_cases/prg{1,2,3}.fa.idx ...),
 *   - the boost-free reference translation units compiled unmodified into
 *     oracle/_ref/libpandora_ref.so (hash64, nt4, KmerHash, Interval, prg::Path,
 *     Minimizer, MiniRecord, MinimizerHit, MinimizerHits::sort, pComp,
 *     clusterComp) by oracle/Makefile.
 * src/seq.cpp, src/utils.cpp and src/index.cpp include boost headers that are
 * absent from this image, so those three files are NOT compiled; their logic is
 * pinned through the reference's unit-test expectations only.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef PANDORA_ORACLE_H
#define PANDORA_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/inthash.cpp:42-63 */
uint32_t orc_nt4(uint8_t c);
/* src/inthash.cpp:115-125 */
uint64_t orc_hash64(uint64_t key, uint64_t mask);
/* src/inthash.cpp:129-162 (KmerHash::kmerhash without the memo table) */
void orc_kmerhash(const char *s, uint32_t k, uint64_t *fwd, uint64_t *rc);

/* src/seq.cpp:91-122 Seq::minimizer_sketch, stateful loop restated.
 * Output in std::set<Minimizer> order (src/minimizer.cpp:18-34):
 * (kmer, start, strand descending).  Returns the sketch size, or -1 if cap is
 * too small. */
int64_t orc_sketch(const char *seq, uint64_t len, uint32_t w, uint32_t k,
                   uint64_t *kmer, uint32_t *start, uint8_t *strand, uint64_t cap);

/* Flat view of Index::minhash (include/index.h:16): key i owns records
 * [rec_off[i], rec_off[i+1]); record j owns intervals
 * [path_off[j], path_off[j+1]) (prg::Path, include/prg/path.h). */
typedef struct {
    uint64_t n_keys;
    const uint64_t *keys;
    const uint64_t *rec_off;
    uint64_t n_recs;
    const uint32_t *rec_prg;
    const uint32_t *rec_knode;
    const uint8_t *rec_strand;
    const uint64_t *path_off;
    const uint32_t *iv_start;
    const uint32_t *iv_len;
} orc_index_t;

/* src/prg/path.cpp:166-183 Path::operator< over flat intervals: returns <0, 0, >0.
 * (0 also when y is a strict prefix of x: operator< is false both ways only for
 * equal paths, a longer path with an equal prefix compares greater.) */
int orc_path_cmp(const uint32_t *as, const uint32_t *al, uint64_t an,
                 const uint32_t *bs, const uint32_t *bl, uint64_t bn);

typedef struct {
    uint32_t w, k;
    int32_t max_diff;
    float fraction;            /* src/utils.cpp:394 */
    uint32_t min_cluster_size;
    uint32_t expected_kmers;   /* src/utils.cpp:395,413-416 */
    uint32_t n_prg;
    const uint32_t *min_path_length; /* KmerGraph::min_path_length() per prg */
} orc_params_t;

typedef struct {
    /* per read */
    uint32_t n_reads;
    uint64_t *sketch_off;      /* n_reads+1: minimizers of read r, position order */
    uint64_t n_mini;
    uint64_t *mini_kmer;
    uint32_t *mini_pos;
    uint8_t *mini_strand;
    /* hits in MinimizerHit::operator< order (src/minihit.cpp:44-66) */
    uint64_t n_hits;
    uint32_t *hit_read, *hit_pos, *hit_prg, *hit_rec;
    uint8_t *hit_fwd;
    uint64_t n_dup_hits;       /* hits dropped by set semantics; the reference asserts 0 (src/minihits.cpp:44-46) */
    /* clusters kept by define_clusters (src/utils.cpp:189-249) in clusterComp order
     * (src/minihits.cpp:95-109) */
    uint64_t n_clusters;
    uint64_t *cl_begin;        /* first hit (index into hit arrays) */
    uint32_t *cl_size, *cl_read, *cl_prg, *cl_first_pos, *cl_last_pos;
    uint8_t *cl_fwd;
    uint8_t *cl_alive;         /* 1 if it survives filter_clusters (src/utils.cpp:251-288) */
    uint64_t n_all_clusters;   /* number of clusters before the size threshold */
} orc_result_t;

/* add_read_hits + MinimizerHits::sort + define_clusters + filter_clusters for a
 * batch of reads (src/utils.cpp:167-288, 354-377).  Read i is
 * bases[offsets[i] .. offsets[i+1]) and has id read_ids[i].  Returns 0 on
 * success.  Free with orc_result_free. */
int orc_map(const orc_index_t *idx, const orc_params_t *prm, const char *bases,
            const uint64_t *offsets, const uint32_t *read_ids, uint32_t n_reads,
            orc_result_t *out);
void orc_result_free(orc_result_t *r);

/* src/utils.cpp:394: 0.5 / exp(e_rate * k) evaluated with the float overload of
 * exp and stored to float. */
float orc_fraction(float e_rate, uint32_t k);
/* src/utils.cpp:211-213,219-220: max((uint32)(min(mpl, expected) * fraction), min_cluster_size) */
uint32_t orc_threshold(uint32_t min_path_length, uint32_t expected_kmers, float fraction,
                       uint32_t min_cluster_size);

/* The read loop of pangraph_from_read_file (src/utils.cpp:396-432): decides which
 * reads are mapped and under which id.  For read i of the file (in get_next()
 * order) writes take[i] = 1 if add_read_hits runs for it, id[i] = the read id it
 * gets.  Returns the number of loop iterations performed (reads consumed);
 * *covg_out = final covg / genome_size; *expected_kmers as in :413-416. */
uint32_t orc_read_loop(const char *bases, const uint64_t *offsets, uint32_t n_reads, uint32_t w,
                       uint32_t k, uint32_t genome_size, uint32_t max_covg, int illumina,
                       uint8_t *take, uint32_t *id, uint32_t *covg_out, uint32_t *expected_kmers);

/* FastaqHandler::get_next state machine (src/fastaq_handler.cpp:43-71) over an
 * in-memory, uncompressed buffer.  Calls cb(name, read, user) once per loop
 * iteration of `while(!fh.eof()) { fh.get_next(); ... }`. */
typedef void (*orc_read_cb)(const char *name, size_t name_len, const char *read, size_t read_len,
                            void *user);
void orc_fastaq_iterate(const char *buf, size_t n, orc_read_cb cb, void *user);

#ifdef __cplusplus
}
#endif
#endif
