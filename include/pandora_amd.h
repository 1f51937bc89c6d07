/*
 * pandora_amd C ABI: the MI355X drop-in boundary for the read -> PRG minimizer
 * path of `pandora map` (leoisl/pandora).
 *
 * The reference has no FFI layer: the path sits behind plain C++ functions
 * (include/utils.h:66-90).  A maintainer replaces the body of
 * pangraph_from_read_file (src/utils.cpp:379-455) with calls to this library;
 * every entry point below names the reference code it stands in for.
 *
 * Conventions: every function returns 0 on success and a non-zero PMAP_E* code
 * on failure; pmap_last_error() then returns a message (thread local).  All
 * pointers in the signatures are HOST pointers unless the name says `dev`.
 * Calls are blocking.
 *
 * Devices and threads.  The library works through mapping contexts: one GPU,
 * the streams the library drives it with, and the launch state of one host
 * thread.  pmap_init(device) makes the first context on that GPU the calling
 * thread's current one; every reads / result / coverage object belongs to the
 * context that was current when it was created, and a call on an object switches
 * the calling thread to the object's context.  Different threads may work in
 * different contexts at the same time (that is how pmap_pangraph_from_read_file
 * uses several GPUs, see pmap_init_devices); one context, and the objects that
 * belong to it, must be used by one thread at a time.  An index may be shared
 * by all contexts: it is uploaded once to every GPU it is used on.
 */
#ifndef PANDORA_AMD_H
#define PANDORA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMAP_OK 0
#define PMAP_EINVAL 1   /* bad argument / malformed input (reference: assert or exit(1)) */
#define PMAP_EHIP 2     /* a HIP runtime call failed */
#define PMAP_EIO 3      /* file could not be opened or parsed (reference: exit(1)) */
#define PMAP_ELIMIT 4   /* a documented size limit was exceeded */
#define PMAP_EDUP 5     /* duplicate (key, prg, path, strand) record: the reference would abort in MinimizerHits::sort (src/minihits.cpp:44-46) */

typedef struct pmap_index pmap_index_t;       /* Index::minhash flattened and resident in HBM */
typedef struct pmap_reads pmap_reads_t;       /* a batch of reads, 2-bit packed, resident in HBM */
typedef struct pmap_result pmap_result_t;     /* minimizers / hits / clusters of one batch, in HBM */
typedef struct pmap_pangraph pmap_pangraph_t; /* host-side pangenome::Graph summary */

const char *pmap_last_error(void);

/* Makes the first mapping context on GPU `device` (created if there is none
 * yet) the calling thread's current one.  device < 0 keeps the thread's
 * context, or makes one on the current HIP device. */
int pmap_init(int device);

/* The mapping contexts pmap_pangraph_from_read_file spreads the batches of a
 * read file over: one per entry of `devices`, in this order; a GPU may be
 * listed more than once (two contexts on one GPU overlap the upload of one
 * batch with the kernels and the host work of another).  n = 0 goes back to
 * the default, four contexts on the calling thread's GPU.  No counterpart in
 * the reference, which is single-threaded: results do not depend on it. */
int pmap_init_devices(const int *devices, int n);

/* ---------------------------------------------------------------- index */

/* Flat view of Index::minhash (include/index.h:16) as the caller holds it:
 * key i owns records [rec_off[i], rec_off[i+1]); record j is a MiniRecord
 * (include/minirecord.h:9-14) whose prg::Path is the intervals
 * [path_off[j], path_off[j+1]) given as (start, length).  min_path_length[p] is
 * prgs[p]->kmer_prg.min_path_length() (src/kmergraph.cpp:664-684). */
typedef struct {
    uint32_t k;
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
    uint32_t n_prg;
    const uint32_t *min_path_length;
} pmap_index_desc_t;

/* Replaces the unordered_map behind Index::minhash / Index::add_record
 * (src/index.cpp:20-37): builds the open-addressing table + CSR record array +
 * per-(prg,path) ranks (order of prg::Path::operator<, src/prg/path.cpp:166-183)
 * on the host; they are uploaded to HBM by the first pmap_map call that uses
 * the index (or by pmap_index_upload_now).  Record indices reported by results
 * are indices into the caller's arrays. */
int pmap_index_create(const pmap_index_desc_t *desc, pmap_index_t **out);
/* Uploads the table and records to HBM now (e.g. outside a timed region). */
int pmap_index_upload_now(pmap_index_t *idx);

/* Replaces Index::load (src/index.cpp:75-124): parses the reference's text
 * `.idx` format.  min_path_length as above (n_prg entries). */
int pmap_index_load(const char *idx_path, uint32_t k, uint32_t n_prg, const uint32_t *min_path_length,
                    pmap_index_t **out);

/* Binary cache of a built index (no counterpart in the reference, whose
 * Index::load re-parses the text every run).  pmap_index_save_cache writes the
 * flat structures; source_path (may be NULL) is the `.idx` file they came
 * from, remembered by size and modification time.  pmap_index_load_cache reads
 * a cache back.  pmap_index_load_cached = pmap_index_load, through the cache at
 * cache_path when it was built from this very idx_path, k and min_path_length
 * (*from_cache = 1), else from the text, after which the cache is (re)written. */
int pmap_index_save_cache(const pmap_index_t *idx, const char *cache_path, const char *source_path);
int pmap_index_load_cache(const char *cache_path, pmap_index_t **out);
int pmap_index_load_cached(const char *idx_path, uint32_t k, uint32_t n_prg, const uint32_t *min_path_length,
                           const char *cache_path, pmap_index_t **out, int *from_cache);

/* Replaces KmerGraph::load + KmerGraph::min_path_length
 * (src/kmergraph.cpp:716-820, 664-684) for one kmer-graph GFA file. */
int pmap_gfa_min_path_length(const char *gfa_path, uint32_t *out);

typedef struct {
    uint32_t k;
    uint64_t n_keys, n_recs, n_paths, table_slots, device_bytes;
    uint32_t n_prg;
    uint32_t filter_bits;           /* log2 of the size of the presence filter in bits, 0 = none */
    uint32_t filter_bits_per_key;   /* 1 or 2 (0 without a filter) */
    uint32_t filter_direct;         /* 1: the filter is a bitmap over the whole key range (bit = key), exact */
} pmap_index_info_t;
int pmap_index_info(const pmap_index_t *idx, pmap_index_info_t *info);

/* Host copy of the flattened records, in the caller's record numbering
 * (what Index::save would write, src/index.cpp:51-68).  Any pointer may be
 * NULL.  path_rank[j] orders records of one prg as prg::Path::operator<. */
int pmap_index_records(const pmap_index_t *idx, uint64_t *key_of_rec, uint32_t *rec_prg, uint32_t *rec_knode,
                       uint8_t *rec_strand, uint32_t *path_rank, uint64_t *path_off);
int pmap_index_intervals(const pmap_index_t *idx, uint32_t *iv_start, uint32_t *iv_len, uint64_t *n_intervals);
void pmap_index_destroy(pmap_index_t *idx);

/* ---------------------------------------------------------------- reads */

/* Uploads a batch of reads (what FastaqHandler::get_next yields,
 * src/fastaq_handler.cpp:43-71): read i is bases[offsets[i] .. offsets[i+1]) and
 * gets read id first_read_id + i.  On the device the batch is encoded with the
 * nt4 table (src/inthash.cpp:42-63) into 2 bits per base plus one "has a
 * non-ACGT letter" flag per read (src/seq.cpp:49-53). */
int pmap_reads_create(const char *bases, const uint64_t *offsets, uint32_t n_reads, uint32_t first_read_id,
                      pmap_reads_t **out);
/* Replaces the batch held by `reads` with another one, reusing its device buffers. */
int pmap_reads_update(pmap_reads_t *reads, const char *bases, const uint64_t *offsets, uint32_t n_reads,
                      uint32_t first_read_id);
/* The same for a batch the caller has packed already: packed holds, read after read, ceil(length / 16)
 * 32-bit words per read, 16 bases each, the first base in the top two bits (A, C, G, T = 0, 1, 2, 3 as nt4
 * gives them, src/inthash.cpp:42-63; unused low bits zero); offsets[i+1] - offsets[i] is the length of read
 * i (offsets as in pmap_reads_create, no bases behind them); acgt_only[i] = 0 marks a read with a letter
 * nt4 maps to 4, which has no sketch (src/seq.cpp:49-53; its words are not looked at).  A quarter of the
 * bytes of pmap_reads_create cross the bus: what pmap_pangraph_from_read_file does with its batches. */
/* Packs reads for pmap_reads_create_packed (on the host): packed must have room for the sum over the reads of
 * ceil(length / 16) words, acgt_only for n_reads bytes. */
int pmap_pack_reads(const char *bases, const uint64_t *offsets, uint32_t n_reads, uint32_t *packed,
                    uint8_t *acgt_only);
int pmap_reads_create_packed(const uint32_t *packed, const uint64_t *offsets, const uint8_t *acgt_only,
                             uint32_t n_reads, uint32_t first_read_id, pmap_reads_t **out);
int pmap_reads_update_packed(pmap_reads_t *reads, const uint32_t *packed, const uint64_t *offsets,
                             const uint8_t *acgt_only, uint32_t n_reads, uint32_t first_read_id);
/* Re-runs only the device encode kernel of an uploaded batch (for measurement). */
int pmap_reads_repack(pmap_reads_t *reads);
int pmap_reads_info(const pmap_reads_t *reads, uint32_t *n_reads, uint64_t *n_bases, uint64_t *device_bytes);
void pmap_reads_destroy(pmap_reads_t *reads);

/* ---------------------------------------------------------------- map */

#define PMAP_STAGE_SKETCH 1   /* Seq::minimizer_sketch              src/seq.cpp:91-122 */
#define PMAP_STAGE_HITS 2     /* add_read_hits                      src/utils.cpp:167-187 */
#define PMAP_STAGE_SORT 3     /* MinimizerHits::sort                src/minihits.cpp:42-53 */
#define PMAP_STAGE_CLUSTER 4  /* define_clusters                    src/utils.cpp:189-249 */

typedef struct {
    uint32_t w, k;
    int32_t max_diff;           /* src/utils.cpp:209 */
    float fraction;             /* fraction_kmers_required_for_cluster, src/utils.cpp:394 */
    uint32_t min_cluster_size;
    uint32_t expected_kmers;    /* expected_number_kmers_in_short_read_sketch, src/utils.cpp:395,413-416 */
    uint32_t last_stage;        /* PMAP_STAGE_*; 0 = all */
    uint32_t profile;           /* 1: HIP-event time of every kernel; 2: of the sketch kernel only (cheapest) */
    uint32_t flags;             /* PMAP_FLAG_* */
} pmap_params_t;

/* By default the index lookup runs inside the sketch kernel and only minimizers
 * that have records are kept.  With this flag the sketch kernel keeps every
 * minimizer (pmap_result_minimizers works) and the lookup runs as its own
 * kernel; hits and clusters are the same either way.  last_stage =
 * PMAP_STAGE_SKETCH implies it. */
#define PMAP_FLAG_UNFUSED 1u
/* Diagnostic choice of the sketch kernel where more than one gives the same results.  k <= 15 with
 * w <= 16 runs by default on the kernel with 16 positions per lane and the windows in
 * registers; PMAP_FLAG_SKETCH_LDS takes the general kernel (windows through LDS) that serves every
 * other (w, k), PMAP_FLAG_SKETCH_NP8 (w = 14 only) the register kernel with 8 positions per lane.
 * The environment variable PMAP_SKETCH=np8|lds, read by pmap_init, sets the default. */
#define PMAP_FLAG_SKETCH_NP8 2u
#define PMAP_FLAG_SKETCH_LDS 4u
/* Diagnostic: sort with the kernel variant for indexes of more than 2^19 loci (keys too wide to share a
 * word with a position) whatever the index. */
#define PMAP_FLAG_SORT_WIDE 8u
/* Diagnostic: reads of middling hit counts (more than 192 hits, 320 with w < 8, and at most 2048 by default; PMAP_SORT_WAVE_MIN / _MAX, read
 * by pmap_init) are sorted by one wavefront each with a network in registers; with PMAP_FLAG_SORT_NO_WAVE they
 * take the rank-counting sort like shorter reads, with PMAP_FLAG_TEST_SORT_WAVE_ALL every read of 2 to 2048
 * hits takes the wavefront sort.  Results are the same. */
#define PMAP_FLAG_SORT_NO_WAVE 32u
#define PMAP_FLAG_TEST_SORT_WAVE_ALL 64u
/* Test hook: the prefix sums that chain the workgroups of the cluster kernels give up at their first wait, as
 * they would after a very long one, so that the batch takes the fallback (one workgroup per kernel) it takes
 * then.  Results are the same. */
#define PMAP_FLAG_TEST_CHAIN_GIVES_UP 16u
/* By default one kernel sorts a read's hits, cuts the clusters and applies the size threshold while the hits are
 * in its registers.  With PMAP_FLAG_TAIL_STAGED the sort and the two cluster kernels run one after the other,
 * each over all hits (the flags above choose among their variants); indexes of more than 2^20 PRGs always take
 * them.  Results are the same.  PMAP_FLAG_TEST_TAIL_NARROW: the fused kernel sorts the reads of a chunk two at a
 * time, as it would many short reads on an index of very many PRGs. */
#define PMAP_FLAG_TAIL_STAGED 128u
#define PMAP_FLAG_TEST_TAIL_NARROW 256u

/* 0.5 / exp(e_rate * k) as src/utils.cpp:394 evaluates it (float). */
float pmap_fraction(float e_rate, uint32_t k);

/* Replaces, for one batch of reads, the chain
 *   Seq::minimizer_sketch -> add_read_hits -> MinimizerHits::sort -> define_clusters
 * (src/seq.cpp:91-122, src/utils.cpp:167-249, src/minihits.cpp:42-53), entirely
 * on the GPU.  *result may point to a result of an earlier call, whose device
 * buffers are then reused (they only grow); pass a pointer to NULL the first
 * time. */
int pmap_map(const pmap_index_t *idx, const pmap_reads_t *reads, const pmap_params_t *prm,
             pmap_result_t **result);

/* The same in two halves, so that the host can prepare and enqueue the next batch
 * while the GPU maps this one (no counterpart in the reference, which is
 * synchronous and single-threaded).  pmap_map_async enqueues the kernels and
 * returns; pmap_map_wait blocks until they are done, maps the batch again if a
 * result buffer turned out too small, and makes the result readable.  idx and
 * reads must stay alive until the wait; every pmap_result_* accessor waits by
 * itself.  Results in flight at the same time must be different objects. */
int pmap_map_async(const pmap_index_t *idx, const pmap_reads_t *reads, const pmap_params_t *prm,
                   pmap_result_t **result);
int pmap_map_wait(pmap_result_t *result);

/* Device-side outputs of pmap_map_noalloc, allocated by the caller: hits in
 * MinimizerHit::operator< order as five arrays of hit_cap entries each (read =
 * index of the read in the batch, key = prg_id << 1 | !is_forward, position in
 * the read, path rank, record index in the caller's numbering) and the clusters
 * kept by define_clusters as rows of eight 32-bit words {first hit, size, read,
 * key, first position, last position, path rank of the first hit, cluster
 * number before the size threshold}.  All pointers are DEVICE pointers. */
typedef struct {
    uint32_t hit_cap;           /* at least 8192; used in multiples of 8192, at most 2^20 */
    uint32_t *hit_read, *hit_key, *hit_pos, *hit_rank, *hit_rec;
    uint32_t cluster_cap;       /* at least 1024; used in multiples of 1024, at most 2^20 */
    uint32_t *cluster_rows;
} pmap_dev_out_t;

#define PMAP_N_TIMERS 8
typedef struct {
    uint32_t n_reads;
    uint64_t n_bases;
    uint64_t n_minimizers;
    uint64_t n_hits;
    uint64_t n_clusters_all;    /* before the size threshold */
    uint64_t n_clusters;        /* kept by define_clusters */
    /* HIP-event times of the last pmap_map with profile=1, milliseconds (sums of
     * the kernels of each phase): 0 sketch, 1 probe (scan+scatter), 2 sort,
     * 3 cluster, 4 first kernel start to last kernel stop.  ms[5] is no time: the number of batches this
     * result object has had to map again without chained prefix sums so far */
    float ms[PMAP_N_TIMERS];
} pmap_counts_t;
int pmap_result_counts(const pmap_result_t *res, pmap_counts_t *out);

/* pmap_map for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops such as adaptive sampling on a running nanopore
 * flow cell, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  Same kernels and results as pmap_map in its default mode, but no
 * hipMalloc / hipFree / event creation inside the call: everything between the
 * inputs and the outputs lives in a 64 MiB array that is part of the library's
 * device image (one copy per GPU, held until the process ends), whose address
 * pmap_init(device) looks up once per device.  Hence the limits, reported as
 * PMAP_ELIMIT: a batch of at most 2^20 tile entry slots (about one million
 * bases after every read is padded to a multiple of 16), 2^17 reads, 2^20 hits
 * (778 240 when some prg's size threshold is 0, i.e. with min_cluster_size = 0:
 * the array then holds a row slot per hit), 2^20 clusters kept, an index of at
 * most 2^20 PRGs.
 * Before the call: pmap_init on this device, pmap_index_upload_now(idx),
 * pmap_reads_create / pmap_reads_update for the batch.  The call blocks until
 * the outputs are complete; *counts receives the totals (also when the outputs
 * were too small: PMAP_ELIMIT, and the counts say what would have fitted).
 * Every call on one GPU uses that GPU's one array, whichever mapping context
 * the reads belong to: a call that finds the array in use by another call
 * returns PMAP_EINVAL at once (nothing of the other call is touched); calls on
 * different GPUs do not meet. */
int pmap_map_noalloc(const pmap_index_t *idx, const pmap_reads_t *reads, const pmap_params_t *prm,
                     const pmap_dev_out_t *out, pmap_counts_t *counts);

/* Per-kernel HIP-event times (milliseconds, on the library's stream) of the
 * last pmap_map call made with profile != 0 (the events ride on the kernel
 * dispatches; each timed kernel costs a few microseconds of dispatch gap).
 * ms has PMAP_N_KERNELS entries, 0 for kernels that were not timed;
 * pmap_kernel_name(i) names entry i. */
#define PMAP_N_KERNELS 16
int pmap_result_kernel_ms(const pmap_result_t *res, float *ms);
const char *pmap_kernel_name(int i);

/* Seq::sketch of every read, concatenated in (read, position) order.  Needs a
 * result of a call with PMAP_FLAG_UNFUSED or last_stage = PMAP_STAGE_SKETCH. */
int pmap_result_minimizers(const pmap_result_t *res, uint64_t *kmer, uint32_t *read_id, uint32_t *pos,
                           uint8_t *strand);
/* MinimizerHits::hits in MinimizerHit::operator< order (src/minihit.cpp:44-66).
 * sorted=0 returns them in emission order (after add_read_hits, before sort). */
int pmap_result_hits(const pmap_result_t *res, int sorted, uint32_t *read_id, uint32_t *pos, uint32_t *prg_id,
                     uint32_t *rec, uint8_t *is_forward);
/* Clusters kept by define_clusters, ordered by their first hit; cluster c is
 * sorted hits [begin[c], begin[c]+size[c]). */
int pmap_result_clusters(const pmap_result_t *res, uint64_t *begin, uint32_t *size, uint32_t *read_id,
                         uint32_t *prg_id, uint8_t *is_forward, uint32_t *first_pos, uint32_t *last_pos,
                         uint32_t *first_rank);
void pmap_result_destroy(pmap_result_t *res);

/* ---------------------------------------------------------------- coverage */

/* KmerNode coverage, the effect of pangenome::Graph::add_hits_to_kmergraphs
 * (src/pangenome/pangraph.cpp:347-383; KmerNode::increment_covg,
 * src/kmernode.cpp:42-49) kept as two device counters per index record (a
 * MiniRecord is one kmer node of its PRG): hits with is_forward = 1 and 0.
 * pmap_coverage_add adds, for every kept cluster c of a mapped batch (the
 * order of pmap_result_clusters), weight[c] to the counter of each of its hits;
 * weight[c] = 0 for a cluster that filter_clusters removed, otherwise the
 * number of surviving clusters of the same (read, prg): the reference walks
 * Node::reads, a multiset with one entry per add_node call, and counts every
 * hit of the read on that node once per entry.  Record numbering is the
 * caller's (pmap_index_create). */
typedef struct pmap_coverage pmap_coverage_t;
int pmap_coverage_create(const pmap_index_t *idx, pmap_coverage_t **out);
int pmap_coverage_add(pmap_coverage_t *cov, const pmap_result_t *res, const uint32_t *weight);
int pmap_coverage_fetch(const pmap_coverage_t *cov, uint32_t *forward, uint32_t *reverse);
void pmap_coverage_destroy(pmap_coverage_t *cov);

/* ---------------------------------------------------------------- host stages */

/* clusterComp order (src/minihits.cpp:95-109) followed by filter_clusters
 * (src/utils.cpp:251-288) over cluster summaries.  order[j] = input index of
 * the j-th cluster in clusterComp order; alive[j] = 1 if that cluster survives. */
int pmap_filter_clusters(uint64_t n, const uint32_t *read_id, const uint32_t *first_pos, const uint32_t *size,
                         const uint32_t *prg_id, const uint32_t *first_rank, const uint8_t *is_forward,
                         const uint32_t *last_pos, uint64_t *order, uint8_t *alive);

/* FastaqHandler (src/fastaq_handler.cpp:12-71) over a plain or gzip file:
 * yields the reads of `while (!fh.eof()) fh.get_next()`.  bases/offsets/names
 * are malloc'd; free with pmap_free. */
int pmap_read_fastaq(const char *path, char **bases, uint64_t **offsets, uint32_t *n_reads, char **names,
                     uint64_t **name_offsets);
void pmap_free(void *p);
/* The same reads (no names) through the reader pmap_pangraph_from_read_file uses
 * for plain files: the file is cut into pieces of about piece_bytes at line
 * starts that look like a record header, every piece is parsed on its own, and
 * pieces whose cut turns out to lie inside a record are parsed again together
 * with the piece before.  The result never depends on piece_bytes.  (No
 * counterpart in the reference, whose reader is sequential.) */
int pmap_read_fastaq_pieces(const char *path, uint64_t piece_bytes, char **bases, uint64_t **offsets,
                            uint32_t *n_reads);

typedef struct {
    uint32_t w, k;
    int32_t max_diff;
    float e_rate;
    uint32_t min_cluster_size;  /* default 10 */
    uint32_t genome_size;       /* default 5000000 */
    int illumina;
    int clean;                  /* must be 0: the de Bruijn cleaning step (src/noise_filtering.cpp) is not part of this library */
    uint32_t max_covg;          /* default 300 */
    uint32_t batch_bases;       /* reads per device batch are cut at this many bases; 0 = default */
    uint32_t flags;             /* PMAP_PG_* */
} pmap_map_args_t;

#define PMAP_PG_NO_HITS 1u      /* do not keep Read::hits (pmap_pangraph_hits returns nothing) */
#define PMAP_PG_COVERAGE 2u     /* also accumulate the kmer node coverage on the device (pmap_pangraph_coverage) */

/* Replaces pangraph_from_read_file (src/utils.cpp:379-455) with clean=false:
 * reads the file, applies the coverage / 10M-read stops, maps every batch on
 * the GPU, runs filter_clusters and performs the add_node sequence
 * (src/pangenome/pangraph.cpp:126-140) into a host summary.  *covg is the
 * function's return value.  Inside the call, parser threads cut a plain file
 * into pieces and parse them side by side (gzip input: one sequential reader),
 * one sequencer applies the read loop's rules in file order, one mapper thread
 * per mapping context (pmap_init_devices; by default four contexts on the
 * calling thread's GPU) uploads and maps whole batches, runs filter_clusters
 * and lists the batch's add_node calls, and the calling thread joins the
 * batches in file order; the call returns when the file is done. */
int pmap_pangraph_from_read_file(const char *reads_path, const pmap_index_t *idx, const pmap_map_args_t *args,
                                 pmap_pangraph_t **out, uint32_t *covg);

typedef struct {
    uint32_t n_nodes, n_reads;
    uint64_t n_read_nodes;      /* sum over reads of len(Read::nodes) */
    uint64_t n_hits;            /* sum over reads and prgs of len(Read::hits[prg]) */
    uint64_t n_add_node_calls;
} pmap_pangraph_info_t;
int pmap_pangraph_info(const pmap_pangraph_t *pg, pmap_pangraph_info_t *info);
/* Nodes sorted by node id: prg_id and covg (pangenome::Node, number of add_node calls). */
int pmap_pangraph_nodes(const pmap_pangraph_t *pg, uint32_t *prg_id, uint32_t *covg);
/* Reads sorted by id; read i owns entries [node_off[i], node_off[i+1]) of
 * node_prg / node_orientation (pangenome::Read::nodes / node_orientations). */
int pmap_pangraph_reads(const pmap_pangraph_t *pg, uint32_t *read_id, uint64_t *node_off, uint32_t *node_prg,
                        uint8_t *node_orientation);
/* Read::hits flattened: sorted by (read id, prg id, pComp_path order
 * src/minihits.cpp:77-93). */
int pmap_pangraph_hits(const pmap_pangraph_t *pg, uint32_t *read_id, uint32_t *prg_id, uint32_t *pos,
                       uint32_t *rec, uint8_t *is_forward);
/* The add_node call sequence: (prg_id, read_id, cluster size, is_forward of first hit). */
int pmap_pangraph_calls(const pmap_pangraph_t *pg, uint32_t *prg_id, uint32_t *read_id, uint32_t *size,
                        uint8_t *is_forward);
/* The cluster handed to every add_node call (add_clusters_to_pangraph,
 * src/utils.cpp:338-352): call j of pmap_pangraph_calls owns entries
 * [hit_off[j], hit_off[j+1]) of pos / rec (hit_off has n_add_node_calls + 1
 * entries, pos / rec n_hits), in MinimizerHit::operator< order, i.e. as the
 * std::set<MinimizerHitPtr, pComp> of that cluster iterates.  Every hit of a
 * call has the call's read_id, prg_id and is_forward; rec is the index record
 * (caller's numbering) that gives prg_path and kmer_node_id.  Any pointer may
 * be NULL.  Not available with PMAP_PG_NO_HITS. */
int pmap_pangraph_call_hits(const pmap_pangraph_t *pg, uint64_t *hit_off, uint32_t *pos, uint32_t *rec);
/* With PMAP_PG_COVERAGE: the coverage every index record's kmer node has after
 * add_hits_to_kmergraphs (src/pangenome/pangraph.cpp:347-383), forward and
 * reverse hits, n_records entries each in the caller's record numbering. */
int pmap_pangraph_coverage(const pmap_pangraph_t *pg, uint32_t *forward, uint32_t *reverse);
void pmap_pangraph_destroy(pmap_pangraph_t *pg);
/* pmap_pangraph_from_read_file keeps the page-locked host buffers of its batches
 * for its next call (at most 1 GiB; PMAP_PINNED_CACHE_BYTES in the environment
 * sets another bound, 0 keeps none).  This frees the ones not in use and returns
 * their bytes.  (No counterpart in the reference, which has no device buffers.) */
uint64_t pmap_release_host_cache(void);

#ifdef __cplusplus
}
#endif
#endif
