// HIP kernels (gfx950) and device-side C ABI of libpandora_amd: the read ->
// PRG minimizer path of `pandora map` on one MI355X.
//
//   encode   nt4 (src/inthash.cpp:42-63) -> 2-bit packed reads + per-read "bad letter" flag
//   sketch   Seq::minimizer_sketch (src/seq.cpp:91-122) in its closed form: k-mer j of a
//            read with n k-mers is a minimizer iff some full window [i, i+w) within [0, n)
//            that contains j has minimum h[j] (every tie kept)
//   probe    Index::minhash lookup of add_read_hits (src/utils.cpp:174-176): one 32-byte slot (key + first record) per probe
//   emit     MinimizerHit construction (src/minihit.cpp:14-22): count / prefix sum / scatter
//   sort     MinimizerHits::sort (src/minihits.cpp:42-53, order src/minihit.cpp:44-66)
//   cluster  define_clusters (src/utils.cpp:189-249): cut flags, prefix sum, size threshold
//
// The batch is cut into tiles of up to 512 k-mer start positions, one wavefront each; sketch,
// probe and hit emission work tile by tile and keep per-tile counts, so the only device-wide
// prefix sum runs over groups of 64 tiles.  Nothing on the path waits for the host: buffer capacities are carried over from the
// previous call, every kernel guards its writes by capacity, and the counters are read back
// once at the end (a too-small buffer is grown and the batch is mapped again).
//
// All kernels are integer / byte work bounded by HBM bandwidth or VALU rate; none uses MFMA.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pmap_internal.h"

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return pmap_fail(PMAP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

// A mapping context: the streams one host thread drives one GPU with.  A process may hold several, on
// different GPUs or on the same one (pmap_init_devices); every reads / result / coverage object belongs to
// the context it was created in, and calls on it switch the calling thread to that context's GPU.
//
// The map path runs on two sketch streams and one or two tail streams.  The sketch kernels of successive
// batches alternate between the sketch streams (they do not depend on each other): the second one fills the
// compute units as the first one drains, and the 15 us between two kernels of one stream (end-of-kernel
// barrier, event, launch) are hidden behind the other stream.  Everything after the sketch kernel runs on
// a tail stream behind an event: a string of short, latency-bound kernels that now runs next to the sketch
// kernels of the following batches instead of in front of them.  One step of the reference workload:
// 0.243 ms as the sum of its kernels, 0.245 ms with one sketch stream, 0.230 ms with two.  On workloads
// with many hits per base that string is longer than a sketch kernel and bounds the step: with two tail
// streams the strings of successive batches overlap as well (+ 3 to 16 % there).  The streams of a process
// must not outnumber its hardware queues, though (GPU_MAX_HW_QUEUES, 4 unless the environment says
// otherwise: with a fifth stream on four queues the reference workload took 0.267 ms), so the second tail
// stream exists only when the environment grants at least 6 queues (PMAP_TAIL_STREAMS=1|2 overrides), and
// coverage and clearing share a tail stream.  Uploads and encodes run on their own stream, so that a batch
// can be brought in while the kernels of the batch before it run.
static const int N_SKETCH_STREAMS = 2;
static int default_tail_streams() {
    if (const char *e = getenv("PMAP_TAIL_STREAMS")) return atoi(e);
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    return q && atoi(q) >= 6 ? 2 : 1;
}
static int g_tail_streams = default_tail_streams();
struct pmap_result;
struct DevCtx {
    int device = 0;
    hipStream_t sketch[N_SKETCH_STREAMS] = {nullptr, nullptr}, tail = nullptr, more_tails[3] = {nullptr, nullptr, nullptr},
                upload = nullptr;
    int n_tails = 1;
    unsigned sketch_turn = 0, tail_turn = 0;
};
// pmap_map_noalloc, per GPU (all contexts of a GPU share it): the address of this device's copy of the scratch
// array, a result object whose buffers are views of it and of the caller's outputs, pinned staging of the
// per-prg thresholds, and what the threshold table in the array was last computed from.  `busy` is held for the
// length of a call: the array cannot serve two calls at once, whichever contexts they come from.
struct NoAlloc {
    std::mutex setup_mu;
    unsigned char *scratch = nullptr;
    pmap_result *view = nullptr;
    uint32_t *thr_stage = nullptr;
    std::atomic<bool> busy{false};
};
static NoAlloc g_noalloc[pmap_index::MAX_DEVICES];
static std::mutex g_ctx_mu;                       // guards the two lists below
static std::vector<DevCtx *> g_all_ctx;           // every context ever made (they live until the process ends)
static std::vector<DevCtx *> g_pool;              // pmap_init_devices: the contexts of pmap_pangraph_from_read_file
static thread_local DevCtx *t_ctx = nullptr;      // the calling thread's current context
static thread_local hipStream_t g_cur = nullptr;  // where KLAUNCH enqueues
// Choice of the sketch kernel where the register-window kernel applies (k <= 15, default w), for
// comparing the variants on a GPU: PMAP_SKETCH=lds keeps the LDS-window kernel, PMAP_SKETCH=np8
// takes 8 instead of 16 positions per lane (PMAP_FLAG_SKETCH_* do the same per call).
enum SketchKind { SKETCH_LDS = 0, SKETCH_DPP8 = 8, SKETCH_DPP16 = 16 };
static SketchKind g_sketch_pref = SKETCH_DPP16;
// reads with more hits than this are sorted by the long-segment kernels (PMAP_SORT_LONG overrides, for experiments)
static uint32_t g_sort_long = 4096;
// reads with more than g_sort_wave_min and at most g_sort_wave_max hits: sort_wave_kernel (one wavefront per read)
static const uint32_t SORT_WAVE_LIMIT = 2048;      // = 64 * SORT_WAVE_E of the kernel
// (PMAP_SORT_WAVE_MIN / _MAX override; by default the lower bound is 192 for sparse sketches, w >= 8, and 320 for
// dense ones, whose batches of very many short reads are bound by memory traffic rather than by instructions)
static uint32_t g_sort_wave_min = 0, g_sort_wave_max = SORT_WAVE_LIMIT;

// ---------------------------------------------------------------------------
// Scratch memory of pmap_map_noalloc, the entry point that allocates nothing: everything pmap_map keeps
// in buffers it allocates itself between its inputs and outputs (counters and the words of the chained
// prefix sums, tile counters, tile entries, hits before the sort, per-read hit bounds, cluster starts,
// per-prg thresholds) lives here instead.  Every GPU the library runs on holds its own copy of the
// array, at its own address, until the process ends.  In host code the array's name is only a handle
// for hipGetSymbolAddress: pmap_init asks for the address once per device and host code uses nothing but
// that pointer (kernel arguments, copies).
static const size_t NA_MAX_SLOTS = 1u << 20;       // tile slots (tiles x slot stride): about 1 Mbases of padded reads
static const size_t NA_MAX_HITS = 1u << 20;
static const size_t NA_MAX_CLUSTERS = 1u << 20;
static const size_t NA_MAX_READS = 1u << 17;
static const size_t NA_MAX_PRG = 1u << 20;
static const size_t NA_MAX_TILES = NA_MAX_SLOTS / 512;   // the smallest slot stride of a tile is 512
static const size_t NA_OFF_STATE = 0;                                                      // 256 KiB
static const size_t NA_OFF_GROUP = NA_OFF_STATE + (256u << 10);                            // 64 KiB
static const size_t NA_OFF_TILECNT = NA_OFF_GROUP + (64u << 10);                           // 16 B per tile
static const size_t NA_OFF_HBOUNDS = NA_OFF_TILECNT + 16 * (NA_MAX_TILES + 16);            // 12 (of 16) B per read
static const size_t NA_OFF_THR = NA_OFF_HBOUNDS + 16 * (NA_MAX_READS + 32);                // 4 B per prg
static const size_t NA_OFF_CSTART = NA_OFF_THR + 4 * (NA_MAX_PRG + 64);                    // 4 B per cluster
static const size_t NA_OFF_CHUNK = NA_OFF_CSTART + 4 * (NA_MAX_CLUSTERS + 64);             // 2 x 16 B per chunk of 128 hits
static const size_t NA_OFF_HITS = NA_OFF_CHUNK + 2 * 16 * (NA_MAX_HITS / 128 + 64);        // 5 x 4 B per hit
static const size_t NA_OFF_ENTS = NA_OFF_HITS + 5 * 4 * NA_MAX_HITS;                       // 32 B per tile slot
static const size_t NA_SCRATCH_BYTES = NA_OFF_ENTS + 32 * NA_MAX_SLOTS;
// The tile entries are dead once the hits have been emitted: the words of the long-segment sort (8 B per hit)
// and the row slots of the fused tail kernels (32 B each) take their place.
static const size_t NA_OFF_ROWS_TMP = NA_OFF_ENTS + 8 * NA_MAX_HITS;
static const size_t NA_ROWS_TMP = (NA_SCRATCH_BYTES - NA_OFF_ROWS_TMP) / 32;
static_assert(NA_SCRATCH_BYTES <= (64u << 20), "the scratch array is documented as 64 MiB");
static_assert(NA_OFF_GROUP % 256 == 0 && NA_OFF_TILECNT % 256 == 0 && NA_OFF_HBOUNDS % 256 == 0 &&
              NA_OFF_THR % 256 == 0 && NA_OFF_CSTART % 256 == 0 && NA_OFF_CHUNK % 256 == 0 && NA_OFF_HITS % 256 == 0 &&
                  NA_OFF_ENTS % 256 == 0 && NA_OFF_ROWS_TMP % 256 == 0,
              "every part of the scratch array starts on a 256-byte boundary");
static __device__ __align__(256) unsigned char pmap_scratch[64u << 20];

static int noalloc_setup(int device);

// makes `ctx` the calling thread's context (and its GPU the thread's current device)
static int bind(DevCtx *ctx) {
    if (t_ctx != ctx) {
        HIP_TRY(hipSetDevice(ctx->device));
        t_ctx = ctx;
    }
    return PMAP_OK;
}

// a new context on `device` (the calling thread's current device afterwards)
static int new_context(int device, DevCtx **out) {
    HIP_TRY(hipSetDevice(device));
    DevCtx *ctx = new DevCtx();
    ctx->device = device;
    // all at the default priority: priorities made no difference to the timing, and with two processes
    // on one GPU a low-priority sketch queue that the tail queue waits for was never scheduled
    for (auto &sk : ctx->sketch) HIP_TRY(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->tail, hipStreamNonBlocking));
    ctx->n_tails = std::max(1, std::min(g_tail_streams, 4));
    for (int i = 1; i < ctx->n_tails; ++i) HIP_TRY(hipStreamCreateWithFlags(&ctx->more_tails[i - 1], hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->upload, hipStreamNonBlocking));
    t_ctx = ctx;
    if (int rc = noalloc_setup(device)) return rc;
    std::lock_guard<std::mutex> l(g_ctx_mu);
    g_all_ctx.push_back(ctx);
    *out = ctx;
    return PMAP_OK;
}

static int device_count(int *n) {
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess || *n == 0)
        return pmap_fail(PMAP_EHIP, "no HIP device: libpandora_amd needs an AMD GPU (gfx950), there is no CPU fallback");
    return PMAP_OK;
}

extern "C" int pmap_init(int device) {
    int n = 0;
    if (int rc = device_count(&n)) return rc;
    const char *sk = getenv("PMAP_SKETCH");
    g_sketch_pref = sk && !strcmp(sk, "lds") ? SKETCH_LDS : sk && !strcmp(sk, "np8") ? SKETCH_DPP8 : SKETCH_DPP16;
    if (const char *sl = getenv("PMAP_SORT_LONG")) g_sort_long = (uint32_t) std::max(1, atoi(sl));
    if (const char *sw = getenv("PMAP_SORT_WAVE_MIN")) g_sort_wave_min = (uint32_t) std::max(1, atoi(sw));
    if (const char *sw = getenv("PMAP_SORT_WAVE_MAX")) g_sort_wave_max = (uint32_t) std::min((int) SORT_WAVE_LIMIT, std::max(0, atoi(sw)));
    if (device < 0) {
        if (t_ctx) return PMAP_OK;
        HIP_TRY(hipGetDevice(&device));
    }
    if (device >= n || device >= pmap_index::MAX_DEVICES) return pmap_fail(PMAP_EINVAL, "no such device");
    // the first context made on that device, or a new one
    DevCtx *ctx = nullptr;
    {
        std::lock_guard<std::mutex> l(g_ctx_mu);
        for (DevCtx *c : g_all_ctx)
            if (c->device == device) { ctx = c; break; }
    }
    if (ctx) return bind(ctx);
    return new_context(device, &ctx);
}

extern "C" int pmap_init_devices(const int *devices, int n_devices) {
    int n = 0;
    if (int rc = device_count(&n)) return rc;
    if (n_devices < 0 || (n_devices > 0 && !devices)) return pmap_fail(PMAP_EINVAL, "null argument");
    if (n_devices > 64) return pmap_fail(PMAP_ELIMIT, "more than 64 mapping contexts");
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= n || devices[i] >= pmap_index::MAX_DEVICES)
            return pmap_fail(PMAP_EINVAL, "no such device");
    DevCtx *before = t_ctx;
    std::vector<DevCtx *> pool;
    for (int i = 0; i < n_devices; ++i) {
        // an entry reuses a context that exists on its device unless an earlier entry took it already
        DevCtx *ctx = nullptr;
        {
            std::lock_guard<std::mutex> l(g_ctx_mu);
            for (DevCtx *c : g_all_ctx)
                if (c->device == devices[i] && std::find(pool.begin(), pool.end(), c) == pool.end()) { ctx = c; break; }
        }
        if (!ctx)
            if (int rc = new_context(devices[i], &ctx)) return rc;
        pool.push_back(ctx);
    }
    {
        std::lock_guard<std::mutex> l(g_ctx_mu);
        g_pool = pool;
    }
    // the calling thread stays where it was (or lands on the first context of the list)
    t_ctx = nullptr;
    if (before) return bind(before);
    if (!pool.empty()) return bind(pool[0]);
    return PMAP_OK;
}

static int ensure_init();

// The contexts pmap_pangraph_from_read_file works with: those of pmap_init_devices, or, by default, four on
// the calling thread's GPU (made on first use and kept).
int pmap_contexts_for_file(std::vector<void *> *out) {
    int rc = ensure_init();
    if (rc) return rc;
    out->clear();
    {
        std::lock_guard<std::mutex> l(g_ctx_mu);
        for (DevCtx *c : g_pool) out->push_back(c);
    }
    if (!out->empty()) return PMAP_OK;
    // no list from pmap_init_devices: FILE_CONTEXTS contexts on the calling thread's GPU, its own among them (a
    // mapper thread spends more time on the host side of its batch than the GPU spends mapping it: per batch of
    // 32 Mbases 0.4 ms upload, 0.26 ms map, 0.7 ms for the kept clusters and filter_clusters, and 1.2 ms more for
    // Read::hits when they are kept.  On 16 cores, without hits: 3, 4, 5 contexts 50 Gbases/s of FASTQ, the
    // parser threads being the limit, 6 contexts 47.5; with hits: 3 contexts 28, 4 contexts 34, 5 and 6 contexts 33)
    static const size_t FILE_CONTEXTS = 4;
    DevCtx *mine = t_ctx;
    out->push_back(mine);
    {
        std::lock_guard<std::mutex> l(g_ctx_mu);
        for (DevCtx *c : g_all_ctx)
            if (c->device == mine->device && c != mine && out->size() < FILE_CONTEXTS) out->push_back(c);
    }
    while (out->size() < FILE_CONTEXTS) {
        DevCtx *more = nullptr;
        rc = new_context(mine->device, &more);
        t_ctx = nullptr;
        if (int rb = bind(mine)) return rb;
        if (rc) return rc;
        out->push_back(more);
    }
    return PMAP_OK;
}

int pmap_bind_context(void *ctx) {
    if (!ctx) return pmap_fail(PMAP_EINVAL, "null argument");
    t_ctx = nullptr;      // a new thread has no device set yet
    return bind((DevCtx *) ctx);
}

void *pmap_pinned_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;     // any GPU may copy from it
    return p;
}

void pmap_pinned_free(void *p) {
    if (p) (void) hipHostFree(p);
}

static int ensure_init() {
    if (t_ctx) return PMAP_OK;
    return pmap_init(-1);
}

// ---------------------------------------------------------------------------
// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool owned = true;      // false: a view of memory that belongs to someone else (never grown, never freed)
    int ensure(size_t bytes) {
        if (bytes <= cap) return PMAP_OK;
        if (!owned) return pmap_fail(PMAP_ELIMIT, "a fixed-size device buffer is too small for this batch");
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return PMAP_OK;
    }
    void release() {
        if (p && owned) (void) hipFree(p);
        p = nullptr;
        cap = 0;
    }
    void view(void *ptr, size_t bytes) {
        p = ptr;
        cap = bytes;
        owned = false;
    }
    template <typename T>
    T *as() const { return (T *) p; }
};

// counters kept in device memory for the whole pipeline, read back once per call
struct Counters {
    unsigned long long n_hits;    // hits of the batch
    unsigned long long n_mini;    // minimizers of the batch
    unsigned long long n_ent;     // entries handed to the scatter (all minimizers, or only those with hits)
    unsigned long long n_cl;      // clusters before the size threshold
    unsigned long long n_kept;    // clusters kept
    uint32_t fail;                // a chained prefix sum gave up waiting
    uint32_t has_long;            // some read has more hits than the rank-counting sort takes: the long-segment kernels have work
    uint32_t dense_marks;         // sum of tile_cnt.w: tiles with more minimizers than the sketch kernel's list holds (1 each: left
                                  // out, the overflow kernel has work; 2 each: done by it)
    uint32_t n_wave;              // reads on the list of the wavefront sort
    uint32_t has_wave;            // some read has a hit count the wavefront sort is for
    uint32_t n_long_list;         // reads on the list of tail_long_kernel
};
static_assert(sizeof(Counters) == 64, "Counters is the head of the state block");

#include "pmap_kernels.hip.h"

// the kernel that sketches with (w, k), the output positions of one of its tiles and the slot
// stride of a tile's outputs
static SketchKind sketch_kind(uint32_t w, uint32_t k, uint32_t flags, uint32_t *out_pos, uint32_t *stride) {
    SketchKind kind = g_sketch_pref;
    if (flags & PMAP_FLAG_SKETCH_LDS) kind = SKETCH_LDS;
    else if (flags & PMAP_FLAG_SKETCH_NP8) kind = SKETCH_DPP8;
    // the register-window kernel serves k <= 15 with w <= 16 (8 positions per lane: the default w only)
    if (k > 15 || w < 1 || w > (uint32_t) DPP_MAX_W) kind = SKETCH_LDS;
    else if (kind == SKETCH_DPP8 && w != 14) kind = SKETCH_DPP16;
    if (kind == SKETCH_DPP16) {
        *out_pos = DppTile<16>::OUT;
        *stride = DppTile<16>::POS;
    } else if (kind == SKETCH_DPP8) {
        *out_pos = DppTile<8>::OUT;
        *stride = DppTile<8>::POS;
    } else {
        const uint32_t a = w - 1, edge = (a >> 2) + ((a & 3) ? 1 : 0);
        *out_pos = 4 * (WAVE_Q - 2 * edge);
        *stride = WAVE_POS;
    }
    return kind;
}


// ---------------------------------------------------------------------------
// host objects

struct pmap_reads {
    DevCtx *ctx = nullptr;
    uint32_t n_reads = 0, first_read_id = 0;
    uint32_t max_len = 0;      // longest read of the batch
    uint64_t n_bases = 0;
    uint32_t n_words = 0;      // packed words, each read padded to a multiple of 16 bases
    uint32_t n_tiles = 0;
    DevBuf ascii, boff, woff, rlen, rflag;
    DevBuf wblk;     // per block of ENC_BLOCK packed words: the read that owns the block's first word
    bool packed = false;            // `ascii` holds words the host packed, `rflag` the host's flags
    std::vector<uint32_t> h_rflag;
    DevBuf words;    // uint2 per packed word: {16 bases, read index}
    DevBuf rmeta;    // uint2 per read: {first position in the padded stream, length or 0 if not sketchable}
    std::vector<uint32_t> h_woff, h_rlen, h_wblk;   // host staging of the per-read tables (kept: the copies are asynchronous)
    std::vector<uint64_t> h_boff;
    uint64_t device_bytes() const {
        return ascii.cap + boff.cap + woff.cap + rlen.cap + rflag.cap + wblk.cap + words.cap + rmeta.cap;
    }
};

enum { KT_SKETCH = 0, KT_PROBE, KT_SCATTER, KT_SORT, KT_CLUSTER_CUT, KT_CLUSTER_KEEP, KT_TAIL, KT_ROWS, KT_COUNT };
static const char *kernel_names[PMAP_N_KERNELS] = {
    "sketch", "probe", "scatter", "sort_rank", "cluster_cut", "cluster_keep", "tail", "rows", "", "", "", "", "", "", "", ""};

struct pmap_result {
    DevCtx *ctx = nullptr;
    pmap_counts_t counts;
    uint32_t first_read_id = 0;
    uint32_t n_tiles = 0;
    uint32_t tile_stride = 0;          // slot stride of a tile's minimizers / entries (set with n_tiles)
    bool have_minimizers = false;     // the last run kept every minimizer (unfused)
    float kernel_ms[PMAP_N_KERNELS];
    uint32_t hit_cap = 0, cl_cap = 0;
    // state block, zeroed before every run: Counters, then the 8-byte words of the three chained prefix sums
    DevBuf state;
    uint32_t n_ablk = 0, n_bblk = 0;   // blocks of the cluster cut and cluster keep kernels
    DevBuf group_sum;                  // uint4 per group of GROUP_TILES tiles: minimizers, entries, hits
    // per tile
    DevBuf tile_cnt, bsum, scratch_total;
    // tile-slotted entries
    DevBuf mkmer, ents;
    // hits before / after sort
    DevBuf h_read, h_key, h_pos, h_rank, h_rec;
    DevBuf hbounds;   // per read: first hit, then ~(one past the last hit); all-ones before every run; behind them the list
                      // of reads for the wavefront sort
    DevBuf s_read, s_key, s_pos, s_rank, s_rec;
    DevBuf long_words;   // sort of reads with very many hits: the hit array's blocks, each sorted (8 bytes per hit)
    // clusters
    DevBuf cstart, k_rows;
    DevBuf chunk_cnt;   // fused tail: uint4 per chunk of TAIL_C hits {clusters, kept clusters, first row slot, 0}
    DevBuf rows_tmp;    // fused tail: rows of the kept clusters in the slots their chunks chose
    DevBuf thr;
    uint32_t thr_minval = 0;          // smallest threshold of any prg
    std::vector<uint32_t> thr_host;
    uint64_t thr_serial = 0;          // pmap_index::serial the table was built from (0 = none yet)
    float thr_fraction = 0;
    uint32_t thr_min = 0, thr_exp = 0;
    hipEvent_t kev[2 * PMAP_N_KERNELS];
    hipEvent_t sketched = nullptr;    // the sketch kernel of the last enqueued run has finished
    hipEvent_t done = nullptr;        // after the counters of the last enqueued run have reached hc
    bool events = false;
    // a run that was enqueued and not yet waited for
    bool pending = false;
    Counters *hc = nullptr;           // pinned host copy of the device counters
    const pmap_index *p_idx = nullptr;
    const pmap_reads *p_rd = nullptr;
    pmap_params_t p_prm;
    bool p_fused = false;
    uint32_t p_last = 0;
    bool long_enabled = false;        // launch the long-segment sort kernels
    bool long_launched = false;       // ... and the last run did
    bool staged_tail = false;         // the last run took the separate sort and cluster kernels
    bool few_hits = false;            // the last batch had so few hits per read that those kernels are the quicker ones
    bool wave_enabled = true;         // launch the wavefront sort kernels (off after a batch without a read for them)
    bool dense_enabled = false;       // launch the sketch kernel for tiles with more minimizers than the usual one's list holds
    bool dense_launched = false;
    bool serial_scans = false;        // this run replaces the chained prefix sums by one workgroup per kernel
    uint32_t n_serial_reruns = 0;     // batches that needed that so far
    uint32_t kt_recorded = 0;
    std::vector<DevBuf *> all() {
        return {&state, &group_sum, &tile_cnt, &bsum, &scratch_total, &mkmer, &ents, &h_read, &h_key, &h_pos, &h_rank, &h_rec,
                &hbounds, &s_read, &s_key, &s_pos, &s_rank, &s_rec, &long_words, &cstart, &k_rows, &chunk_cnt, &rows_tmp, &thr};
    }
};

static int new_result(pmap_result **out);

// ---------------------------------------------------------------------------
// index upload

// The index's copy on the calling thread's GPU, made on first use.  Every caller comes through here (and so
// through upload_mu) before it reads the device pointers: several mapper threads share one index.
int pmap_index_upload(pmap_index *idx) {
    int rc = ensure_init();
    if (rc) return rc;
    pmap_index::OnDevice &od = idx->dev[t_ctx->device];
    std::lock_guard<std::mutex> l(idx->upload_mu);
    if (od.slots) return PMAP_OK;
    const size_t sb = idx->slots.size() * sizeof(PmapSlot), rb = (idx->recs.size() + 1) * sizeof(PmapRec);
    const size_t fb = idx->filter.size() * sizeof(uint32_t);
    PmapSlot *slots = nullptr;
    auto upload = [&]() -> int {
        HIP_TRY(hipMalloc((void **) &od.recs, rb));
        if (!idx->recs.empty())
            HIP_TRY(hipMemcpy(od.recs, idx->recs.data(), idx->recs.size() * sizeof(PmapRec), hipMemcpyHostToDevice));
        if (fb) {
            HIP_TRY(hipMalloc((void **) &od.filter, fb));
            HIP_TRY(hipMemcpy(od.filter, idx->filter.data(), fb, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMalloc((void **) &slots, sb));
        HIP_TRY(hipMemcpy(slots, idx->slots.data(), sb, hipMemcpyHostToDevice));
        return PMAP_OK;
    };
    if ((rc = upload())) {
        // nothing of a copy that did not complete stays behind: the next call starts over
        if (od.recs) (void) hipFree(od.recs);
        if (od.filter) (void) hipFree(od.filter);
        if (slots) (void) hipFree(slots);
        od.recs = nullptr;
        od.filter = nullptr;
        return rc;
    }
    od.slots = slots;       // last: a set `slots` says the whole copy is there
    idx->device_bytes = sb + rb + fb;
    return PMAP_OK;
}

// (for callers that must not allocate)
static bool index_is_on_device(const pmap_index *idx, int device) {
    std::lock_guard<std::mutex> l(const_cast<pmap_index *>(idx)->upload_mu);
    return idx->dev[device].slots != nullptr;
}

extern "C" int pmap_index_upload_now(pmap_index_t *idx) {
    if (!idx) return pmap_fail(PMAP_EINVAL, "null argument");
    return pmap_index_upload(idx);
}

void pmap_index_release_device(pmap_index *idx) {
    int before = -1;
    (void) hipGetDevice(&before);
    for (int d = 0; d < pmap_index::MAX_DEVICES; ++d) {
        pmap_index::OnDevice &od = idx->dev[d];
        if (!od.slots && !od.recs && !od.filter) continue;
        (void) hipSetDevice(d);
        if (od.slots) (void) hipFree(od.slots);
        if (od.recs) (void) hipFree(od.recs);
        if (od.filter) (void) hipFree(od.filter);
        od = pmap_index::OnDevice();
    }
    if (before >= 0) (void) hipSetDevice(before);
}

// ---------------------------------------------------------------------------
// reads

static int launch_encode(pmap_reads *rd) {
    hipStream_t g_upload = rd->ctx->upload;
    // (a packed batch brings its bad-letter flags with it)
    if (!rd->packed) HIP_TRY(hipMemsetAsync(rd->rflag.p, 0, sizeof(uint32_t) * (rd->n_reads + 1), g_upload));
    if (rd->n_words) {
        const uint32_t n_blk = (rd->n_words + ENC_BLOCK - 1) / ENC_BLOCK;
        if (rd->packed)
            hipLaunchKernelGGL(encode_kernel<true>, dim3((n_blk + 3) / 4), dim3(256), 0, g_upload,
                               rd->ascii.as<uint32_t>(), rd->boff.as<uint64_t>(), rd->woff.as<uint32_t>(),
                               rd->rlen.as<uint32_t>(), rd->wblk.as<uint32_t>(), rd->n_reads, rd->n_words,
                               rd->words.as<uint2>(), rd->rflag.as<uint32_t>());
        else
            hipLaunchKernelGGL(encode_kernel<false>, dim3((n_blk + 3) / 4), dim3(256), 0, g_upload,
                               rd->ascii.as<uint32_t>(), rd->boff.as<uint64_t>(), rd->woff.as<uint32_t>(),
                               rd->rlen.as<uint32_t>(), rd->wblk.as<uint32_t>(), rd->n_reads, rd->n_words,
                               rd->words.as<uint2>(), rd->rflag.as<uint32_t>());
        HIP_TRY(hipGetLastError());
    }
    if (rd->n_reads) {
        hipLaunchKernelGGL(read_meta_kernel, dim3((rd->n_reads + 255) / 256), dim3(256), 0, g_upload,
                           rd->woff.as<uint32_t>(), rd->rlen.as<uint32_t>(), rd->rflag.as<uint32_t>(), rd->n_reads,
                           rd->rmeta.as<uint2>());
        HIP_TRY(hipGetLastError());
    }
    return PMAP_OK;
}

// fills a reads object (new or reused: its buffers only grow) with a batch and waits for the upload
// packed != nullptr: the batch comes as 2-bit words from the host (`bases` is not looked at), acgt_only[r] = 0
// marks a read with a letter outside ACGT
static int reads_fill(pmap_reads *rd, const char *bases, const uint64_t *offsets, uint32_t n_reads,
                      uint32_t first_read_id, const uint32_t *packed = nullptr, const uint8_t *acgt_only = nullptr) {
    int rc;
    if (!offsets || (n_reads && !bases && !packed && offsets[n_reads] != offsets[0]))
        return pmap_fail(PMAP_EINVAL, "null argument");
    if (packed && n_reads && !acgt_only) return pmap_fail(PMAP_EINVAL, "null argument");
    if ((uint64_t) first_read_id + n_reads > 0xFFFFFFFEull)
        return pmap_fail(PMAP_ELIMIT, "read ids exceed 32 bits (src/minihit.cpp:18-19)");
    if (int rcb = bind(rd->ctx)) return rcb;
    hipStream_t g_upload = rd->ctx->upload;
    std::vector<uint32_t> &woff = rd->h_woff, &rlen = rd->h_rlen;
    std::vector<uint64_t> &boff = rd->h_boff;
    std::vector<uint32_t> &wblk = rd->h_wblk;
    wblk.clear();
    woff.assign(n_reads + 1, 0);
    rlen.assign(n_reads + 1, 0);
    boff.assign(n_reads + 1, 0);
    uint64_t words = 0;
    uint32_t max_len = 0;
    const uint64_t base0 = offsets[0];
    for (uint32_t r = 0; r < n_reads; ++r) {
        if (offsets[r + 1] < offsets[r]) return pmap_fail(PMAP_EINVAL, "read offsets not monotone");
        uint64_t len = offsets[r + 1] - offsets[r];
        if (len >= 0x7FFFFFFEull) return pmap_fail(PMAP_ELIMIT, "a read is longer than 2^31-3 bases");
        woff[r] = (uint32_t) words;
        rlen[r] = (uint32_t) len;
        max_len = std::max(max_len, (uint32_t) len);
        boff[r] = offsets[r] - base0;
        // the blocks of the encode kernel whose first word is one of this read's
        const uint64_t after = words + (len + 15) / 16;
        for (uint64_t b = (words + ENC_BLOCK - 1) / ENC_BLOCK; b * ENC_BLOCK < after; ++b) wblk.push_back(r);
        words = after;
        if (words * 16 >= 0xFFFFF000ull)
            return pmap_fail(PMAP_ELIMIT, "batch exceeds 2^32 padded bases: split it into several batches");
    }
    woff[n_reads] = (uint32_t) words;
    boff[n_reads] = offsets[n_reads] - base0;
    const uint64_t n_bases = offsets[n_reads] - base0;

    rd->n_reads = n_reads;
    rd->first_read_id = first_read_id;
    rd->n_bases = n_bases;
    rd->n_words = (uint32_t) words;
    rd->max_len = max_len;
    rd->n_tiles = 0;   // the tiling depends on w: pmap_map computes it
    rd->packed = packed != nullptr;
    const size_t ascii_bytes = packed ? 4 * (size_t) words + 64 : ((n_bases + 3) & ~3ull) + 64;
    const size_t word_slots = (size_t) words + 16;
    if ((rc = rd->ascii.ensure(ascii_bytes)) || (rc = rd->boff.ensure(sizeof(uint64_t) * (n_reads + 1))) ||
        (rc = rd->woff.ensure(sizeof(uint32_t) * (n_reads + 1))) ||
        (rc = rd->rlen.ensure(sizeof(uint32_t) * (n_reads + 1))) ||
        (rc = rd->rflag.ensure(sizeof(uint32_t) * (n_reads + 1))) ||
        (rc = rd->wblk.ensure(sizeof(uint32_t) * (wblk.size() + 1))) ||
        (rc = rd->words.ensure(sizeof(uint2) * word_slots)) ||
        (rc = rd->rmeta.ensure(sizeof(uint2) * (n_reads + 1))))
        return rc;
    hipError_t e = hipSuccess;
    if (!packed) e = hipMemsetAsync((char *) rd->ascii.p + ((n_bases) & ~3ull), 0, ascii_bytes - (n_bases & ~3ull), g_upload);
    // the words after the last one are read by the sketch kernel of the last tile: keep them empty
    if (e == hipSuccess) e = hipMemsetAsync(rd->words.as<uint2>() + words, 0, sizeof(uint2) * 16, g_upload);
    if (packed) {
        rd->h_rflag.resize(n_reads + 1);
        for (uint32_t r = 0; r < n_reads; ++r) rd->h_rflag[r] = acgt_only[r] ? 0u : 1u;
        rd->h_rflag[n_reads] = 0;
        if (e == hipSuccess && words) e = hipMemcpyAsync(rd->ascii.p, packed, 4 * (size_t) words, hipMemcpyHostToDevice, g_upload);
        if (e == hipSuccess) e = hipMemcpyAsync(rd->rflag.p, rd->h_rflag.data(), sizeof(uint32_t) * (n_reads + 1), hipMemcpyHostToDevice, g_upload);
    } else if (e == hipSuccess && n_bases) {
        e = hipMemcpyAsync(rd->ascii.p, bases + base0, n_bases, hipMemcpyHostToDevice, g_upload);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rd->boff.p, boff.data(), sizeof(uint64_t) * (n_reads + 1), hipMemcpyHostToDevice, g_upload);
    if (e == hipSuccess) e = hipMemcpyAsync(rd->woff.p, woff.data(), sizeof(uint32_t) * (n_reads + 1), hipMemcpyHostToDevice, g_upload);
    if (e == hipSuccess) e = hipMemcpyAsync(rd->rlen.p, rlen.data(), sizeof(uint32_t) * (n_reads + 1), hipMemcpyHostToDevice, g_upload);
    if (e == hipSuccess && !wblk.empty())
        e = hipMemcpyAsync(rd->wblk.p, wblk.data(), sizeof(uint32_t) * wblk.size(), hipMemcpyHostToDevice, g_upload);
    if (e != hipSuccess) return pmap_fail(PMAP_EHIP, std::string("read upload: ") + hipGetErrorString(e));
    if ((rc = launch_encode(rd))) return rc;
    e = hipStreamSynchronize(g_upload);
    if (e != hipSuccess) return pmap_fail(PMAP_EHIP, std::string("read encode: ") + hipGetErrorString(e));
    return PMAP_OK;
}

extern "C" int pmap_reads_create(const char *bases, const uint64_t *offsets, uint32_t n_reads,
                                 uint32_t first_read_id, pmap_reads_t **out) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!out) return pmap_fail(PMAP_EINVAL, "null argument");
    pmap_reads *rd = new pmap_reads();
    rd->ctx = t_ctx;
    if ((rc = reads_fill(rd, bases, offsets, n_reads, first_read_id))) {
        pmap_reads_destroy(rd);
        return rc;
    }
    *out = rd;
    return PMAP_OK;
}

extern "C" int pmap_reads_update(pmap_reads_t *rd, const char *bases, const uint64_t *offsets, uint32_t n_reads,
                                 uint32_t first_read_id) {
    if (!rd) return pmap_fail(PMAP_EINVAL, "null argument");
    return reads_fill(rd, bases, offsets, n_reads, first_read_id);
}

extern "C" int pmap_reads_create_packed(const uint32_t *packed, const uint64_t *offsets, const uint8_t *acgt_only,
                                        uint32_t n_reads, uint32_t first_read_id, pmap_reads_t **out) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!out || !packed) return pmap_fail(PMAP_EINVAL, "null argument");
    pmap_reads *rd = new pmap_reads();
    rd->ctx = t_ctx;
    if ((rc = reads_fill(rd, nullptr, offsets, n_reads, first_read_id, packed, acgt_only))) {
        pmap_reads_destroy(rd);
        return rc;
    }
    *out = rd;
    return PMAP_OK;
}

extern "C" int pmap_reads_update_packed(pmap_reads_t *rd, const uint32_t *packed, const uint64_t *offsets,
                                        const uint8_t *acgt_only, uint32_t n_reads, uint32_t first_read_id) {
    if (!rd || !packed) return pmap_fail(PMAP_EINVAL, "null argument");
    return reads_fill(rd, nullptr, offsets, n_reads, first_read_id, packed, acgt_only);
}

extern "C" int pmap_reads_repack(pmap_reads_t *rd) {
    if (!rd) return pmap_fail(PMAP_EINVAL, "null argument");
    int rc = bind(rd->ctx);
    if (rc || (rc = launch_encode(rd))) return rc;
    HIP_TRY(hipStreamSynchronize(rd->ctx->upload));
    return PMAP_OK;
}

extern "C" int pmap_reads_info(const pmap_reads_t *rd, uint32_t *n_reads, uint64_t *n_bases, uint64_t *device_bytes) {
    if (!rd) return pmap_fail(PMAP_EINVAL, "null argument");
    if (n_reads) *n_reads = rd->n_reads;
    if (n_bases) *n_bases = rd->n_bases;
    if (device_bytes) *device_bytes = rd->device_bytes();
    return PMAP_OK;
}

extern "C" void pmap_reads_destroy(pmap_reads_t *rd) {
    if (!rd) return;
    if (rd->ctx) (void) bind(rd->ctx);
    rd->ascii.release(); rd->boff.release(); rd->woff.release(); rd->rlen.release(); rd->rflag.release(); rd->wblk.release();
    rd->words.release(); rd->rmeta.release();
    delete rd;
}

// ---------------------------------------------------------------------------
// map

// Per-kernel timing: the start / stop events ride on the kernel dispatches themselves
// (hipExtLaunchKernelGGL), so timing does not put extra packets between the kernels.
struct KTimer {
    pmap_result *res;
    bool on;
    uint32_t mask;           // kernels (bit per KT_*) that get events
    uint32_t recorded = 0;
    int cur = -1;
    bool first = false;
    void begin(int k) { if (on && (mask >> k & 1u)) { cur = k; first = true; } }
    void end(int k) {
        if (on && cur == k) {
            if (!first) recorded |= 1u << k;   // at least one kernel was launched in the group
            cur = -1;
        }
    }
    // events for the next launch: the group's first launch carries the start event, every launch the stop event
    void events(hipEvent_t *s, hipEvent_t *e);
};
static thread_local KTimer *g_kt = nullptr;

#define KLAUNCH_B(kernel, grid, block, ...)                                                                 \
    do {                                                                                                    \
        hipEvent_t s_ = nullptr, e_ = nullptr;                                                              \
        if (g_kt) g_kt->events(&s_, &e_);                                                                   \
        if (s_ || e_) hipExtLaunchKernelGGL(kernel, grid, dim3(block), 0, g_cur, s_, e_, 0, __VA_ARGS__);    \
        else kernel<<<grid, dim3(block), 0, g_cur>>>(__VA_ARGS__);                                          \
    } while (0)
// (not written in terms of KLAUNCH_B: a second expansion would split template arguments at their commas)
// LDS = dynamic LDS bytes on top of the kernel's own
#define KLAUNCH_L(kernel, grid, LDS, ...)                                                                 \
    do {                                                                                                  \
        hipEvent_t s_ = nullptr, e_ = nullptr;                                                            \
        if (g_kt) g_kt->events(&s_, &e_);                                                                 \
        if (s_ || e_) hipExtLaunchKernelGGL(kernel, grid, dim3(256), LDS, g_cur, s_, e_, 0, __VA_ARGS__);  \
        else kernel<<<grid, dim3(256), LDS, g_cur>>>(__VA_ARGS__);                                        \
    } while (0)
#define KLAUNCH(kernel, grid, ...)                                                                        \
    do {                                                                                                  \
        hipEvent_t s_ = nullptr, e_ = nullptr;                                                            \
        if (g_kt) g_kt->events(&s_, &e_);                                                                 \
        if (s_ || e_) hipExtLaunchKernelGGL(kernel, grid, dim3(256), 0, g_cur, s_, e_, 0, __VA_ARGS__);    \
        else kernel<<<grid, dim3(256), 0, g_cur>>>(__VA_ARGS__);                                          \
    } while (0)

void KTimer::events(hipEvent_t *s, hipEvent_t *e) {
    *s = *e = nullptr;
    if (!on || cur < 0) return;
    if (first) { *s = res->kev[2 * cur]; first = false; }
    *e = res->kev[2 * cur + 1];
}

// exclusive scan (of values or popcounts) of n uint32 into out; the total stays on the device.
// out may be null (total only).
template <bool POPC>
static int device_scan(pmap_result *res, const uint32_t *in, uint32_t *out, uint64_t n, unsigned long long *d_total) {
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), g_cur));
        return PMAP_OK;
    }
    const uint32_t nblk = (uint32_t) ((n + SCAN_TILE - 1) / SCAN_TILE);
    int rc;
    if ((rc = res->bsum.ensure(sizeof(uint32_t) * nblk))) return rc;
    KLAUNCH(HIP_KERNEL_NAME(scan_reduce_kernel<POPC>), dim3(nblk), in, n,
                       res->bsum.as<uint32_t>());
    KLAUNCH(scan_spine_kernel, dim3(1), res->bsum.as<uint32_t>(), nblk, d_total);
    if (out)
        KLAUNCH(HIP_KERNEL_NAME(scan_apply_kernel<POPC>), dim3(nblk), in, n,
                           res->bsum.as<uint32_t>(), out);
    HIP_TRY(hipGetLastError());
    return PMAP_OK;
}

// v rounded up to a multiple of m, clamped to the largest multiple of m that fits 32 bits
static uint32_t round_up(uint64_t v, uint32_t m) {
    const uint64_t r = (v + m - 1) / m * m, top = 0xFFFFFFFFull / m * m;
    return (uint32_t) (r < top ? r : top);
}

// enqueues the whole pipeline once with the current capacities; the counters arrive in res->hc
// when res->done has happened
// extra dynamic LDS per workgroup of the register-window sketch kernels: lowers how many of them share a
// compute unit (6 with none; for experiments)
// wavefronts of tail_chunk_kernel (each takes chunks in turn; PMAP_TAIL_GRID, for experiments): by default a
// quarter of the chunks the hit buffers have room for, at least 4096 (4 per SIMD of a GPU of 256 compute units: small
// batches gain 3 % from a grid that is resident at once) and at most 32768
// Batches of many reads with few hits each (150-base reads under a sparse sketch: 8 hits per read) go through the
// separate sort and cluster kernels: a chunk of the fused tail kernel then holds 15 reads, its wavefront's chain of
// dependent loads is as long as for two reads, and the rank counting of the separate sort kernel costs next to
// nothing (measured on 200 000 reads of 150 bases, w = 14: 245 against 190 Gbases/s).  A result object decides by
// its previous batch.  PMAP_TAIL_AUTO: hits per read below which it does so (default 16; 0: never).
static unsigned g_tail_auto = getenv("PMAP_TAIL_AUTO") ? (unsigned) std::max(0, atoi(getenv("PMAP_TAIL_AUTO"))) : 16u;
static unsigned g_tail_grid = getenv("PMAP_TAIL_GRID") ? (unsigned) std::max(64, atoi(getenv("PMAP_TAIL_GRID"))) : 0u;
static unsigned g_sketch_pad = getenv("PMAP_SKETCH_LDS_PAD") ? (unsigned) std::max(0, atoi(getenv("PMAP_SKETCH_LDS_PAD"))) : 0;
static int run_pipeline(pmap_result *res) {
    int rc;
    const pmap_index *idx = res->p_idx;
    const pmap_reads *rd = res->p_rd;
    const pmap_params_t *prm = &res->p_prm;
    const bool fused = res->p_fused;
    const uint32_t last = res->p_last;
    KTimer kt{res, prm->profile >= 1, prm->profile == 2 ? 1u << KT_SKETCH : ~0u};
    DevCtx *ctx = res->ctx;
    if ((rc = bind(ctx))) return rc;
    const pmap_index::OnDevice &od = idx->dev[ctx->device];
    // (the kernels after the sketch kernel of successive batches do not depend on each other either)
    const unsigned turn = ctx->tail_turn++ % (unsigned) ctx->n_tails;
    hipStream_t g_tail = turn ? ctx->more_tails[turn - 1] : ctx->tail;
    hipStream_t sketch_stream = ctx->sketch[ctx->sketch_turn++ % N_SKETCH_STREAMS];
    g_cur = sketch_stream;
    const uint32_t k = prm->k, w = prm->w;
    const bool k32 = k <= 15;
    const uint32_t n_tiles = res->n_tiles;
    // state block: counters + one 8-byte word per block of each chained prefix sum
    res->n_ablk = res->hit_cap / CUT_BLOCK;
    res->n_bblk = (res->cl_cap + KEEP_BLOCK - 1) / KEEP_BLOCK;        // (the kernel guards by the capacity itself)
    const uint32_t n_chunks_cap = res->hit_cap / TAIL_C;                        // chunks of the fused tail kernels
    const uint32_t n_cgrp = (n_chunks_cap + TAIL_GROUP - 1) / TAIL_GROUP + 1;   // their groups: two 8-byte sums each
    const size_t state_bytes =
        (sizeof(Counters) + sizeof(unsigned long long) * ((size_t) res->n_ablk + res->n_bblk + 2 + 2 * (size_t) n_cgrp) + 15) &
        ~(size_t) 15;
    if ((rc = res->state.ensure(state_bytes))) return rc;
    Counters *dc = res->state.as<Counters>();
    unsigned long long *gran_a = (unsigned long long *) (dc + 1), *gran_b = gran_a + res->n_ablk;
    unsigned long long *cg_sum = gran_b + res->n_bblk + 2;
    const uint32_t n_groups = (uint32_t) (((size_t) n_tiles + GROUP_TILES - 1) / GROUP_TILES);
    if ((rc = res->group_sum.ensure(sizeof(uint4) * ((size_t) n_groups + 1)))) return rc;
    if (n_tiles == 0) {           // empty batch: nothing to launch
        memset(res->hc, 0, sizeof(Counters));
        res->kt_recorded = 0;
        HIP_TRY(hipEventRecord(res->done, g_tail));
        return PMAP_OK;
    }
    // the sketch kernel clears the state block and the read bounds for the kernels after it
    SketchAux aux;
    aux.state = res->state.as<unsigned long long>();
    aux.n_state_words = (uint32_t) (state_bytes / 8);
    aux.hbounds = res->hbounds.as<uint32_t>();
    aux.n_hb = 2 * (rd->n_reads + 1);
    // the timer is registered for the launches of this call only, whichever way the function is left
    struct TimerScope {
        explicit TimerScope(KTimer *t) { g_kt = t; }
        ~TimerScope() { g_kt = nullptr; }
    } timer_scope(&kt);

    // ---- sketch (+ probe when fused)
    if (n_tiles) {
        res->dense_launched = res->dense_enabled;
        kt.begin(KT_SKETCH);
#define SKETCH_ARGS                                                                                                   \
    rd->words.as<uint2>(), rd->rmeta.as<uint2>(), rd->n_words, n_tiles, w, k, od.slots, idx->log2cap,                 \
        od.filter, idx->filter_bits, res->mkmer.as<uint64_t>(), res->ents.as<uint4>(), res->tile_cnt.as<uint4>(), \
        aux
#define SKETCH_ARGS_NOAUX                                                                                             \
    rd->words.as<uint2>(), rd->rmeta.as<uint2>(), rd->n_words, n_tiles, w, k, od.slots, idx->log2cap,                 \
        od.filter, idx->filter_bits, res->mkmer.as<uint64_t>(), res->ents.as<uint4>(), res->tile_cnt.as<uint4>()
#define SKETCH_LAUNCH(HT, FUSED, WC)                                                                                  \
    KLAUNCH(HIP_KERNEL_NAME(sketch_wave_kernel<HT, FUSED, WC>), dim3((n_tiles + 3) / 4), SKETCH_ARGS)
        // w = 14 (the default of pandora map) and w = 1 (dense sketches) have their own instantiations
#define SKETCH_DISPATCH(HT, FUSED)                                                                                    \
    do {                                                                                                              \
        if (w == 14) SKETCH_LAUNCH(HT, FUSED, 14);                                                                    \
        else if (w == 1) SKETCH_LAUNCH(HT, FUSED, 1);                                                                 \
        else SKETCH_LAUNCH(HT, FUSED, 0);                                                                             \
    } while (0)
        // k <= 15 with w <= 16: windows in registers (sketch_dpp_kernel); the default w = 14, k = 15 at compile time
#define SKETCH_DPP(FUSED, WC, KC, NP)                                                                                 \
    do {                                                                                                              \
        KLAUNCH_L(HIP_KERNEL_NAME(sketch_dpp_kernel<FUSED, WC, KC, NP>), dim3((n_tiles + 3) / 4),                     \
                  (DppList<WC, NP>::BYTES + g_sketch_pad), SKETCH_ARGS);                                              \
        if (res->dense_launched && DppList<WC, NP>::CAP < DppTile<NP>::POS)                                           \
            KLAUNCH_L(HIP_KERNEL_NAME(sketch_dpp_overflow_kernel<FUSED, WC, KC, NP>),                                 \
                      dim3(std::min((n_tiles + 255) / 256, 1024u)), (4u * DppTile<NP>::POS * 8u), SKETCH_ARGS_NOAUX); \
    } while (0)
#define SKETCH_DPP_NP(NP)                                                                                             \
    do {                                                                                                              \
        if (fused && k == 15) SKETCH_DPP(true, 14, 15, NP);                                                           \
        else if (fused) SKETCH_DPP(true, 14, 0, NP);                                                                  \
        else if (k == 15) SKETCH_DPP(false, 14, 15, NP);                                                              \
        else SKETCH_DPP(false, 14, 0, NP);                                                                            \
    } while (0)
        // any other w of 1..16: w at compile time, k from the argument
#define SKETCH_DPP_W(WC)                                                                                              \
    case WC:                                                                                                          \
        if (fused) SKETCH_DPP(true, WC, 0, 16);                                                                       \
        else SKETCH_DPP(false, WC, 0, 16);                                                                            \
        break;
        uint32_t out_pos_, stride_;
        const SketchKind kind = sketch_kind(w, k, prm->flags, &out_pos_, &stride_);
        if (kind == SKETCH_DPP16 && w == 14) {
            SKETCH_DPP_NP(16);
        } else if (kind == SKETCH_DPP16) {
            switch (w) {
                SKETCH_DPP_W(1) SKETCH_DPP_W(2) SKETCH_DPP_W(3) SKETCH_DPP_W(4) SKETCH_DPP_W(5) SKETCH_DPP_W(6) SKETCH_DPP_W(7)
                SKETCH_DPP_W(8) SKETCH_DPP_W(9) SKETCH_DPP_W(10) SKETCH_DPP_W(11) SKETCH_DPP_W(12) SKETCH_DPP_W(13)
                SKETCH_DPP_W(15) SKETCH_DPP_W(16)
                default: return pmap_fail(PMAP_EINVAL, "no register-window sketch kernel for this w");
            }
        } else if (kind == SKETCH_DPP8) {
            SKETCH_DPP_NP(8);
        } else if (k32 && fused) SKETCH_DISPATCH(uint32_t, true);
        else if (k32) SKETCH_DISPATCH(uint32_t, false);
        else if (fused) SKETCH_DISPATCH(uint64_t, true);
        else SKETCH_DISPATCH(uint64_t, false);
#undef SKETCH_DPP_W
#undef SKETCH_DPP_NP
#undef SKETCH_DPP
#undef SKETCH_DISPATCH
#undef SKETCH_LAUNCH
#undef SKETCH_ARGS
#undef SKETCH_ARGS_NOAUX
        kt.end(KT_SKETCH);
        HIP_TRY(hipGetLastError());
    }
    // everything else follows on the tail stream
    HIP_TRY(hipEventRecord(res->sketched, sketch_stream));
    HIP_TRY(hipStreamWaitEvent(g_tail, res->sketched, 0));
    g_cur = g_tail;

    // ---- probe (unfused), batch totals, hit emission
    if (n_tiles) {
        const bool emit = last >= PMAP_STAGE_HITS;
        if (!fused && emit) {
            kt.begin(KT_PROBE);
            KLAUNCH(probe_wave_kernel, dim3((n_tiles + 3) / 4), res->mkmer.as<uint64_t>(), n_tiles, od.slots,
                    idx->log2cap, od.filter, idx->filter_bits, res->ents.as<uint4>(), res->tile_cnt.as<uint4>(),
                    res->tile_stride);
            kt.end(KT_PROBE);
        }
        kt.begin(KT_SCATTER);
        const uint32_t n_waves = (n_tiles + SCATTER_TILES - 1) / SCATTER_TILES;
        KLAUNCH(group_reduce_kernel, dim3((n_groups + 3) / 4), res->tile_cnt.as<uint4>(), n_tiles,
                res->group_sum.as<uint4>());
        KLAUNCH(scatter_group_kernel, dim3((n_waves + 3) / 4), res->tile_cnt.as<uint4>(), n_tiles, res->tile_stride,
                res->ents.as<uint4>(),
                od.recs, res->hit_cap, emit ? 1u : 0u, dc, res->group_sum.as<uint4>(), rd->n_reads + 1,
                res->hbounds.as<uint32_t>(),
                res->h_read.as<uint32_t>(), res->h_key.as<uint32_t>(), res->h_pos.as<uint32_t>(),
                res->h_rank.as<uint32_t>(), res->h_rec.as<uint32_t>());
        kt.end(KT_SCATTER);
        HIP_TRY(hipGetLastError());
    }

    // ---- sort + cluster in one pass (tail_chunk_kernel), unless the separate kernels are asked for or the keys
    // are too wide to share a word with a place
    uint32_t key_bits = 1;
    while (key_bits < 32 && (2ull * idx->n_prg) > (1ull << key_bits)) ++key_bits;
    const bool staged = (prm->flags & PMAP_FLAG_TAIL_STAGED) || key_bits > TAIL_MAX_KEY_BITS ||
                        (res->few_hits && !(prm->flags & PMAP_FLAG_TEST_TAIL_NARROW));
    res->staged_tail = staged;
    if (!staged && last >= PMAP_STAGE_SORT && n_tiles) {
        const bool cluster = last >= PMAP_STAGE_CLUSTER;
        const uint32_t sort_long = std::max(std::min(g_sort_long, TAIL_BIG_MAX), TAIL_C);
        // (a read has at most one minimizer per base and a minimizer at most max_cnt hits)
        const uint64_t most_hits = (uint64_t) rd->max_len * idx->max_cnt;
        res->long_launched = res->long_enabled && most_hits > sort_long;
        TailArgs ta;
        ta.dc = dc;
        ta.hit_cap = res->hit_cap;
        ta.n_reads = rd->n_reads + 1;
        ta.key_bits = key_bits;
        ta.ord_bits = (prm->flags & PMAP_FLAG_TEST_TAIL_NARROW) ? 1u : std::min(8u, 32u - 8u - key_bits);
        ta.do_cluster = cluster ? 1u : 0u;
        ta.max_diff = (int) prm->max_diff;
        ta.min_thr = cluster ? res->thr_minval : 0u;
        ta.slot_div = ta.min_thr + 1;
        ta.rows_tmp_cap = res->hit_cap / ta.slot_div + 2;
        ta.sort_long = sort_long;
        if ((rc = res->chunk_cnt.ensure(2 * sizeof(uint4) * ((size_t) n_chunks_cap + 64))) ||
            (cluster && (rc = res->rows_tmp.ensure(32 * (size_t) ta.rows_tmp_cap))))
            return rc;
        ta.h_read = res->h_read.as<uint32_t>(); ta.h_key = res->h_key.as<uint32_t>(); ta.h_pos = res->h_pos.as<uint32_t>();
        ta.h_rank = res->h_rank.as<uint32_t>(); ta.h_rec = res->h_rec.as<uint32_t>();
        ta.hbounds = res->hbounds.as<uint32_t>();
        ta.thr = res->thr.as<uint32_t>();
        ta.s_read = res->s_read.as<uint32_t>(); ta.s_key = res->s_key.as<uint32_t>(); ta.s_pos = res->s_pos.as<uint32_t>();
        ta.s_rank = res->s_rank.as<uint32_t>(); ta.s_rec = res->s_rec.as<uint32_t>();
        ta.chunk_cnt = res->chunk_cnt.as<uint4>();
        ta.chunk_cont = ta.chunk_cnt + n_chunks_cap + 64;
        ta.cg_sum = cg_sum;
        ta.rows_tmp = res->rows_tmp.as<uint32_t>();
        ta.big_list = res->hbounds.as<uint32_t>() + 2 * ((size_t) rd->n_reads + 1);     // behind the bounds
        ta.long_list = res->hbounds.as<uint32_t>() + 3 * ((size_t) rd->n_reads + 1);
        kt.begin(KT_TAIL);
        const uint32_t tail_grid = g_tail_grid ? g_tail_grid : std::min(std::max(n_chunks_cap / 4, 4096u), 32768u);
        KLAUNCH_B(tail_chunk_kernel, dim3(std::min(n_chunks_cap, tail_grid)), 64, ta);
        if (most_hits > TAIL_WAVE_MAX && sort_long > TAIL_WAVE_MAX)
            KLAUNCH_B(tail_big_kernel, dim3(std::min(rd->n_reads + 1, 1024u)), 64, ta);
        // Reads with more hits than a wavefront sorts: the long-segment kernels, launched only once a batch has
        // turned out to hold such a read (pmap_map_wait then maps the batch again with them, and they stay on
        // until a batch comes without one)
        if (res->long_launched) {
            KLAUNCH(long_presort_kernel, dim3(std::min(res->hit_cap / SORT_LB, 1024u)), dc, res->hit_cap, res->h_read.as<uint32_t>(),
                    res->h_key.as<uint32_t>(), res->hbounds.as<uint32_t>(), rd->n_reads + 1,
                    res->long_words.as<unsigned long long>(), sort_long);
            KLAUNCH(long_rank_kernel, dim3(std::min(res->hit_cap / 256, 2048u)), dc, res->hit_cap, res->h_read.as<uint32_t>(),
                    res->h_key.as<uint32_t>(), res->h_pos.as<uint32_t>(), res->h_rank.as<uint32_t>(),
                    res->h_rec.as<uint32_t>(), res->hbounds.as<uint32_t>(), rd->n_reads + 1,
                    res->long_words.as<unsigned long long>(), res->s_read.as<uint32_t>(), res->s_key.as<uint32_t>(),
                    res->s_pos.as<uint32_t>(), res->s_rank.as<uint32_t>(), res->s_rec.as<uint32_t>(), sort_long);
            if (cluster) KLAUNCH_B(tail_long_kernel, dim3(8 * std::min(rd->n_reads + 1, 128u)), 64, ta);
        }
        kt.end(KT_TAIL);
        if (cluster) {
            kt.begin(KT_ROWS);
            KLAUNCH_B(rows_gather_kernel, dim3((n_chunks_cap + 63) / 64), 64, dc, res->hit_cap, ta.chunk_cnt, ta.chunk_cont, cg_sum,
                      res->rows_tmp.as<uint32_t>(), ta.rows_tmp_cap, res->k_rows.as<uint32_t>(), res->cl_cap);
            kt.end(KT_ROWS);
        }
        HIP_TRY(hipGetLastError());
    }

    // ---- sort (separate kernels)
    if (staged && last >= PMAP_STAGE_SORT && n_tiles) {
        kt.begin(KT_SORT);
        // (a read has at most one minimizer per base and a minimizer at most max_cnt hits)
        res->long_launched = res->long_enabled && (uint64_t) rd->max_len * idx->max_cnt > g_sort_long;
        // Reads of middling hit counts go to sort_wave_kernel, whose words hold a key and a place of up to
        // 6 + log2 SORT_WAVE_E bits (PMAP_FLAG_SORT_NO_WAVE: none do; PMAP_FLAG_TEST_SORT_WAVE_ALL: every read
        // of 2 hits or more up to the kernel's limit does)
        uint32_t wave_min = g_sort_wave_min ? g_sort_wave_min : w >= 8 ? 192u : 320u, wave_max = g_sort_wave_max;
        uint32_t *wave_list = res->hbounds.as<uint32_t>() + 2 * ((size_t) rd->n_reads + 1);     // behind the bounds
        if (prm->flags & PMAP_FLAG_TEST_SORT_WAVE_ALL) wave_min = 1, wave_max = SORT_WAVE_LIMIT;
        // (a read has at most one minimizer per base and a minimizer at most max_cnt hits)
        const uint64_t most_hits = (uint64_t) rd->max_len * idx->max_cnt;
        if ((prm->flags & PMAP_FLAG_SORT_NO_WAVE) || 2ull * idx->n_prg * SORT_WAVE_LIMIT > (1ull << 32) || wave_min >= wave_max ||
            most_hits <= wave_min)
            wave_min = wave_max = 0;
        const bool wave_launched = wave_max && (res->wave_enabled || (prm->flags & PMAP_FLAG_TEST_SORT_WAVE_ALL));
#define SORT_ARGS                                                                                                     \
    &dc->n_hits, res->hit_cap, res->h_read.as<uint32_t>(), res->h_key.as<uint32_t>(), res->h_pos.as<uint32_t>(),      \
        res->h_rank.as<uint32_t>(), res->h_rec.as<uint32_t>(), res->hbounds.as<uint32_t>(), rd->n_reads + 1,          \
        res->s_read.as<uint32_t>(), res->s_key.as<uint32_t>(), res->s_pos.as<uint32_t>(), res->s_rank.as<uint32_t>(), \
        res->s_rec.as<uint32_t>(), &dc->has_long, g_sort_long, res->long_launched ? 1u : 0u, wave_min, wave_max,      \
        wave_launched ? 1u : 0u, &dc->has_wave
        // keys are prg << 1 | strand: narrow enough to share a word with a place in the staged chunk?
        if (2ull * idx->n_prg <= (1ull << (31 - SORT_CH_BITS)) && !(prm->flags & PMAP_FLAG_SORT_WIDE))
            KLAUNCH(sort_rank_kernel<true>, dim3(res->hit_cap / 256), SORT_ARGS);
        else
            KLAUNCH(sort_rank_kernel<false>, dim3(res->hit_cap / 256), SORT_ARGS);
#undef SORT_ARGS
        static_assert(SORT_WAVE_LIMIT == 64 * SORT_WAVE_E, "the kernel's largest network");
#define SORT_WAVE_ARGS                                                                                                \
    &dc->n_hits, res->hit_cap, res->h_key.as<uint32_t>(), res->h_pos.as<uint32_t>(), res->h_rank.as<uint32_t>(),      \
        res->h_rec.as<uint32_t>(), res->hbounds.as<uint32_t>(), rd->n_reads + 1, res->s_read.as<uint32_t>(),          \
        res->s_key.as<uint32_t>(), res->s_pos.as<uint32_t>(), res->s_rank.as<uint32_t>(), res->s_rec.as<uint32_t>(),  \
        g_sort_long, wave_min, wave_max, &dc->n_wave, wave_list
        if (wave_launched) {
            // at most a read per wave_min hits is on the list; a wavefront takes several in turn beyond 32768 reads
            const uint32_t wave_grid = (uint32_t) std::min<uint64_t>(8192, std::min<uint64_t>(rd->n_reads, res->hit_cap / std::max(wave_min, 1u)) / 4 + 1);
            KLAUNCH(sort_wave_list_kernel, dim3((rd->n_reads + 1 + 255) / 256), res->hbounds.as<uint32_t>(), rd->n_reads + 1,
                    g_sort_long, wave_min, wave_max, &dc->n_wave, wave_list);
            KLAUNCH(sort_wave_kernel<false>, dim3(wave_grid), SORT_WAVE_ARGS);
            if (wave_max > SORT_WAVE_LIMIT / 2 && most_hits > SORT_WAVE_LIMIT / 2)
                KLAUNCH(sort_wave_kernel<true>, dim3(wave_grid), SORT_WAVE_ARGS);
        }
#undef SORT_WAVE_ARGS
        // Reads with more hits than the rank counting takes.  Such reads are rare: the two kernels are launched
        // only once a batch has turned out to hold one (pmap_map_wait then maps the batch again with them, and
        // they stay on until a batch comes without such a read).
        if (res->long_launched) {
            KLAUNCH(long_presort_kernel, dim3(std::min(res->hit_cap / SORT_LB, 1024u)), dc, res->hit_cap, res->h_read.as<uint32_t>(),
                    res->h_key.as<uint32_t>(), res->hbounds.as<uint32_t>(), rd->n_reads + 1,
                    res->long_words.as<unsigned long long>(), g_sort_long);
            KLAUNCH(long_rank_kernel, dim3(std::min(res->hit_cap / 256, 2048u)), dc, res->hit_cap, res->h_read.as<uint32_t>(),
                    res->h_key.as<uint32_t>(), res->h_pos.as<uint32_t>(), res->h_rank.as<uint32_t>(),
                    res->h_rec.as<uint32_t>(), res->hbounds.as<uint32_t>(), rd->n_reads + 1,
                    res->long_words.as<unsigned long long>(), res->s_read.as<uint32_t>(), res->s_key.as<uint32_t>(),
                    res->s_pos.as<uint32_t>(), res->s_rank.as<uint32_t>(), res->s_rec.as<uint32_t>(), g_sort_long);
        }
        kt.end(KT_SORT);
        HIP_TRY(hipGetLastError());
    }

    // ---- cluster (separate kernels)
    if (staged && last >= PMAP_STAGE_CLUSTER && n_tiles) {
        // the chained prefix sums wait for earlier workgroups; a batch whose wait ran out is mapped again with
        // one workgroup per kernel that walks the blocks itself (pmap_map_wait sets serial_scans)
        const uint32_t spin = (prm->flags & PMAP_FLAG_TEST_CHAIN_GIVES_UP) ? 0u : CHAIN_SPIN_LIMIT;
#define CUT_ARGS                                                                                                      \
    dc, res->hit_cap, res->s_read.as<uint32_t>(), res->s_key.as<uint32_t>(), res->s_pos.as<uint32_t>(),              \
        (int) prm->max_diff, gran_a, res->cl_cap, res->cstart.as<uint32_t>(), res->n_ablk, spin
#define KEEP_ARGS                                                                                                     \
    dc, res->cl_cap, res->hit_cap, res->cstart.as<uint32_t>(), res->s_read.as<uint32_t>(), res->s_key.as<uint32_t>(), \
        res->s_pos.as<uint32_t>(), res->s_rank.as<uint32_t>(), res->thr.as<uint32_t>(), idx->n_prg + 1, gran_b,      \
        res->k_rows.as<uint32_t>(), res->n_bblk, spin
        kt.begin(KT_CLUSTER_CUT);
        if (res->serial_scans) KLAUNCH_B(cluster_cut_kernel<true>, dim3(1), CUT_THREADS, CUT_ARGS);
        else KLAUNCH_B(cluster_cut_kernel<false>, dim3(res->n_ablk), CUT_THREADS, CUT_ARGS);
        kt.end(KT_CLUSTER_CUT);
        kt.begin(KT_CLUSTER_KEEP);
        if (res->serial_scans) KLAUNCH_B(cluster_keep_kernel<true>, dim3(1), KEEP_THREADS, KEEP_ARGS);
        else KLAUNCH_B(cluster_keep_kernel<false>, dim3(res->n_bblk), KEEP_THREADS, KEEP_ARGS);
        kt.end(KT_CLUSTER_KEEP);
#undef CUT_ARGS
#undef KEEP_ARGS
        HIP_TRY(hipGetLastError());
    }
    res->kt_recorded = kt.recorded;
    HIP_TRY(hipMemcpyAsync(res->hc, dc, sizeof(Counters), hipMemcpyDeviceToHost, g_tail));
    HIP_TRY(hipEventRecord(res->done, g_tail));
    return PMAP_OK;
}

static int ensure_hit_buffers(pmap_result *res, uint32_t hit_cap) {
    int rc;
    const size_t hb = sizeof(uint32_t) * (size_t) hit_cap;
    if ((rc = res->h_read.ensure(hb)) || (rc = res->h_key.ensure(hb)) || (rc = res->h_pos.ensure(hb)) ||
        (rc = res->h_rank.ensure(hb)) || (rc = res->h_rec.ensure(hb)) || (rc = res->s_read.ensure(hb)) ||
        (rc = res->s_key.ensure(hb)) || (rc = res->s_pos.ensure(hb)) || (rc = res->s_rank.ensure(hb)) ||
        (rc = res->s_rec.ensure(hb)) || (rc = res->long_words.ensure(2 * hb)))
        return rc;
    res->hit_cap = hit_cap;
    return PMAP_OK;
}

static int ensure_cluster_buffers(pmap_result *res, uint32_t cl_cap) {
    int rc;
    const size_t cb = sizeof(uint32_t) * ((size_t) cl_cap + 1);
    if ((rc = res->cstart.ensure(cb)) || (rc = res->k_rows.ensure(cb * 8))) return rc;
    res->cl_cap = cl_cap;
    return PMAP_OK;
}

extern "C" int pmap_map_wait(pmap_result_t *res);

extern "C" int pmap_map_async(const pmap_index_t *idx, const pmap_reads_t *rd, const pmap_params_t *prm,
                              pmap_result_t **result) {
    int rc;
    if (!idx || !rd || !prm || !result) return pmap_fail(PMAP_EINVAL, "null argument");
    if (prm->k != idx->k) return pmap_fail(PMAP_EINVAL, "k differs from the k the index was built with");
    if (prm->w < 1 || prm->w > (uint32_t) SK_MAX_W)
        return pmap_fail(PMAP_EINVAL, "w must be in [1,64]");
    // the batch's context decides where the call runs
    DevCtx *ctx = rd->ctx;
    if ((rc = bind(ctx))) return rc;
    if ((rc = pmap_index_upload(const_cast<pmap_index *>(idx)))) return rc;
    pmap_result *res = *result;
    if (res && res->ctx != ctx)
        return pmap_fail(PMAP_EINVAL, "the result object belongs to another mapping context than the reads");
    if (!res) {
        if ((rc = new_result(result))) {
            pmap_result_destroy(*result);
            *result = nullptr;
            return rc;
        }
        res = *result;
    }
    hipStream_t g_tail = ctx->tail;
    if (res->pending && (rc = pmap_map_wait(res))) return rc;   // the buffers are about to be reused
    const uint32_t last = prm->last_stage ? prm->last_stage : PMAP_STAGE_CLUSTER;
    const bool fused = !(prm->flags & PMAP_FLAG_UNFUSED) && last >= PMAP_STAGE_HITS;
    pmap_counts_t &cn = res->counts;
    memset(&cn, 0, sizeof(cn));
    memset(res->kernel_ms, 0, sizeof(res->kernel_ms));
    cn.n_reads = rd->n_reads;
    cn.n_bases = rd->n_bases;
    res->first_read_id = rd->first_read_id;
    {
        uint32_t out_pos = 0;
        sketch_kind(prm->w, prm->k, prm->flags, &out_pos, &res->tile_stride);
        res->n_tiles = (uint32_t) (((uint64_t) rd->n_words * 16 + out_pos - 1) / out_pos);
    }
    res->have_minimizers = false;

    // buffers whose size follows from the batch alone
    const uint32_t n_tiles = res->n_tiles;
    const size_t sb = sizeof(uint32_t) * (size_t) n_tiles * res->tile_stride;
    if ((rc = res->tile_cnt.ensure(sizeof(uint4) * ((size_t) n_tiles + 1))) || (rc = res->ents.ensure(8 * sb)) ||
        (rc = res->hbounds.ensure(sizeof(uint32_t) * 4 * ((size_t) rd->n_reads + 1))))
        return rc;
    if (!fused && (rc = res->mkmer.ensure(2 * sb))) return rc;

    // per-prg size threshold (src/utils.cpp:211-213, 219-220), refreshed when its inputs change
    if (last >= PMAP_STAGE_CLUSTER &&
        (res->thr_serial != idx->serial || res->thr_fraction != prm->fraction || res->thr_min != prm->min_cluster_size ||
         res->thr_exp != prm->expected_kmers)) {
        res->thr_host.assign(idx->n_prg + 1, 0);
        uint32_t lowest = idx->n_prg ? 0xFFFFFFFFu : 0u;
        for (uint32_t p = 0; p < idx->n_prg; ++p) {
            res->thr_host[p] = pmap_threshold(idx->min_path_length[p], prm->expected_kmers, prm->fraction,
                                              prm->min_cluster_size);
            lowest = std::min(lowest, res->thr_host[p]);
        }
        res->thr_minval = lowest;
        if ((rc = res->thr.ensure(sizeof(uint32_t) * res->thr_host.size()))) return rc;
        HIP_TRY(hipMemcpyAsync(res->thr.p, res->thr_host.data(), sizeof(uint32_t) * res->thr_host.size(),
                               hipMemcpyHostToDevice, g_tail));
        HIP_TRY(hipStreamSynchronize(g_tail));   // thr_host is pageable and may change before the copy ran
        res->thr_serial = idx->serial;
        res->thr_fraction = prm->fraction;
        res->thr_min = prm->min_cluster_size;
        res->thr_exp = prm->expected_kmers;
    }
    // capacities of the data-dependent buffers: carried over from earlier calls, grown on overflow
    if (res->hit_cap == 0 && (rc = ensure_hit_buffers(res, round_up(std::max<uint64_t>((uint64_t) n_tiles * (res->tile_stride / 64), 4096), CUT_BLOCK))))
        return rc;
    if (res->cl_cap == 0 && (rc = ensure_cluster_buffers(res, round_up(std::max<uint64_t>(res->hit_cap / 8, 1024), KEEP_BLOCK))))
        return rc;

    res->p_idx = idx;
    res->p_rd = rd;
    res->p_prm = *prm;
    res->p_fused = fused;
    res->p_last = last;
    if ((rc = run_pipeline(res))) return rc;
    res->pending = true;
    return PMAP_OK;
}

extern "C" int pmap_map_wait(pmap_result_t *res) {
    if (!res) return pmap_fail(PMAP_EINVAL, "null argument");
    if (!res->pending) return PMAP_OK;
    res->pending = false;
    int rc;
    if ((rc = bind(res->ctx))) return rc;
    const uint32_t last = res->p_last;
    pmap_counts_t &cn = res->counts;
    unsigned long long n_mini = 0, n_hits = 0;
    for (int attempt = 0;; ++attempt) {
        HIP_TRY(hipEventSynchronize(res->done));
        const Counters &hc = *res->hc;
        if (hc.fail) {
            // a workgroup waited in vain for the prefix of an earlier one (another process on the GPU, or a
            // dispatch order the chain does not expect): the same batch once more without a chain
            if (res->serial_scans) {
                res->serial_scans = false;
                return pmap_fail(PMAP_EHIP, "the cluster kernels failed without a chained prefix sum");
            }
            res->serial_scans = true;
            res->n_serial_reruns++;
            rc = run_pipeline(res);
            if (rc) { res->serial_scans = false; return rc; }
            continue;
        }
        res->serial_scans = false;
        if (res->p_last >= PMAP_STAGE_SORT && hc.has_long && !res->long_launched) {
            // a read with more hits than the rank counting sorts: once more with the long-segment kernels
            res->long_enabled = true;
            if ((rc = run_pipeline(res))) return rc;
            continue;
        }
        res->long_enabled = hc.has_long != 0;
        res->wave_enabled = hc.has_wave != 0;
        if (hc.dense_marks && !res->dense_launched) {
            // a tile with more minimizers than the sketch kernel's list holds (low-complexity sequence) was left
            // out: once more with the kernel that takes such tiles behind it; it stays on until a batch comes
            // without such a tile
            res->dense_enabled = true;
            if ((rc = run_pipeline(res))) return rc;
            continue;
        }
        res->dense_enabled = hc.dense_marks != 0;
        n_mini = hc.n_mini;
        n_hits = hc.n_hits;
        if (n_hits >= 0xFFFFFF00ull) return pmap_fail(PMAP_ELIMIT, "more than 2^32-257 hits in one batch");
        bool again = false;
        if (last >= PMAP_STAGE_HITS && n_hits > res->hit_cap) {
            const uint32_t want = round_up(n_hits + n_hits / 8, CUT_BLOCK);
            if (want < n_hits) return pmap_fail(PMAP_ELIMIT, "too many hits in one batch for 32-bit hit buffers");
            if ((rc = ensure_hit_buffers(res, want))) return rc;
            again = true;
        }
        // at most one cluster per hit; when the hits overflowed the cluster count is only a lower bound
        // (the fused tail keeps no list of all clusters: only the rows of the kept ones need room)
        const uint64_t n_rows = res->staged_tail ? hc.n_cl : hc.n_kept;
        if (last >= PMAP_STAGE_CLUSTER && (n_rows > res->cl_cap || (again && res->cl_cap < res->hit_cap / 8))) {
            const uint64_t want = std::max<uint64_t>(n_rows + n_rows / 8, res->hit_cap / 8);
            if ((rc = ensure_cluster_buffers(res, round_up(std::min<uint64_t>(want, res->hit_cap), KEEP_BLOCK)))) return rc;
            again = true;
        }
        if (!again) break;
        if (attempt > 6) return pmap_fail(PMAP_EHIP, "buffer sizing did not converge");
        // a buffer was too small: it has been grown (freeing waits for the device), map the batch again
        if ((rc = run_pipeline(res))) return rc;
    }
    const Counters &hc = *res->hc;
    res->few_hits = last >= PMAP_STAGE_CLUSTER && res->p_rd->n_reads >= 1024 && n_hits < (unsigned long long) g_tail_auto * res->p_rd->n_reads;
    cn.n_minimizers = n_mini;
    cn.n_hits = last >= PMAP_STAGE_HITS ? n_hits : 0;
    cn.n_clusters_all = last >= PMAP_STAGE_CLUSTER ? hc.n_cl : 0;
    cn.n_clusters = last >= PMAP_STAGE_CLUSTER ? hc.n_kept : 0;
    res->have_minimizers = !res->p_fused;
    cn.ms[5] = (float) res->n_serial_reruns;
    if (res->p_prm.profile >= 1) {
        const uint32_t recorded = res->kt_recorded;
        for (int i = 0; i < KT_COUNT; ++i) {
            float ms = 0;
            // (a kernel that did not run in this call reads 0, not the time of an earlier call)
            if (!(recorded >> i & 1u) || hipEventElapsedTime(&ms, res->kev[2 * i], res->kev[2 * i + 1]) != hipSuccess) ms = 0;
            res->kernel_ms[i] = ms;
        }
        // stage times = sums of their kernels; ms[4] = first kernel start to last kernel stop
        const float *km = res->kernel_ms;
        cn.ms[0] = km[KT_SKETCH];
        cn.ms[1] = km[KT_PROBE] + km[KT_SCATTER];
        cn.ms[2] = km[KT_SORT] + km[KT_TAIL];
        cn.ms[3] = km[KT_CLUSTER_CUT] + km[KT_CLUSTER_KEEP] + km[KT_ROWS];
        int lastk = -1;
        for (int i = 0; i < KT_COUNT; ++i)
            if (recorded >> i & 1u) lastk = i;
        if ((recorded & 1u) && lastk >= 0)
            (void) hipEventElapsedTime(&cn.ms[4], res->kev[0], res->kev[2 * lastk + 1]);
    }
    return PMAP_OK;
}

extern "C" int pmap_map(const pmap_index_t *idx, const pmap_reads_t *rd, const pmap_params_t *prm,
                        pmap_result_t **result) {
    int rc = pmap_map_async(idx, rd, prm, result);
    if (rc) return rc;
    return pmap_map_wait(*result);
}

// ---------------------------------------------------------------------------
// pmap_map without allocations

static int new_result(pmap_result **out) {     // in the calling thread's context
    pmap_result *res = new pmap_result();
    res->ctx = t_ctx;
    memset(&res->counts, 0, sizeof(res->counts));
    memset(res->kernel_ms, 0, sizeof(res->kernel_ms));
    *out = res;
    for (auto &e : res->kev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreateWithFlags(&res->sketched, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&res->done, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void **) &res->hc, sizeof(Counters), hipHostMallocDefault));
    res->events = true;
    return PMAP_OK;
}

// Once per device, with that device current and a context of it bound (pmap_init / pmap_init_devices make the
// first context of a GPU): the address of this device's copy of the scratch array, and everything else
// pmap_map_noalloc would otherwise have to create per call.
static int noalloc_setup(int device) {
    NoAlloc &na = g_noalloc[device];
    std::lock_guard<std::mutex> l(na.setup_mu);
    if (na.scratch) return PMAP_OK;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(pmap_scratch));
    if (e != hipSuccess || !ptr)
        return pmap_fail(PMAP_EHIP, std::string("hipGetSymbolAddress(pmap_scratch): ") + hipGetErrorString(e));
    size_t bytes = 0;
    HIP_TRY(hipGetSymbolSize(&bytes, HIP_SYMBOL(pmap_scratch)));
    if (bytes < NA_SCRATCH_BYTES) return pmap_fail(PMAP_EHIP, "the scratch array is smaller than its layout");
    HIP_TRY(hipHostMalloc((void **) &na.thr_stage, 4 * (NA_MAX_PRG + 64), hipHostMallocDefault));
    int rc = new_result(&na.view);
    if (rc) return rc;
    unsigned char *base = (unsigned char *) ptr;
    pmap_result *v = na.view;
    v->state.view(base + NA_OFF_STATE, NA_OFF_GROUP - NA_OFF_STATE);
    v->group_sum.view(base + NA_OFF_GROUP, NA_OFF_TILECNT - NA_OFF_GROUP);
    v->tile_cnt.view(base + NA_OFF_TILECNT, NA_OFF_HBOUNDS - NA_OFF_TILECNT);
    v->hbounds.view(base + NA_OFF_HBOUNDS, NA_OFF_THR - NA_OFF_HBOUNDS);
    v->thr.view(base + NA_OFF_THR, NA_OFF_CSTART - NA_OFF_THR);
    v->cstart.view(base + NA_OFF_CSTART, NA_OFF_CHUNK - NA_OFF_CSTART);
    v->chunk_cnt.view(base + NA_OFF_CHUNK, NA_OFF_HITS - NA_OFF_CHUNK);
    DevBuf *unsorted[5] = {&v->h_read, &v->h_key, &v->h_pos, &v->h_rank, &v->h_rec};
    for (int i = 0; i < 5; ++i) unsorted[i]->view(base + NA_OFF_HITS + (size_t) i * 4 * NA_MAX_HITS, 4 * NA_MAX_HITS);
    v->ents.view(base + NA_OFF_ENTS, 32 * NA_MAX_SLOTS);
    static_assert(NA_OFF_ROWS_TMP + 32 * NA_ROWS_TMP <= NA_SCRATCH_BYTES && NA_ROWS_TMP >= NA_MAX_HITS / 2,
                  "the long-segment sort's words and the row slots fit the entry area");
    v->long_words.view(base + NA_OFF_ENTS, 8 * NA_MAX_HITS);
    v->rows_tmp.view(base + NA_OFF_ROWS_TMP, 32 * NA_ROWS_TMP);
    // never used on this path (unfused mode, minimizer gather); empty views make any use fail instead of allocate
    v->mkmer.view(nullptr, 0);
    v->bsum.view(nullptr, 0);
    v->scratch_total.view(nullptr, 0);
    na.scratch = base;
    return PMAP_OK;
}

extern "C" int pmap_map_noalloc(const pmap_index_t *idx, const pmap_reads_t *rd, const pmap_params_t *prm,
                                const pmap_dev_out_t *out, pmap_counts_t *counts) {
    if (!idx || !rd || !prm || !out || !counts) return pmap_fail(PMAP_EINVAL, "null argument");
    if (!t_ctx) return pmap_fail(PMAP_EINVAL, "call pmap_init first: pmap_map_noalloc sets nothing up by itself");
    DevCtx *ctx = rd->ctx;
    if (int rcb = bind(ctx)) return rcb;
    NoAlloc &na = g_noalloc[ctx->device];
    if (!na.scratch || !na.view)
        return pmap_fail(PMAP_EINVAL, "the reads' GPU has no scratch array");
    // one call at a time per GPU, whichever contexts and threads the calls come from
    if (na.busy.exchange(true, std::memory_order_acquire))
        return pmap_fail(PMAP_EINVAL, "the scratch array of this GPU is in use by another pmap_map_noalloc call");
    struct Release {
        std::atomic<bool> &b;
        ~Release() { b.store(false, std::memory_order_release); }
    } release{na.busy};
    pmap_result *res = na.view;
    res->ctx = ctx;                 // the streams of this call
    hipStream_t g_tail = ctx->tail;
    if (prm->k != idx->k) return pmap_fail(PMAP_EINVAL, "k differs from the k the index was built with");
    if (prm->w < 1 || prm->w > (uint32_t) SK_MAX_W) return pmap_fail(PMAP_EINVAL, "w must be in [1,64]");
    if ((prm->flags & PMAP_FLAG_UNFUSED) || (prm->last_stage && prm->last_stage != PMAP_STAGE_CLUSTER))
        return pmap_fail(PMAP_EINVAL, "pmap_map_noalloc runs the whole path in its default mode only");
    if (!index_is_on_device(idx, ctx->device))
        return pmap_fail(PMAP_EINVAL, "the index is not on the device: call pmap_index_upload_now first (this entry point does not allocate)");
    if (!out->hit_read || !out->hit_key || !out->hit_pos || !out->hit_rank || !out->hit_rec || !out->cluster_rows)
        return pmap_fail(PMAP_EINVAL, "null output buffer");
    // capacities as the kernels' grids want them: whole blocks of the cluster kernels
    uint32_t hit_cap = (uint32_t) std::min<uint64_t>(out->hit_cap / CUT_BLOCK * (uint64_t) CUT_BLOCK, NA_MAX_HITS);
    const uint32_t cl_cap = (uint32_t) std::min<uint64_t>(out->cluster_cap / 1024u * (uint64_t) 1024u, NA_MAX_CLUSTERS);   // as documented: multiples of 1024
    if (hit_cap == 0 || cl_cap == 0)
        return pmap_fail(PMAP_EINVAL, "output buffers must hold at least 8192 hits and 1024 clusters");
    uint32_t out_pos = 0, stride = 0;
    sketch_kind(prm->w, prm->k, prm->flags, &out_pos, &stride);
    const uint64_t n_tiles = ((uint64_t) rd->n_words * 16 + out_pos - 1) / out_pos;
    if (n_tiles * stride > NA_MAX_SLOTS)
        return pmap_fail(PMAP_ELIMIT, "batch too large for pmap_map_noalloc: its tiles need more than 2^20 entry slots "
                                      "(about one million padded bases); split the batch or use pmap_map");
    if (rd->n_reads > NA_MAX_READS) return pmap_fail(PMAP_ELIMIT, "more than 2^17 reads in a batch for pmap_map_noalloc");
    if (idx->n_prg > NA_MAX_PRG) return pmap_fail(PMAP_ELIMIT, "more than 2^20 PRGs in the index for pmap_map_noalloc");
    int rc;
    if (res->pending && (rc = pmap_map_wait(res))) return rc;

    memset(&res->counts, 0, sizeof(res->counts));
    res->counts.n_reads = rd->n_reads;
    res->counts.n_bases = rd->n_bases;
    res->first_read_id = rd->first_read_id;
    res->n_tiles = (uint32_t) n_tiles;
    res->tile_stride = stride;
    res->have_minimizers = false;
    const size_t hb = 4 * (size_t) hit_cap;
    res->s_read.view(out->hit_read, hb);
    res->s_key.view(out->hit_key, hb);
    res->s_pos.view(out->hit_pos, hb);
    res->s_rank.view(out->hit_rank, hb);
    res->s_rec.view(out->hit_rec, hb);
    res->k_rows.view(out->cluster_rows, 32 * (size_t) cl_cap);
    // per-prg size threshold (src/utils.cpp:211-213, 219-220) through the pinned staging buffer
    if (res->thr_serial != idx->serial || res->thr_fraction != prm->fraction || res->thr_min != prm->min_cluster_size ||
        res->thr_exp != prm->expected_kmers) {
        HIP_TRY(hipStreamSynchronize(g_tail));        // an earlier copy out of the staging buffer has finished
        uint32_t lowest = idx->n_prg ? 0xFFFFFFFFu : 0u;
        for (uint32_t p = 0; p < idx->n_prg; ++p) {
            na.thr_stage[p] = pmap_threshold(idx->min_path_length[p], prm->expected_kmers, prm->fraction,
                                             prm->min_cluster_size);
            lowest = std::min(lowest, na.thr_stage[p]);
        }
        res->thr_minval = lowest;
        na.thr_stage[idx->n_prg] = 0;
        HIP_TRY(hipMemcpyAsync(res->thr.p, na.thr_stage, 4 * ((size_t) idx->n_prg + 1), hipMemcpyHostToDevice, g_tail));
        HIP_TRY(hipStreamSynchronize(g_tail));        // the kernels that read the table may run on the other tail stream
        res->thr_serial = idx->serial;
        res->thr_fraction = prm->fraction;
        res->thr_min = prm->min_cluster_size;
        res->thr_exp = prm->expected_kmers;
    }
    // the fused tail kernels keep a row slot per (smallest threshold + 1) hits: with thresholds of zero the
    // slots the array has room for bound the hits
    if (hit_cap / (res->thr_minval + 1) + 2 > NA_ROWS_TMP)
        hit_cap = (uint32_t) ((NA_ROWS_TMP - 2) * (uint64_t) (res->thr_minval + 1) / CUT_BLOCK * CUT_BLOCK);
    res->hit_cap = hit_cap;
    res->cl_cap = cl_cap;
    res->p_idx = idx;
    res->p_rd = rd;
    res->p_prm = *prm;
    res->p_prm.profile = 0;
    res->p_fused = true;
    res->p_last = PMAP_STAGE_CLUSTER;
    // as pmap_map_wait: the batch is mapped again without a chain between workgroups when a chained prefix sum
    // gave up, with the kernel for tiles full of minimizers when the usual one left a tile out, with the
    // long-segment sort kernels when a read had more hits than the rank counting takes
    for (;;) {
        if ((rc = run_pipeline(res))) { res->serial_scans = false; return rc; }
        HIP_TRY(hipEventSynchronize(res->done));
        if (res->hc->fail) {
            if (res->serial_scans) {
                res->serial_scans = false;
                return pmap_fail(PMAP_EHIP, "the cluster kernels failed without a chained prefix sum");
            }
            res->serial_scans = true;
            continue;
        }
        res->serial_scans = false;
        if (res->hc->dense_marks && !res->dense_launched) {
            res->dense_enabled = true;
            continue;
        }
        if (res->hc->has_long && !res->long_launched) {
            res->long_enabled = true;
            continue;
        }
        break;
    }
    res->dense_enabled = res->hc->dense_marks != 0;
    res->long_enabled = res->hc->has_long != 0;
    res->wave_enabled = res->hc->has_wave != 0;
    const Counters &hc = *res->hc;
    counts->n_reads = rd->n_reads;
    counts->n_bases = rd->n_bases;
    counts->n_minimizers = hc.n_mini;
    counts->n_hits = hc.n_hits;
    counts->n_clusters_all = hc.n_cl;
    counts->n_clusters = hc.n_kept;
    memset(counts->ms, 0, sizeof(counts->ms));
    if (hc.n_hits > hit_cap)
        return pmap_fail(PMAP_ELIMIT, "the batch has more hits than the output buffers (or pmap_map_noalloc's 2^20) hold; "
                                      "the counts say how many");
    if ((res->staged_tail ? hc.n_cl : hc.n_kept) > cl_cap)
        return pmap_fail(PMAP_ELIMIT, "the batch has more clusters than the output buffers (or pmap_map_noalloc's 2^20) hold; "
                                      "the counts say how many");
    return PMAP_OK;
}

// every accessor of a result first waits for a run that is still in flight
static int settled(const pmap_result_t *res) {
    if (!res) return pmap_fail(PMAP_EINVAL, "null argument");
    if (int rc = bind(res->ctx)) return rc;      // the copies that follow run on the result's GPU
    return res->pending ? pmap_map_wait(const_cast<pmap_result_t *>(res)) : PMAP_OK;
}

extern "C" int pmap_result_counts(const pmap_result_t *res, pmap_counts_t *out) {
    if (!res || !out) return pmap_fail(PMAP_EINVAL, "null argument");
    if (int rc = settled(res)) return rc;
    *out = res->counts;
    return PMAP_OK;
}

extern "C" int pmap_result_kernel_ms(const pmap_result_t *res, float *ms) {
    if (!res || !ms) return pmap_fail(PMAP_EINVAL, "null argument");
    if (int rc = settled(res)) return rc;
    memcpy(ms, res->kernel_ms, sizeof(res->kernel_ms));
    return PMAP_OK;
}

extern "C" const char *pmap_kernel_name(int i) {
    if (i < 0 || i >= PMAP_N_KERNELS) return "";
    return kernel_names[i];
}

template <typename T>
static int fetch(const void *dev, size_t n, std::vector<T> *out) {
    out->resize(n);
    if (n) HIP_TRY(hipMemcpy(out->data(), dev, n * sizeof(T), hipMemcpyDeviceToHost));
    return PMAP_OK;
}

extern "C" int pmap_result_minimizers(const pmap_result_t *cres, uint64_t *kmer, uint32_t *read_id, uint32_t *pos,
                                      uint8_t *strand) {
    if (!cres) return pmap_fail(PMAP_EINVAL, "null argument");
    if (int rc0 = settled(cres)) return rc0;
    pmap_result *res = const_cast<pmap_result *>(cres);
    if (!res->have_minimizers)
        return pmap_fail(PMAP_EINVAL, "the complete minimizer list was not kept: map with PMAP_FLAG_UNFUSED or last_stage = PMAP_STAGE_SKETCH");
    const size_t n = res->counts.n_minimizers;
    if (n == 0) return PMAP_OK;
    int rc;
    // gather the tile slots into one list
    DevBuf nmini, moff, okm, ord, ops;
    auto done = [&](int code) {
        nmini.release(); moff.release(); okm.release(); ord.release(); ops.release();
        return code;
    };
    if ((rc = nmini.ensure(sizeof(uint32_t) * (res->n_tiles + 1))) ||
        (rc = moff.ensure(sizeof(uint32_t) * (res->n_tiles + 1))) || (rc = okm.ensure(sizeof(uint64_t) * n)) ||
        (rc = ord.ensure(sizeof(uint32_t) * n)) || (rc = ops.ensure(sizeof(uint32_t) * n)) ||
        (rc = res->scratch_total.ensure(sizeof(unsigned long long))))
        return done(rc);
    g_cur = res->ctx->tail;
    hipLaunchKernelGGL(tile_field_kernel, dim3((res->n_tiles + 255) / 256), dim3(256), 0, g_cur,
                       res->tile_cnt.as<uint4>(), res->n_tiles, 0u, nmini.as<uint32_t>());
    if ((rc = device_scan<false>(res, nmini.as<uint32_t>(), moff.as<uint32_t>(), res->n_tiles,
                                 res->scratch_total.as<unsigned long long>())))
        return done(rc);
    hipLaunchKernelGGL(compact_minimizers_kernel, dim3((res->n_tiles + 3) / 4), dim3(256), 0, g_cur,
                       nmini.as<uint32_t>(), moff.as<uint32_t>(), res->n_tiles, res->tile_stride, res->mkmer.as<uint64_t>(),
                       res->ents.as<uint4>(), okm.as<uint64_t>(), ord.as<uint32_t>(),
                       ops.as<uint32_t>());
    if (hipStreamSynchronize(g_cur) != hipSuccess) return done(pmap_fail(PMAP_EHIP, "minimizer gather failed"));
    if (kmer && hipMemcpy(kmer, okm.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess)
        return done(pmap_fail(PMAP_EHIP, "copy failed"));
    std::vector<uint32_t> v;
    if (read_id) {
        if ((rc = fetch(ord.p, n, &v))) return done(rc);
        for (size_t i = 0; i < n; ++i) read_id[i] = v[i] + res->first_read_id;
    }
    if (pos || strand) {
        if ((rc = fetch(ops.p, n, &v))) return done(rc);
        for (size_t i = 0; i < n; ++i) {
            if (pos) pos[i] = v[i] & 0x7FFFFFFFu;
            if (strand) strand[i] = (uint8_t) (v[i] >> 31);
        }
    }
    return done(PMAP_OK);
}

extern "C" int pmap_result_hits(const pmap_result_t *res, int sorted, uint32_t *read_id, uint32_t *pos,
                                uint32_t *prg_id, uint32_t *rec, uint8_t *is_forward) {
    if (!res) return pmap_fail(PMAP_EINVAL, "null argument");
    if (int rc0 = settled(res)) return rc0;
    const size_t n = res->counts.n_hits;
    if (n == 0) return PMAP_OK;
    int rc;
    std::vector<uint32_t> v;
    if (read_id) {
        if ((rc = fetch((sorted ? res->s_read : res->h_read).p, n, &v))) return rc;
        for (size_t i = 0; i < n; ++i) read_id[i] = v[i] + res->first_read_id;
    }
    if (pos) HIP_TRY(hipMemcpy(pos, (sorted ? res->s_pos : res->h_pos).p, n * 4, hipMemcpyDeviceToHost));
    if (rec) HIP_TRY(hipMemcpy(rec, (sorted ? res->s_rec : res->h_rec).p, n * 4, hipMemcpyDeviceToHost));
    if (prg_id || is_forward) {
        if ((rc = fetch((sorted ? res->s_key : res->h_key).p, n, &v))) return rc;
        for (size_t i = 0; i < n; ++i) {
            if (prg_id) prg_id[i] = v[i] >> 1;
            if (is_forward) is_forward[i] = (uint8_t) (1u - (v[i] & 1u));
        }
    }
    return PMAP_OK;
}

// (library-internal) the path ranks of the sorted hits, for the read-file pipeline: one more column from the device
// costs less than a lookup in the index's rank table per hit on the host
int pmap_result_hit_ranks(const pmap_result *res, uint32_t *rank) {
    if (!res || !rank) return pmap_fail(PMAP_EINVAL, "null argument");
    if (int rc0 = settled(res)) return rc0;
    const size_t n = res->counts.n_hits;
    if (n) HIP_TRY(hipMemcpy(rank, res->s_rank.p, n * 4, hipMemcpyDeviceToHost));
    return PMAP_OK;
}

extern "C" int pmap_result_clusters(const pmap_result_t *res, uint64_t *begin, uint32_t *size, uint32_t *read_id,
                                    uint32_t *prg_id, uint8_t *is_forward, uint32_t *first_pos, uint32_t *last_pos,
                                    uint32_t *first_rank) {
    if (!res) return pmap_fail(PMAP_EINVAL, "null argument");
    if (int rc0 = settled(res)) return rc0;
    const size_t n = res->counts.n_clusters;
    std::vector<uint32_t> rows;
    int rc;
    if ((rc = fetch(res->k_rows.p, n * 8, &rows))) return rc;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t *r = rows.data() + i * 8;
        if (begin) begin[i] = r[0];
        if (size) size[i] = r[1];
        if (read_id) read_id[i] = r[2] + res->first_read_id;
        if (prg_id) prg_id[i] = r[3] >> 1;
        if (is_forward) is_forward[i] = (uint8_t) (1u - (r[3] & 1u));
        if (first_pos) first_pos[i] = r[4];
        if (last_pos) last_pos[i] = r[5];
        if (first_rank) first_rank[i] = r[6];
    }
    return PMAP_OK;
}

// ---------------------------------------------------------------------------
// coverage

struct pmap_coverage {
    DevCtx *ctx = nullptr;
    uint32_t n_recs = 0;
    DevBuf covg, weight;
};

extern "C" int pmap_coverage_create(const pmap_index_t *idx, pmap_coverage_t **out) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!idx || !out) return pmap_fail(PMAP_EINVAL, "null argument");
    if (idx->n_recs >= 0x7FFFFFFFull) return pmap_fail(PMAP_ELIMIT, "too many records for 32-bit record indices");
    pmap_coverage *cov = new pmap_coverage();
    cov->ctx = t_ctx;
    hipStream_t g_stream = t_ctx->tail;
    cov->n_recs = (uint32_t) idx->n_recs;
    const size_t bytes = sizeof(uint32_t) * 2 * ((size_t) cov->n_recs + 1);
    if ((rc = cov->covg.ensure(bytes))) {
        delete cov;
        return rc;
    }
    HIP_TRY(hipMemsetAsync(cov->covg.p, 0, bytes, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    *out = cov;
    return PMAP_OK;
}

extern "C" int pmap_coverage_add(pmap_coverage_t *cov, const pmap_result_t *res, const uint32_t *weight) {
    if (!cov || !res) return pmap_fail(PMAP_EINVAL, "null argument");
    if (int rc0 = settled(res)) return rc0;
    const uint32_t n = (uint32_t) res->counts.n_clusters;
    if (n == 0) return PMAP_OK;
    if (!weight) return pmap_fail(PMAP_EINVAL, "null argument");
    if (res->p_idx == nullptr || res->p_idx->n_recs != cov->n_recs)
        return pmap_fail(PMAP_EINVAL, "the result was mapped against another index than the coverage belongs to");
    if (res->ctx != cov->ctx && res->ctx->device != cov->ctx->device)
        return pmap_fail(PMAP_EINVAL, "the result and the coverage counters live on different GPUs");
    hipStream_t g_stream = res->ctx->tail;
    int rc;
    if ((rc = cov->weight.ensure(sizeof(uint32_t) * n))) return rc;
    // blocking copy: the caller's array may change as soon as this call returns
    HIP_TRY(hipMemcpy(cov->weight.p, weight, sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(coverage_kernel, dim3((n + 3) / 4), dim3(256), 0, g_stream, res->k_rows.as<uint32_t>(),
                       cov->weight.as<uint32_t>(), n, res->s_rec.as<uint32_t>(), cov->n_recs, cov->covg.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    // the weights may be overwritten by the next call: wait (the kernel is short)
    HIP_TRY(hipStreamSynchronize(g_stream));
    return PMAP_OK;
}

extern "C" int pmap_coverage_fetch(const pmap_coverage_t *cov, uint32_t *forward, uint32_t *reverse) {
    if (!cov) return pmap_fail(PMAP_EINVAL, "null argument");
    if (int rc = bind(cov->ctx)) return rc;
    HIP_TRY(hipDeviceSynchronize());     // adds may have come from results of several contexts on this GPU
    const size_t b = sizeof(uint32_t) * cov->n_recs;
    if (forward && b) HIP_TRY(hipMemcpy(forward, cov->covg.p, b, hipMemcpyDeviceToHost));
    if (reverse && b) HIP_TRY(hipMemcpy(reverse, cov->covg.as<uint32_t>() + cov->n_recs, b, hipMemcpyDeviceToHost));
    return PMAP_OK;
}

extern "C" void pmap_coverage_destroy(pmap_coverage_t *cov) {
    if (!cov) return;
    if (cov->ctx) (void) bind(cov->ctx);
    cov->covg.release();
    cov->weight.release();
    delete cov;
}

extern "C" void pmap_result_destroy(pmap_result_t *res) {
    if (!res) return;
    if (res->ctx) (void) bind(res->ctx);
    if (res->pending) (void) hipEventSynchronize(res->done);
    for (DevBuf *b : res->all()) b->release();
    if (res->events) {
        for (auto &e : res->kev) (void) hipEventDestroy(e);
        (void) hipEventDestroy(res->done);
        (void) hipEventDestroy(res->sketched);
        (void) hipHostFree(res->hc);
    }
    delete res;
}
