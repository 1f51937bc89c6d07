// HIP kernels (gfx950) and device-side C ABI of libpandora_amd: the read ->
// PRG minimizer path of `pandora map` on one MI355X.
//
//   encode   nt4 (src/inthash.cpp:42-63) -> 2-bit packed reads + per-read "bad letter" flag
//   sketch   Seq::minimizer_sketch (src/seq.cpp:91-122) in its closed form: k-mer j of a
//            read with n k-mers is a minimizer iff some full window [i, i+w) within [0, n)
//            that contains j has minimum h[j] (every tie kept)
//   probe    Index::minhash lookup of add_read_hits (src/utils.cpp:174-176): one 16-byte slot per probe
//   emit     MinimizerHit construction (src/minihit.cpp:14-22): count / prefix sum / scatter
//   sort     MinimizerHits::sort (src/minihits.cpp:42-53, order src/minihit.cpp:44-66)
//   cluster  define_clusters (src/utils.cpp:189-249): cut flags, prefix sum, size threshold
//
// All kernels are integer / byte work bounded by HBM bandwidth or VALU rate; none uses MFMA.
#include <hip/hip_runtime.h>

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

static hipStream_t g_stream = nullptr;
static bool g_ready = false;

extern "C" int pmap_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return pmap_fail(PMAP_EHIP, "no HIP device: libpandora_amd needs an AMD GPU (gfx950), there is no CPU fallback");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    if (!g_stream) HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_ready = true;
    return PMAP_OK;
}

static int ensure_init() {
    if (g_ready) return PMAP_OK;
    return pmap_init(-1);
}

// ---------------------------------------------------------------------------
// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PMAP_OK;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return PMAP_OK;
    }
    void release() {
        if (p) (void) hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T *as() const { return (T *) p; }
};

// ---------------------------------------------------------------------------
// device helpers

// src/inthash.cpp:42-63
__device__ __forceinline__ uint32_t nt4_dev(uint32_t c) {
    if (c < 4) return c;
    uint32_t u = c & 0xDFu;
    return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u;
}

// src/inthash.cpp:115-125.  For HT = uint32_t the mask has at most 30 bits (k <= 15): every
// step is masked to 2k bits, so arithmetic modulo 2^32 gives the same low 2k bits as modulo 2^64.
template <typename HT>
__device__ __forceinline__ HT hash64_dev(HT key, HT mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    if (sizeof(HT) == 8) key = (key + (key << 31)) & mask;
    // for a <=30-bit mask (key << 31) & mask == 0
    return key;
}

// The k-mer starting at bit offset `sh` (even, < 32) of the big-endian 2-bit stream w0:w1:w2.
__device__ __forceinline__ uint64_t kmer_at_u64(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t sh, uint32_t k) {
    uint64_t hi = ((uint64_t) w0 << 32) | w1;
    uint64_t x = hi << sh;
    if (sh) x |= (uint64_t) w2 >> (32 - sh);
    return x >> (64 - 2 * k);
}
__device__ __forceinline__ uint32_t kmer_at_u32(uint32_t w0, uint32_t w1, uint32_t sh, uint32_t k) {
    uint64_t hi = ((uint64_t) w0 << 32) | w1;
    return (uint32_t) ((hi << sh) >> (64 - 2 * k));
}

// reverse complement of a k-mer held in the low 2k bits (src/seq.cpp:44)
__device__ __forceinline__ uint64_t revcomp_u64(uint64_t fwd, uint32_t k) {
    uint64_t r = __brevll(~fwd);
    r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
    return r >> (64 - 2 * k);
}
__device__ __forceinline__ uint32_t revcomp_u32(uint32_t fwd, uint32_t k) {
    uint32_t r = __brev(~fwd);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    return r >> (32 - 2 * k);
}

template <typename HT>
struct Canon {
    HT h;
    uint32_t strand;
};

// canonical hash and strand of a k-mer (src/seq.cpp:109-111)
__device__ __forceinline__ Canon<uint64_t> canon_u64(uint64_t fwd, uint32_t k, uint64_t mask) {
    uint64_t hf = hash64_dev<uint64_t>(fwd, mask), hr = hash64_dev<uint64_t>(revcomp_u64(fwd, k), mask);
    Canon<uint64_t> c;
    c.h = hf < hr ? hf : hr;
    c.strand = hf <= hr;
    return c;
}
__device__ __forceinline__ Canon<uint32_t> canon_u32(uint32_t fwd, uint32_t k, uint32_t mask) {
    uint32_t hf = hash64_dev<uint32_t>(fwd, mask), hr = hash64_dev<uint32_t>(revcomp_u32(fwd, k), mask);
    Canon<uint32_t> c;
    c.h = hf < hr ? hf : hr;
    c.strand = hf <= hr;
    return c;
}

// exclusive scan of one value per thread over a 256-thread block; *total = block sum
__device__ __forceinline__ uint32_t block_scan_256(uint32_t v, uint32_t *lds4, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(incl, d);
        if (lane >= (uint32_t) d) incl += t;
    }
    if (lane == 63) lds4[wv] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        uint32_t s = lds4[i];
        if (i < wv) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + incl - v;
}

// ---------------------------------------------------------------------------
// device-wide exclusive scan of uint32 (optionally of popcounts), three launches
static const uint32_t SCAN_TILE = 2048;   // 256 threads x 8

template <bool POPC>
__global__ __launch_bounds__(256) void scan_reduce_kernel(const uint32_t *__restrict__ in, uint64_t n,
                                                          uint32_t *__restrict__ bsum) {
    __shared__ uint32_t lds4[4];
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    uint32_t s = 0;
    if (base + 8 <= n) {
        uint4 a = *(const uint4 *) (in + base), b = *(const uint4 *) (in + base + 4);
        if (POPC) s = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
        else s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    } else {
        for (uint32_t i = 0; i < 8; ++i)
            if (base + i < n) s += POPC ? __popc(in[base + i]) : in[base + i];
    }
    uint32_t tot;
    block_scan_256(s, lds4, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// one block: exclusive scan of the block sums in place, grand total to *total
__global__ __launch_bounds__(256) void scan_spine_kernel(uint32_t *__restrict__ bsum, uint32_t nblk,
                                                         uint64_t *__restrict__ total) {
    __shared__ uint32_t lds4[4];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < nblk; base += 256) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < nblk ? bsum[i] : 0, tot;
        uint32_t ex = block_scan_256(v, lds4, &tot);
        if (i < nblk) bsum[i] = (uint32_t) (carry + ex);
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

template <bool POPC>
__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t *__restrict__ in, uint64_t n,
                                                         const uint32_t *__restrict__ bsum,
                                                         uint32_t *__restrict__ out) {
    __shared__ uint32_t lds4[4];
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    uint32_t v[8];
    if (base + 8 <= n) {
        uint4 a = *(const uint4 *) (in + base), b = *(const uint4 *) (in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) v[i] = base + i < n ? in[base + i] : 0;
    }
    uint32_t s = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
        if (POPC) v[i] = __popc(v[i]);
        s += v[i];
    }
    uint32_t tot;
    uint32_t ex = block_scan_256(s, lds4, &tot) + bsum[blockIdx.x];
    uint32_t o[8];
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) { o[i] = ex; ex += v[i]; }
    if (base + 8 <= n) {
        *(uint4 *) (out + base) = make_uint4(o[0], o[1], o[2], o[3]);
        *(uint4 *) (out + base + 4) = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i)
            if (base + i < n) out[base + i] = o[i];
    }
}

// ---------------------------------------------------------------------------
// encode: one thread per packed word (16 bases)
__global__ __launch_bounds__(256) void encode_kernel(const uint32_t *__restrict__ ascii32,
                                                     const uint64_t *__restrict__ boff,
                                                     const uint32_t *__restrict__ woff,
                                                     const uint32_t *__restrict__ rlen, uint32_t n_reads,
                                                     uint32_t n_words, uint32_t *__restrict__ packed,
                                                     uint32_t *__restrict__ wread, uint32_t *__restrict__ rflag) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_words) return;
    // the read that owns word t: the largest r with woff[r] <= t (woff[n_reads] = n_words > t)
    uint32_t lo = 0, hi = n_reads;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (woff[mid] <= t) lo = mid; else hi = mid;
    }
    const uint32_t r = lo;
    const uint32_t j0 = (t - woff[r]) * 16, len = rlen[r];
    const uint64_t a = boff[r] + j0;
    const uint64_t wi = a >> 2;
    const uint32_t sh = (uint32_t) (a & 3) * 8;
    uint32_t x[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) x[q] = ascii32[wi + q];
    uint32_t word = 0, bad = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t v = sh ? (x[q] >> sh) | (x[q + 1] << (32 - sh)) : x[q];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t i = q * 4 + b;
            uint32_t code = nt4_dev((v >> (8 * b)) & 0xFFu);
            if (j0 + i >= len) code = 0;
            else if (code > 3) { bad = 1; code = 0; }
            word |= code << (30 - 2 * i);
        }
    }
    packed[t] = word;
    wread[t] = r;
    if (bad) rflag[r] = 1;   // every writer stores the same value
}

// ---------------------------------------------------------------------------
// sketch, pass A: one workgroup per tile of 1024 k-mer start positions of the padded base
// stream; writes one flag bit per position ("this k-mer is in the read's sketch").
static const int SK_TILE = 1024;
static const int SK_HALO = 64;                                        // >= w-1
static const int SK_NW = (SK_HALO + SK_TILE + SK_HALO + 32) / 16 + 3; // packed words held in LDS
static const int SK_NH = SK_TILE + 2 * SK_HALO;

template <typename HT>
__global__ __launch_bounds__(256) void sketch_flags_kernel(const uint32_t *__restrict__ packed,
                                                           const uint32_t *__restrict__ wread,
                                                           const uint32_t *__restrict__ woff,
                                                           const uint32_t *__restrict__ rlen,
                                                           const uint32_t *__restrict__ rflag,
                                                           uint32_t n_words, uint32_t w, uint32_t k,
                                                           unsigned long long *__restrict__ flags64) {
    __shared__ uint32_t W[SK_NW];
    __shared__ uint32_t mstart[SK_NW];   // first base (global position) of the read owning the word
    __shared__ uint32_t mlen[SK_NW];     // its length, 0 if the read is unusable
    __shared__ HT h[SK_NH];
    __shared__ uint32_t jv[SK_NH];       // k-mer index within its read, 0xFFFFFFFF if not a k-mer
    __shared__ HT Lv[SK_NH];

    const HT INVALID = (HT) ~(HT) 0;
    const uint32_t tid = threadIdx.x;
    const int64_t T0 = (int64_t) blockIdx.x * SK_TILE;
    const int64_t wbase = (T0 >> 4) - (SK_HALO >> 4);
    for (int i = tid; i < SK_NW; i += 256) {
        int64_t wi = wbase + i;
        uint32_t word = 0, ms = 0, ml = 0;
        if (wi >= 0 && wi < (int64_t) n_words) {
            word = packed[wi];
            uint32_t r = wread[wi];
            ms = woff[r] * 16;
            ml = rflag[r] ? 0 : rlen[r];
        }
        W[i] = word;
        mstart[i] = ms;
        mlen[i] = ml;
    }
    __syncthreads();

    const HT mask = (HT) (((uint64_t) 1 << (2 * k)) - 1);
    const int nh = SK_TILE + 2 * ((int) w - 1);
    const int64_t p0 = T0 - ((int64_t) w - 1);       // global position of local index 0
    const int q0 = SK_HALO - ((int) w - 1);          // offset of local index 0 inside the LDS words
    for (int i = tid; i < nh; i += 256) {
        const int q = q0 + i;
        const int wi = q >> 4;
        const uint32_t sh = (uint32_t) (q & 15) * 2;
        const int64_t p = p0 + i;
        HT hv = INVALID;
        uint32_t j = 0xFFFFFFFFu;
        const uint32_t len = mlen[wi];
        // reads shorter than w+k-1 have no sketch (src/seq.cpp:92-94)
        if (p >= 0 && len + 1 >= w + k) {
            const uint32_t jj = (uint32_t) p - mstart[wi];
            if (jj + k <= len) {
                j = jj;
                if (sizeof(HT) == 4) hv = (HT) canon_u32(kmer_at_u32(W[wi], W[wi + 1], sh, k), k, (uint32_t) mask).h;
                else hv = (HT) canon_u64(kmer_at_u64(W[wi], W[wi + 1], W[wi + 2], sh, k), k, (uint64_t) mask).h;
            }
        }
        h[i] = hv;
        jv[i] = j;
    }
    __syncthreads();

    // Lv[t] = minimum of the full window ending at k-mer t, 0 if that window is not inside one read
    for (int t = tid + (int) w - 1; t < nh; t += 256) {
        HT m = 0;
        const uint32_t j = jv[t];
        if (j != 0xFFFFFFFFu && j >= w - 1) {
            m = h[t];
            for (uint32_t d = 1; d < w; ++d) {
                HT o = h[t - d];
                m = o < m ? o : m;
            }
        }
        Lv[t] = m;
    }
    __syncthreads();

    // k-mer i is a minimizer iff it equals the minimum of one of the (full) windows containing it,
    // i.e. the largest of those minima
#pragma unroll
    for (int c = 0; c < SK_TILE / 256; ++c) {
        const int lp = tid + 256 * c;
        const int i = lp + (int) w - 1;
        const HT hv = h[i];
        bool f = false;
        if (jv[i] != 0xFFFFFFFFu) {
            HT mx = Lv[i];
            for (uint32_t d = 1; d < w; ++d) {
                HT o = Lv[i + d];
                mx = o > mx ? o : mx;
            }
            f = (mx == hv);
        }
        unsigned long long b = __ballot(f);
        if ((tid & 63) == 0) flags64[(T0 + lp) >> 6] = b;
    }
}

// sketch, pass B: one thread per 32-position flag word; writes the minimizers in (read, position) order
template <typename HT>
__global__ __launch_bounds__(256) void sketch_emit_kernel(const uint32_t *__restrict__ flags,
                                                          const uint32_t *__restrict__ foffs, uint32_t n_fw,
                                                          const uint32_t *__restrict__ packed,
                                                          const uint32_t *__restrict__ wread,
                                                          const uint32_t *__restrict__ woff, uint32_t k,
                                                          uint64_t *__restrict__ mkmer, uint32_t *__restrict__ mread,
                                                          uint32_t *__restrict__ mps) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_fw) return;
    uint32_t bits = flags[g];
    if (!bits) return;
    uint32_t o = foffs[g];
    const HT mask = (HT) (((uint64_t) 1 << (2 * k)) - 1);
    while (bits) {
        const uint32_t b = __ffs(bits) - 1;
        bits &= bits - 1;
        const uint32_t p = g * 32 + b;
        const uint32_t wi = p >> 4, sh = (p & 15) * 2;
        const uint32_t r = wread[wi];
        const uint32_t j = p - woff[r] * 16;
        uint64_t hv;
        uint32_t strand;
        if (sizeof(HT) == 4) {
            Canon<uint32_t> c = canon_u32(kmer_at_u32(packed[wi], packed[wi + 1], sh, k), k, (uint32_t) mask);
            hv = c.h; strand = c.strand;
        } else {
            Canon<uint64_t> c = canon_u64(kmer_at_u64(packed[wi], packed[wi + 1], packed[wi + 2], sh, k), k, (uint64_t) mask);
            hv = c.h; strand = c.strand;
        }
        mkmer[o] = hv;
        mread[o] = r;
        mps[o] = j | (strand << 31);
        ++o;
    }
}

// ---------------------------------------------------------------------------
// probe: one thread per minimizer, linear probing over 16-byte slots
__global__ __launch_bounds__(256) void probe_kernel(const uint64_t *__restrict__ mkmer, uint32_t n_mini,
                                                    const PmapSlot *__restrict__ slots, uint32_t log2cap,
                                                    uint32_t *__restrict__ mcnt, uint32_t *__restrict__ mroff) {
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n_mini) return;
    const uint64_t key = mkmer[m];
    const uint32_t capmask = (1u << log2cap) - 1;
    uint32_t s = (uint32_t) ((key * 0x9E3779B97F4A7C15ULL) >> (64 - log2cap));
    uint32_t cnt = 0, off = 0;
    for (uint32_t step = 0; step <= capmask; ++step) {   // the table is at most half full: ends at an empty slot
        const uint4 v = *(const uint4 *) (slots + s);
        const uint64_t sk = ((uint64_t) v.y << 32) | v.x;
        if (sk == key) { off = v.z; cnt = v.w; break; }
        if (sk == PMAP_EMPTY_KEY) break;
        s = (s + 1) & capmask;
    }
    mcnt[m] = cnt;
    mroff[m] = off;
}

// scatter: one thread per minimizer writes its hits (MinimizerHit ctor, src/minihit.cpp:14-22)
__global__ __launch_bounds__(256) void scatter_kernel(uint32_t n_mini, const uint32_t *__restrict__ mread,
                                                      const uint32_t *__restrict__ mps,
                                                      const uint32_t *__restrict__ mcnt,
                                                      const uint32_t *__restrict__ mroff,
                                                      const uint32_t *__restrict__ mhoff,
                                                      const PmapRec *__restrict__ recs, uint32_t *__restrict__ h_read,
                                                      uint32_t *__restrict__ h_key, uint32_t *__restrict__ h_pos,
                                                      uint32_t *__restrict__ h_rank, uint32_t *__restrict__ h_rec) {
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n_mini) return;
    const uint32_t c = mcnt[m];
    if (!c) return;
    const uint32_t ro = mroff[m], ho = mhoff[m], r = mread[m], ps = mps[m];
    const uint32_t pos = ps & 0x7FFFFFFFu, strand = ps >> 31;
    for (uint32_t q = 0; q < c; ++q) {
        const uint4 rec = *(const uint4 *) (recs + ro + q);   // prg, knode<<1|strand, rank, orig
        const uint32_t fwd = (strand == (rec.y & 1u));
        h_read[ho + q] = r;
        h_key[ho + q] = rec.x << 1 | (fwd ? 0u : 1u);         // forward hits sort first (src/minihit.cpp:54-55)
        h_pos[ho + q] = pos;
        h_rank[ho + q] = rec.z;
        h_rec[ho + q] = rec.w;
    }
}

// first / one-past-last hit of every read that has hits
__global__ __launch_bounds__(256) void seg_bounds_kernel(const uint32_t *__restrict__ h_read, uint32_t n_hits,
                                                         uint32_t *__restrict__ hstart, uint32_t *__restrict__ hend) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_hits) return;
    const uint32_t r = h_read[i];
    if (i == 0) hstart[r] = 0;
    else {
        const uint32_t rp = h_read[i - 1];
        if (rp != r) { hstart[r] = i; hend[rp] = i; }
    }
    if (i == n_hits - 1) hend[r] = n_hits;
}

// ---------------------------------------------------------------------------
// sort: hits arrive ordered by (read, position, prg, path rank).  A stable sort of each read's
// segment by key = prg << 1 | !forward yields MinimizerHit::operator< order.  Each hit counts the
// hits of its read that must precede it (rank by counting) and is written to that place.
static const int SORT_CH = 2048;

__global__ __launch_bounds__(256) void sort_rank_kernel(uint32_t n_hits, const uint32_t *__restrict__ h_read,
                                                        const uint32_t *__restrict__ h_key,
                                                        const uint32_t *__restrict__ h_pos,
                                                        const uint32_t *__restrict__ h_rank,
                                                        const uint32_t *__restrict__ h_rec,
                                                        const uint32_t *__restrict__ hstart,
                                                        const uint32_t *__restrict__ hend,
                                                        uint32_t *__restrict__ s_read, uint32_t *__restrict__ s_key,
                                                        uint32_t *__restrict__ s_pos, uint32_t *__restrict__ s_rank,
                                                        uint32_t *__restrict__ s_rec) {
    __shared__ uint32_t keys[SORT_CH];
    __shared__ uint32_t range[2];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n_hits;
    uint32_t r = 0, key = 0, s = 0, e = 0;
    if (active) {
        r = h_read[i];
        key = h_key[i];
        s = hstart[r];
        e = hend[r];
    }
    if (threadIdx.x == 0) range[0] = s;
    if (active && (threadIdx.x == 255 || i == n_hits - 1)) range[1] = e;
    __syncthreads();
    const uint32_t smin = range[0], emax = range[1];
    uint32_t c = 0;
    for (uint32_t base = smin; base < emax; base += SORT_CH) {
        const uint32_t n = min((uint32_t) SORT_CH, emax - base);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < n; t += 256) keys[t] = h_key[base + t];
        __syncthreads();
        if (active) {
            const uint32_t a = max(s, base), b = min(e, base + n);
            const uint32_t mid = min(max(i, a), b);   // [a, mid) precede hit i, [mid', b) follow it
            for (uint32_t j = a; j < mid; ++j) c += keys[j - base] <= key;
            for (uint32_t j = max(a, i + 1); j < b; ++j) c += keys[j - base] < key;
        }
    }
    if (active) {
        const uint32_t d = s + c;
        s_read[d] = r;
        s_key[d] = key;
        s_pos[d] = h_pos[i];
        s_rank[d] = h_rank[i];
        s_rec[d] = h_rec[i];
    }
}

// ---------------------------------------------------------------------------
// cluster cut (src/utils.cpp:206-209): one flag bit per sorted hit that starts a cluster
__global__ __launch_bounds__(256) void cluster_flags_kernel(uint32_t n_hits, const uint32_t *__restrict__ s_read,
                                                            const uint32_t *__restrict__ s_key,
                                                            const uint32_t *__restrict__ s_pos, int max_diff,
                                                            unsigned long long *__restrict__ cflags64) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    bool f = false;
    if (i < n_hits) {
        if (i == 0) f = true;
        else {
            const int d = (int) s_pos[i] - (int) s_pos[i - 1];
            f = s_read[i] != s_read[i - 1] || s_key[i] != s_key[i - 1] || (d < 0 ? -d : d) > max_diff;
        }
    }
    unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0) cflags64[i >> 6] = b;
}

// positions of the set bits of a flag array, in order (offs = exclusive scan of the popcounts)
__global__ __launch_bounds__(256) void bit_positions_kernel(const uint32_t *__restrict__ flags,
                                                            const uint32_t *__restrict__ offs, uint32_t n_fw,
                                                            uint32_t *__restrict__ out) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_fw) return;
    uint32_t bits = flags[g];
    uint32_t o = offs[g];
    while (bits) {
        const uint32_t b = __ffs(bits) - 1;
        bits &= bits - 1;
        out[o++] = g * 32 + b;
    }
}

// one thread per cluster: summary row + "kept" flag (size > threshold, src/utils.cpp:211-220)
__global__ __launch_bounds__(256) void cluster_table_kernel(uint32_t n_cl, uint32_t n_hits,
                                                            const uint32_t *__restrict__ cstart,
                                                            const uint32_t *__restrict__ s_read,
                                                            const uint32_t *__restrict__ s_key,
                                                            const uint32_t *__restrict__ s_pos,
                                                            const uint32_t *__restrict__ s_rank,
                                                            const uint32_t *__restrict__ thr,
                                                            uint32_t *__restrict__ c_size, uint32_t *__restrict__ c_read,
                                                            uint32_t *__restrict__ c_key, uint32_t *__restrict__ c_first,
                                                            uint32_t *__restrict__ c_last, uint32_t *__restrict__ c_rank,
                                                            unsigned long long *__restrict__ kflags64) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    bool kept = false;
    if (c < n_cl) {
        const uint32_t s = cstart[c];
        const uint32_t e = c + 1 < n_cl ? cstart[c + 1] : n_hits;
        const uint32_t size = e - s, key = s_key[s];
        c_size[c] = size;
        c_read[c] = s_read[s];
        c_key[c] = key;
        c_first[c] = s_pos[s];
        c_last[c] = s_pos[e - 1];
        c_rank[c] = s_rank[s];
        kept = size > thr[key >> 1];
    }
    unsigned long long b = __ballot(kept);
    if ((threadIdx.x & 63) == 0) kflags64[c >> 6] = b;
}

// gathers the rows of the kept clusters
__global__ __launch_bounds__(256) void cluster_gather_kernel(uint32_t n_kept, const uint32_t *__restrict__ kidx,
                                                             const uint32_t *__restrict__ cstart,
                                                             const uint32_t *__restrict__ c_size,
                                                             const uint32_t *__restrict__ c_read,
                                                             const uint32_t *__restrict__ c_key,
                                                             const uint32_t *__restrict__ c_first,
                                                             const uint32_t *__restrict__ c_last,
                                                             const uint32_t *__restrict__ c_rank,
                                                             uint32_t *__restrict__ k_rows) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_kept) return;
    const uint32_t c = kidx[t];
    uint32_t *row = k_rows + (uint64_t) t * 8;
    *(uint4 *) row = make_uint4(cstart[c], c_size[c], c_read[c], c_key[c]);
    *(uint4 *) (row + 4) = make_uint4(c_first[c], c_last[c], c_rank[c], c);
}

// ---------------------------------------------------------------------------
// host objects

struct pmap_reads {
    uint32_t n_reads = 0, first_read_id = 0;
    uint64_t n_bases = 0;
    uint32_t n_words = 0;      // packed words, each read padded to a multiple of 16 bases
    uint32_t n_tiles = 0;
    DevBuf ascii, boff, woff, rlen, rflag, packed, wread;
    uint64_t device_bytes() const {
        return ascii.cap + boff.cap + woff.cap + rlen.cap + rflag.cap + packed.cap + wread.cap;
    }
};

enum {
    KT_SKETCH_FLAGS = 0, KT_SKETCH_SCAN, KT_SKETCH_EMIT, KT_PROBE, KT_HIT_SCAN, KT_SCATTER, KT_SEG_BOUNDS,
    KT_SORT, KT_CLUSTER_FLAGS, KT_CLUSTER_SCAN, KT_CLUSTER_STARTS, KT_CLUSTER_TABLE, KT_KEPT_SCAN,
    KT_KEPT_GATHER, KT_COUNT
};
static const char *kernel_names[PMAP_N_KERNELS] = {
    "sketch_flags", "sketch_scan", "sketch_emit", "probe", "hit_scan", "scatter", "seg_bounds", "sort_rank",
    "cluster_flags", "cluster_scan", "cluster_starts", "cluster_table", "kept_scan", "kept_gather", "", ""};

struct pmap_result {
    pmap_counts_t counts;
    uint32_t first_read_id = 0;
    float kernel_ms[PMAP_N_KERNELS];
    // sketch
    DevBuf flags, foffs, bsum, total;
    DevBuf mkmer, mread, mps, mcnt, mroff, mhoff;
    // hits before / after sort
    DevBuf h_read, h_key, h_pos, h_rank, h_rec, hstart, hend;
    DevBuf s_read, s_key, s_pos, s_rank, s_rec;
    // clusters
    DevBuf cflags, coffs, cstart, c_size, c_read, c_key, c_first, c_last, c_rank, kflags, koffs, kidx, k_rows;
    DevBuf thr;
    const pmap_index *thr_idx = nullptr;
    float thr_fraction = 0;
    uint32_t thr_min = 0, thr_exp = 0;
    hipEvent_t ev[8];
    hipEvent_t kev[2 * PMAP_N_KERNELS];
    bool events = false;
    DevBuf *all[40] = {&flags, &foffs, &bsum, &total, &mkmer, &mread, &mps, &mcnt, &mroff, &mhoff, &h_read, &h_key,
                       &h_pos, &h_rank, &h_rec, &hstart, &hend, &s_read, &s_key, &s_pos, &s_rank, &s_rec, &cflags,
                       &coffs, &cstart, &c_size, &c_read, &c_key, &c_first, &c_last, &c_rank, &kflags, &koffs, &kidx,
                       &k_rows, &thr, nullptr, nullptr, nullptr, nullptr};
};

// ---------------------------------------------------------------------------
// index upload

int pmap_index_upload(pmap_index *idx) {
    int rc = ensure_init();
    if (rc) return rc;
    if (idx->d_slots) return PMAP_OK;
    size_t sb = idx->slots.size() * sizeof(PmapSlot), rb = (idx->recs.size() + 1) * sizeof(PmapRec);
    HIP_TRY(hipMalloc((void **) &idx->d_slots, sb));
    HIP_TRY(hipMalloc((void **) &idx->d_recs, rb));
    HIP_TRY(hipMemcpy(idx->d_slots, idx->slots.data(), sb, hipMemcpyHostToDevice));
    if (!idx->recs.empty())
        HIP_TRY(hipMemcpy(idx->d_recs, idx->recs.data(), idx->recs.size() * sizeof(PmapRec), hipMemcpyHostToDevice));
    idx->device_bytes = sb + rb;
    return PMAP_OK;
}

extern "C" int pmap_index_upload_now(pmap_index_t *idx) {
    if (!idx) return pmap_fail(PMAP_EINVAL, "null argument");
    return pmap_index_upload(idx);
}

void pmap_index_release_device(pmap_index *idx) {
    if (idx->d_slots) (void) hipFree(idx->d_slots);
    if (idx->d_recs) (void) hipFree(idx->d_recs);
    if (idx->d_thr) (void) hipFree(idx->d_thr);
    idx->d_slots = nullptr;
    idx->d_recs = nullptr;
    idx->d_thr = nullptr;
}

// ---------------------------------------------------------------------------
// reads

static int launch_encode(pmap_reads *rd) {
    HIP_TRY(hipMemsetAsync(rd->rflag.p, 0, sizeof(uint32_t) * (rd->n_reads + 1), g_stream));
    if (rd->n_words) {
        hipLaunchKernelGGL(encode_kernel, dim3((rd->n_words + 255) / 256), dim3(256), 0, g_stream,
                           rd->ascii.as<uint32_t>(), rd->boff.as<uint64_t>(), rd->woff.as<uint32_t>(),
                           rd->rlen.as<uint32_t>(), rd->n_reads, rd->n_words, rd->packed.as<uint32_t>(),
                           rd->wread.as<uint32_t>(), rd->rflag.as<uint32_t>());
        HIP_TRY(hipGetLastError());
    }
    return PMAP_OK;
}

extern "C" int pmap_reads_create(const char *bases, const uint64_t *offsets, uint32_t n_reads,
                                 uint32_t first_read_id, pmap_reads_t **out) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!offsets || !out || (n_reads && !bases && offsets[n_reads] != offsets[0]))
        return pmap_fail(PMAP_EINVAL, "null argument");
    if ((uint64_t) first_read_id + n_reads > 0xFFFFFFFEull)
        return pmap_fail(PMAP_ELIMIT, "read ids exceed 32 bits (src/minihit.cpp:18-19)");
    std::vector<uint32_t> woff(n_reads + 1), rlen(n_reads + 1, 0);
    std::vector<uint64_t> boff(n_reads + 1);
    uint64_t words = 0;
    const uint64_t base0 = offsets[0];
    for (uint32_t r = 0; r < n_reads; ++r) {
        if (offsets[r + 1] < offsets[r]) return pmap_fail(PMAP_EINVAL, "read offsets not monotone");
        uint64_t len = offsets[r + 1] - offsets[r];
        if (len >= 0x7FFFFFFFull) return pmap_fail(PMAP_ELIMIT, "a read is longer than 2^31-2 bases");
        woff[r] = (uint32_t) words;
        rlen[r] = (uint32_t) len;
        boff[r] = offsets[r] - base0;
        words += (len + 15) / 16;
        if (words * 16 >= 0xFFFFF000ull)
            return pmap_fail(PMAP_ELIMIT, "batch exceeds 2^32 padded bases: split it into several batches");
    }
    woff[n_reads] = (uint32_t) words;
    boff[n_reads] = offsets[n_reads] - base0;
    const uint64_t n_bases = offsets[n_reads] - base0;

    pmap_reads *rd = new pmap_reads();
    rd->n_reads = n_reads;
    rd->first_read_id = first_read_id;
    rd->n_bases = n_bases;
    rd->n_words = (uint32_t) words;
    rd->n_tiles = (uint32_t) ((words * 16 + SK_TILE - 1) / SK_TILE);
    const size_t ascii_bytes = ((n_bases + 3) & ~3ull) + 64;
    const size_t word_slots = (size_t) words + 16;
    auto fail = [&](int code) {
        pmap_reads_destroy(rd);
        return code;
    };
    if ((rc = rd->ascii.ensure(ascii_bytes)) || (rc = rd->boff.ensure(sizeof(uint64_t) * (n_reads + 1))) ||
        (rc = rd->woff.ensure(sizeof(uint32_t) * (n_reads + 1))) ||
        (rc = rd->rlen.ensure(sizeof(uint32_t) * (n_reads + 1))) ||
        (rc = rd->rflag.ensure(sizeof(uint32_t) * (n_reads + 1))) ||
        (rc = rd->packed.ensure(sizeof(uint32_t) * word_slots)) ||
        (rc = rd->wread.ensure(sizeof(uint32_t) * word_slots)))
        return fail(rc);
    hipError_t e;
    e = hipMemsetAsync((char *) rd->ascii.p + ((n_bases) & ~3ull), 0, ascii_bytes - (n_bases & ~3ull), g_stream);
    if (e == hipSuccess) e = hipMemsetAsync(rd->packed.p, 0, sizeof(uint32_t) * word_slots, g_stream);
    if (e == hipSuccess) e = hipMemsetAsync(rd->wread.p, 0, sizeof(uint32_t) * word_slots, g_stream);
    if (e == hipSuccess && n_bases) e = hipMemcpyAsync(rd->ascii.p, bases + base0, n_bases, hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rd->boff.p, boff.data(), sizeof(uint64_t) * (n_reads + 1), hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rd->woff.p, woff.data(), sizeof(uint32_t) * (n_reads + 1), hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rd->rlen.p, rlen.data(), sizeof(uint32_t) * (n_reads + 1), hipMemcpyHostToDevice, g_stream);
    if (e != hipSuccess) {
        pmap_fail(PMAP_EHIP, std::string("read upload: ") + hipGetErrorString(e));
        return fail(PMAP_EHIP);
    }
    if ((rc = launch_encode(rd))) return fail(rc);
    e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) {
        pmap_fail(PMAP_EHIP, std::string("read encode: ") + hipGetErrorString(e));
        return fail(PMAP_EHIP);
    }
    *out = rd;
    return PMAP_OK;
}

extern "C" int pmap_reads_repack(pmap_reads_t *rd) {
    if (!rd) return pmap_fail(PMAP_EINVAL, "null argument");
    int rc = launch_encode(rd);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(g_stream));
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
    rd->ascii.release(); rd->boff.release(); rd->woff.release(); rd->rlen.release(); rd->rflag.release();
    rd->packed.release(); rd->wread.release();
    delete rd;
}

// ---------------------------------------------------------------------------
// map

struct KTimer {
    pmap_result *res;
    bool on;
    uint32_t recorded = 0;
    void begin(int k) { if (on) (void) hipEventRecord(res->kev[2 * k], g_stream); }
    void end(int k) {
        if (on) {
            (void) hipEventRecord(res->kev[2 * k + 1], g_stream);
            recorded |= 1u << k;
        }
    }
};

// exclusive scan (of values or popcounts) of n uint32 into out; the total is read back
template <bool POPC>
static int device_scan(pmap_result *res, const uint32_t *in, uint32_t *out, uint64_t n, uint64_t *total) {
    if (n == 0) { *total = 0; return PMAP_OK; }
    const uint32_t nblk = (uint32_t) ((n + SCAN_TILE - 1) / SCAN_TILE);
    int rc;
    if ((rc = res->bsum.ensure(sizeof(uint32_t) * nblk)) || (rc = res->total.ensure(sizeof(uint64_t)))) return rc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_reduce_kernel<POPC>), dim3(nblk), dim3(256), 0, g_stream, in, n,
                       res->bsum.as<uint32_t>());
    hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(256), 0, g_stream, res->bsum.as<uint32_t>(), nblk,
                       res->total.as<uint64_t>());
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_apply_kernel<POPC>), dim3(nblk), dim3(256), 0, g_stream, in, n,
                       res->bsum.as<uint32_t>(), out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(total, res->total.p, sizeof(uint64_t), hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return PMAP_OK;
}

extern "C" int pmap_map(const pmap_index_t *idx, const pmap_reads_t *rd, const pmap_params_t *prm,
                        pmap_result_t **result) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!idx || !rd || !prm || !result) return pmap_fail(PMAP_EINVAL, "null argument");
    if (prm->k != idx->k) return pmap_fail(PMAP_EINVAL, "k differs from the k the index was built with");
    if (!idx->d_slots && (rc = pmap_index_upload(const_cast<pmap_index *>(idx)))) return rc;
    if (prm->w < 1 || prm->w > (uint32_t) SK_HALO + 1)
        return pmap_fail(PMAP_EINVAL, "w must be in [1,65]");
    pmap_result *res = *result;
    if (!res) {
        res = new pmap_result();
        memset(&res->counts, 0, sizeof(res->counts));
        memset(res->kernel_ms, 0, sizeof(res->kernel_ms));
        for (auto &e : res->ev) HIP_TRY(hipEventCreate(&e));
        for (auto &e : res->kev) HIP_TRY(hipEventCreate(&e));
        res->events = true;
        *result = res;
    }
    const uint32_t last = prm->last_stage ? prm->last_stage : PMAP_STAGE_CLUSTER;
    const bool prof = prm->profile != 0;
    KTimer kt{res, prm->profile >= 2};
    pmap_counts_t &cn = res->counts;
    memset(&cn, 0, sizeof(cn));
    memset(res->kernel_ms, 0, sizeof(res->kernel_ms));
    cn.n_reads = rd->n_reads;
    cn.n_bases = rd->n_bases;
    res->first_read_id = rd->first_read_id;
    const uint32_t k = prm->k, w = prm->w;
    const bool k32 = k <= 15;
    int stage_done = 0;

    if (prof) HIP_TRY(hipEventRecord(res->ev[0], g_stream));
    // ---- sketch
    const uint64_t n_pos = (uint64_t) rd->n_tiles * SK_TILE;
    const uint32_t n_fw = (uint32_t) (n_pos / 32);
    uint64_t n_mini = 0;
    if (n_fw) {
        if ((rc = res->flags.ensure(sizeof(uint32_t) * n_fw)) || (rc = res->foffs.ensure(sizeof(uint32_t) * n_fw)))
            return rc;
        kt.begin(KT_SKETCH_FLAGS);
        if (k32)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(sketch_flags_kernel<uint32_t>), dim3(rd->n_tiles), dim3(256), 0,
                               g_stream, rd->packed.as<uint32_t>(), rd->wread.as<uint32_t>(),
                               rd->woff.as<uint32_t>(), rd->rlen.as<uint32_t>(), rd->rflag.as<uint32_t>(),
                               rd->n_words, w, k, res->flags.as<unsigned long long>());
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(sketch_flags_kernel<uint64_t>), dim3(rd->n_tiles), dim3(256), 0,
                               g_stream, rd->packed.as<uint32_t>(), rd->wread.as<uint32_t>(),
                               rd->woff.as<uint32_t>(), rd->rlen.as<uint32_t>(), rd->rflag.as<uint32_t>(),
                               rd->n_words, w, k, res->flags.as<unsigned long long>());
        kt.end(KT_SKETCH_FLAGS);
        kt.begin(KT_SKETCH_SCAN);
        if ((rc = device_scan<true>(res, res->flags.as<uint32_t>(), res->foffs.as<uint32_t>(), n_fw, &n_mini)))
            return rc;
        kt.end(KT_SKETCH_SCAN);
    }
    if (n_mini >= 0xFFFFFFFFull) return pmap_fail(PMAP_ELIMIT, "more than 2^32-2 minimizers in one batch");
    cn.n_minimizers = n_mini;
    if (n_mini) {
        if ((rc = res->mkmer.ensure(sizeof(uint64_t) * n_mini)) || (rc = res->mread.ensure(sizeof(uint32_t) * n_mini)) ||
            (rc = res->mps.ensure(sizeof(uint32_t) * n_mini)))
            return rc;
        kt.begin(KT_SKETCH_EMIT);
        if (k32)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(sketch_emit_kernel<uint32_t>), dim3((n_fw + 255) / 256), dim3(256), 0,
                               g_stream, res->flags.as<uint32_t>(), res->foffs.as<uint32_t>(), n_fw,
                               rd->packed.as<uint32_t>(), rd->wread.as<uint32_t>(), rd->woff.as<uint32_t>(), k,
                               res->mkmer.as<uint64_t>(), res->mread.as<uint32_t>(), res->mps.as<uint32_t>());
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(sketch_emit_kernel<uint64_t>), dim3((n_fw + 255) / 256), dim3(256), 0,
                               g_stream, res->flags.as<uint32_t>(), res->foffs.as<uint32_t>(), n_fw,
                               rd->packed.as<uint32_t>(), rd->wread.as<uint32_t>(), rd->woff.as<uint32_t>(), k,
                               res->mkmer.as<uint64_t>(), res->mread.as<uint32_t>(), res->mps.as<uint32_t>());
        kt.end(KT_SKETCH_EMIT);
        HIP_TRY(hipGetLastError());
    }
    if (prof) HIP_TRY(hipEventRecord(res->ev[1], g_stream));
    stage_done = PMAP_STAGE_SKETCH;

    // ---- probe + hit emission
    uint64_t n_hits = 0;
    if (last >= PMAP_STAGE_HITS) {
        if (n_mini) {
            const uint32_t nm = (uint32_t) n_mini;
            if ((rc = res->mcnt.ensure(sizeof(uint32_t) * nm)) || (rc = res->mroff.ensure(sizeof(uint32_t) * nm)) ||
                (rc = res->mhoff.ensure(sizeof(uint32_t) * nm)))
                return rc;
            kt.begin(KT_PROBE);
            hipLaunchKernelGGL(probe_kernel, dim3((nm + 255) / 256), dim3(256), 0, g_stream,
                               res->mkmer.as<uint64_t>(), nm, idx->d_slots, idx->log2cap, res->mcnt.as<uint32_t>(),
                               res->mroff.as<uint32_t>());
            kt.end(KT_PROBE);
            kt.begin(KT_HIT_SCAN);
            if ((rc = device_scan<false>(res, res->mcnt.as<uint32_t>(), res->mhoff.as<uint32_t>(), nm, &n_hits)))
                return rc;
            kt.end(KT_HIT_SCAN);
            if (n_hits >= 0xFFFFFFFFull) return pmap_fail(PMAP_ELIMIT, "more than 2^32-2 hits in one batch");
            if (n_hits) {
                const size_t hb = sizeof(uint32_t) * n_hits;
                if ((rc = res->h_read.ensure(hb)) || (rc = res->h_key.ensure(hb)) || (rc = res->h_pos.ensure(hb)) ||
                    (rc = res->h_rank.ensure(hb)) || (rc = res->h_rec.ensure(hb)) ||
                    (rc = res->hstart.ensure(sizeof(uint32_t) * (rd->n_reads + 1))) ||
                    (rc = res->hend.ensure(sizeof(uint32_t) * (rd->n_reads + 1))))
                    return rc;
                kt.begin(KT_SCATTER);
                hipLaunchKernelGGL(scatter_kernel, dim3((nm + 255) / 256), dim3(256), 0, g_stream, nm,
                                   res->mread.as<uint32_t>(), res->mps.as<uint32_t>(), res->mcnt.as<uint32_t>(),
                                   res->mroff.as<uint32_t>(), res->mhoff.as<uint32_t>(), idx->d_recs,
                                   res->h_read.as<uint32_t>(), res->h_key.as<uint32_t>(), res->h_pos.as<uint32_t>(),
                                   res->h_rank.as<uint32_t>(), res->h_rec.as<uint32_t>());
                kt.end(KT_SCATTER);
                kt.begin(KT_SEG_BOUNDS);
                hipLaunchKernelGGL(seg_bounds_kernel, dim3(((uint32_t) n_hits + 255) / 256), dim3(256), 0, g_stream,
                                   res->h_read.as<uint32_t>(), (uint32_t) n_hits, res->hstart.as<uint32_t>(),
                                   res->hend.as<uint32_t>());
                kt.end(KT_SEG_BOUNDS);
                HIP_TRY(hipGetLastError());
            }
        }
        stage_done = PMAP_STAGE_HITS;
    }
    cn.n_hits = n_hits;
    if (prof) HIP_TRY(hipEventRecord(res->ev[2], g_stream));

    // ---- sort
    if (last >= PMAP_STAGE_SORT) {
        if (n_hits) {
            const uint32_t nh = (uint32_t) n_hits;
            const size_t hb = sizeof(uint32_t) * n_hits;
            if ((rc = res->s_read.ensure(hb)) || (rc = res->s_key.ensure(hb)) || (rc = res->s_pos.ensure(hb)) ||
                (rc = res->s_rank.ensure(hb)) || (rc = res->s_rec.ensure(hb)))
                return rc;
            kt.begin(KT_SORT);
            hipLaunchKernelGGL(sort_rank_kernel, dim3((nh + 255) / 256), dim3(256), 0, g_stream, nh,
                               res->h_read.as<uint32_t>(), res->h_key.as<uint32_t>(), res->h_pos.as<uint32_t>(),
                               res->h_rank.as<uint32_t>(), res->h_rec.as<uint32_t>(), res->hstart.as<uint32_t>(),
                               res->hend.as<uint32_t>(), res->s_read.as<uint32_t>(), res->s_key.as<uint32_t>(),
                               res->s_pos.as<uint32_t>(), res->s_rank.as<uint32_t>(), res->s_rec.as<uint32_t>());
            kt.end(KT_SORT);
            HIP_TRY(hipGetLastError());
        }
        stage_done = PMAP_STAGE_SORT;
    }
    if (prof) HIP_TRY(hipEventRecord(res->ev[3], g_stream));

    // ---- cluster
    if (last >= PMAP_STAGE_CLUSTER) {
        uint64_t n_cl = 0, n_kept = 0;
        if (n_hits) {
            const uint32_t nh = (uint32_t) n_hits;
            // per-prg size threshold (src/utils.cpp:211-213, 219-220), refreshed when its inputs change
            if (res->thr_idx != idx || res->thr_fraction != prm->fraction || res->thr_min != prm->min_cluster_size ||
                res->thr_exp != prm->expected_kmers) {
                std::vector<uint32_t> thr(idx->n_prg + 1, 0);
                for (uint32_t p = 0; p < idx->n_prg; ++p)
                    thr[p] = pmap_threshold(idx->min_path_length[p], prm->expected_kmers, prm->fraction,
                                            prm->min_cluster_size);
                if ((rc = res->thr.ensure(sizeof(uint32_t) * thr.size()))) return rc;
                HIP_TRY(hipMemcpyAsync(res->thr.p, thr.data(), sizeof(uint32_t) * thr.size(), hipMemcpyHostToDevice,
                                       g_stream));
                HIP_TRY(hipStreamSynchronize(g_stream));   // thr is a stack vector
                res->thr_idx = idx;
                res->thr_fraction = prm->fraction;
                res->thr_min = prm->min_cluster_size;
                res->thr_exp = prm->expected_kmers;
            }
            const uint32_t n_cfw = ((nh + 255) / 256) * 8;   // 32-bit flag words, whole blocks
            if ((rc = res->cflags.ensure(sizeof(uint32_t) * n_cfw)) || (rc = res->coffs.ensure(sizeof(uint32_t) * n_cfw)))
                return rc;
            kt.begin(KT_CLUSTER_FLAGS);
            hipLaunchKernelGGL(cluster_flags_kernel, dim3((nh + 255) / 256), dim3(256), 0, g_stream, nh,
                               res->s_read.as<uint32_t>(), res->s_key.as<uint32_t>(), res->s_pos.as<uint32_t>(),
                               (int) prm->max_diff, res->cflags.as<unsigned long long>());
            kt.end(KT_CLUSTER_FLAGS);
            kt.begin(KT_CLUSTER_SCAN);
            if ((rc = device_scan<true>(res, res->cflags.as<uint32_t>(), res->coffs.as<uint32_t>(), n_cfw, &n_cl)))
                return rc;
            kt.end(KT_CLUSTER_SCAN);
            const uint32_t nc = (uint32_t) n_cl;
            const size_t cb = sizeof(uint32_t) * (n_cl + 1);
            const uint32_t n_kfw = ((nc + 255) / 256) * 8;
            if ((rc = res->cstart.ensure(cb)) || (rc = res->c_size.ensure(cb)) || (rc = res->c_read.ensure(cb)) ||
                (rc = res->c_key.ensure(cb)) || (rc = res->c_first.ensure(cb)) || (rc = res->c_last.ensure(cb)) ||
                (rc = res->c_rank.ensure(cb)) || (rc = res->kflags.ensure(sizeof(uint32_t) * n_kfw)) ||
                (rc = res->koffs.ensure(sizeof(uint32_t) * n_kfw)))
                return rc;
            kt.begin(KT_CLUSTER_STARTS);
            hipLaunchKernelGGL(bit_positions_kernel, dim3((n_cfw + 255) / 256), dim3(256), 0, g_stream,
                               res->cflags.as<uint32_t>(), res->coffs.as<uint32_t>(), n_cfw,
                               res->cstart.as<uint32_t>());
            kt.end(KT_CLUSTER_STARTS);
            kt.begin(KT_CLUSTER_TABLE);
            hipLaunchKernelGGL(cluster_table_kernel, dim3((nc + 255) / 256), dim3(256), 0, g_stream, nc, nh,
                               res->cstart.as<uint32_t>(), res->s_read.as<uint32_t>(), res->s_key.as<uint32_t>(),
                               res->s_pos.as<uint32_t>(), res->s_rank.as<uint32_t>(), res->thr.as<uint32_t>(),
                               res->c_size.as<uint32_t>(), res->c_read.as<uint32_t>(), res->c_key.as<uint32_t>(),
                               res->c_first.as<uint32_t>(), res->c_last.as<uint32_t>(), res->c_rank.as<uint32_t>(),
                               res->kflags.as<unsigned long long>());
            kt.end(KT_CLUSTER_TABLE);
            kt.begin(KT_KEPT_SCAN);
            if ((rc = device_scan<true>(res, res->kflags.as<uint32_t>(), res->koffs.as<uint32_t>(), n_kfw, &n_kept)))
                return rc;
            kt.end(KT_KEPT_SCAN);
            if (n_kept) {
                const uint32_t nk = (uint32_t) n_kept;
                if ((rc = res->kidx.ensure(sizeof(uint32_t) * nk)) || (rc = res->k_rows.ensure(sizeof(uint32_t) * 8 * nk)))
                    return rc;
                kt.begin(KT_KEPT_GATHER);
                hipLaunchKernelGGL(bit_positions_kernel, dim3((n_kfw + 255) / 256), dim3(256), 0, g_stream,
                                   res->kflags.as<uint32_t>(), res->koffs.as<uint32_t>(), n_kfw,
                                   res->kidx.as<uint32_t>());
                hipLaunchKernelGGL(cluster_gather_kernel, dim3((nk + 255) / 256), dim3(256), 0, g_stream, nk,
                                   res->kidx.as<uint32_t>(), res->cstart.as<uint32_t>(), res->c_size.as<uint32_t>(),
                                   res->c_read.as<uint32_t>(), res->c_key.as<uint32_t>(), res->c_first.as<uint32_t>(),
                                   res->c_last.as<uint32_t>(), res->c_rank.as<uint32_t>(), res->k_rows.as<uint32_t>());
                kt.end(KT_KEPT_GATHER);
            }
            HIP_TRY(hipGetLastError());
        }
        cn.n_clusters_all = n_cl;
        cn.n_clusters = n_kept;
        stage_done = PMAP_STAGE_CLUSTER;
    }
    if (prof) HIP_TRY(hipEventRecord(res->ev[4], g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    (void) stage_done;
    if (prof) {
        for (int i = 0; i < 4; ++i) HIP_TRY(hipEventElapsedTime(&cn.ms[i], res->ev[i], res->ev[i + 1]));
        HIP_TRY(hipEventElapsedTime(&cn.ms[4], res->ev[0], res->ev[4]));
    }
    if (kt.on) {
        for (int i = 0; i < KT_COUNT; ++i) {
            float ms = 0;
            if ((kt.recorded >> i & 1u) && hipEventElapsedTime(&ms, res->kev[2 * i], res->kev[2 * i + 1]) == hipSuccess)
                res->kernel_ms[i] = ms;
        }
    }
    return PMAP_OK;
}

extern "C" int pmap_result_counts(const pmap_result_t *res, pmap_counts_t *out) {
    if (!res || !out) return pmap_fail(PMAP_EINVAL, "null argument");
    *out = res->counts;
    return PMAP_OK;
}

extern "C" int pmap_result_kernel_ms(const pmap_result_t *res, float *ms) {
    if (!res || !ms) return pmap_fail(PMAP_EINVAL, "null argument");
    memcpy(ms, res->kernel_ms, sizeof(res->kernel_ms));
    return PMAP_OK;
}

extern "C" const char *pmap_kernel_name(int i) {
    if (i < 0 || i >= PMAP_N_KERNELS) return "";
    return kernel_names[i];
}

template <typename T>
static int fetch(const DevBuf &b, size_t n, std::vector<T> *out) {
    out->resize(n);
    if (n) HIP_TRY(hipMemcpy(out->data(), b.p, n * sizeof(T), hipMemcpyDeviceToHost));
    return PMAP_OK;
}

extern "C" int pmap_result_minimizers(const pmap_result_t *res, uint64_t *kmer, uint32_t *read_id, uint32_t *pos,
                                      uint8_t *strand) {
    if (!res) return pmap_fail(PMAP_EINVAL, "null argument");
    const size_t n = res->counts.n_minimizers;
    int rc;
    if (kmer && n) HIP_TRY(hipMemcpy(kmer, res->mkmer.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (read_id) {
        std::vector<uint32_t> v;
        if ((rc = fetch(res->mread, n, &v))) return rc;
        for (size_t i = 0; i < n; ++i) read_id[i] = v[i] + res->first_read_id;
    }
    if (pos || strand) {
        std::vector<uint32_t> v;
        if ((rc = fetch(res->mps, n, &v))) return rc;
        for (size_t i = 0; i < n; ++i) {
            if (pos) pos[i] = v[i] & 0x7FFFFFFFu;
            if (strand) strand[i] = (uint8_t) (v[i] >> 31);
        }
    }
    return PMAP_OK;
}

extern "C" int pmap_result_hits(const pmap_result_t *res, int sorted, uint32_t *read_id, uint32_t *pos,
                                uint32_t *prg_id, uint32_t *rec, uint8_t *is_forward) {
    if (!res) return pmap_fail(PMAP_EINVAL, "null argument");
    const size_t n = res->counts.n_hits;
    int rc;
    std::vector<uint32_t> v;
    if (read_id) {
        if ((rc = fetch(sorted ? res->s_read : res->h_read, n, &v))) return rc;
        for (size_t i = 0; i < n; ++i) read_id[i] = v[i] + res->first_read_id;
    }
    if (pos && n) HIP_TRY(hipMemcpy(pos, (sorted ? res->s_pos : res->h_pos).p, n * 4, hipMemcpyDeviceToHost));
    if (rec && n) HIP_TRY(hipMemcpy(rec, (sorted ? res->s_rec : res->h_rec).p, n * 4, hipMemcpyDeviceToHost));
    if (prg_id || is_forward) {
        if ((rc = fetch(sorted ? res->s_key : res->h_key, n, &v))) return rc;
        for (size_t i = 0; i < n; ++i) {
            if (prg_id) prg_id[i] = v[i] >> 1;
            if (is_forward) is_forward[i] = (uint8_t) (1u - (v[i] & 1u));
        }
    }
    return PMAP_OK;
}

extern "C" int pmap_result_clusters(const pmap_result_t *res, uint64_t *begin, uint32_t *size, uint32_t *read_id,
                                    uint32_t *prg_id, uint8_t *is_forward, uint32_t *first_pos, uint32_t *last_pos,
                                    uint32_t *first_rank) {
    if (!res) return pmap_fail(PMAP_EINVAL, "null argument");
    const size_t n = res->counts.n_clusters;
    std::vector<uint32_t> rows;
    int rc;
    if ((rc = fetch(res->k_rows, n * 8, &rows))) return rc;
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

extern "C" void pmap_result_destroy(pmap_result_t *res) {
    if (!res) return;
    for (DevBuf *b : res->all)
        if (b) b->release();
    if (res->events) {
        for (auto &e : res->ev) (void) hipEventDestroy(e);
        for (auto &e : res->kev) (void) hipEventDestroy(e);
    }
    delete res;
}
