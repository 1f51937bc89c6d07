// Device code of libpandora_amd (gfx950): included by pmap_device.hip only.
#ifndef PMAP_KERNELS_HIP_H
#define PMAP_KERNELS_HIP_H

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

// reverse complement of a k-mer held in the low 2k bits (src/seq.cpp:44)
__device__ __forceinline__ uint64_t revcomp_dev(uint64_t fwd, uint32_t k) {
    uint64_t r = __brevll(~fwd);
    r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
    return r >> (64 - 2 * k);
}
__device__ __forceinline__ uint32_t revcomp_dev(uint32_t fwd, uint32_t k) {
    uint32_t r = __brev(~fwd);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    return r >> (32 - 2 * k);
}

// canonical hash (src/seq.cpp:109-111): h = min(h_fwd, h_rc), strand = (h_fwd <= h_rc)
template <typename HT>
__device__ __forceinline__ HT canon_dev(HT fwd, uint32_t k, HT mask, uint32_t *strand) {
    const HT hf = hash64_dev<HT>(fwd, mask), hr = hash64_dev<HT>(revcomp_dev(fwd, k), mask);
    *strand = hf <= hr;
    return hf < hr ? hf : hr;
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
// device-wide exclusive scan of uint32 (optionally of popcounts): reduce, spine, apply
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
                                                         unsigned long long *__restrict__ total) {
    __shared__ uint32_t lds4[4];
    unsigned long long carry = 0;
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
                                                     uint32_t n_words, uint2 *__restrict__ words,
                                                     uint32_t *__restrict__ rflag) {
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
    words[t] = make_uint2(word, r);
    if (bad) rflag[r] = 1;   // every writer stores the same value
}

// rmeta[r] = {first position of read r in the padded stream, its length, or 0 for a read with a
// letter outside ACGT (such a read has no sketch, src/seq.cpp:48-52)}
__global__ __launch_bounds__(256) void read_meta_kernel(const uint32_t *__restrict__ woff,
                                                        const uint32_t *__restrict__ rlen,
                                                        const uint32_t *__restrict__ rflag, uint32_t n_reads,
                                                        uint2 *__restrict__ rmeta) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_reads) rmeta[r] = make_uint2(woff[r] * 16u, rflag[r] ? 0u : rlen[r]);
}

#include "pmap_sketch.hip.h"

#include "pmap_emit.hip.h"

__device__ __forceinline__ uint32_t clamp_count(const unsigned long long *p, uint32_t cap) {
    const unsigned long long v = *p;
    return v < cap ? (uint32_t) v : cap;
}

// first / one-past-last hit of every read that has hits
__global__ __launch_bounds__(256) void seg_bounds_kernel(const uint32_t *__restrict__ h_read,
                                                         const unsigned long long *__restrict__ n_hits_p,
                                                         uint32_t hit_cap, uint32_t *__restrict__ hstart,
                                                         uint32_t *__restrict__ hend) {
    const uint32_t n_hits = clamp_count(n_hits_p, hit_cap);
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

__global__ __launch_bounds__(256) void sort_rank_kernel(const unsigned long long *__restrict__ n_hits_p,
                                                        uint32_t hit_cap, const uint32_t *__restrict__ h_read,
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
    const uint32_t n_hits = clamp_count(n_hits_p, hit_cap);
    if (blockIdx.x * 256 >= n_hits) return;
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
// cluster cut (src/utils.cpp:206-209): one flag bit per sorted hit that starts a cluster; bits beyond
// the last hit are written as zero so that the flag array can be scanned over its whole capacity
__global__ __launch_bounds__(256) void cluster_flags_kernel(const unsigned long long *__restrict__ n_hits_p,
                                                            uint32_t hit_cap, const uint32_t *__restrict__ s_read,
                                                            const uint32_t *__restrict__ s_key,
                                                            const uint32_t *__restrict__ s_pos, int max_diff,
                                                            unsigned long long *__restrict__ cflags64) {
    const uint32_t n_hits = clamp_count(n_hits_p, hit_cap);
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
                                                            uint32_t out_cap, uint32_t *__restrict__ out) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_fw) return;
    uint32_t bits = flags[g];
    uint32_t o = offs[g];
    while (bits) {
        const uint32_t b = __ffs(bits) - 1;
        bits &= bits - 1;
        if (o < out_cap) out[o] = g * 32 + b;
        ++o;
    }
}

// one thread per cluster: summary row + "kept" flag (size > threshold, src/utils.cpp:211-220)
__global__ __launch_bounds__(256) void cluster_table_kernel(const unsigned long long *__restrict__ n_cl_p,
                                                            uint32_t cl_cap,
                                                            const unsigned long long *__restrict__ n_hits_p,
                                                            uint32_t hit_cap, const uint32_t *__restrict__ cstart,
                                                            const uint32_t *__restrict__ s_read,
                                                            const uint32_t *__restrict__ s_key,
                                                            const uint32_t *__restrict__ s_pos,
                                                            const uint32_t *__restrict__ s_rank,
                                                            const uint32_t *__restrict__ thr,
                                                            uint32_t *__restrict__ c_size, uint32_t *__restrict__ c_read,
                                                            uint32_t *__restrict__ c_key, uint32_t *__restrict__ c_first,
                                                            uint32_t *__restrict__ c_last, uint32_t *__restrict__ c_rank,
                                                            unsigned long long *__restrict__ kflags64) {
    const uint32_t n_cl = clamp_count(n_cl_p, cl_cap), n_hits = clamp_count(n_hits_p, hit_cap);
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
__global__ __launch_bounds__(256) void cluster_gather_kernel(const unsigned long long *__restrict__ n_kept_p,
                                                             uint32_t cl_cap, const uint32_t *__restrict__ kidx,
                                                             const uint32_t *__restrict__ cstart,
                                                             const uint32_t *__restrict__ c_size,
                                                             const uint32_t *__restrict__ c_read,
                                                             const uint32_t *__restrict__ c_key,
                                                             const uint32_t *__restrict__ c_first,
                                                             const uint32_t *__restrict__ c_last,
                                                             const uint32_t *__restrict__ c_rank,
                                                             uint32_t *__restrict__ k_rows) {
    const uint32_t n_kept = clamp_count(n_kept_p, cl_cap);
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_kept) return;
    const uint32_t c = kidx[t];
    uint32_t *row = k_rows + (uint64_t) t * 8;
    *(uint4 *) row = make_uint4(cstart[c], c_size[c], c_read[c], c_key[c]);
    *(uint4 *) (row + 4) = make_uint4(c_first[c], c_last[c], c_rank[c], c);
}

#endif
