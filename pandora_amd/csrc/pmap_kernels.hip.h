// Device code of libpandora_amd (gfx950): included by pmap_device.hip only.
#ifndef PMAP_KERNELS_HIP_H
#define PMAP_KERNELS_HIP_H

// ---------------------------------------------------------------------------
// device helpers

// four consecutive 32-bit words at any 4-byte boundary
struct __attribute__((aligned(4))) Quad4 {
    uint32_t x, y, z, w;
};

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
// encode (src/inthash.cpp:42-63): one wavefront per ENC_BLOCK packed words (1024 bases), one lane per word of 16
// bases.  wblk[b] = the read that owns the block's first word (from the host, which lays the words out anyway);
// the lanes walk on from there over the reads that start inside the block (a handful at most: wave-uniform
// loop over the word offsets of the next reads, held one per lane).  The letters are handled four at a time
// inside a 32-bit register: code = ((c >> 1) ^ (c >> 2)) & 3 for A, C, G, T in either case, the byte itself for
// the raw codes 0..3; a letter is good if it is a raw code or equals the letter its code stands for (looked up
// with one byte permute); four codes are squeezed into a byte with three shifts.
static const uint32_t ENC_BLOCK = 256;      // four words per lane: what a wavefront waits for is asked for four at a time

// bit 7 of every byte of z that is not zero
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t z) {
    return (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
}

// the 16 letters from letter j0 of a read of len letters that starts at byte a0 of ascii32: packed word, bad-letter flag
struct EncIn {
    Quad4 q4;
    uint32_t x4, sh, j0, len;
};

__device__ __forceinline__ EncIn encode_fetch(const uint32_t *__restrict__ ascii32, uint64_t a0, uint32_t j0, uint32_t len) {
    const uint64_t a = a0 + j0, wi = a >> 2;
    EncIn e;
    e.sh = (uint32_t) (a & 3);
    e.q4 = *(const Quad4 *) (ascii32 + wi);
    e.x4 = ascii32[wi + 4];
    e.j0 = j0;
    e.len = len;
    return e;
}

__device__ __forceinline__ uint32_t encode_word(const EncIn &e, uint32_t *bad_out) {
    const uint32_t x[5] = {e.q4.x, e.q4.y, e.q4.z, e.q4.w, e.x4};
    uint32_t word = 0, bad = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // the 4 letters j0 + 4 q ..: bytes sh .. sh + 3 of the pair x[q + 1] : x[q]
        const uint32_t v = __builtin_amdgcn_alignbyte(x[q + 1], x[q], e.sh);
        const uint32_t left = e.len > e.j0 + 4 * q ? e.len - (e.j0 + 4 * q) : 0u;   // letters of the read from here on
        const uint32_t in = left >= 4 ? 0xFFFFFFFFu : (1u << (8 * left)) - 1u;      // their bytes
        const uint32_t raw7 = ~nonzero_bytes(v & 0xFCFCFCFCu) & 0x80808080u;        // bit 7 of the bytes 0..3
        const uint32_t raw3 = (raw7 >> 7) | (raw7 >> 6);                            // 3 in those bytes
        uint32_t code = ((v >> 1) ^ (v >> 2)) & 0x03030303u;
        // upper-case letter against the letter of its code ("ACGT" by code, one byte permute)
        const uint32_t want = __builtin_amdgcn_perm(0u, 0x54474341u, code);
        bad |= nonzero_bytes((v & 0xDFDFDFDFu) ^ want) & ~raw7 & in;
        code = ((code & ~raw3) | (v & raw3)) & in;
        // first letter on top: byte 0 -> bits 7:6, byte 1 -> 5:4, byte 2 -> 3:2, byte 3 -> 1:0
        const uint32_t byte = ((code << 6) | (code >> 4) | (code >> 14) | (code >> 24)) & 0xFFu;
        word |= byte << (24 - 8 * q);
    }
    *bad_out = bad;
    return word;
}

// PACKED: the batch came 2-bit packed from the host (pmap_reads_update_packed): ascii32 holds the packed words,
// one per 16 bases, and the kernel only pairs every word with its read.
template <bool PACKED>
__global__ __launch_bounds__(256) void encode_kernel(const uint32_t *__restrict__ ascii32,
                                                     const uint64_t *__restrict__ boff,
                                                     const uint32_t *__restrict__ woff,
                                                     const uint32_t *__restrict__ rlen,
                                                     const uint32_t *__restrict__ wblk, uint32_t n_reads,
                                                     uint32_t n_words, uint2 *__restrict__ words,
                                                     uint32_t *__restrict__ rflag) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t blk = blockIdx.x * 4 + (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint32_t t0 = blk * ENC_BLOCK;
    if (t0 >= n_words) return;
    constexpr int NW = ENC_BLOCK / 64;          // words per lane: t0 + lane, t0 + 64 + lane, ...
    const uint32_t t_last = t0 + ENC_BLOCK - 1;
    // the read that owns a word t: the largest r with woff[r] <= t (woff[n_reads] = n_words is above every word)
    uint32_t r[NW];
    r[0] = (uint32_t) __builtin_amdgcn_readfirstlane((int) wblk[blk]);
#pragma unroll
    for (int u = 1; u < NW; ++u) r[u] = r[0];
    for (uint32_t base = r[0] + 1;; base += 64) {               // wave-uniform
        const uint32_t wv = woff[min(base + lane, n_reads)];
        const uint32_t cnt = (uint32_t) __popcll(__builtin_amdgcn_ballot_w64(wv <= t_last && base + lane <= n_reads));
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t wj = (uint32_t) __builtin_amdgcn_readlane((int) wv, (int) j);
#pragma unroll
            for (int u = 0; u < NW; ++u) r[u] += wj <= t0 + 64 * u + lane ? 1u : 0u;
        }
        if (cnt < 64) break;
    }
    uint32_t w0[NW], len[NW];
    uint64_t b0[NW];
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        r[u] = min(r[u], n_reads - 1);                          // (only for words beyond the last one)
        w0[u] = woff[r[u]];
        len[u] = rlen[r[u]];
        b0[u] = boff[r[u]];
    }
    if constexpr (PACKED) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const uint32_t t = t0 + 64 * u + lane;
            if (t < n_words) words[t] = make_uint2(ascii32[t], r[u]);
        }
        return;
    }
    EncIn e[NW];
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const uint32_t t = t0 + 64 * u + lane;
        if (t < n_words) e[u] = encode_fetch(ascii32, b0[u], (t - w0[u]) * 16, len[u]);
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const uint32_t t = t0 + 64 * u + lane;
        if (t < n_words) {
            uint32_t bad;
            const uint32_t word = encode_word(e[u], &bad);
            words[t] = make_uint2(word, r[u]);
            if (bad) rflag[r[u]] = 1;   // every writer stores the same value
        }
    }
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

// ---------------------------------------------------------------------------
// Chained prefix sum over the blocks of one launch.  Block b = blockIdx.x adds `total` and gets
// back the sum of the totals of blocks 0..b-1.  Workgroups are dispatched in index order on every
// XCD, so the lowest-numbered unfinished block is always resident and never waits: the chain
// cannot stall itself, and a block publishes its own total before it looks back.  gran[b] is one 8-byte word
// {state << 32 | value}: state 1 = value is the block's own total, 2 = value is the inclusive
// prefix up to and including the block; the word is written and read whole with agent-scope
// atomics, so the value never needs a separate flag or fence.  gran must be zero before the launch.
// The wait is bounded and reports through *fail (no hang if the dispatch order ever differs).
// Called by every thread of the block; lds2 = two words of LDS.
static const uint32_t CHAIN_SPIN_LIMIT = 1u << 18;
static const int CHAIN_SLABS = 2;     // one look-back round reads 2 x 64 predecessors, both loads in flight together

__device__ __forceinline__ uint32_t chained_scan(unsigned long long *gran, uint32_t b, uint32_t total,
                                                 uint32_t *fail, uint32_t *lds2, uint32_t spin_limit) {
    if (threadIdx.x < 64) {
        const uint32_t lane = threadIdx.x;
        uint32_t excl = 0;
        if (b != 0) {
            if (lane == 0)
                __hip_atomic_store(gran + b, (1ull << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t top = (int64_t) b - 1;          // nearest predecessor not yet added
            for (;;) {
                // slab i, lane l looks at block top - (64 i + l): nearest first
                unsigned long long g[CHAIN_SLABS];
                uint32_t spins = 0;
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int i = 0; i < CHAIN_SLABS; ++i) {
                        const int64_t j = top - (int64_t) (64 * i + lane);
                        g[i] = 2ull << 32;          // before block 0: an inclusive prefix of 0
                        if (j >= 0) g[i] = __hip_atomic_load(gran + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ready = ready && (g[i] >> 32) != 0;
                    }
                    if (__all(ready)) break;
                    if (++spins > spin_limit) {                 // uniform: every lane counts alike
                        if (lane == 0) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int i = 0; i < CHAIN_SLABS; ++i)
                            if ((g[i] >> 32) == 0) g[i] = 2ull << 32;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                bool done = false;
#pragma unroll
                for (int i = 0; i < CHAIN_SLABS; ++i) {
                    if (!done) {
                        const unsigned long long incl = ballot64((g[i] >> 32) == 2);
                        if (incl) {
                            const uint32_t stop = (uint32_t) __builtin_ctzll(incl);   // nearest block that knows its prefix
                            excl += wave_sum(lane <= stop ? (uint32_t) g[i] : 0u);
                            done = true;
                        } else {
                            excl += wave_sum((uint32_t) g[i]);
                        }
                    }
                }
                if (done) break;
                top -= 64 * CHAIN_SLABS;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(gran + b, (2ull << 32) | (uint32_t) (excl + total), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            lds2[0] = excl;
        }
    }
    __syncthreads();
    return lds2[0];
}

#include "pmap_emit.hip.h"

__device__ __forceinline__ uint32_t clamp_count(const unsigned long long *p, uint32_t cap) {
    const unsigned long long v = *p;
    return v < cap ? (uint32_t) v : cap;
}

// ---------------------------------------------------------------------------
// sort: hits arrive ordered by (read, position, prg, path rank).  A stable sort of each read's
// segment by key = prg << 1 | !forward yields MinimizerHit::operator< order.  Each hit counts the
// hits of its read that must precede it (rank by counting) and is written to that place.
//
// The keys of the block's reads are staged through LDS in chunks of SORT_CH.  NARROW (keys below
// 2^(31 - SORT_CH_BITS), i.e. every index of up to half a million loci): a staged key carries its
// place in the chunk below it, key << (SORT_CH_BITS + 1) | place, and "precedes hit i" is one
// comparison with the hit's own (key, place) -- place 0 / all-ones for chunks after / before the
// one that holds the hit, which counts smaller keys only / smaller and equal ones.
static const int SORT_CH_BITS = 11, SORT_CH = 1 << SORT_CH_BITS;

// A read with more than SORT_LONG hits is left to the long-segment kernels below (rank counting costs a
// comparison per pair of hits of a read: fine for the hundreds of hits of a usual read, not for the 10^5 of
// a megabase read on an index with several records per key); *has_long tells them that there is work.
static const uint32_t SORT_LONG_DEFAULT = 4096;

template <bool NARROW>
__global__ __launch_bounds__(256) void sort_rank_kernel(const unsigned long long *__restrict__ n_hits_p,
                                                        uint32_t hit_cap, const uint32_t *__restrict__ h_read,
                                                        const uint32_t *__restrict__ h_key,
                                                        const uint32_t *__restrict__ h_pos,
                                                        const uint32_t *__restrict__ h_rank,
                                                        const uint32_t *__restrict__ h_rec,
                                                        const uint32_t *__restrict__ hbounds, uint32_t n_reads,
                                                        uint32_t *__restrict__ s_read, uint32_t *__restrict__ s_key,
                                                        uint32_t *__restrict__ s_pos, uint32_t *__restrict__ s_rank,
                                                        uint32_t *__restrict__ s_rec, uint32_t *__restrict__ has_long, uint32_t sort_long,
                                                        uint32_t long_follows, uint32_t wave_min, uint32_t wave_max,
                                                        uint32_t wave_follows, uint32_t *__restrict__ has_wave) {
    __shared__ uint32_t keys[SORT_CH];
    __shared__ uint32_t wlo[4], whi[4];
    const uint32_t n_hits = clamp_count(n_hits_p, hit_cap);
    if (blockIdx.x * 256 >= n_hits) return;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    bool active = i < n_hits;
    uint32_t r = 0, key = 0, s = 0, e = 0;
    if (active) {
        r = h_read[i];
        key = h_key[i];
        s = hbounds[r];
        e = ~hbounds[n_reads + r];
        if (e - s > sort_long) {
            active = false;
            *has_long = 1u;             // every writer stores the same value
            if (!long_follows) {
                // The long-segment kernels are not part of this run: the host will map the batch again with
                // them.  Until then the kernels behind this one must find a well-formed hit in every place
                // (they index tables by its fields): the hit stays where it is, unsorted.
                s_read[i] = r;
                s_key[i] = key;
                s_pos[i] = h_pos[i];
                s_rank[i] = h_rank[i];
                s_rec[i] = h_rec[i];
            }
        } else if (e - s > wave_min && e - s <= wave_max) {
            // sort_wave_kernel's kind of read: left to it when it is part of this run, and the host hears of it
            // either way (it runs that kernel only while batches hold such reads)
            if (wave_follows) active = false;
            *has_wave = 1u;             // every writer stores the same value
        }
    }
    // the stretch of hits this block's reads cover: from the first active thread's read to the last one's
    // (hits are ordered by read, so are the bounds)
    {
        const unsigned long long m = ballot64(active);
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (m == 0) {
            if (lane == 0) { wlo[wv] = 0xFFFFFFFFu; whi[wv] = 0u; }
        } else {
            if (lane == (uint32_t) __builtin_ctzll(m)) wlo[wv] = s;
            if (lane == 63u - (uint32_t) __builtin_clzll(m)) whi[wv] = e;
        }
    }
    __syncthreads();
    const uint32_t smin = min(min(wlo[0], wlo[1]), min(wlo[2], wlo[3]));
    const uint32_t emax = max(max(whi[0], whi[1]), max(whi[2], whi[3]));
    const uint32_t PLACE = (2u << SORT_CH_BITS) - 1;      // SORT_CH_BITS + 1 bits: all-ones is above every place
    uint32_t c = 0;
    for (uint32_t base = smin; base < emax; base += SORT_CH) {
        const uint32_t n = min((uint32_t) SORT_CH, emax - base);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < n; t += 256)
            keys[t] = NARROW ? (h_key[base + t] << (SORT_CH_BITS + 1) | t) : h_key[base + t];
        __syncthreads();
        if (active) {
            const uint32_t a = max(s, base), b = min(e, base + n);
            if (NARROW) {
                const uint32_t place = i < base ? 0u : i - base < n ? i - base : PLACE;
                const uint32_t mine = key << (SORT_CH_BITS + 1) | place;
                for (uint32_t j = a; j < b; ++j) c += keys[j - base] < mine;
            } else {
                const uint32_t mid = min(max(i, a), b);   // [a, mid) precede hit i, [mid', b) follow it
                for (uint32_t j = a; j < mid; ++j) c += keys[j - base] <= key;
                for (uint32_t j = max(a, i + 1); j < b; ++j) c += keys[j - base] < key;
            }
        }
    }
    if (active) {
        const uint32_t d = s + c;
        if (d >= n_hits) return;          // only with inconsistent segment bounds: never write out of range
        s_read[d] = r;
        s_key[d] = key;
        s_pos[d] = h_pos[i];
        s_rank[d] = h_rank[i];
        s_rec[d] = h_rec[i];
    }
}

// ---------------------------------------------------------------------------
// Reads with more hits than a comparison per pair of hits is good for (wave_min) and at most 64 * SORT_WAVE_E
// (wave_max): one wavefront per read sorts the words {key, place in the read's segment} with a bitonic network
// held in registers, E words per lane (word i of the segment in lane i / E, register i % E), then moves every
// hit to the place its word ended at.  About 100 instructions per word for 512 hits, whatever the keys are,
// against 2 per pair of hits for the rank counting (the 470 hits of a read on a 200 000-locus index: 930).
// The network is the all-ascending form: a merge of blocks of 2m words starts with the step that compares word
// i with word i ^ (2m - 1) (its mirror image in the block) and goes on with steps i ^ m/2, i ^ m/4, ..., 1.
// Steps inside a lane are min / max of two registers; steps across lanes take the other word through a DPP
// permutation (lanes ^ 1, 2, 3, 7, 8, 15), a ds_swizzle (^ 4, 16, 31) or a ds_bpermute (^ 32, 63).
static const int SORT_WAVE_E = 32;

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_perm(uint32_t x) {
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, CTRL, 0xf, 0xf, true);      // every lane has a source lane
}

// x of lane (lane ^ XM)
template <int XM>
__device__ __forceinline__ uint32_t lane_xor(uint32_t x, uint32_t lane) {
    if constexpr (XM == 1) return dpp_perm<0xB1>(x);                // quad_perm [1,0,3,2]
    else if constexpr (XM == 2) return dpp_perm<0x4E>(x);           // quad_perm [2,3,0,1]
    else if constexpr (XM == 3) return dpp_perm<0x1B>(x);           // quad_perm [3,2,1,0]
    else if constexpr (XM == 7) return dpp_perm<0x141>(x);          // row_half_mirror
    else if constexpr (XM == 8) return dpp_perm<0x128>(x);          // row_ror:8
    else if constexpr (XM == 15) return dpp_perm<0x140>(x);         // row_mirror
    else if constexpr (XM == 4) return (uint32_t) __builtin_amdgcn_ds_swizzle((int) x, 0x101F);    // and 0x1f, xor 4
    else if constexpr (XM == 16) return (uint32_t) __builtin_amdgcn_ds_swizzle((int) x, 0x401F);   // xor 16
    else if constexpr (XM == 31) return (uint32_t) __builtin_amdgcn_ds_swizzle((int) x, 0x7C1F);   // xor 31
    else {
        static_assert(XM == 32 || XM == 63, "no permutation for this lane distance");
        return (uint32_t) __builtin_amdgcn_ds_bpermute((int) ((lane ^ (uint32_t) XM) << 2), (int) x);
    }
}

// compare-exchange with lane ^ XM: lanes with BIT clear keep the smaller word.  MIRROR: the other word is
// register E-1-j of that lane instead of register j.
template <int E, int XM, int BIT, bool MIRROR>
__device__ __forceinline__ void sortw_lane_step(uint32_t (&v)[E], uint32_t lane) {
    const bool lower = (lane & (uint32_t) BIT) == 0;
    uint32_t o[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const uint32_t p = lane_xor<XM>(v[MIRROR ? E - 1 - j : j], lane);
        const uint32_t lo = min(v[j], p), hi = max(v[j], p);
        o[j] = lower ? lo : hi;
    }
#pragma unroll
    for (int j = 0; j < E; ++j) v[j] = o[j];
}

// compare-exchange of registers j and j ^ X (X = STRIDE, or SIZE - 1 for the mirror step) inside every lane
template <int E, int X, int LOWBIT>
__device__ __forceinline__ void sortw_reg_step(uint32_t (&v)[E]) {
#pragma unroll
    for (int j = 0; j < E; ++j)
        if ((j & LOWBIT) == 0) {
            const int jj = j ^ X;
            const uint32_t a = v[j], b = v[jj];
            v[j] = min(a, b);
            v[jj] = max(a, b);
        }
}

// the steps i ^ STRIDE, i ^ STRIDE/2, ..., i ^ 1 inside the lanes
template <int E, int STRIDE>
__device__ __forceinline__ void sortw_reg_tail(uint32_t (&v)[E]) {
    if constexpr (STRIDE >= 1) {
        sortw_reg_step<E, STRIDE, STRIDE>(v);
        sortw_reg_tail<E, STRIDE / 2>(v);
    }
}

// merges of blocks of SIZE = 2, 4, ..., E registers inside every lane
template <int E, int SIZE>
__device__ __forceinline__ void sortw_reg_merges(uint32_t (&v)[E]) {
    if constexpr (SIZE <= E) {
        sortw_reg_step<E, SIZE - 1, SIZE / 2>(v);
        sortw_reg_tail<E, SIZE / 4>(v);
        sortw_reg_merges<E, 2 * SIZE>(v);
    }
}

// the steps lane ^ S, lane ^ S/2, ..., lane ^ 1
template <int E, int S>
__device__ __forceinline__ void sortw_lane_tail(uint32_t (&v)[E], uint32_t lane) {
    if constexpr (S >= 1) {
        sortw_lane_step<E, S, S, false>(v, lane);
        sortw_lane_tail<E, S / 2>(v, lane);
    }
}

// merges of blocks of LB = 2, 4, ..., 64 lanes
template <int E, int LB>
__device__ __forceinline__ void sortw_lane_merges(uint32_t (&v)[E], uint32_t lane) {
    if constexpr (LB <= 64) {
        sortw_lane_step<E, LB - 1, LB / 2, true>(v, lane);
        sortw_lane_tail<E, LB / 4>(v, lane);
        sortw_reg_tail<E, E / 2>(v);
        sortw_lane_merges<E, 2 * LB>(v, lane);
    }
}

// The reads sort_wave_kernel takes (wave_min < hits <= wave_max, and not beyond sort_long), as a list: one thread
// per read, one atomic per wavefront.  (One atomic per read from sort_rank_kernel, whose threads see the same
// bounds, doubled that kernel's time: 15 000 of them on one word wait for each other.)
__global__ __launch_bounds__(256) void sort_wave_list_kernel(const uint32_t *__restrict__ hbounds, uint32_t n_reads,
                                                             uint32_t sort_long, uint32_t wave_min, uint32_t wave_max,
                                                             uint32_t *__restrict__ wave_n, uint32_t *__restrict__ wave_list) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    bool take = false;
    if (r < n_reads) {
        const uint32_t s = hbounds[r], e = ~hbounds[n_reads + r];       // a read without hits has s = all-ones, e = 0
        take = s < e && e - s > wave_min && e - s <= wave_max && e - s <= sort_long;
    }
    const unsigned long long m = ballot64(take);
    if (m == 0) return;
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(wave_n, (uint32_t) __popcll(m));
    base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
    if (take) wave_list[base + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0))] = r;
}

template <int E>
__device__ __forceinline__ void sort_wave_segment(uint32_t lane, uint32_t r, uint32_t s, uint32_t n,
                                                  const uint32_t *__restrict__ h_key, const uint32_t *__restrict__ h_pos,
                                                  const uint32_t *__restrict__ h_rank, const uint32_t *__restrict__ h_rec,
                                                  uint32_t *__restrict__ s_read, uint32_t *__restrict__ s_key,
                                                  uint32_t *__restrict__ s_pos, uint32_t *__restrict__ s_rank,
                                                  uint32_t *__restrict__ s_rec) {
    constexpr uint32_t PB = 6 + (E == 1 ? 0 : E == 2 ? 1 : E == 4 ? 2 : E == 8 ? 3 : E == 16 ? 4 : 5), PM = (1u << PB) - 1;
    static_assert(E == 1 || E == 2 || E == 4 || E == 8 || E == 16 || E == 32, "a power of two");
    uint32_t v[E];
    // a lane's E words are consecutive in memory: 16-byte accesses where the whole group of 4 is inside the segment
    // (the segment starts at any multiple of 4 bytes)
    const uint32_t i0 = lane * E;
    if constexpr (E >= 4) {
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            const uint32_t i = i0 + 4 * q;
            uint32_t kk[4] = {0, 0, 0, 0};
            if (i + 4 <= n) {
                const Quad4 t = *(const Quad4 *) (h_key + s + i);
                kk[0] = t.x; kk[1] = t.y; kk[2] = t.z; kk[3] = t.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u < n) kk[u] = h_key[s + i + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) v[4 * q + u] = i + u < n ? (kk[u] << PB | (i + u)) : 0xFFFFFFFFu;
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t i = i0 + j;
            v[j] = i < n ? (h_key[s + i] << PB | i) : 0xFFFFFFFFu;      // words past the end sort behind every hit
        }
    }
    sortw_reg_merges<E, 2>(v);
    sortw_lane_merges<E, 2>(v, lane);
    if constexpr (E >= 4) {
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            const uint32_t i = i0 + 4 * q, d = s + i;
            if (i >= n) continue;
            uint32_t from[4], ps[4], rk[4], rc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                from[u] = s + (v[4 * q + u] & PM);
                const bool in = i + u < n;
                ps[u] = in ? h_pos[from[u]] : 0u;
                rk[u] = in ? h_rank[from[u]] : 0u;
                rc[u] = in ? h_rec[from[u]] : 0u;
            }
            if (i + 4 <= n) {
                *(Quad4 *) (s_read + d) = Quad4{r, r, r, r};
                *(Quad4 *) (s_key + d) = Quad4{v[4 * q] >> PB, v[4 * q + 1] >> PB, v[4 * q + 2] >> PB, v[4 * q + 3] >> PB};
                *(Quad4 *) (s_pos + d) = Quad4{ps[0], ps[1], ps[2], ps[3]};
                *(Quad4 *) (s_rank + d) = Quad4{rk[0], rk[1], rk[2], rk[3]};
                *(Quad4 *) (s_rec + d) = Quad4{rc[0], rc[1], rc[2], rc[3]};
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u < n) {
                        s_read[d + u] = r;
                        s_key[d + u] = v[4 * q + u] >> PB;
                        s_pos[d + u] = ps[u];
                        s_rank[d + u] = rk[u];
                        s_rec[d + u] = rc[u];
                    }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t i = i0 + j;
            if (i < n) {
                const uint32_t from = s + (v[j] & PM), d = s + i;
                s_read[d] = r;
                s_key[d] = v[j] >> PB;
                s_pos[d] = h_pos[from];
                s_rank[d] = h_rank[from];
                s_rec[d] = h_rec[from];
            }
        }
    }
}

// The reads of wave_list[0 .. *wave_n) (sort_wave_list_kernel), one wavefront per read at a time, any grid.
// BIG = false: reads of up to 1024 hits; BIG = true: of 1025 to 2048 hits (its 32 words per lane take 92 registers
// instead of 56: a kernel of its own, so that the shorter reads keep 8 wavefronts per SIMD to hide their loads).
// Keys must be below 2^(32 - 6 - log2 SORT_WAVE_E) (the host checks the index).
template <bool BIG>
__global__ __launch_bounds__(256) void sort_wave_kernel(const unsigned long long *__restrict__ n_hits_p, uint32_t hit_cap,
                                                        const uint32_t *__restrict__ h_key,
                                                        const uint32_t *__restrict__ h_pos,
                                                        const uint32_t *__restrict__ h_rank,
                                                        const uint32_t *__restrict__ h_rec,
                                                        const uint32_t *__restrict__ hbounds, uint32_t n_reads,
                                                        uint32_t *__restrict__ s_read, uint32_t *__restrict__ s_key,
                                                        uint32_t *__restrict__ s_pos, uint32_t *__restrict__ s_rank,
                                                        uint32_t *__restrict__ s_rec, uint32_t sort_long,
                                                        uint32_t wave_min, uint32_t wave_max,
                                                        const uint32_t *__restrict__ wave_n,
                                                        const uint32_t *__restrict__ wave_list) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint32_t n_hits = clamp_count(n_hits_p, hit_cap);
    const uint32_t n_list = min(*wave_n, n_reads);
    for (uint32_t t = wave; t < n_list; t += gridDim.x * 4) {       // wave-uniform
        const uint32_t r = (uint32_t) __builtin_amdgcn_readfirstlane((int) wave_list[t]);
        if (r >= n_reads) continue;
        const uint32_t s = (uint32_t) __builtin_amdgcn_readfirstlane((int) hbounds[r]);
        const uint32_t e = (uint32_t) __builtin_amdgcn_readfirstlane((int) ~hbounds[n_reads + r]);
        if (s >= e || e > n_hits) continue;       // (a read without hits has s = all-ones, e = 0)
        const uint32_t n = e - s;
        if (n <= wave_min || n > wave_max || n > sort_long) continue;
#define SORT_WAVE_CASE(E)                                                                                             \
    if (n <= 64u * E) {                                                                                               \
        sort_wave_segment<E>(lane, r, s, n, h_key, h_pos, h_rank, h_rec, s_read, s_key, s_pos, s_rank, s_rec);        \
        continue;                                                                                                     \
    }
        if constexpr (BIG) {
            if (n <= 64u * 16) continue;
            SORT_WAVE_CASE(32)
        } else {
            SORT_WAVE_CASE(1)
            SORT_WAVE_CASE(2)
            SORT_WAVE_CASE(4)
            SORT_WAVE_CASE(8)
            SORT_WAVE_CASE(16)
        }
#undef SORT_WAVE_CASE
    }
    static_assert(SORT_WAVE_E == 32, "one case per power of two up to SORT_WAVE_E");
}

// ---------------------------------------------------------------------------
// Long segments.  The hit array is cut into fixed blocks of SORT_LB hits.  long_presort_kernel sorts, for
// every block that touches a long segment, the words {read - first read of the block, key, place in the
// block} (ascending: hits are ordered by read already, so this orders each read's part of the block by key
// and keeps the emission order inside a key).  long_rank_kernel then finds a hit's place in its read's
// sorted segment by binary searches, two per block the segment touches, instead of one comparison per hit
// of the segment: the hits of blocks before its own with keys <= its key, of later blocks with keys < its
// key, and its own position among its read's words in its own block.
static const uint32_t SORT_LB_BITS = 11, SORT_LB = 1u << SORT_LB_BITS;

__device__ __forceinline__ unsigned long long long_word(uint32_t read_local, uint32_t key, uint32_t place) {
    return ((unsigned long long) read_local << (32 + SORT_LB_BITS)) | ((unsigned long long) key << SORT_LB_BITS) | place;
}

__device__ __forceinline__ bool in_long_segment(uint32_t i, const uint32_t *__restrict__ h_read,
                                                const uint32_t *__restrict__ hbounds, uint32_t n_reads, uint32_t sort_long) {
    const uint32_t r = h_read[i];
    return ~hbounds[n_reads + r] - hbounds[r] > sort_long;
}

__global__ __launch_bounds__(256) void long_presort_kernel(const Counters *__restrict__ dc, uint32_t hit_cap,
                                                           const uint32_t *__restrict__ h_read,
                                                           const uint32_t *__restrict__ h_key,
                                                           const uint32_t *__restrict__ hbounds, uint32_t n_reads,
                                                           unsigned long long *__restrict__ words, uint32_t sort_long) {
    if (!dc->has_long) return;       // the usual case: a small grid that leaves at once
    __shared__ unsigned long long w[SORT_LB];
    const uint32_t n_hits = clamp_count(&dc->n_hits, hit_cap);
    for (uint32_t b0 = blockIdx.x * SORT_LB; b0 < n_hits; b0 += gridDim.x * SORT_LB) {
    __syncthreads();                 // the block before has been written out
    const uint32_t n = min(SORT_LB, n_hits - b0);
    // does a long segment touch the block?  One that is longer than a block holds the block's first or last hit
    // if it touches it at all; shorter ones (a low threshold) may lie inside: look at every hit then
    bool touched = in_long_segment(b0, h_read, hbounds, n_reads, sort_long) ||
                   in_long_segment(b0 + n - 1, h_read, hbounds, n_reads, sort_long);
    if (sort_long < SORT_LB) {
        bool mine = false;
        for (uint32_t t = threadIdx.x; t < n; t += 256) mine = mine || in_long_segment(b0 + t, h_read, hbounds, n_reads, sort_long);
        touched = __syncthreads_or(mine) != 0;
    }
    if (!touched) continue;
    const uint32_t r0 = h_read[b0];
    for (uint32_t t = threadIdx.x; t < SORT_LB; t += 256)
        w[t] = t < n ? long_word(h_read[b0 + t] - r0, h_key[b0 + t], t) : ~0ull;
    __syncthreads();
    // bitonic sort of SORT_LB words, 256 threads, 4 compare-exchanges each per step
    for (uint32_t size = 2; size <= SORT_LB; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < SORT_LB / 2; t += 256) {
                const uint32_t lo = 2 * t - (t & (stride - 1));        // index with bit `stride` clear
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = w[lo], b = w[hi];
                if ((a > b) == up) { w[lo] = b; w[hi] = a; }
            }
            __syncthreads();
        }
    for (uint32_t t = threadIdx.x; t < SORT_LB; t += 256) words[b0 + t] = w[t];
    }
}

// number of words of the sorted block below `x` (the block has SORT_LB words, padded with all-ones)
__device__ __forceinline__ uint32_t long_lower_bound(const unsigned long long *__restrict__ blk, unsigned long long x) {
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = SORT_LB >> 1; step > 0; step >>= 1)
        if (blk[lo + step - 1] < x) lo += step;
    return lo + (blk[lo] < x ? 1u : 0u);
}

__global__ __launch_bounds__(256) void long_rank_kernel(const Counters *__restrict__ dc, uint32_t hit_cap,
                                                        const uint32_t *__restrict__ h_read,
                                                        const uint32_t *__restrict__ h_key,
                                                        const uint32_t *__restrict__ h_pos,
                                                        const uint32_t *__restrict__ h_rank,
                                                        const uint32_t *__restrict__ h_rec,
                                                        const uint32_t *__restrict__ hbounds, uint32_t n_reads,
                                                        const unsigned long long *__restrict__ words,
                                                        uint32_t *__restrict__ s_read, uint32_t *__restrict__ s_key,
                                                        uint32_t *__restrict__ s_pos, uint32_t *__restrict__ s_rank,
                                                        uint32_t *__restrict__ s_rec, uint32_t sort_long) {
    if (!dc->has_long) return;       // the usual case: a small grid that leaves at once
    const uint32_t n_hits = clamp_count(&dc->n_hits, hit_cap);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_hits; i += gridDim.x * 256) {
    const uint32_t r = h_read[i], key = h_key[i];
    const uint32_t s = hbounds[r], e = ~hbounds[n_reads + r];
    if (e - s <= sort_long || e > n_hits) continue;
    const uint32_t own = i >> SORT_LB_BITS;
    uint32_t c = 0;
    for (uint32_t blk = s >> SORT_LB_BITS; blk <= (e - 1) >> SORT_LB_BITS; ++blk) {
        const unsigned long long *wb = words + (size_t) blk * SORT_LB;
        const uint32_t rl = r - h_read[blk << SORT_LB_BITS];         // this read's number inside the block
        const uint32_t first = long_lower_bound(wb, long_word(rl, 0u, 0u));
        uint32_t upto;
        if (blk < own) upto = long_lower_bound(wb, long_word(rl, key, SORT_LB - 1) + 1);        // keys <= key
        else if (blk > own) upto = long_lower_bound(wb, long_word(rl, key, 0u));                // keys < key
        else upto = long_lower_bound(wb, long_word(rl, key, i & (SORT_LB - 1)));                // before hit i in its own block
        c += upto - first;
    }
    const uint32_t d = s + c;
    if (d >= n_hits) continue;
    s_read[d] = r;
    s_key[d] = key;
    s_pos[d] = h_pos[i];
    s_rank[d] = h_rank[i];
    s_rec[d] = h_rec[i];
    }
}

// ---------------------------------------------------------------------------
// cluster cut (src/utils.cpp:206-209).  A block of 512 threads takes 8192 sorted hits (16 rounds of
// 512), flags the hits that start a cluster, and writes their indices to cstart at the chained
// prefix sum of the blocks' flag counts.  The last block writes the cluster total.  Many hits per
// block keep the chain short (its cost grows with the number of blocks that poll each other's
// words).  8 wavefronts per block: a 1024-thread block needs 4 free slots on every SIMD of a
// compute unit at once, and next to a long kernel of small blocks (the sketch of the next batch,
// which leaves 3) it could not start before that kernel had drained: 136 instead of 11 us.
static const uint32_t CUT_THREADS = 512, CUT_WAVES = CUT_THREADS / 64, CUT_ROUNDS = 16, CUT_BLOCK = CUT_THREADS * CUT_ROUNDS;

// SERIAL: the same work by ONE workgroup that takes the blocks in turn and carries the prefix itself, no
// chain between workgroups: what a batch falls back to when a chained prefix sum gave up waiting (it
// relies on workgroups being dispatched in index order, which nothing guarantees).
template <bool SERIAL>
__global__ __launch_bounds__(CUT_THREADS) void cluster_cut_kernel(Counters *__restrict__ dc, uint32_t hit_cap,
                                                                  const uint32_t *__restrict__ s_read,
                                                                  const uint32_t *__restrict__ s_key,
                                                                  const uint32_t *__restrict__ s_pos, int max_diff,
                                                                  unsigned long long *__restrict__ gran,
                                                                  uint32_t cl_cap, uint32_t *__restrict__ cstart,
                                                                  uint32_t n_blocks, uint32_t spin_limit) {
    __shared__ uint32_t s_cnt[CUT_ROUNDS * CUT_WAVES], s_half[2], s_scan[2];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n_hits = clamp_count(&dc->n_hits, hit_cap);
    uint32_t carry = 0;
    for (uint32_t b = SERIAL ? 0u : blockIdx.x; b < n_blocks; b += SERIAL ? 1u : n_blocks) {
    if (SERIAL) __syncthreads();          // the counts of the block before have been used
    uint32_t fbits = 0;
    uint8_t rank[CUT_ROUNDS];
    static_assert(CUT_ROUNDS <= 32, "one flag bit per round");
#pragma unroll
    for (uint32_t r = 0; r < CUT_ROUNDS; ++r) {
        const uint32_t i = b * CUT_BLOCK + r * CUT_THREADS + threadIdx.x;
        bool f = false;
        if (i < n_hits) {
            if (i == 0) f = true;
            else {
                const int d = (int) s_pos[i] - (int) s_pos[i - 1];
                f = s_read[i] != s_read[i - 1] || s_key[i] != s_key[i - 1] || (d < 0 ? -d : d) > max_diff;
            }
        }
        const unsigned long long m = ballot64(f);
        rank[r] = (uint8_t) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0));
        if (f) fbits |= 1u << r;
        if (lane == 0) s_cnt[r * CUT_WAVES + wv] = (uint32_t) __popcll(m);
    }
    __syncthreads();
    // exclusive scan of the 128 (round, wavefront) counts, which are in hit order, by wavefronts 0 and 1
    static_assert(CUT_ROUNDS * CUT_WAVES == 128 && CUT_WAVES >= 2, "two wavefronts scan the counts");
    uint32_t ex = 0;
    if (wv < 2) {
        uint32_t tot;
        ex = wave_scan(s_cnt[threadIdx.x], &tot);
        if (lane == 0) s_half[wv] = tot;
    }
    __syncthreads();
    const uint32_t total = s_half[0] + s_half[1];
    if (wv < 2) s_cnt[threadIdx.x] = ex + (wv ? s_half[0] : 0u);
    uint32_t excl;
    if (SERIAL) {
        excl = carry;
        carry += total;
        __syncthreads();
    } else {
        excl = chained_scan(gran, b, total, &dc->fail, s_scan, spin_limit);   // has a block barrier after the writes above
    }
#pragma unroll
    for (uint32_t r = 0; r < CUT_ROUNDS; ++r)
        if (fbits >> r & 1u) {
            const uint32_t c = excl + s_cnt[r * CUT_WAVES + wv] + rank[r];
            if (c < cl_cap) cstart[c] = b * CUT_BLOCK + r * CUT_THREADS + threadIdx.x;
        }
    // a serial pass stops behind the last hit instead of walking the whole capacity
    if (SERIAL && (uint64_t) b * CUT_BLOCK + CUT_BLOCK >= n_hits) b = n_blocks - 1;
    if (b == n_blocks - 1 && threadIdx.x == 0) dc->n_cl = (unsigned long long) excl + total;
    }
}

// One thread per cluster, 4 rounds of 512 per block: size threshold (src/utils.cpp:211-220) and,
// for the clusters kept, the summary row {first hit, size, read, prg << 1 | !forward, first
// position, last position, path rank of the first hit, cluster number} at the chained prefix sum of
// the kept counts.  The last block writes the kept total.
static const uint32_t KEEP_THREADS = 512, KEEP_WAVES = KEEP_THREADS / 64, KEEP_ROUNDS = 4, KEEP_BLOCK = KEEP_THREADS * KEEP_ROUNDS;

template <bool SERIAL>
__global__ __launch_bounds__(KEEP_THREADS) void cluster_keep_kernel(Counters *__restrict__ dc, uint32_t cl_cap,
                                                                    uint32_t hit_cap, const uint32_t *__restrict__ cstart,
                                                                    const uint32_t *__restrict__ s_read,
                                                                    const uint32_t *__restrict__ s_key,
                                                                    const uint32_t *__restrict__ s_pos,
                                                                    const uint32_t *__restrict__ s_rank,
                                                                    const uint32_t *__restrict__ thr, uint32_t thr_n,
                                                                    unsigned long long *__restrict__ gran,
                                                                    uint32_t *__restrict__ k_rows, uint32_t n_blocks,
                                                                    uint32_t spin_limit) {
    __shared__ uint32_t s_cnt[KEEP_ROUNDS * KEEP_WAVES], s_scan[2];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n_cl = clamp_count(&dc->n_cl, cl_cap), n_hits = clamp_count(&dc->n_hits, hit_cap);
    uint32_t carry = 0;
    for (uint32_t b = SERIAL ? 0u : blockIdx.x; b < n_blocks; b += SERIAL ? 1u : n_blocks) {
    if (SERIAL) __syncthreads();
    bool kept[KEEP_ROUNDS];
    uint32_t s[KEEP_ROUNDS], e[KEEP_ROUNDS], key[KEEP_ROUNDS], rank[KEEP_ROUNDS];
#pragma unroll
    for (uint32_t r = 0; r < KEEP_ROUNDS; ++r) {
        const uint32_t c = b * KEEP_BLOCK + r * KEEP_THREADS + threadIdx.x;
        kept[r] = false;
        s[r] = e[r] = key[r] = 0;
        if (c < n_cl) {
            s[r] = cstart[c];
            e[r] = c + 1 < n_cl ? cstart[c + 1] : n_hits;
            // When a chained prefix sum of the kernel before gave up waiting, cluster starts were written to wrong
            // places and some entries of cstart hold whatever the buffer held before: the batch will be mapped
            // again, but nothing here may be used as an index unchecked.
            if (s[r] < e[r] && e[r] <= n_hits) {
                key[r] = s_key[s[r]];
                kept[r] = e[r] - s[r] > thr[min(key[r] >> 1, thr_n - 1)];
            } else {
                s[r] = e[r] = 0;
            }
        }
        const unsigned long long m = ballot64(kept[r]);
        rank[r] = __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0));
        if (lane == 0) s_cnt[r * KEEP_WAVES + wv] = (uint32_t) __popcll(m);
    }
    __syncthreads();
    // (round, wavefront) counts are in cluster order
    uint32_t total = 0, before[KEEP_ROUNDS];
#pragma unroll
    for (uint32_t v = 0; v < KEEP_ROUNDS * KEEP_WAVES; ++v) {
#pragma unroll
        for (uint32_t r = 0; r < KEEP_ROUNDS; ++r)
            if (v == r * KEEP_WAVES + wv) before[r] = total;
        total += s_cnt[v];
    }
    uint32_t excl;
    if (SERIAL) {
        excl = carry;
        carry += total;
    } else {
        excl = chained_scan(gran, b, total, &dc->fail, s_scan, spin_limit);
    }
#pragma unroll
    for (uint32_t r = 0; r < KEEP_ROUNDS; ++r)
        if (kept[r]) {
            const uint32_t c = b * KEEP_BLOCK + r * KEEP_THREADS + threadIdx.x;
            uint32_t *row = k_rows + (uint64_t) (excl + before[r] + rank[r]) * 8;
            *(uint4 *) row = make_uint4(s[r], e[r] - s[r], s_read[s[r]], key[r]);
            *(uint4 *) (row + 4) = make_uint4(s_pos[s[r]], s_pos[e[r] - 1], s_rank[s[r]], c);
        }
    // a serial pass stops behind the last cluster instead of walking the whole capacity
    if (SERIAL && (uint64_t) b * KEEP_BLOCK + KEEP_BLOCK >= n_cl) b = n_blocks - 1;
    if (b == n_blocks - 1 && threadIdx.x == 0) dc->n_kept = (unsigned long long) excl + total;
    }
}

#include "pmap_tail.hip.h"

#endif
