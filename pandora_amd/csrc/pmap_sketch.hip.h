// Sketch kernel of libpandora_amd (gfx950): Seq::minimizer_sketch (src/seq.cpp:91-122) in its
// closed form, optionally fused with the index lookup of add_read_hits (src/utils.cpp:174-176).
// Included by pmap_kernels.hip.h only.
//
// Every wavefront works alone on 128 quads (512 consecutive k-mer start positions) of the padded
// base stream and never meets a workgroup barrier: it stages the packed words it needs in its own
// slice of LDS, hashes the 4 positions of two quads per lane (quads `lane` and `lane + 64`), takes
// window minima and the "equals the largest window minimum" test with 16-byte LDS reads, compacts
// the minimizers with ballots and (fused mode) looks them up in the index at once.  With
// a = w-1 = 4f+p and edge = f + (p>0), the first and last `edge` quads are halo: the wavefront
// emits the minimizers of its middle 128 - 2*edge quads, and consecutive wavefronts overlap by
// the halo.
#ifndef PMAP_SKETCH_HIP_H
#define PMAP_SKETCH_HIP_H

#include <type_traits>

static const int SK_NQ = 2;                        // quads per lane
static const int WAVE_Q = 64 * SK_NQ;              // quads per wavefront window
static const int WAVE_POS = 4 * WAVE_Q;            // 512 positions, also the slot stride per wavefront
static const int SK_NW = WAVE_POS / 16 + 4;        // packed words held in LDS (window + misalignment + extraction overrun)
static const int SK_MAX_W = 64;                    // edge <= 16 leaves at least 96 output quads
static const uint32_t SK_NOKMER = 0xFFFFFFFFu;

template <typename HT>
struct Quad {
    HT v[4];
};

// 16-byte LDS reads of whole quads, issued by hand: written as plain loads the compiler narrows
// them to the components a window shape needs, and dword reads at a 16-byte lane stride hit 8 of
// the 32 banks (4-way conflicts), while a full ds_read_b128 is conflict-free.  ldq_issue starts the
// read of quad q; the data may be used only after lds_wait() and then lds_arrive() on the quad (the
// wait covers every read issued so far; lds_arrive ties the quad's registers to a point after it,
// so that no use of them can be scheduled before the wait).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename HT>
struct QuadRaw {
    u32x4 lo, hi;   // hi only for 8-byte elements
};

__device__ __forceinline__ u32x4 lds_read_b128(const void *p) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((uint32_t) (uintptr_t) p) : "memory");
    return r;
}

template <typename HT>
__device__ __forceinline__ QuadRaw<HT> ldq_issue(const HT *a, int q) {
    QuadRaw<HT> r;
    r.lo = lds_read_b128(a + 4 * q);
    if (sizeof(HT) == 8) r.hi = lds_read_b128(a + 4 * q + 2);
    return r;
}

__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <typename HT>
__device__ __forceinline__ void lds_arrive(QuadRaw<HT> &r) {
    if (sizeof(HT) == 8) asm volatile("" : "+v"(r.lo), "+v"(r.hi));
    else asm volatile("" : "+v"(r.lo));
}

template <typename HT>
__device__ __forceinline__ Quad<HT> unraw(const QuadRaw<HT> &r) {
    Quad<HT> q;
    if (sizeof(HT) == 4) {
        q.v[0] = (HT) r.lo.x; q.v[1] = (HT) r.lo.y; q.v[2] = (HT) r.lo.z; q.v[3] = (HT) r.lo.w;
    } else {
        q.v[0] = (HT) (((uint64_t) r.lo.y << 32) | r.lo.x); q.v[1] = (HT) (((uint64_t) r.lo.w << 32) | r.lo.z);
        q.v[2] = (HT) (((uint64_t) r.hi.y << 32) | r.hi.x); q.v[3] = (HT) (((uint64_t) r.hi.w << 32) | r.hi.z);
    }
    return q;
}

// the quads base + step * i, i = 0..n-1 (n <= SK_MAXRUN), read together with the quads x0, x1, x2
static const int SK_MAXRUN = 4;
template <typename HT>
struct QuadSet {
    Quad<HT> x0, x1, x2, run[SK_MAXRUN];
};

template <typename HT>
__device__ __forceinline__ QuadSet<HT> ldq_set(const HT *a, int q0, int q1, int q2, int run0, int n) {
    QuadRaw<HT> r0 = ldq_issue(a, q0), r1 = ldq_issue(a, q1), r2 = ldq_issue(a, q2), rr[SK_MAXRUN];
#pragma unroll
    for (int i = 0; i < SK_MAXRUN; ++i)
        if (i < n) rr[i] = ldq_issue(a, run0 + i);
    lds_wait();
    lds_arrive(r0);
    lds_arrive(r1);
    lds_arrive(r2);
    QuadSet<HT> s;
    s.x0 = unraw(r0); s.x1 = unraw(r1); s.x2 = unraw(r2);
#pragma unroll
    for (int i = 0; i < SK_MAXRUN; ++i)
        if (i < n) {
            lds_arrive(rr[i]);
            s.run[i] = unraw(rr[i]);
        }
    return s;
}

// one quad, waited for at once (window shapes with long runs)
template <typename HT>
__device__ __forceinline__ Quad<HT> ldq(const HT *a, int q) {
    QuadRaw<HT> r = ldq_issue(a, q);
    lds_wait();
    lds_arrive(r);
    return unraw(r);
}

template <typename HT>
__device__ __forceinline__ void stq(HT *a, int q, const Quad<HT> &r) {
    if (sizeof(HT) == 4) {
        *(uint4 *) (a + 4 * q) = make_uint4((uint32_t) r.v[0], (uint32_t) r.v[1], (uint32_t) r.v[2], (uint32_t) r.v[3]);
    } else {
        *(ulonglong2 *) (a + 4 * q) = make_ulonglong2((unsigned long long) r.v[0], (unsigned long long) r.v[1]);
        *(ulonglong2 *) (a + 4 * q + 2) = make_ulonglong2((unsigned long long) r.v[2], (unsigned long long) r.v[3]);
    }
}

// Lanes of the wavefront where the predicate holds.  (__ballot takes an int: a bool goes through a
// 0/1 register and a second compare on its way there.)
__device__ __forceinline__ unsigned long long ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

template <typename HT> __device__ __forceinline__ HT tmin(HT a, HT b) { return a < b ? a : b; }
template <typename HT> __device__ __forceinline__ HT tmax(HT a, HT b) { return a > b ? a : b; }

// Orders this wavefront's LDS writes before its later LDS reads.  LDS executes one wavefront's
// instructions in order; the fence keeps the compiler from moving accesses across this point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (a << S) + b in one full-rate instruction.  Written out because the compiler otherwise folds the
// shift-add chains of the hash into quarter-rate 32-bit multiplies.
template <int S>
__device__ __forceinline__ uint32_t lshl_add(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(S), "v"(b));
    return r;
}

// Bit widths of the three right shifts of hash64 under a 2k-bit mask: (key & mask) >> s has
// max(2k - s, 0) significant bits.
struct HashWidths {
    uint32_t w24, w14, w28;
};
__device__ __forceinline__ HashWidths hash_widths(uint32_t k) {
    const uint32_t kb = 2 * k;
    HashWidths r;
    r.w24 = kb > 24 ? kb - 24 : 0;
    r.w14 = kb > 14 ? kb - 14 : 0;
    r.w28 = kb > 28 ? kb - 28 : 0;
    return r;
}

// src/inthash.cpp:115-125 for a mask of at most 30 bits (k <= 15).  Additions and left shifts
// carry only upwards, so the low 2k bits of every step are right whatever sits above them: the
// input need not be masked, each `key ^ key >> s` takes its shifted operand with a bit-field
// extract of the masked part, and the mask itself is applied once at the end.  The last step of
// the reference, (key + (key << 31)) & mask, leaves a <= 30-bit key unchanged.
__device__ __forceinline__ uint32_t hash64_u32(uint32_t key, uint32_t mask, const HashWidths &hw) {
    key = lshl_add<21>(key, ~key);
    key ^= __builtin_amdgcn_ubfe(key, 24u, hw.w24);
    key = lshl_add<8>(key, lshl_add<3>(key, key));
    key ^= __builtin_amdgcn_ubfe(key, 14u, hw.w14);
    key = lshl_add<4>(key, lshl_add<2>(key, key));
    // (key ^ field) & mask in one three-input bit operation (truth table 0x28 = (a ^ b) & c)
    const uint32_t t = __builtin_amdgcn_ubfe(key, 28u, hw.w28);
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x28" : "=v"(r) : "v"(key), "v"(t), "v"(mask));
    return r;
}

// reverse complement of 16 bases held in 32 bits (2 bits per base, first base on top)
__device__ __forceinline__ uint32_t revcomp16(uint32_t v) {
    const uint32_t r = __brev(v);   // reverses the bases and swaps the two bits of each
    return ~(((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1));
}

// inclusive prefix sum over the lanes of the wavefront
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    // inclusive scan inside each row of 16 lanes, then row totals carried across rows
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return v;
}

// sum over the wavefront, valid in every lane
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return (uint32_t) __builtin_amdgcn_readlane((int) wave_incl_scan(v), 63);
}

// Minimum over the a+1 positions ending at each of the 4 positions of quad Q (backward window),
// a = 4f + p >= 4.  Window of position 4Q+j starts at 4(Q-f) + j - p.
template <typename HT>
__device__ __forceinline__ Quad<HT> window_min_back(const HT *h, int Q, int f, int p) {
    // own quad, the two boundary quads, and the f-1 whole quads in between
    const int nmid = f - 1;
    const QuadSet<HT> qs = ldq_set(h, Q, Q - f, p ? Q - f - 1 : Q - f, Q - f + 1, nmid < SK_MAXRUN ? nmid : SK_MAXRUN);
    const Quad<HT> own = qs.x0, v = qs.x1, u = qs.x2;
    HT mid = (HT) ~(HT) 0;
#pragma unroll
    for (int i = 0; i < SK_MAXRUN; ++i)
        if (i < nmid) {
            const Quad<HT> &m = qs.run[i];
            mid = tmin(mid, tmin(tmin(m.v[0], m.v[1]), tmin(m.v[2], m.v[3])));
        }
    for (int q = Q - f + 1 + SK_MAXRUN; q < Q; ++q) {
        const Quad<HT> m = ldq(h, q);
        mid = tmin(mid, tmin(tmin(m.v[0], m.v[1]), tmin(m.v[2], m.v[3])));
    }
    // suffix minima of the two boundary quads, prefix minima of the own quad
    HT sv[4], su[4], pw[4];
    sv[3] = v.v[3]; sv[2] = tmin(v.v[2], sv[3]); sv[1] = tmin(v.v[1], sv[2]); sv[0] = tmin(v.v[0], sv[1]);
    su[3] = u.v[3]; su[2] = tmin(u.v[2], su[3]); su[1] = tmin(u.v[1], su[2]); su[0] = tmin(u.v[0], su[1]);
    pw[0] = own.v[0]; pw[1] = tmin(pw[0], own.v[1]); pw[2] = tmin(pw[1], own.v[2]); pw[3] = tmin(pw[2], own.v[3]);
    Quad<HT> out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = j - p;     // window start offset inside quad Q-f (negative: starts in quad Q-f-1)
        HT edge;
        if (d >= 0) edge = d == 0 ? sv[0] : d == 1 ? sv[1] : d == 2 ? sv[2] : sv[3];
        else edge = tmin(sv[0], d == -1 ? su[3] : d == -2 ? su[2] : su[1]);
        out.v[j] = tmin(tmin(pw[j], mid), edge);
    }
    return out;
}

// Maximum over the a+1 positions starting at each of the 4 positions of quad Q (forward window).
// Window of position 4Q+j ends at 4(Q+f) + j + p.
template <typename HT>
__device__ __forceinline__ Quad<HT> window_max_fwd(const HT *L, int Q, int f, int p) {
    const int nmid = f - 1;
    const QuadSet<HT> qs = ldq_set(L, Q, Q + f, p ? Q + f + 1 : Q + f, Q + 1, nmid < SK_MAXRUN ? nmid : SK_MAXRUN);
    const Quad<HT> own = qs.x0, v = qs.x1, u = qs.x2;
    HT mid = 0;
#pragma unroll
    for (int i = 0; i < SK_MAXRUN; ++i)
        if (i < nmid) {
            const Quad<HT> &m = qs.run[i];
            mid = tmax(mid, tmax(tmax(m.v[0], m.v[1]), tmax(m.v[2], m.v[3])));
        }
    for (int q = Q + 1 + SK_MAXRUN; q < Q + f; ++q) {
        const Quad<HT> m = ldq(L, q);
        mid = tmax(mid, tmax(tmax(m.v[0], m.v[1]), tmax(m.v[2], m.v[3])));
    }
    HT pv[4], pu[4], sw[4];
    pv[0] = v.v[0]; pv[1] = tmax(pv[0], v.v[1]); pv[2] = tmax(pv[1], v.v[2]); pv[3] = tmax(pv[2], v.v[3]);
    pu[0] = u.v[0]; pu[1] = tmax(pu[0], u.v[1]); pu[2] = tmax(pu[1], u.v[2]); pu[3] = tmax(pu[2], u.v[3]);
    sw[3] = own.v[3]; sw[2] = tmax(own.v[2], sw[3]); sw[1] = tmax(own.v[1], sw[2]); sw[0] = tmax(own.v[0], sw[1]);
    Quad<HT> out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = j + p;     // window end offset inside quad Q+f (>= 4: ends in quad Q+f+1)
        HT edge;
        if (e <= 3) edge = e == 0 ? pv[0] : e == 1 ? pv[1] : e == 2 ? pv[2] : pv[3];
        else edge = tmax(pv[3], e == 4 ? pu[0] : e == 5 ? pu[1] : pu[2]);
        out.v[j] = tmax(tmax(sw[j], mid), edge);
    }
    return out;
}

// One lookup of the index: the presence filter (pmap_filter_word / pmap_filter_mask: one or two bits
// per key in a bitmap small enough to stay in L2) turns most absent keys away before the slot table
// is touched.
__device__ __forceinline__ void index_lookup(const PmapSlot *__restrict__ slots, uint32_t log2cap,
                                             const uint32_t *__restrict__ filt, uint32_t fbits, uint64_t key,
                                             uint32_t *cnt, uint32_t *off, uint4 *first) {
    *cnt = 0;
    *off = 0;
    *first = make_uint4(0, 0, 0, 0);
    const uint32_t hm = pmap_key_mix(key);
    if (fbits) {
        const uint32_t m = pmap_filter_mask(key, hm, fbits);
        if ((filt[pmap_filter_word(key, hm, fbits)] & m) != m) return;
    }
    const uint32_t capmask = (1u << log2cap) - 1;
    uint32_t s = hm >> (32 - log2cap);
    for (uint32_t step = 0; step <= capmask; ++step) {   // at most half full: ends at an empty slot
        const uint4 v = *(const uint4 *) (slots + s);
        const uint64_t sk = ((uint64_t) v.y << 32) | v.x;
        if (sk == key) {
            // the record half of the slot shares the 64-byte sector that has just been fetched
            *off = v.z;
            *cnt = v.w;
            *first = *((const uint4 *) (slots + s) + 1);
            return;
        }
        if (sk == PMAP_EMPTY_KEY) return;
        s = (s + 1) & capmask;
    }
}

// Tiles are summed in groups of GROUP_TILES (group_reduce_kernel) so that the hit emission can find
// its offsets without a device-wide prefix sum.  (Adding the tile counts to group counters with
// atomics from this kernel was tried: ~3 atomics per wavefront on words shared by the wavefronts
// running at the same time tripled the kernel's time.)
static const uint32_t GROUP_TILES = 256;

struct SketchAux {
    unsigned long long *state;       // cleared by the first blocks: counters + words of the chained prefix sums
    uint32_t n_state_words;
    uint32_t *hbounds;               // set to all-ones by the first blocks
    uint32_t n_hb;
};

// FUSED = false: every minimizer of the wavefront goes to its slots of (mkmer, ents).
// FUSED = true : every minimizer is looked up in the index at once; only those with records go
//                to the slots of ents; the wavefront's hit count is kept.
// WC > 0 fixes w at compile time (window shapes unroll); WC = 0 takes w from the argument.
// Wavefront g owns output positions [g * 4 * out_q, (g + 1) * 4 * out_q), out_q = 128 - 2 * edge.
// words[t] = {16 packed bases, read index}; rmeta[r] = {first position of read r in the padded
// stream, its length or 0 if it has a letter outside ACGT}.
// tile_cnt[g] = {minimizers, entries, hits, 0} of the wavefront.
template <typename HT, bool FUSED, int WC>
__global__ __launch_bounds__(256) void sketch_wave_kernel(
    const uint2 *__restrict__ words, const uint2 *__restrict__ rmeta, uint32_t n_words, uint32_t n_tiles,
    uint32_t w_arg, uint32_t k, const PmapSlot *__restrict__ slots, uint32_t log2cap,
    const uint32_t *__restrict__ filt, uint32_t fbits, uint64_t *__restrict__ mkmer, uint4 *__restrict__ ents,
    uint4 *__restrict__ tile_cnt, SketchAux aux) {
    const uint32_t w = WC ? (uint32_t) WC : w_arg;
    // housekeeping for the kernels that follow in the stream, by the first blocks only: zero the
    // counters and the words of the chained prefix sums, set the per-read hit bounds to all-ones
    if (blockIdx.x * 256u < (aux.n_state_words > aux.n_hb ? aux.n_state_words : aux.n_hb)) {
        const uint32_t gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
        for (uint32_t i = gid; i < aux.n_state_words; i += gsz) aux.state[i] = 0ull;
        for (uint32_t i = gid; i < aux.n_hb; i += gsz) aux.hbounds[i] = 0xFFFFFFFFu;
    }
    __shared__ uint32_t sW[4][SK_NW], sStart[4][SK_NW], sLen[4][SK_NW], sRead[4][SK_NW];
    __shared__ __align__(16) HT sH[4][WAVE_POS];
    __shared__ __align__(16) HT sL[4][WAVE_POS + 8];

    // the wavefront's number is the same in every lane: kept in a scalar register, so is everything
    // derived from it (tile position, LDS slice addresses)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint32_t g = blockIdx.x * 4 + wv;
    if (g >= n_tiles) return;                       // whole wavefront leaves; nothing below uses a block barrier
    uint32_t *W = sW[wv], *mstart = sStart[wv], *mlen = sLen[wv], *wr = sRead[wv];
    HT *h = sH[wv], *Lv = sL[wv];

    const int a = (int) w - 1, f = a >> 2, p = a & 3;
    const int edge = f + (p ? 1 : 0);
    const int out_q = WAVE_Q - 2 * edge;
    const int64_t P0 = (int64_t) g * (4 * out_q) - 4 * edge;   // global position of quad 0, a multiple of 4
    const int64_t wbase = P0 >> 4;                              // floor: P0 is negative for g = 0
    const uint32_t off0 = (uint32_t) (P0 & 15);                 // 0, 4, 8 or 12
    const uint32_t P0lo = (uint32_t) P0;                        // positions are below 2^32; wrapping is harmless
    if (lane < SK_NW) {
        const int32_t wi = (int32_t) wbase + lane;      // word indices are below 2^28
        uint32_t word = 0, ms = 0, ml = 0, ri = 0;
        if (wi >= 0 && wi < (int32_t) n_words) {
            const uint2 wd = words[wi];
            word = wd.x;
            ri = wd.y;
            const uint2 rm = rmeta[ri];
            ms = rm.x;
            ml = rm.y;
            if (ml + 1 < w + k) ml = 0;             // too short to sketch (src/seq.cpp:92-94)
        }
        W[lane] = word;
        mstart[lane] = ms;
        mlen[lane] = ml;
        wr[lane] = ri;
    }
    if (lane < 2) stq(Lv, WAVE_Q + lane, Quad<HT>{{0, 0, 0, 0}});
    wave_lds_sync();

    // ---- canonical hashes of the lane's quads (src/seq.cpp:41-47, 109-111)
    const HT mask = (HT) (((uint64_t) 1 << (2 * k)) - 1);
    const HashWidths hw = hash_widths(k);
    // byte selector that funnels the 4 bytes starting (lane's misalignment) bytes into a word pair;
    // the misalignment is the same for both quads of the lane (they are 64 quads = 16 words apart)
    const uint32_t mis = ((uint32_t) lane + (off0 >> 2)) & 3u;     // quad inside its packed word
    const uint32_t bsel = mis == 0 ? 0x07060504u : mis == 1 ? 0x06050403u : mis == 2 ? 0x05040302u : 0x04030201u;
    // Positions that are no k-mer of a read (padding, reads too short or with a bad letter) get
    // hashes of whatever bits are there: they never reach a result, because a window minimum counts
    // only if its whole window lies inside one read, and a flag only at a real k-mer.
    Quad<HT> own[SK_NQ];
    int lim[SK_NQ], lo[SK_NQ];      // k-mer j of the quad exists iff j <= lim; its full backward window iff also j >= lo
    uint32_t nstrand = 0;           // bit 7 - (4 hq + j): h_fwd > h_rc at position j of quad hq, i.e. strand = 0
#pragma unroll
    for (int hq = 0; hq < SK_NQ; ++hq) {
        const int Q = lane + 64 * hq;
        const int wi = (int) ((4u * (uint32_t) Q + off0) >> 4);
        const uint32_t len = mlen[wi];
        const uint32_t j0 = P0lo + 4u * (uint32_t) Q - mstart[wi];     // position of the quad inside its read
        lim[hq] = len ? (int) (len - k - j0) : -1;
        lo[hq] = a - (int) j0;
        const uint32_t w0 = W[wi], w1 = W[wi + 1], w2 = W[wi + 2];
        // 64 stream bits starting at the quad's first base
        const uint32_t x0 = __builtin_amdgcn_perm(w0, w1, bsel), x1 = __builtin_amdgcn_perm(w1, w2, bsel);
        if constexpr (sizeof(HT) == 4) {
            // k <= 15.  With Y = reverse complement of the 32 bases x0:x1, the reverse complement of
            // the k-mer at base j is (Y >> 2j) & mask; hash64_u32 ignores what lies above the mask.
            const uint32_t ylo = revcomp16(x0), yhi = revcomp16(x1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t top = j ? __builtin_amdgcn_alignbit(x0, x1, 32 - 2 * j) : x0;
                const uint32_t fwd = top >> (32 - 2 * k);
                const uint32_t rc = j ? __builtin_amdgcn_alignbit(yhi, ylo, 2 * j) : ylo;
                const uint32_t hf = hash64_u32(fwd, mask, hw), hr = hash64_u32(rc, mask, hw);
                // hashes are below 2^30: the sign of hr - hf tells hf > hr; strand = (hf <= hr)
                nstrand = __builtin_amdgcn_alignbit(nstrand, hr - hf, 31);   // nstrand << 1 | sign
                own[hq].v[j] = tmin(hf, hr);
            }
        } else {
            const uint32_t w3 = W[wi + 3];
            const uint32_t x2 = __builtin_amdgcn_perm(w2, w3, bsel);   // the next 32 bits
            const uint64_t x = ((uint64_t) x0 << 32) | x1;
            uint64_t rc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint64_t z = x << (2 * j);
                if (j) z |= (uint64_t) (x2 >> (32 - 2 * j));
                const uint64_t fwd = z >> (64 - 2 * k);
                // reverse complement: from scratch for the first k-mer, then one base at a time (src/seq.cpp:44)
                if (j == 0) rc = revcomp_dev(fwd, k);
                else rc = (rc >> 2) | ((uint64_t) (3u ^ ((uint32_t) fwd & 3u)) << (2 * k - 2));
                const uint64_t hf = hash64_dev<uint64_t>(fwd, mask), hr = hash64_dev<uint64_t>(rc, mask);
                own[hq].v[j] = tmin(hf, hr);
                nstrand = nstrand << 1 | (hf > hr ? 1u : 0u);
            }
        }
        stq(h, Q, own[hq]);
    }
    wave_lds_sync();

    // ---- Lv[t] = minimum of the full window ending at k-mer t, 0 if that window is not inside one read
#pragma unroll
    for (int hq = 0; hq < SK_NQ; ++hq) {
        const int Q = lane + 64 * hq;
        const int Qc = Q < edge ? edge : Q;         // halo quads: any in-range window will do, their result is never used
        Quad<HT> m;
        if (a >= 4) {
            m = window_min_back<HT>(h, Qc, f, p);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                HT x = h[4 * Qc + j];
                for (int d = 1; d <= a; ++d) x = tmin(x, h[4 * Qc + j - d]);
                m.v[j] = x;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j > lim[hq] || j < lo[hq]) m.v[j] = 0;
        stq(Lv, Q, m);
    }
    wave_lds_sync();

    // ---- k-mer i is a minimizer iff it equals the largest of the minima of the windows containing it
    bool fl[SK_NQ][4];
#pragma unroll
    for (int hq = 0; hq < SK_NQ; ++hq) {
        const int Q = lane + 64 * hq;
        const bool inr = (uint32_t) (Q - edge) < (uint32_t) out_q;
        const int Qc = Q > WAVE_Q - 1 - edge ? WAVE_Q - 1 - edge : Q;
        Quad<HT> mx;
        if (a >= 4) {
            mx = window_max_fwd<HT>(Lv, Qc, f, p);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                HT x = Lv[4 * Qc + j];
                for (int d = 1; d <= a; ++d) x = tmax(x, Lv[4 * Qc + j + d]);
                mx.v[j] = x;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) fl[hq][j] = inr && j <= lim[hq] && mx.v[j] == own[hq].v[j];
    }
    wave_lds_sync();                                // every lane has read Lv: it becomes the list of minimizer positions

    // ---- compaction in position order with ballots: list[e] = index (4 * quad + j) of the e-th
    // minimizer, with its strand in bit 16
    uint32_t *list = (uint32_t *) Lv;
    uint32_t n = 0;
#pragma unroll
    for (int hq = 0; hq < SK_NQ; ++hq) {
        uint32_t o = n, c = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long b = ballot64(fl[hq][j]);
            // flags of lower lanes are earlier positions whatever their j
            o += __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0));
            c += __popcll(b);
        }
        n += c;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (fl[hq][j])
                list[o++] = (4u * (uint32_t) (lane + 64 * hq) + j) | ((~nstrand >> (7 - (4 * hq + j)) & 1u) << 16);
    }
    wave_lds_sync();

    const uint64_t slot0 = (uint64_t) g * WAVE_POS;
    uint32_t nent = 0, hits = 0;
    for (uint32_t base = 0; base < n; base += 64) {     // n is wave-uniform
        const uint32_t e = base + lane;
        uint32_t cnt = 0, off = 0, ps = 0, rd = 0;
        HT hv = 0;
        if (e < n) {
            const uint32_t le = list[e], i = le & 0xFFFFu, wi = (i + off0) >> 4;
            hv = h[i];
            ps = (P0lo + i - mstart[wi]) | (le >> 16) << 31;     // position inside the read, strand on top
            rd = wr[wi];
        }
        if (!FUSED) {
            if (e < n) {
                mkmer[slot0 + e] = (uint64_t) hv;
                ents[2 * (slot0 + e)] = make_uint4(rd, ps, 0, 0);
            }
        } else {
            uint4 first;
            if (e < n) index_lookup(slots, log2cap, filt, fbits, (uint64_t) hv, &cnt, &off, &first);
            const unsigned long long b = ballot64(cnt != 0);
            if (cnt) {
                const uint64_t s = slot0 + nent +
                                   __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0));
                ents[2 * s] = make_uint4(rd, ps, off, cnt);
                ents[2 * s + 1] = first;
            }
            nent += __popcll(b);
            hits += cnt;
        }
    }
    if (FUSED) hits = wave_sum(hits);
    else nent = n;
    if (lane == 0) tile_cnt[g] = make_uint4(n, nent, hits, 0);
}

// ---------------------------------------------------------------------------------------------
// k <= 15 and 2 <= w <= 16: tiles with their windows kept in registers.
//
// Lane l holds NP consecutive positions NP*l .. NP*l + NP-1 of the wavefront's 64 * NP.  A window of
// w positions is built by doubling: after "v[t] = min(v[t], v[t-S])" with S = 1, 2, 4, ... every
// v[t] covers 2S positions, and a last step with S = w - (covered) completes it.  No step reaches
// further back than 8 positions, so the operand is either another register of the lane or a
// register of the lane before it, taken inside the min / max instruction itself by a DPP
// whole-wavefront shift (wave_shr:1 / wave_shl:1, full rate on gfx950).  The hashes never go
// through LDS for the windows: 4 instructions per position for the window minimum and 4 for the
// window maximum.
//
// The tile starts 16 positions before its first output position (a multiple of 16: lane l's
// bases start at a word or half-word boundary of the packed stream) and has 16 halo positions at
// the end.  NP = 16 halves what is paid per tile (halo, set-up) and fills the 64 lanes of the
// lookup rounds better (about 132 minimizers in 3 rounds instead of 66 in 2) at 6 instead of 8
// wavefronts per SIMD.
static const int DPP_HALO = 16;
static const int DPP_MAX_W = 16;
// Minimizers of a tile go through a list of {hash, position in the tile} in LDS, one list per wavefront.
// Sparse sketches (w >= 8: about one position in eight) get a list of CAP = 5/8 of the tile's positions,
// which makes 6 workgroups fit a compute unit (measured best for the whole pipeline: 8 leave the kernels
// of the batch before no room next to this one, 5 hide less latency); a tile with more minimizers than
// that (low-complexity sequence) is left to sketch_dpp_overflow_kernel, which the host adds once a batch
// had such a tile.  Dense sketches get a full list.
// The lists are DYNAMIC LDS, sized by the launch: the compiler pads a kernel's register allocation up
// to what the occupancy allowed by its STATIC LDS leaves room for (24 KB of static LDS: 6 waves per SIMD,
// 73 registers allocated for the 50 in use), and those registers are then missing for the kernels
// of the batch before, which run next to this one (0.225 instead of 0.19 ms per step of the pipeline).
template <int WC, int NP>
struct DppList {
    static const int CAP = WC >= 8 ? 64 * NP * 5 / 8 : 64 * NP;
    static const unsigned BYTES = 4u * CAP * 8u;          // 4 wavefronts per workgroup, 8 bytes per entry
};

template <int NP>
struct DppTile {
    static const int POS = 64 * NP;                 // positions of a tile = slot stride of its outputs
    static const int OUT = POS - 2 * DPP_HALO;      // output positions
    static const int NWD = POS / 16 + 4;            // packed words staged in LDS
};

// value of x in the lane before / after; `fill` where there is none (lane 0 / lane 63)
__device__ __forceinline__ uint32_t lane_before(uint32_t x, uint32_t fill) {
    return (uint32_t) __builtin_amdgcn_update_dpp((int) fill, (int) x, 0x138, 0xf, 0xf, false);   // wave_shr:1
}
__device__ __forceinline__ uint32_t lane_after(uint32_t x, uint32_t fill) {
    return (uint32_t) __builtin_amdgcn_update_dpp((int) fill, (int) x, 0x130, 0xf, 0xf, false);   // wave_shl:1
}

// v[t] = min(v[t], v[t - S]) over the wavefront's positions, 1 <= S <= N
template <int S, int N>
__device__ __forceinline__ void slide_min(uint32_t (&v)[N]) {
    uint32_t o[N];
#pragma unroll
    for (int j = 0; j < N; ++j)
        o[j] = j >= S ? tmin(v[j], v[j >= S ? j - S : 0]) : tmin(v[j], lane_before(v[j >= S ? 0 : N - S + j], 0xFFFFFFFFu));
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = o[j];
}
// v[t] = max(v[t], v[t + S])
template <int S, int N>
__device__ __forceinline__ void slide_max(uint32_t (&v)[N]) {
    uint32_t o[N];
#pragma unroll
    for (int j = 0; j < N; ++j)
        o[j] = j + S < N ? tmax(v[j], v[j + S < N ? j + S : 0]) : tmax(v[j], lane_after(v[j + S < N ? 0 : j + S - N], 0u));
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = o[j];
}
// v[t] = min(v[t-W+1 .. t]); C = positions covered so far
template <int W, int N, int C = 1>
__device__ __forceinline__ void window_min_regs(uint32_t (&v)[N]) {
    if constexpr (2 * C <= W) {
        slide_min<C, N>(v);
        window_min_regs<W, N, 2 * C>(v);
    } else if constexpr (C < W) {
        slide_min<W - C, N>(v);
    }
}
// v[t] = max(v[t .. t+W-1])
template <int W, int N, int C = 1>
__device__ __forceinline__ void window_max_regs(uint32_t (&v)[N]) {
    if constexpr (2 * C <= W) {
        slide_max<C, N>(v);
        window_max_regs<W, N, 2 * C>(v);
    } else if constexpr (C < W) {
        slide_max<W - C, N>(v);
    }
}

// One tile of sketch_dpp_kernel, by one wavefront: tile g of the batch, its minimizers through `list` (CAP
// entries of LDS).  Returns false, with nothing written, when the tile has more than CAP minimizers;
// otherwise writes the tile's outputs and tile_cnt[g] = {minimizers, entries, hits, mark}.
// W, mstart, mlen, wr: the wavefront's DppTile<NP>::NWD words of LDS each; strands: 64 words.
template <bool FUSED, int WC, int KC, int NP, int CAP>
__device__ __forceinline__ bool sketch_dpp_tile(
    const uint32_t g, const int lane, uint32_t *W, uint32_t *mstart, uint32_t *mlen, uint32_t *wr, uint32_t *strands,
    uint2 *list, const uint32_t mark, const uint2 *__restrict__ words, const uint2 *__restrict__ rmeta, uint32_t n_words,
    uint32_t k_arg, const PmapSlot *__restrict__ slots, uint32_t log2cap, const uint32_t *__restrict__ filt,
    uint32_t fbits, uint64_t *__restrict__ mkmer, uint4 *__restrict__ ents, uint4 *__restrict__ tile_cnt) {
    typedef DppTile<NP> T;
    const uint32_t w = WC, k = KC ? (uint32_t) KC : k_arg;
    const int a = WC - 1;
    const int64_t P0 = (int64_t) g * T::OUT - DPP_HALO;        // position 0 of the tile, a multiple of 16
    const int32_t wbase = (int32_t) (P0 >> 4);                  // -1 for g = 0
    const uint32_t P0lo = (uint32_t) P0;
    // the tile's POS / 16 packed words with their read's start / length / index, one per lane, and the word
    // behind them (the k-mers of the last positions reach into it) by one more lane
    const int NW = T::POS / 16;
    if (lane < NW) {
        const int32_t wi = wbase + lane;
        uint32_t word = 0, ms = 0, ml = 0, ri = 0;
        if (wi >= 0 && wi < (int32_t) n_words) {
            const uint2 wd = words[wi];
            word = wd.x;
            ri = wd.y;
            const uint2 rm = rmeta[ri];
            ms = rm.x;
            ml = rm.y;
            if (ml + 1 < w + k) ml = 0;             // too short to sketch (src/seq.cpp:92-94)
        }
        W[lane] = word;
        mstart[lane] = ms;
        mlen[lane] = ml;
        wr[lane] = ri;
    }
    if (lane == (NW < 64 ? NW : 0)) {
        const int32_t wi = wbase + NW;
        W[NW] = wi >= 0 && wi < (int32_t) n_words ? words[wi].x : 0u;
    }
    wave_lds_sync();

    // ---- canonical hashes of the lane's NP positions (src/seq.cpp:41-47, 109-111)
    const uint32_t mask = (uint32_t) (((uint64_t) 1 << (2 * k)) - 1);
    const HashWidths hw = hash_widths(k);
    constexpr bool LAZY_STRAND = FUSED && WC >= 8;
    const uint64_t slot0 = (uint64_t) g * T::POS;
    uint32_t n = 0, nent = 0, hits = 0;
    const int wi = NP == 16 ? lane : lane >> 1;
    const uint32_t len = mlen[wi];
    const uint32_t j0 = P0lo + (uint32_t) NP * (uint32_t) lane - mstart[wi];   // position of the lane's first base inside its read
    const int lim = len ? (int) (len - k - j0) : -1;                // k-mer j exists iff j <= lim
    const int lo = a - (int) j0;                                    // its full backward window iff also j >= lo
    // the 32 bases from the lane's first one (NP = 8: odd lanes start in mid-word and only the top
    // half of x1 is filled, which is as far as their k-mers reach)
    const uint32_t w0 = W[wi], w1 = W[wi + 1];
    uint32_t x0, x1;
    if (NP == 16) {
        x0 = w0;
        x1 = w1;
    } else {
        const uint32_t bsel = (lane & 1) ? 0x05040302u : 0x07060504u;
        x0 = __builtin_amdgcn_perm(w0, w1, bsel);
        x1 = __builtin_amdgcn_perm(w1, 0u, bsel);
    }
    // With Y = reverse complement of the 32 bases x0:x1, the reverse complement of the k-mer at
    // base j is (Y >> 2j) & mask; hash64_u32 ignores what lies above the mask.
    const uint32_t ylo = revcomp16(x0), yhi = revcomp16(x1);
    uint32_t own[NP];
    uint32_t nstrand = 0;           // bit NP-1 - j: h_fwd > h_rc at position j, i.e. strand = 0
    // Sparse sketches in fused mode (LAZY_STRAND): only the few minimizers that have records need their
    // strand, and hashing their forward k-mer once more (strand = the canonical hash is the forward one)
    // costs less than keeping two instructions per position here.
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        uint32_t fwd;
        if constexpr (KC > 0) {
            // the k-mer sits 64 - 2k - 2j bits up in x0:x1; one shift brings it down (bits above it do not matter)
            const int sh = 64 - 2 * KC - 2 * j;     // a constant once the loop is unrolled
            if (sh > 32) fwd = x0 >> (sh - 32);
            else if (sh == 32) fwd = x0;
            else fwd = __builtin_amdgcn_alignbit(x0, x1, (uint32_t) sh);
        } else {
            const uint32_t top = j ? __builtin_amdgcn_alignbit(x0, x1, 32 - 2 * j) : x0;
            fwd = top >> (32 - 2 * k);
        }
        const uint32_t rc = j ? __builtin_amdgcn_alignbit(yhi, ylo, 2 * j) : ylo;
        const uint32_t hf = hash64_u32(fwd, mask, hw), hr = hash64_u32(rc, mask, hw);
        // hashes are below 2^30: the sign of hr - hf tells hf > hr; strand = (hf <= hr)
        if constexpr (!LAZY_STRAND) nstrand = __builtin_amdgcn_alignbit(nstrand, hr - hf, 31);   // nstrand << 1 | sign
        own[j] = tmin(hf, hr);
    }
    if constexpr (!LAZY_STRAND) strands[lane] = ~nstrand;

    // ---- L[t] = minimum of the full window ending at k-mer t, 0 if that window is not inside one read;
    //      k-mer i is a minimizer iff it equals the largest of the minima of the windows containing it.
    // A wavefront whose every position is a k-mer with its whole backward window inside the read
    // (the usual case in long reads) needs no masking; the first 16 positions are halo, their windows never count.
    uint32_t L[NP], cmp[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) L[j] = cmp[j] = own[j];
    window_min_regs<WC, NP>(L);
    const unsigned long long HALO_LANES = (1ull << (DPP_HALO / NP)) - 1;
    const bool plain = ballot64(lim >= NP - 1) == ~0ull && (ballot64(lo <= 0) | HALO_LANES) == ~0ull;
    if (!plain) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (j > lim || j < lo) L[j] = 0;
            if (j > lim) cmp[j] = 0xFFFFFFFFu;      // no k-mer here: equals no window minimum (hashes are below 2^30)
        }
    }
    window_max_regs<WC, NP>(L);

    // ---- compaction in position order: list[e] = {hash, index in the tile} of the e-th minimizer of the pass
    const bool inr = (uint32_t) (lane - DPP_HALO / NP) < (uint32_t) (T::OUT / NP);
    // every lane counts its own flags (one add-with-carry per position), a prefix sum over the lanes
    // places the lane's first minimizer
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) mine += (inr && L[j] == cmp[j]) ? 1u : 0u;
    const uint32_t upto = wave_incl_scan(mine);
    n = (uint32_t) __builtin_amdgcn_readlane((int) upto, 63);
    uint32_t o = upto - mine;
    const uint32_t laneNP = (uint32_t) NP * (uint32_t) lane;
    asm volatile("" : "+v"(o));         // keeps the sum whole: one scaling to a byte offset, not one per term
    if (n > (uint32_t) CAP) return false;       // wave-uniform; never so for dense sketches
    {
        uint2 *lp = list + o;
#pragma unroll
        for (int j = 0; j < NP; ++j)
            if (inr && L[j] == cmp[j]) *lp++ = make_uint2(cmp[j], laneNP | (uint32_t) j);
    }
    wave_lds_sync();
    const uint32_t np = n;

    if constexpr (!FUSED) {
        for (uint32_t base = 0; base < np; base += 64) {     // np is wave-uniform
            const uint32_t e = base + lane;
            if (e < np) {
                const uint2 le = list[e];
                const uint32_t i = le.y, wj = i >> 4;
                // strand of position j of a lane: bit NP-1 - j of the lane's word; moved to bit 31
                const uint32_t st = strands[i / NP] << (32 - NP + (i & (NP - 1)));
                mkmer[slot0 + e] = (uint64_t) le.x;
                ents[2 * (slot0 + e)] = make_uint4(wr[wj], (P0lo + i - mstart[wj]) | (st & 0x80000000u), 0, 0);
            }
        }
    } else {
        // ---- presence filter for every minimizer: up to 4 rounds of 64 at a time, their loads in
        // flight together; what passes stays in the list (compacted in place, position order kept)
        uint32_t n2 = 0;
        for (uint32_t base = 0; base < np; base += 256) {
            uint2 le[4];
            uint32_t fw[4], fb[4], fb2[4];
            const uint32_t fsh = (32 - (fbits & 0xFFu)) & 31u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                le[r] = make_uint2(0, 0);
                fw[r] = 0;          // lanes past the end pass nothing
                fb[r] = fb2[r] = 0;
                if (base + 64 * r < np) {
                    const uint32_t e = base + 64 * r + lane;
                    if (e < np) {
                        le[r] = list[e];
                        fw[r] = 0xFFFFFFFFu;
                        if (fbits) {
                            const uint32_t hm = pmap_key_mix((uint64_t) le[r].x);
                            fb[r] = (fbits & PMAP_FILTER_DIRECT) ? le[r].x : hm >> fsh;     // pmap_filter_bit
                            if (fbits & PMAP_FILTER_TWO) fb2[r] = hm >> 3;
                            fw[r] = filt[fb[r] >> 5];
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (base + 64 * r < np) {
                    // bit fb & 31 of the word and, with two bits per key, bit fb2 & 31 as well
                    uint32_t t = __builtin_amdgcn_ubfe(fw[r], fb[r], 1u);
                    if (fbits & PMAP_FILTER_TWO) t &= fw[r] >> (fb2[r] & 31u);
                    const bool pass = t != 0;
                    const unsigned long long b = ballot64(pass);
                    if (pass)
                        list[n2 + __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0))] = le[r];
                    n2 += __popcll(b);
                }
        }
        wave_lds_sync();
        // ---- table lookup of those, entries of the ones with records
        for (uint32_t base = 0; base < n2; base += 64) {
            const uint32_t e = base + lane;
            uint32_t cnt = 0, off = 0, ps = 0, rd = 0;
            uint4 first = make_uint4(0, 0, 0, 0);
            if (e < n2) {
                const uint2 le = list[e];
                const uint32_t i = le.y, wj = i >> 4, hv = le.x;
                index_lookup(slots, log2cap, nullptr, 0u, (uint64_t) hv, &cnt, &off, &first);
                uint32_t st;
                if constexpr (LAZY_STRAND) {
                    // the forward k-mer at position i: 2 (i & 15) bits into the words wj, wj + 1
                    const uint64_t xx = ((uint64_t) W[wj] << 32) | W[wj + 1];
                    const uint32_t top = (uint32_t) ((xx << (2u * (i & 15u))) >> 32);
                    st = hash64_u32(top >> (32 - 2 * k), mask, hw) == hv ? 0x80000000u : 0u;   // h_fwd <= h_rc
                } else {
                    st = strands[i / NP] << (32 - NP + (i & (NP - 1)));
                }
                ps = (P0lo + i - mstart[wj]) | (st & 0x80000000u);     // position inside the read, strand on top
                rd = wr[wj];
            }
            const unsigned long long b = ballot64(cnt != 0);
            if (cnt) {
                const uint64_t s = slot0 + nent +
                                   __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0));
                ents[2 * s] = make_uint4(rd, ps, off, cnt);
                ents[2 * s + 1] = first;
            }
            nent += __popcll(b);
            hits += cnt;
        }
    }
    if (FUSED) hits = wave_sum(hits);
    else nent = n;
    if (lane == 0) tile_cnt[g] = make_uint4(n, nent, hits, mark);
    return true;
}


// Same arguments and outputs as sketch_wave_kernel<uint32_t, FUSED, WC>; a tile has
// DppTile<NP>::OUT output positions and its outputs sit at a slot stride of DppTile<NP>::POS.
// KC > 0 fixes k at compile time.  Dynamic LDS: DppList<WC, NP>::BYTES.
// A tile with more minimizers than its list holds gets tile_cnt = {0, 0, 0, 1} and nothing else: the
// batch totals then tell the host to map the batch again with sketch_dpp_overflow_kernel behind this one.
template <bool FUSED, int WC, int KC, int NP>
__global__ __launch_bounds__(256) void sketch_dpp_kernel(
    const uint2 *__restrict__ words, const uint2 *__restrict__ rmeta, uint32_t n_words, uint32_t n_tiles,
    uint32_t w_arg, uint32_t k_arg, const PmapSlot *__restrict__ slots, uint32_t log2cap,
    const uint32_t *__restrict__ filt, uint32_t fbits, uint64_t *__restrict__ mkmer, uint4 *__restrict__ ents,
    uint4 *__restrict__ tile_cnt, SketchAux aux) {
    static_assert(WC >= 1 && WC <= DPP_MAX_W && WC - 1 <= DPP_HALO, "window steps must stay within the neighbouring lane");
    static_assert(NP == 8 || NP == 16, "a lane starts at a word or half-word boundary");
    typedef DppTile<NP> T;
    (void) w_arg;
    if (blockIdx.x * 256u < (aux.n_state_words > aux.n_hb ? aux.n_state_words : aux.n_hb)) {
        const uint32_t gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
        for (uint32_t i = gid; i < aux.n_state_words; i += gsz) aux.state[i] = 0ull;
        for (uint32_t i = gid; i < aux.n_hb; i += gsz) aux.hbounds[i] = 0xFFFFFFFFu;
    }
    constexpr int CAP = DppList<WC, NP>::CAP;
    __shared__ uint32_t sW[4][T::NWD], sStart[4][T::NWD], sLen[4][T::NWD], sRead[4][T::NWD];
    extern __shared__ __align__(8) unsigned char sketch_dyn_lds[];      // DppList<WC, NP>::BYTES, from the launch
    __shared__ uint32_t sStrand[4][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint32_t g = blockIdx.x * 4 + wv;
    if (g >= n_tiles) return;
    const bool done = sketch_dpp_tile<FUSED, WC, KC, NP, CAP>(
        g, lane, sW[wv], sStart[wv], sLen[wv], sRead[wv], sStrand[wv], (uint2 *) sketch_dyn_lds + wv * CAP, 0u, words, rmeta,
        n_words, k_arg, slots, log2cap, filt, fbits, mkmer, ents, tile_cnt);
    if (!done && lane == 0) tile_cnt[g] = make_uint4(0, 0, 0, 1);
}

// The tiles sketch_dpp_kernel left out (tile_cnt[g].w == 1), with a list as long as the tile; marks them
// with tile_cnt[g].w = 2, so that the batch totals still tell that the batch had such tiles.  Any grid:
// a wavefront looks at 64 tiles at a time.  Dynamic LDS: 4 * DppTile<NP>::POS * 8 bytes.
template <bool FUSED, int WC, int KC, int NP>
__global__ __launch_bounds__(256) void sketch_dpp_overflow_kernel(
    const uint2 *__restrict__ words, const uint2 *__restrict__ rmeta, uint32_t n_words, uint32_t n_tiles,
    uint32_t w_arg, uint32_t k_arg, const PmapSlot *__restrict__ slots, uint32_t log2cap,
    const uint32_t *__restrict__ filt, uint32_t fbits, uint64_t *__restrict__ mkmer, uint4 *__restrict__ ents,
    uint4 *__restrict__ tile_cnt) {
    typedef DppTile<NP> T;
    (void) w_arg;
    __shared__ uint32_t sW[4][T::NWD], sStart[4][T::NWD], sLen[4][T::NWD], sRead[4][T::NWD];
    extern __shared__ __align__(8) unsigned char sketch_dyn_lds[];
    __shared__ uint32_t sStrand[4][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    for (uint64_t t0 = ((uint64_t) blockIdx.x * 4 + wv) * 64; t0 < n_tiles; t0 += (uint64_t) gridDim.x * 4 * 64) {
        const uint64_t t = t0 + lane;
        unsigned long long todo = ballot64(t < n_tiles && tile_cnt[t].w == 1u);
        while (todo) {                  // wave-uniform
            const uint32_t g = (uint32_t) t0 + (uint32_t) __builtin_ctzll(todo);
            todo &= todo - 1;
            sketch_dpp_tile<FUSED, WC, KC, NP, T::POS>(g, lane, sW[wv], sStart[wv], sLen[wv], sRead[wv], sStrand[wv],
                                                      (uint2 *) sketch_dyn_lds + wv * T::POS, 2u, words, rmeta, n_words,
                                                      k_arg, slots, log2cap, filt, fbits, mkmer, ents, tile_cnt);
            wave_lds_sync();            // the next tile writes the same LDS
        }
    }
}

#endif
