// Sketch kernel of libpandora_amd (gfx950): Seq::minimizer_sketch (src/seq.cpp:91-122) in its
// closed form, optionally fused with the index lookup of add_read_hits (src/utils.cpp:174-176).
// Included by pmap_kernels.hip.h only.
//
// Every wavefront works alone on 64 quads (256 consecutive k-mer start positions) of the padded
// base stream and never meets a workgroup barrier: it stages the packed words it needs in its own
// slice of LDS, hashes the 4 positions of one quad per lane, takes window minima and the "equals
// the largest window minimum" test with 16-byte LDS reads, compacts the minimizers with ballots
// and (fused mode) looks them up in the index at once.  With a = w-1 = 4f+p and edge = f + (p>0),
// the first and last `edge` quads are halo: the wavefront emits the minimizers of its middle
// 64 - 2*edge quads, and consecutive wavefronts overlap by the halo.
#ifndef PMAP_SKETCH_HIP_H
#define PMAP_SKETCH_HIP_H

static const int WAVE_Q = 64;                      // quads per wavefront window
static const int WAVE_POS = 4 * WAVE_Q;            // 256 positions, also the slot stride per wavefront
static const int SK_NW = WAVE_POS / 16 + 4;        // packed words held in LDS (window + misalignment + extraction overrun)
static const int SK_MAX_W = 64;                    // edge <= 16 leaves at least 32 output quads
static const uint32_t SK_NOKMER = 0xFFFFFFFFu;

template <typename HT>
struct Quad {
    HT v[4];
};

template <typename HT>
__device__ __forceinline__ Quad<HT> ldq(const HT *a, int q) {
    Quad<HT> r;
    if (sizeof(HT) == 4) {
        const uint4 x = *(const uint4 *) (a + 4 * q);
        r.v[0] = (HT) x.x; r.v[1] = (HT) x.y; r.v[2] = (HT) x.z; r.v[3] = (HT) x.w;
    } else {
        const ulonglong2 x = *(const ulonglong2 *) (a + 4 * q), y = *(const ulonglong2 *) (a + 4 * q + 2);
        r.v[0] = (HT) x.x; r.v[1] = (HT) x.y; r.v[2] = (HT) y.x; r.v[3] = (HT) y.y;
    }
    return r;
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

// src/inthash.cpp:115-125 for a mask of at most 30 bits (k <= 15): every step is masked to 2k
// bits, so arithmetic modulo 2^32 gives the same low 2k bits as modulo 2^64, and the last step
// (key + (key << 31)) & mask leaves key unchanged.
__device__ __forceinline__ uint32_t hash64_u32(uint32_t key, uint32_t mask) {
    key = lshl_add<21>(key, ~key) & mask;
    key = key ^ key >> 24;
    key = lshl_add<8>(key, lshl_add<3>(key, key)) & mask;
    key = key ^ key >> 14;
    key = lshl_add<4>(key, lshl_add<2>(key, key)) & mask;
    key = key ^ key >> 28;
    return key;
}

__device__ __forceinline__ uint32_t hash_any(uint32_t key, uint32_t mask) { return hash64_u32(key, mask); }
__device__ __forceinline__ uint64_t hash_any(uint64_t key, uint64_t mask) { return hash64_dev<uint64_t>(key, mask); }

// Minimum over the a+1 positions ending at each of the 4 positions of quad Q (backward window),
// a = 4f + p >= 4.  Window of position 4Q+j starts at 4(Q-f) + j - p.
template <typename HT>
__device__ __forceinline__ Quad<HT> window_min_back(const HT *h, int Q, int f, int p) {
    const Quad<HT> own = ldq(h, Q), v = ldq(h, Q - f);
    Quad<HT> u = v;
    if (p) u = ldq(h, Q - f - 1);
    HT mid = (HT) ~(HT) 0;
    for (int q = Q - f + 1; q < Q; ++q) {
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
    const Quad<HT> own = ldq(L, Q), v = ldq(L, Q + f);
    Quad<HT> u = v;
    if (p) u = ldq(L, Q + f + 1);
    HT mid = 0;
    for (int q = Q + 1; q < Q + f; ++q) {
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

// FUSED = false: every minimizer of the wavefront goes to its slots of (mkmer, ents).
// FUSED = true : every minimizer is looked up in the index at once; only those with records go
//                to the slots of ents; the wavefront's hit count is kept.
// WC > 0 fixes w at compile time (window shapes unroll); WC = 0 takes w from the argument.
// Wavefront g owns output positions [g * 4 * out_q, (g + 1) * 4 * out_q), out_q = 64 - 2 * edge.
template <typename HT, bool FUSED, int WC>
__global__ __launch_bounds__(256) void sketch_wave_kernel(
    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ wread, const uint32_t *__restrict__ woff,
    const uint32_t *__restrict__ rlen, const uint32_t *__restrict__ rflag, uint32_t n_words, uint32_t n_tiles,
    uint32_t w_arg, uint32_t k, const PmapSlot *__restrict__ slots, uint32_t log2cap, uint64_t *__restrict__ mkmer,
    uint4 *__restrict__ ents, uint32_t *__restrict__ tile_nmini, uint32_t *__restrict__ tile_nent,
    uint32_t *__restrict__ tile_nhits) {
    const uint32_t w = WC ? (uint32_t) WC : w_arg;
    __shared__ uint32_t sW[4][SK_NW], sStart[4][SK_NW], sLen[4][SK_NW], sRead[4][SK_NW];
    __shared__ __align__(16) HT sH[4][WAVE_POS];
    __shared__ __align__(16) uint32_t sJ[4][WAVE_POS];
    __shared__ __align__(16) HT sL[4][WAVE_POS + 8];

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x * 4 + wv;
    if (g >= n_tiles) return;                       // whole wavefront leaves; nothing below uses a block barrier
    uint32_t *W = sW[wv], *mstart = sStart[wv], *mlen = sLen[wv], *wr = sRead[wv];
    HT *h = sH[wv], *Lv = sL[wv];
    uint32_t *jv = sJ[wv];
    const HT INVALID = (HT) ~(HT) 0;

    const int a = (int) w - 1, f = a >> 2, p = a & 3;
    const int edge = f + (p ? 1 : 0);
    const int out_q = WAVE_Q - 2 * edge;
    const int64_t P0 = (int64_t) g * (4 * out_q) - 4 * edge;   // global position of quad 0, a multiple of 4
    const int64_t wbase = P0 >> 4;                              // floor: P0 is negative for g = 0
    if (lane < SK_NW) {
        const int64_t wi = wbase + lane;
        uint32_t word = 0, ms = 0, ml = 0, ri = 0;
        if (wi >= 0 && wi < (int64_t) n_words) {
            word = packed[wi];
            ri = wread[wi];
            ms = woff[ri] * 16;
            ml = rflag[ri] ? 0 : rlen[ri];
            if (ml + 1 < w + k) ml = 0;             // too short to sketch (src/seq.cpp:92-94)
        }
        W[lane] = word;
        mstart[lane] = ms;
        mlen[lane] = ml;
        wr[lane] = ri;
    }
    wave_lds_sync();

    // ---- canonical hashes of the lane's quad (src/seq.cpp:41-47, 109-111)
    const HT mask = (HT) (((uint64_t) 1 << (2 * k)) - 1);
    const int64_t pq = P0 + 4 * lane;               // global position of the quad's first base
    const int wi = (int) ((pq >> 4) - wbase);
    const uint32_t len = mlen[wi];
    const uint32_t rd = wr[wi];
    Quad<HT> own;
    own.v[0] = own.v[1] = own.v[2] = own.v[3] = INVALID;
    uint32_t jj[4] = {SK_NOKMER, SK_NOKMER, SK_NOKMER, SK_NOKMER};
    if (len != 0) {
        const uint32_t j0 = (uint32_t) pq - mstart[wi];
        const uint32_t sh0 = (uint32_t) (pq & 15) * 2;          // 0, 8, 16 or 24
        const uint32_t w0 = W[wi], w1 = W[wi + 1], w2 = W[wi + 2];
        // 64 stream bits starting at the quad's first base
        const uint32_t x0 = sh0 ? (w0 << sh0) | (w1 >> (32 - sh0)) : w0;
        const uint32_t x1 = sh0 ? (w1 << sh0) | (w2 >> (32 - sh0)) : w1;
        uint32_t x2 = 0;                                        // the next 32 bits, only needed for k > 15
        if (sizeof(HT) == 8) {
            const uint32_t w3 = W[wi + 3];
            x2 = sh0 ? (w2 << sh0) | (w3 >> (32 - sh0)) : w2;
        }
        HT rc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            HT fwd;
            if (sizeof(HT) == 4) {
                // k <= 15: 2k bits at bit offset 2j <= 6 of x0:x1
                const uint32_t top = j ? (x0 << (2 * j)) | (x1 >> (32 - 2 * j)) : x0;
                fwd = (HT) (top >> (32 - 2 * k));
            } else {
                const uint64_t x = ((uint64_t) x0 << 32) | x1;
                uint64_t z = x << (2 * j);
                if (j) z |= (uint64_t) (x2 >> (32 - 2 * j));
                fwd = (HT) (z >> (64 - 2 * k));
            }
            // reverse complement: from scratch for the first k-mer, then one base at a time (src/seq.cpp:44)
            if (j == 0) rc = (HT) revcomp_dev(fwd, k);
            else rc = (HT) ((rc >> 2) | ((HT) (3u ^ ((uint32_t) fwd & 3u)) << (2 * k - 2)));
            const HT hf = hash_any(fwd, mask), hr = hash_any(rc, mask);
            const bool ok = j0 + j + k <= len;
            own.v[j] = ok ? tmin(hf, hr) : INVALID;
            jj[j] = ok ? ((j0 + j) | ((hf <= hr ? 1u : 0u) << 31)) : SK_NOKMER;
        }
    }
    stq(h, lane, own);
    *(uint4 *) (jv + 4 * lane) = make_uint4(jj[0], jj[1], jj[2], jj[3]);
    if (lane < 2) stq(Lv, WAVE_Q + lane, Quad<HT>{{0, 0, 0, 0}});   // the forward window of the last quads may read 2 quads past the end
    wave_lds_sync();

    // ---- Lv[t] = minimum of the full window ending at k-mer t, 0 if that window is not inside one read
    {
        Quad<HT> m;
        m.v[0] = m.v[1] = m.v[2] = m.v[3] = 0;
        if (lane >= edge) {
            if (a >= 4) {
                m = window_min_back<HT>(h, lane, f, p);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    HT x = own.v[j];
                    for (int d = 1; d <= a; ++d) x = tmin(x, h[4 * lane + j - d]);
                    m.v[j] = x;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (jj[j] == SK_NOKMER || (jj[j] & 0x7FFFFFFFu) < (uint32_t) a) m.v[j] = 0;
        }
        stq(Lv, lane, m);
    }
    wave_lds_sync();

    // ---- k-mer i is a minimizer iff it equals the largest of the minima of the windows containing it
    bool fl[4] = {false, false, false, false};
    if (lane >= edge && lane < WAVE_Q - edge) {
        Quad<HT> mx;
        if (a >= 4) {
            mx = window_max_fwd<HT>(Lv, lane, f, p);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                HT x = Lv[4 * lane + j];
                for (int d = 1; d <= a; ++d) x = tmax(x, Lv[4 * lane + j + d]);
                mx.v[j] = x;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) fl[j] = jj[j] != SK_NOKMER && mx.v[j] == own.v[j];
    }
    wave_lds_sync();                                // every lane has read its quads: the arrays can be reused

    // ---- compaction in position order with ballots; Lv / jv / h become the staging lists
    HT *c_h = Lv;
    uint32_t *c_ps = jv, *c_rd = (uint32_t *) h;
    uint32_t n = 0, o = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long b = __ballot(fl[j]);
        // flags of lower lanes are earlier positions whatever their j
        o += __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0));
        n += __popcll(b);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (fl[j]) {
            c_h[o] = own.v[j];
            c_ps[o] = jj[j];
            c_rd[o] = rd;
            ++o;
        }
    wave_lds_sync();

    const uint64_t slot0 = (uint64_t) g * WAVE_POS;
    if (!FUSED) {
        for (uint32_t e = lane; e < n; e += 64) {
            mkmer[slot0 + e] = (uint64_t) c_h[e];
            ents[slot0 + e] = make_uint4(c_rd[e], c_ps[e], 0, 0);
        }
        if (lane == 0) {
            tile_nmini[g] = n;
            tile_nent[g] = n;
        }
    } else {
        uint32_t nent = 0, hits = 0;
        for (uint32_t base = 0; base < n; base += 64) {     // n is wave-uniform
            const uint32_t e = base + lane;
            uint32_t cnt = 0, off = 0;
            if (e < n) table_lookup(slots, log2cap, (uint64_t) c_h[e], &cnt, &off);
            const unsigned long long b = __ballot(cnt != 0);
            if (cnt) {
                const uint64_t s = slot0 + nent +
                                   __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0));
                ents[s] = make_uint4(c_rd[e], c_ps[e], off, cnt);
            }
            nent += __popcll(b);
            hits += cnt;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) hits += __shfl_xor(hits, d);
        if (lane == 0) {
            tile_nmini[g] = n;
            tile_nent[g] = nent;
            tile_nhits[g] = hits;
        }
    }
}

#endif
