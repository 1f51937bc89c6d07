// Sort, cluster cut and size threshold in one pass over the hits (gfx950).  Included by pmap_kernels.hip.h only.
//
//   MinimizerHits::sort      src/minihits.cpp:42-53, order src/minihit.cpp:44-66
//   define_clusters          src/utils.cpp:189-249
//
// Reads never share a cluster (src/utils.cpp:206), and the hits of a read are one contiguous stretch [s, e) of
// the emitted hits (hbounds[]), before and after the sort.  The hit array is cut into chunks of TAIL_C hits; one
// wavefront per chunk takes the reads whose first hit lies in its chunk:
//   * the reads of at most TAIL_C hits among them - they are consecutive, and together fewer than 2 TAIL_C hits -
//     as ONE segmented sort: words {read's ordinal in the chunk, key, place} through a bitonic network held in
//     registers (4 words per lane);
//   * the one read with more hits that may start in the chunk (it is the last to start there) by itself, words
//     {key, place}, 4 to 16 words per lane; reads of 1025 to 2048 hits go on a list for tail_big_kernel (32 words
//     per lane: a kernel of its own because of its registers), longer ones are left to the long-segment kernels
//     (long_presort_kernel, long_rank_kernel) and tail_long_kernel.
// With the sorted words still in its registers the wavefront writes every hit to its place, compares each word
// with the one before it (previous register, or last register of the previous lane through a DPP shift) to flag
// the starts of clusters, measures each cluster by the distance to the next flag, applies the size threshold of
// the cluster's prg, and writes the rows of the clusters it keeps.  No workgroup waits for another one:
// what is global - the number of a cluster among all clusters and the place of a kept cluster's row among all
// rows - is left to rows_gather_kernel, which sums the per-chunk counts (by groups of TAIL_GROUP chunks, the
// group totals added up with atomics that return nothing) and moves the rows from the slots the wavefronts
// chose for them: a kept cluster has at least T = (smallest threshold of any prg) + 1 hits, so the j-th kept
// cluster of a chunk whose first own hit is a fits slot a / T + j without meeting the rows of any other chunk.
#ifndef PMAP_TAIL_HIP_H
#define PMAP_TAIL_HIP_H

static const uint32_t TAIL_C = 128;             // hits per chunk
static const uint32_t TAIL_GROUP = 256;         // chunks per group of rows_gather_kernel
static const uint32_t TAIL_WAVE_MAX = 1024;     // largest read tail_chunk_kernel sorts itself
static const uint32_t TAIL_BIG_MAX = 2048;      // largest read tail_big_kernel sorts (64 lanes x 32 words)
static const uint32_t TAIL_MAX_KEY_BITS = 21;   // key and an 11-bit place share a word

struct TailArgs {
    Counters *dc;
    uint32_t hit_cap;
    uint32_t n_reads;           // entries of each half of hbounds
    uint32_t key_bits;          // bits of the largest key, prg << 1 | !forward
    uint32_t ord_bits;          // bits left for a read's ordinal in the words of a packed item
    uint32_t do_cluster;        // 0: sort only
    int max_diff;
    uint32_t min_thr;           // smallest threshold of any prg
    uint32_t slot_div;          // T = min_thr + 1
    uint32_t rows_tmp_cap;      // rows that rows_tmp holds
    uint32_t sort_long;         // reads with more hits belong to the long-segment kernels (>= TAIL_C, <= TAIL_BIG_MAX)
    const uint32_t *h_read, *h_key, *h_pos, *h_rank, *h_rec, *hbounds, *thr;
    uint32_t *s_read, *s_key, *s_pos, *s_rank, *s_rec;
    uint4 *chunk_cnt;               // per chunk: clusters, kept clusters, first row slot, 0 of the reads that start in it
    uint4 *chunk_cont;              // the same of the long read that runs through the chunk (tail_long_kernel)
    unsigned long long *cg_sum;     // per group of chunks: clusters, kept clusters (zero before the run)
    uint32_t *rows_tmp;
    uint32_t *big_list, *long_list; // chunks in which a read for tail_big_kernel / tail_long_kernel starts
};

template <int E>
struct TailWord {
    static const uint32_t PB = 6 + (E == 4 ? 2 : E == 8 ? 3 : E == 16 ? 4 : 5);   // bits of a place
    static const uint32_t PM = (1u << PB) - 1;
    static_assert(E == 4 || E == 8 || E == 16 || E == 32, "whole quads, a power of two");
};

__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) x); }

// Word j (dynamic, per lane) of the lane's E words.
template <int E>
__device__ __forceinline__ uint32_t tail_pick(const uint32_t (&v)[E], const uint32_t j) {
    // (a chain of selects; the empty asm keeps the compiler from turning it into an array in scratch memory)
    uint32_t w = v[0];
#pragma unroll
    for (int jj = 1; jj < E; ++jj) {
        w = j == (uint32_t) jj ? v[jj] : w;
        asm volatile("" : "+v"(w));
    }
    return w;
}

// The words v[] of one item - m of them valid, the others all-ones - are sorted, every hit is written to its
// place dst + (its rank in the item), clusters are cut, counted (ncl, nkept: running counts of the chunk) and the
// kept ones get their rows (slot_base: the chunk's first row slot, chosen with its first kept cluster).  A word is
// {group, place}: the hit is h_*[src + place]; hits of one group belong to one read, prg and strand.  PACKED: the
// group's low key_bits are the key and the read comes from h_read; otherwise the group is the key and the read
// is r_single.
// Everything a wavefront waits for is asked for in as few rounds as possible: the hits of two quads of words
// together, the thresholds of all clusters that might be kept (one per lane and round), the fields of the rows.
template <int E, bool PACKED>
__device__ __forceinline__ void tail_sorted(const TailArgs &A, const uint32_t *__restrict__ h_read,
                                            const uint32_t *__restrict__ h_pos, const uint32_t *__restrict__ h_rank,
                                            const uint32_t *__restrict__ h_rec, uint32_t *__restrict__ s_read,
                                            uint32_t *__restrict__ s_key, uint32_t *__restrict__ s_pos,
                                            uint32_t *__restrict__ s_rank, uint32_t *__restrict__ s_rec,
                                            const uint32_t lane, const uint32_t src, const uint32_t dst, const uint32_t m,
                                            const uint32_t r_single, uint32_t (&v)[E], uint32_t &slot_base, uint32_t &ncl,
                                            uint32_t &nkept) {
    constexpr uint32_t PB = TailWord<E>::PB, PM = TailWord<E>::PM;
    const uint32_t KM = PACKED ? (1u << A.key_bits) - 1u : 0xFFFFFFFFu;
    sortw_reg_merges<E, 2>(v);
    sortw_lane_merges<E, 2>(v, lane);
    const uint32_t i0 = lane * E;
    // the word before the lane's first one is the last word of the lane before
    constexpr int QG = (E == 8 || E == 32) ? 2 : 1;     // quads whose hits are fetched together (E = 16: registers)
    constexpr bool ONE_ROUND = E / 4 == QG;             // all of a lane's hits arrive together: its last position is among them
    uint32_t prev_pos = 0, prev_grp = 0;
    if (A.do_cluster) {
        if constexpr (!ONE_ROUND) {
            const uint32_t plast = i0 + E - 1 < m ? h_pos[src + (v[E - 1] & PM)] : 0u;
            prev_pos = lane_before(plast, 0u);
        }
        prev_grp = lane_before(v[E - 1], 0u) >> PB;
    }
    uint32_t fmask = 0;         // bit j: word j of the lane starts a cluster
#pragma unroll
    for (int q0 = 0; q0 < E / 4; q0 += QG) {
        uint32_t ps[QG][4], rk[QG][4], rc[QG][4], rd[QG][4];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const uint32_t i = i0 + 4 * (q0 + g);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t from = src + (v[4 * (q0 + g) + u] & PM);
                const bool in = i + u < m;
                ps[g][u] = in ? h_pos[from] : 0u;
                rk[g][u] = in ? h_rank[from] : 0u;
                rc[g][u] = in ? h_rec[from] : 0u;
                rd[g][u] = PACKED ? (in ? h_read[from] : 0u) : r_single;
            }
        }
        if constexpr (ONE_ROUND) {
            if (A.do_cluster) prev_pos = lane_before(ps[QG - 1][3], 0u);
        }
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const int q = q0 + g;
            const uint32_t i = i0 + 4 * q, d = dst + i;
            if (i + 4 <= m) {
                *(Quad4 *) (s_read + d) = Quad4{rd[g][0], rd[g][1], rd[g][2], rd[g][3]};
                *(Quad4 *) (s_key + d) = Quad4{(v[4 * q] >> PB) & KM, (v[4 * q + 1] >> PB) & KM, (v[4 * q + 2] >> PB) & KM,
                                               (v[4 * q + 3] >> PB) & KM};
                *(Quad4 *) (s_pos + d) = Quad4{ps[g][0], ps[g][1], ps[g][2], ps[g][3]};
                *(Quad4 *) (s_rank + d) = Quad4{rk[g][0], rk[g][1], rk[g][2], rk[g][3]};
                *(Quad4 *) (s_rec + d) = Quad4{rc[g][0], rc[g][1], rc[g][2], rc[g][3]};
            } else if (i < m) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u < m) {
                        s_read[d + u] = rd[g][u];
                        s_key[d + u] = (v[4 * q + u] >> PB) & KM;
                        s_pos[d + u] = ps[g][u];
                        s_rank[d + u] = rk[g][u];
                        s_rec[d + u] = rc[g][u];
                    }
            }
            if (A.do_cluster) {
                // src/utils.cpp:206-209: a new cluster where read, prg or strand change or the positions lie apart
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t t = i + u, grp = v[4 * q + u] >> PB;
                    const int dd = (int) ps[g][u] - (int) prev_pos;
                    const bool f = t < m && (t == 0 || grp != prev_grp || (dd < 0 ? -dd : dd) > A.max_diff);
                    fmask |= (f ? 1u : 0u) << (4 * q + u);
                    prev_grp = grp;
                    prev_pos = ps[g][u];
                }
            }
        }
    }
    if (!A.do_cluster) return;
    // a cluster ends where the next one starts: the next flag in the lane, else the first flag of the next lane
    // that has one, else the end of the item
    const uint32_t first_flag = fmask ? i0 + (uint32_t) __builtin_ctz(fmask) : 0u;
    const unsigned long long anyf = ballot64(fmask != 0);
    const unsigned long long higher = lane == 63 ? 0ull : anyf & ~((2ull << lane) - 1ull);
    const uint32_t next_lane = higher ? (uint32_t) __builtin_ctzll(higher) : lane;
    const uint32_t ff_next = (uint32_t) __shfl((int) first_flag, (int) next_lane);
    const uint32_t beyond = higher ? ff_next : m;
    // the word after the cluster that starts at word j of the lane
#define TAIL_NEXT(j) ((((fmask >> (j)) >> 1) != 0) ? i0 + (j) + 1 + (uint32_t) __builtin_ctz((fmask >> (j)) >> 1) : beyond)
    // src/utils.cpp:211-220.  Most clusters are too small for any prg and never look their threshold up; the
    // others do, one per lane at a time
    uint32_t cand = 0;
#pragma unroll
    for (int j = 0; j < E; ++j)
        if ((fmask >> j & 1u) && TAIL_NEXT(j) - (i0 + j) > A.min_thr) cand |= 1u << j;
    uint32_t kmask = 0;         // bit j: the cluster that starts at word j is kept
    while (ballot64(cand != 0)) {                       // wave-uniform
        if (cand) {
            const uint32_t j = (uint32_t) __builtin_ctz(cand);
            const uint32_t key = (tail_pick<E>(v, j) >> PB) & KM;
            if (TAIL_NEXT(j) - (i0 + j) > A.thr[key >> 1]) kmask |= 1u << j;
            cand &= cand - 1;
        }
    }
    const uint32_t both = (uint32_t) __popc(kmask) << 16 | (uint32_t) __popc(fmask);      // at most 2048 each
    const uint32_t incl = wave_incl_scan(both);
    const uint32_t tot = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63), excl = incl - both;
    const uint32_t cl0 = ncl, k0 = nkept;
    ncl += tot & 0xFFFFu;
    nkept += tot >> 16;
    if ((tot >> 16) == 0) return;
    if (k0 == 0) {
        // the chunk's first kept cluster chooses the chunk's slots: cluster j of the chunk starts at least j T
        // hits behind it, so slot (its first hit) / T + j lies before the slots any later chunk chooses
        const uint32_t fl = (uint32_t) __builtin_ctzll(ballot64(kmask != 0));
        const uint32_t first_kept = (uint32_t) __builtin_amdgcn_readlane((int) (i0 + (kmask ? (uint32_t) __builtin_ctz(kmask) : 0u)), (int) fl);
        slot_base = (dst + first_kept) / A.slot_div;
    }
    // rows of the kept clusters, by the lanes that hold their first words, one per lane at a time.  A cluster's
    // last word sits in another lane and register: fetched with one cross-lane read per register.
    uint32_t todo = kmask;
    while (ballot64(todo != 0)) {                       // wave-uniform
        const bool mine = todo != 0;
        const uint32_t j = mine ? (uint32_t) __builtin_ctz(todo) : 0u;
        const uint32_t next = TAIL_NEXT(j);
        const uint32_t t2 = mine ? next - 1 : i0;
        const uint32_t l2 = t2 / E, j2 = t2 % E;
        uint32_t w2 = 0;
#pragma unroll
        for (int jj = 0; jj < E; ++jj) {
            const uint32_t o = (uint32_t) __shfl((int) v[jj], (int) l2);
            if (j2 == (uint32_t) jj) w2 = o;
        }
        if (mine) {
            const uint32_t wj = tail_pick<E>(v, j);
            const uint32_t sf = src + (wj & PM), sl = src + (w2 & PM);
            const uint32_t slot = slot_base + k0 + (excl >> 16) + (uint32_t) __popc(kmask & ((1u << j) - 1u));
            if (slot < A.rows_tmp_cap) {
                uint32_t *row = A.rows_tmp + (uint64_t) slot * 8;
                *(uint4 *) row = make_uint4(dst + i0 + j, next - (i0 + j), PACKED ? h_read[sf] : r_single, (wj >> PB) & KM);
                *(uint4 *) (row + 4) = make_uint4(h_pos[sf], h_pos[sl], h_rank[sf],
                                                  cl0 + (excl & 0xFFFFu) + (uint32_t) __popc(fmask & ((1u << j) - 1u)));
            }
            todo &= todo - 1;
        }
    }
#undef TAIL_NEXT
}

#define TAIL_SORTED_ARGS A, A.h_read, A.h_pos, A.h_rank, A.h_rec, A.s_read, A.s_key, A.s_pos, A.s_rank, A.s_rec

// one read of m <= 64 E hits, [a, a + m)
template <int E>
__device__ __forceinline__ void tail_single(const TailArgs &A, const uint32_t lane, const uint32_t a, const uint32_t m,
                                            const uint32_t r, uint32_t &slot_base, uint32_t &ncl, uint32_t &nkept) {
    constexpr uint32_t PB = TailWord<E>::PB;
    uint32_t v[E];
    const uint32_t i0 = lane * E;
#pragma unroll
    for (int q = 0; q < E / 4; ++q) {
        const uint32_t i = i0 + 4 * q;
        uint32_t kk[4] = {0, 0, 0, 0};
        if (i + 4 <= m) {
            const Quad4 t = *(const Quad4 *) (A.h_key + a + i);
            kk[0] = t.x; kk[1] = t.y; kk[2] = t.z; kk[3] = t.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u < m) kk[u] = A.h_key[a + i + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) v[4 * q + u] = i + u < m ? (kk[u] << PB | (i + u)) : 0xFFFFFFFFu;
    }
    tail_sorted<E, false>(TAIL_SORTED_ARGS, lane, a, a, m, r, v, slot_base, ncl, nkept);
}

// the consecutive whole reads [a, a + m), m <= 256, as one segmented sort
__device__ __forceinline__ void tail_packed(const TailArgs &A, const uint32_t lane, const uint32_t a, const uint32_t m,
                                            uint32_t &slot_base, uint32_t &ncl, uint32_t &nkept) {
    constexpr int E = 4;
    constexpr uint32_t PB = TailWord<E>::PB;
    const uint32_t i0 = lane * E;
    uint32_t key[E] = {0, 0, 0, 0}, rd[E] = {0, 0, 0, 0};
    if (i0 + 4 <= m) {
        const Quad4 t = *(const Quad4 *) (A.h_key + a + i0), r = *(const Quad4 *) (A.h_read + a + i0);
        key[0] = t.x; key[1] = t.y; key[2] = t.z; key[3] = t.w;
        rd[0] = r.x; rd[1] = r.y; rd[2] = r.z; rd[3] = r.w;
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u < m) {
                key[u] = A.h_key[a + i0 + u];
                rd[u] = A.h_read[a + i0 + u];
            }
    }
    // ordinal of a hit's read among the reads of the item
    const uint32_t rd_before = lane_before(rd[E - 1], 0u);
    uint32_t ord[E], cnt = 0;
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const uint32_t p = u ? rd[u - 1] : rd_before;
        cnt += (i0 + u < m && i0 + u > 0 && rd[u] != p) ? 1u : 0u;
        ord[u] = cnt;
    }
    const uint32_t incl = wave_incl_scan(cnt);
    const uint32_t n_ord = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63) + 1;
#pragma unroll
    for (int u = 0; u < E; ++u) ord[u] += incl - cnt;
    const uint32_t R = 1u << A.ord_bits, shift = A.key_bits + PB;
    uint32_t v[E];
    if (n_ord <= R) {
#pragma unroll
        for (int u = 0; u < E; ++u) v[u] = i0 + u < m ? (ord[u] << shift | key[u] << PB | (i0 + u)) : 0xFFFFFFFFu;
        tail_sorted<E, true>(TAIL_SORTED_ARGS, lane, a, a, m, 0u, v, slot_base, ncl, nkept);
        return;
    }
    // more reads than a word has ordinals for: R reads at a time (hits of other reads sort behind, as words of all-ones)
    uint32_t done = 0;              // hits of the reads already sorted
    for (uint32_t o0 = 0; o0 < n_ord; o0 += R) {        // wave-uniform
        uint32_t inside = 0;
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const bool in = i0 + u < m && ord[u] - o0 < R;
            inside += in ? 1u : 0u;
            v[u] = in ? ((ord[u] - o0) << shift | key[u] << PB | (i0 + u)) : 0xFFFFFFFFu;
        }
        const uint32_t m_it = wave_sum(inside);
        tail_sorted<E, true>(TAIL_SORTED_ARGS, lane, a, a + done, m_it, 0u, v, slot_base, ncl, nkept);
        done += m_it;
    }
}

// what a chunk's wavefront has to know about the chunk: its own hits start at a; the last read that starts in it is rl = [sl, el)
struct TailChunk {
    bool any;           // some read starts in the chunk
    uint32_t a, rl, sl, el;
};
// ... found with two rounds of loads that depend on each other: the reads of the chunk's first and last hit,
// then their bounds.  Kept apart so that a wavefront can ask for them while it works on an earlier chunk.
struct TailProbe1 {
    uint32_t r0, rl;
};
struct TailProbe2 {
    uint32_t s0, e0, sl, el;
};

__device__ __forceinline__ TailProbe1 tail_probe1(const TailArgs &A, const uint32_t c, const uint32_t n_hits) {
    const uint32_t lo = c * TAIL_C, hi = min(lo + TAIL_C, n_hits);
    TailProbe1 p;
    p.r0 = A.h_read[lo];
    p.rl = A.h_read[hi - 1];
    return p;
}

__device__ __forceinline__ TailProbe2 tail_probe2(const TailArgs &A, const TailProbe1 &p) {
    const uint32_t NR = A.n_reads;
    const uint32_t r0 = min(rfl(p.r0), NR - 1), rl = min(rfl(p.rl), NR - 1);      // (read numbers are below NR: never index out of range)
    TailProbe2 q;
    q.s0 = A.hbounds[r0];
    q.e0 = ~A.hbounds[NR + r0];
    q.sl = A.hbounds[rl];
    q.el = ~A.hbounds[NR + rl];
    return q;
}

__device__ __forceinline__ TailChunk tail_chunk_from(const TailArgs &A, const uint32_t c, const uint32_t n_hits,
                                                     const TailProbe1 &p, const TailProbe2 &q) {
    TailChunk k;
    const uint32_t lo = c * TAIL_C, hi = min(lo + TAIL_C, n_hits);
    k.rl = min(rfl(p.rl), A.n_reads - 1);
    k.sl = rfl(q.sl);
    k.el = rfl(q.el);
    k.a = rfl(q.s0) == lo ? lo : rfl(q.e0);
    // (the last conditions hold whenever the bounds are consistent with the hits)
    k.any = k.a < hi && k.sl >= lo && k.el > k.sl && k.el <= n_hits;
    return k;
}

__device__ __forceinline__ TailChunk tail_chunk_of(const TailArgs &A, const uint32_t c, const uint32_t n_hits) {
    const TailProbe1 p = tail_probe1(A, c, n_hits);
    const TailProbe2 q = tail_probe2(A, p);
    return tail_chunk_from(A, c, n_hits, p, q);
}

__device__ __forceinline__ void tail_add_group(const TailArgs &A, const uint32_t c, const uint32_t dcl, const uint32_t dkept) {
    unsigned long long *g = A.cg_sum + 2 * (size_t) (c / TAIL_GROUP);
    if (dcl) (void) __hip_atomic_fetch_add(g, (unsigned long long) dcl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (dkept) (void) __hip_atomic_fetch_add(g + 1, (unsigned long long) dkept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The chunks of TAIL_C hits, one wavefront per chunk at a time, any grid: wavefront b takes chunks b, b + grid,
// ...  What it has to know about a chunk takes two dependent rounds of loads; it asks for them one and two
// chunks ahead, so that they arrive while it sorts.
__global__ __launch_bounds__(64) void tail_chunk_kernel(const TailArgs A) {
    const uint32_t lane = threadIdx.x, G = gridDim.x;
    const uint32_t n_hits = clamp_count(&A.dc->n_hits, A.hit_cap);
    const uint32_t n_chunks = (uint32_t) (((uint64_t) n_hits + TAIL_C - 1) / TAIL_C);
    uint32_t c = blockIdx.x;
    if (c >= n_chunks) return;
    TailProbe1 p_cur = tail_probe1(A, c, n_hits), p_next = p_cur;
    TailProbe2 q_cur = tail_probe2(A, p_cur), q_next = q_cur;
    if (c + G < n_chunks) p_next = tail_probe1(A, c + G, n_hits);
    for (; c < n_chunks; c += G) {                  // wave-uniform
        const TailChunk k = tail_chunk_from(A, c, n_hits, p_cur, q_cur);
        TailProbe1 p_after = p_next;
        if (c + G < n_chunks) q_next = tail_probe2(A, p_next);
        if ((uint64_t) c + 2 * (uint64_t) G < n_chunks) p_after = tail_probe1(A, c + 2 * G, n_hits);
        uint32_t ncl = 0, nkept = 0, base = 0;
        if (k.any) {
            const uint32_t nl = k.el - k.sl;
            const bool big = nl > TAIL_C;
            const uint32_t b = big ? k.sl : k.el;
            if (b > k.a && b - k.a <= 256u) tail_packed(A, lane, k.a, b - k.a, base, ncl, nkept);
            if (big) {
                if (nl > A.sort_long) {
                    if (lane == 0) {
                        A.dc->has_long = 1u;
                        const uint32_t at = atomicAdd(&A.dc->n_long_list, 1u);
                        if (at < A.n_reads) A.long_list[at] = c;
                    }
                } else if (nl <= 256u) tail_single<4>(A, lane, k.sl, nl, k.rl, base, ncl, nkept);
                else if (nl <= 512u) tail_single<8>(A, lane, k.sl, nl, k.rl, base, ncl, nkept);
                else if (nl <= TAIL_WAVE_MAX) tail_single<16>(A, lane, k.sl, nl, k.rl, base, ncl, nkept);
                else if (lane == 0) {
                    const uint32_t at = atomicAdd(&A.dc->n_wave, 1u);
                    if (at < A.n_reads) A.big_list[at] = c;
                }
            }
        }
        if (A.do_cluster && lane == 0) {
            A.chunk_cnt[c] = make_uint4(ncl, nkept, base, 0u);
            A.chunk_cont[c] = make_uint4(0u, 0u, 0u, 0u);       // (tail_long_kernel fills it where a long read runs through the chunk)
            tail_add_group(A, c, ncl, nkept);
        }
        p_cur = p_next;
        q_cur = q_next;
        p_next = p_after;
    }
}

// The reads of 1025 to 2048 hits (big_list, written by tail_chunk_kernel): one wavefront per read at a time, any grid.
__global__ __launch_bounds__(64) void tail_big_kernel(const TailArgs A) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n_hits = clamp_count(&A.dc->n_hits, A.hit_cap);
    const uint32_t n_list = min(A.dc->n_wave, A.n_reads);
    for (uint32_t t = blockIdx.x; t < n_list; t += gridDim.x) {
        const uint32_t c = rfl(A.big_list[t]);
        if ((uint64_t) c * TAIL_C >= n_hits) continue;
        const TailChunk k = tail_chunk_of(A, c, n_hits);
        if (!k.any) continue;
        const uint32_t nl = k.el - k.sl;
        if (nl <= TAIL_WAVE_MAX || nl > TAIL_BIG_MAX || nl > A.sort_long) continue;
        uint32_t ncl = 0, nkept = 0, base = 0;
        if (A.do_cluster) {
            const uint4 cc = A.chunk_cnt[c];
            ncl = rfl(cc.x); nkept = rfl(cc.y); base = rfl(cc.z);
        }
        const uint32_t cl0 = ncl, k0 = nkept;
        tail_single<32>(A, lane, k.sl, nl, k.rl, base, ncl, nkept);
        if (A.do_cluster && lane == 0) {
            A.chunk_cnt[c] = make_uint4(ncl, nkept, base, 0u);
            tail_add_group(A, c, ncl - cl0, nkept - k0);
        }
    }
}

// The reads the long-segment kernels have sorted (long_list): clusters cut from the sorted hits in memory, one
// wavefront per chunk of TAIL_C hits the read runs through (any grid), two hits per lane.  The wavefront takes
// the clusters that START in its chunk: a cluster's size is the distance to the next start, looked for beyond
// the chunk where the chunk holds none.  The chunk the read starts in counts them with the reads that start
// before it there (chunk_cnt); the chunks it runs through have a count of their own for it (chunk_cont),
// whose rows come before those of the reads that start in that chunk.
__global__ __launch_bounds__(64) void tail_long_kernel(const TailArgs A) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n_hits = clamp_count(&A.dc->n_hits, A.hit_cap);
    const uint32_t n_list = min(A.dc->n_long_list, A.n_reads);
    if (!A.do_cluster) return;
    // teams of TEAM wavefronts take the reads in turn, the members of a team the chunks of a read
    constexpr uint32_t TEAM = 8;
    const uint32_t team = blockIdx.x / TEAM, member = blockIdx.x % TEAM, n_teams = max(gridDim.x / TEAM, 1u);
    if (team >= n_teams) return;
    for (uint32_t t = team; t < n_list; t += n_teams) {
        const uint32_t c0 = rfl(A.long_list[t]);
        if ((uint64_t) c0 * TAIL_C >= n_hits) continue;
        const TailChunk k = tail_chunk_of(A, c0, n_hits);
        if (!k.any || k.el - k.sl <= A.sort_long) continue;
        const uint32_t sl = k.sl, el = k.el, c1 = (el - 1) / TAIL_C;
        for (uint32_t c = c0 + member; c <= c1; c += TEAM) {                    // wave-uniform
            const uint32_t p0 = max(sl, c * TAIL_C), p1 = min(el, (c + 1) * TAIL_C);
            // hit i starts a cluster (src/utils.cpp:206-209)
            bool f[2];
            uint32_t key[2] = {0, 0}, pos[2] = {0, 0};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t i = p0 + 64 * u + lane;
                f[u] = false;
                if (i < p1) {
                    key[u] = A.s_key[i];
                    pos[u] = A.s_pos[i];
                    if (i == sl) f[u] = true;
                    else {
                        const int dd = (int) pos[u] - (int) A.s_pos[i - 1];
                        f[u] = key[u] != A.s_key[i - 1] || (dd < 0 ? -dd : dd) > A.max_diff;
                    }
                }
            }
            const unsigned long long fm[2] = {ballot64(f[0]), ballot64(f[1])};
            if ((fm[0] | fm[1]) == 0) {         // no cluster starts here
                if (c != c0 && lane == 0) A.chunk_cont[c] = make_uint4(0u, 0u, 0u, 0u);
                continue;
            }
            // the first start behind the chunk
            uint32_t beyond = el;
            for (uint32_t q = p1; q < el; q += 64) {                           // wave-uniform
                const uint32_t i = q + lane;
                bool g = false;
                if (i < el) {
                    const int dd = (int) A.s_pos[i] - (int) A.s_pos[i - 1];
                    g = A.s_key[i] != A.s_key[i - 1] || (dd < 0 ? -dd : dd) > A.max_diff;
                }
                const unsigned long long gm = ballot64(g);
                if (gm) {
                    beyond = q + (uint32_t) __builtin_ctzll(gm);
                    break;
                }
            }
            uint32_t ncl = 0, nkept = 0, base = 0;
            if (c == c0) {
                const uint4 cc = A.chunk_cnt[c];
                ncl = rfl(cc.x); nkept = rfl(cc.y); base = rfl(cc.z);
            }
            const uint32_t cl0 = ncl, k0 = nkept;
            bool kept[2];
            uint32_t next[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t i = p0 + 64 * u + lane;
                // the next start: a higher lane of this half, the other half, beyond the chunk
                const unsigned long long up = lane == 63 ? 0ull : fm[u] & ~((2ull << lane) - 1ull);
                next[u] = up ? p0 + 64 * u + (uint32_t) __builtin_ctzll(up)
                             : (u == 0 && fm[1]) ? p0 + 64 + (uint32_t) __builtin_ctzll(fm[1]) : beyond;
                const uint32_t size = next[u] - i;
                kept[u] = f[u] && size > A.min_thr && size > A.thr[key[u] >> 1];       // src/utils.cpp:211-220
            }
            const unsigned long long km[2] = {ballot64(kept[0]), ballot64(kept[1])};
            const uint32_t nk = (uint32_t) (__popcll(km[0]) + __popcll(km[1]));
            if (nk && k0 == 0) {
                const uint32_t first_kept = km[0] ? p0 + (uint32_t) __builtin_ctzll(km[0]) : p0 + 64 + (uint32_t) __builtin_ctzll(km[1]);
                base = first_kept / A.slot_div;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (kept[u]) {
                    const uint32_t i = p0 + 64 * u + lane;
                    const uint32_t below_k = (u ? (uint32_t) __popcll(km[0]) : 0u) +
                                             __builtin_amdgcn_mbcnt_hi((uint32_t) (km[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) km[u], 0));
                    const uint32_t below_f = (u ? (uint32_t) __popcll(fm[0]) : 0u) +
                                             __builtin_amdgcn_mbcnt_hi((uint32_t) (fm[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) fm[u], 0));
                    const uint32_t slot = base + k0 + below_k;
                    if (slot < A.rows_tmp_cap) {
                        uint32_t *row = A.rows_tmp + (uint64_t) slot * 8;
                        *(uint4 *) row = make_uint4(i, next[u] - i, k.rl, key[u]);
                        *(uint4 *) (row + 4) = make_uint4(pos[u], A.s_pos[next[u] - 1], A.s_rank[i], cl0 + below_f);
                    }
                }
            ncl += (uint32_t) (__popcll(fm[0]) + __popcll(fm[1]));
            nkept += nk;
            if (lane == 0) {
                if (c == c0) A.chunk_cnt[c] = make_uint4(ncl, nkept, base, 0u);
                else A.chunk_cont[c] = make_uint4(ncl, nkept, base, 0u);
                tail_add_group(A, c, ncl - cl0, nkept - k0);
            }
        }
    }
}

// Rows of the kept clusters from the chunks' slots to their places: one wavefront per 64 chunks, one lane per
// chunk.  The clusters and kept clusters before a chunk: the totals of the groups of TAIL_GROUP chunks before
// its own (cg_sum, summed by brute force: a few hundred words), the counts of the chunks before it in its
// group, and a prefix sum over the lanes.  The last wavefront leaves the totals in the counters.
__global__ __launch_bounds__(64) void rows_gather_kernel(Counters *__restrict__ dc, uint32_t hit_cap,
                                                         const uint4 *__restrict__ chunk_cnt,
                                                         const uint4 *__restrict__ chunk_cont,
                                                         const unsigned long long *__restrict__ cg_sum,
                                                         const uint32_t *__restrict__ rows_tmp, uint32_t rows_tmp_cap,
                                                         uint32_t *__restrict__ k_rows, uint32_t row_cap) {
    static_assert(TAIL_GROUP % 64 == 0, "a wavefront stays inside one group");
    const uint32_t lane = threadIdx.x;
    const uint32_t n_hits = clamp_count(&dc->n_hits, hit_cap);
    const uint32_t n_chunks = (uint32_t) (((uint64_t) n_hits + TAIL_C - 1) / TAIL_C);
    const uint32_t w0 = blockIdx.x * 64;            // the wavefront's first chunk
    if (w0 >= n_chunks) return;                     // (no hits: the counters are zero already)
    const uint32_t g = w0 / TAIL_GROUP;
    unsigned long long pc = 0, pk = 0;
    for (uint32_t i = lane; i < g; i += 64) {
        pc += cg_sum[2 * (size_t) i];
        pk += cg_sum[2 * (size_t) i + 1];
    }
    for (uint32_t b = g * TAIL_GROUP; b < w0; b += 64) {        // wave-uniform, at most TAIL_GROUP / 64 - 1 rounds
        const uint4 x = chunk_cnt[b + lane], y = chunk_cont[b + lane];
        pc += x.x + y.x;
        pk += x.y + y.y;
    }
    pc = wave_sum64(pc);
    pk = wave_sum64(pk);
    const uint32_t c = w0 + lane;
    uint4 own = make_uint4(0, 0, 0, 0), cont = make_uint4(0, 0, 0, 0);
    if (c < n_chunks) {
        own = chunk_cnt[c];
        cont = chunk_cont[c];
    }
    uint32_t totc, totk;
    const uint32_t exc = wave_scan(own.x + cont.x, &totc), exk = wave_scan(own.y + cont.y, &totk);
    // the rows of the read that runs through the chunk, then those of the reads that start in it
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const uint4 cc = part ? own : cont;
        const unsigned long long to0 = pk + exk + (part ? cont.y : 0u);
        const uint32_t cl0 = (uint32_t) pc + exc + (part ? cont.x : 0u);
        for (uint32_t j = 0; j < cc.y; ++j) {
            const unsigned long long to = to0 + j;
            const uint32_t from = cc.z + j;
            if (to < row_cap && from < rows_tmp_cap) {
                const uint4 *s = (const uint4 *) (rows_tmp + (uint64_t) from * 8);
                uint4 lo = s[0], hi = s[1];
                hi.w += cl0;                        // the cluster's number among all clusters of the batch
                uint4 *d = (uint4 *) (k_rows + to * 8);
                d[0] = lo;
                d[1] = hi;
            }
        }
    }
    if ((uint64_t) w0 + 64 >= n_chunks && lane == 0) {
        dc->n_cl = pc + totc;
        dc->n_kept = pk + totk;
    }
}

#endif
