// Probe (unfused mode), hit emission and minimizer gather kernels of libpandora_amd (gfx950).
// Included by pmap_kernels.hip.h only.
//
// The sketch kernel leaves, per wavefront tile g, tile_cnt[g] = {minimizers, entries, hits, 0} and
// the entries themselves in the tile's slots of `ents` (slot stride = positions of a tile), 32 bytes each:
// {read index, position | strand << 31, first record's index, record count} {the first record}.
#ifndef PMAP_EMIT_HIP_H
#define PMAP_EMIT_HIP_H

// exclusive scan of one value per lane over the wavefront; *total = wavefront sum
__device__ __forceinline__ uint32_t wave_scan(uint32_t v, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(incl, d);
        if (lane >= (uint32_t) d) incl += t;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}

// 64-bit sum over the wavefront, valid in every lane
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// probe as its own kernel (unfused mode): one wavefront per tile looks up the tile's minimizers
// (add_read_hits' index->minhash.find, src/utils.cpp:175)
__global__ __launch_bounds__(256) void probe_wave_kernel(const uint64_t *__restrict__ mkmer, uint32_t n_tiles,
                                                         const PmapSlot *__restrict__ slots, uint32_t log2cap,
                                                         const uint32_t *__restrict__ filt, uint32_t fbits,
                                                         uint4 *__restrict__ ents, uint4 *__restrict__ tile_cnt,
                                                         uint32_t stride) {
    const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_tiles) return;
    const uint32_t n = tile_cnt[g].y;
    const uint64_t slot0 = (uint64_t) g * stride;
    uint32_t mine = 0;
    for (uint32_t e = lane; e < n; e += 64) {
        uint32_t cnt, off;
        uint4 first;
        index_lookup(slots, log2cap, filt, fbits, mkmer[slot0 + e], &cnt, &off, &first);
        uint32_t *ent = (uint32_t *) (ents + 2 * (slot0 + e));
        ent[2] = off;
        ent[3] = cnt;
        ents[2 * (slot0 + e) + 1] = first;
        mine += cnt;
    }
    mine = wave_sum(mine);
    if (lane == 0) ((uint32_t *) (tile_cnt + g))[2] = mine;
}

// group_sum[G] = {minimizers, entries, hits, marks} of the GROUP_TILES tiles of group G (marks: tile_cnt.w, set for
// tiles with more minimizers than the sketch kernel's list holds): one wavefront per group
__global__ __launch_bounds__(256) void group_reduce_kernel(const uint4 *__restrict__ tile_cnt, uint32_t n_tiles,
                                                           uint4 *__restrict__ group_sum) {
    const uint32_t lane = threadIdx.x & 63, G = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t t0 = G * GROUP_TILES;
    if (t0 >= n_tiles) return;
    uint32_t x = 0, y = 0, z = 0, v = 0;
#pragma unroll
    for (uint32_t i = 0; i < GROUP_TILES / 64; ++i) {
        const uint32_t t = t0 + i * 64 + lane;
        if (t < n_tiles) {
            const uint4 c = tile_cnt[t];
            x += c.x; y += c.y; z += c.z; v += c.w;
        }
    }
    x = wave_sum(x); y = wave_sum(y); z = wave_sum(z); v = wave_sum(v);
    if (lane == 0) group_sum[G] = make_uint4(x, y, z, v);
}

// Hit emission (MinimizerHit ctor, src/minihit.cpp:14-22).  A wavefront takes SCATTER_TILES tiles.
// Its first hit = the hit counts of all groups before its own (group_reduce_kernel; summed here by
// brute force: a few hundred L2-resident words) + the hit counts of the tiles of its own group
// before its first tile.  No block depends on another one, so the grid is as fine as the
// latency of a wavefront's dependent loads asks for.  The wavefront numbers its entries
// consecutively (tile by tile, so still in (read, position) order); lane e takes entry e, finds its
// tile by a search over the per-tile prefix sums held in the lanes, and writes the entry's hits at
// the running offset + the prefix sum of the record counts.
// hbounds[r] = first hit of read r, hbounds[n_reads + r] = ~(one past its last hit); both start as
// all-ones and are lowered with atomic minima where the read index changes between consecutive
// entries and at the first and last entry of every wavefront.
// The last wavefront of the grid also folds the batch totals into the counters.
static const uint32_t SCATTER_TILES = 16;
static_assert(GROUP_TILES % SCATTER_TILES == 0 && SCATTER_TILES <= 64, "a wavefront stays inside one group");

__global__ __launch_bounds__(256) void scatter_group_kernel(
    const uint4 *__restrict__ tile_cnt, uint32_t n_tiles, uint32_t stride, const uint4 *__restrict__ ents,
    const PmapRec *__restrict__ recs, uint32_t hit_cap, uint32_t emit, Counters *__restrict__ dc,
    const uint4 *__restrict__ group_sum, uint32_t n_reads,
    uint32_t *__restrict__ hbounds, uint32_t *__restrict__ h_read, uint32_t *__restrict__ h_key,
    uint32_t *__restrict__ h_pos, uint32_t *__restrict__ h_rank, uint32_t *__restrict__ h_rec) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t t0 = wave * SCATTER_TILES;
    if (t0 >= n_tiles) return;
    const uint32_t G = t0 / GROUP_TILES;
    // hits before the wavefront's first tile
    unsigned long long before64 = 0;
    // 8 independent loads per lane in flight at a time: the sum is latency-bound, not bandwidth-bound
    for (uint32_t i0 = 0; i0 < G; i0 += 64 * 8) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t i = i0 + u * 64 + lane;
            v[u] = i < G ? group_sum[i].z : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) before64 += v[u];
    }
    for (uint32_t t = G * GROUP_TILES + lane; t < t0; t += 64) before64 += tile_cnt[t].z;
    uint32_t run = wave_sum((uint32_t) before64);    // offsets are 32-bit: wraps only in a batch the host rejects
    const uint32_t t = t0 + lane;
    uint4 c = make_uint4(0, 0, 0, 0);
    if (lane < SCATTER_TILES && t < n_tiles) c = tile_cnt[t];
    uint32_t E;
    const uint32_t estart = wave_scan(c.y, &E);
    if (t0 + SCATTER_TILES >= n_tiles) {
        // last wavefront: batch totals, in 64 bits, for the kernels that follow and for the host
        const unsigned long long h = wave_sum64(before64 + c.z);
        unsigned long long m = 0, en = 0, mk = 0;
        for (uint32_t i = lane; i <= G; i += 64) {
            const uint4 gs = group_sum[i];
            m += gs.x;
            en += gs.y;
            mk += gs.w;
        }
        m = wave_sum64(m);
        en = wave_sum64(en);
        mk = wave_sum64(mk);
        if (lane == 0) {
            dc->n_hits = h;
            dc->n_mini = m;
            dc->n_ent = en;
            dc->dense_marks = (uint32_t) (mk < 0xFFFFFFFFull ? mk : 0xFFFFFFFFull);
        }
    }
    if (!emit || E == 0) return;
    uint32_t *nhend = hbounds + n_reads;
    bool have_prev = false;
    uint32_t prev_r = 0, prev_end = 0;
    for (uint32_t base = 0; base < E; base += 64) {
        const uint32_t e = base + lane;
        // the tile of entry e: the largest T with estart[T] <= e
        uint32_t T = 0;
#pragma unroll
        for (uint32_t step = SCATTER_TILES / 2; step > 0; step >>= 1) {
            const uint32_t cand = T + step;
            const uint32_t v = __shfl(estart, cand & 63);
            if (v <= e) T = cand;
        }
        const uint32_t first = __shfl(estart, T);
        uint4 ent = make_uint4(0, 0, 0, 0), rec0 = make_uint4(0, 0, 0, 0);
        if (e < E) {
            const uint4 *slot = ents + 2 * ((uint64_t) (t0 + T) * stride + (e - first));
            ent = slot[0];
            rec0 = slot[1];
        }
        const uint32_t cn = ent.w;
        uint32_t tot;
        const uint32_t ho = run + wave_scan(cn, &tot);
        const uint32_t pos = ent.y & 0x7FFFFFFFu, strand = ent.y >> 31;
        for (uint32_t q = 0; q < cn; ++q) {
            const uint32_t d = ho + q;
            if (d >= hit_cap) break;
            // prg, knode<<1|strand, rank, orig: the first record came with the entry
            const uint4 rec = q == 0 ? rec0 : *(const uint4 *) (recs + ent.z + q);
            const uint32_t fwd = (strand == (rec.y & 1u));
            h_read[d] = ent.x;
            h_key[d] = rec.x << 1 | (fwd ? 0u : 1u);                 // forward hits sort first (src/minihit.cpp:54-55)
            h_pos[d] = pos;
            h_rank[d] = rec.z;
            h_rec[d] = rec.w;
        }
        // segment bounds: the entry with hits before this one ends where this one starts
        const unsigned long long act = ballot64(cn != 0);
        if (act) {
            const unsigned long long lower = act & ((1ull << lane) - 1ull);
            const int P = lower ? 63 - __clzll((long long) lower) : 0;
            const uint32_t rp_lane = __shfl(ent.x, P);
            if (cn != 0) {
                const bool has = lower != 0 || have_prev;
                const uint32_t rp = lower ? rp_lane : prev_r;
                const uint32_t at = ho < hit_cap ? ho : hit_cap;
                if (!has || rp != ent.x) atomicMin(&hbounds[ent.x], at);
                if (has && rp != ent.x) atomicMin(&nhend[rp], ~at);
            }
            const int last = 63 - __clzll((long long) act);
            prev_r = __shfl(ent.x, last);
            prev_end = __shfl(ho + cn, last);
            have_prev = true;
        }
        run += tot;
    }
    if (have_prev && lane == 0) atomicMin(&nhend[prev_r], ~(prev_end < hit_cap ? prev_end : hit_cap));
}

// one field (0 minimizers, 1 entries, 2 hits) of the tile counters as a plain array
__global__ __launch_bounds__(256) void tile_field_kernel(const uint4 *__restrict__ tile_cnt, uint32_t n_tiles,
                                                         uint32_t field, uint32_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < n_tiles) out[t] = ((const uint32_t *) (tile_cnt + t))[field];
}

// gathers the slotted minimizers into one contiguous list (only when a caller asks for them)
__global__ __launch_bounds__(256) void compact_minimizers_kernel(const uint32_t *__restrict__ tile_nmini,
                                                                 const uint32_t *__restrict__ tile_moff,
                                                                 uint32_t n_tiles, uint32_t stride,
                                                                 const uint64_t *__restrict__ mkmer,
                                                                 const uint4 *__restrict__ ents,
                                                                 uint64_t *__restrict__ o_kmer,
                                                                 uint32_t *__restrict__ o_read,
                                                                 uint32_t *__restrict__ o_ps) {
    const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_tiles) return;
    const uint32_t n = tile_nmini[g], o = tile_moff[g];
    const uint64_t slot0 = (uint64_t) g * stride;
    for (uint32_t e = lane; e < n; e += 64) {
        const uint4 ent = ents[2 * (slot0 + e)];
        o_kmer[o + e] = mkmer[slot0 + e];
        o_read[o + e] = ent.x;
        o_ps[o + e] = ent.y;
    }
}

// KmerNode coverage (pangenome::Graph::add_hits_to_kmergraphs, src/pangenome/pangraph.cpp:347-383,
// KmerNode::increment_covg, src/kmernode.cpp:42-49): one wavefront per kept cluster adds the
// cluster's weight to the counter of every hit's record, forward hits in covg[0, n_recs), reverse
// hits in covg[n_recs, 2 n_recs).  weight 0 = the cluster did not survive filter_clusters.
__global__ __launch_bounds__(256) void coverage_kernel(const uint32_t *__restrict__ k_rows,
                                                       const uint32_t *__restrict__ weight, uint32_t n_kept,
                                                       const uint32_t *__restrict__ s_rec, uint32_t n_recs,
                                                       uint32_t *__restrict__ covg) {
    const uint32_t lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_kept) return;
    const uint32_t wgt = weight[c];
    if (wgt == 0) return;
    const uint4 row = *(const uint4 *) (k_rows + (uint64_t) c * 8);   // first hit, size, read, prg << 1 | !forward
    uint32_t *dst = covg + ((row.w & 1u) ? n_recs : 0u);
    for (uint32_t i = lane; i < row.y; i += 64) atomicAdd(&dst[s_rec[row.x + i]], wgt);
}

#endif
