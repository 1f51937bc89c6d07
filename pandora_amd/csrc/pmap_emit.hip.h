// Probe (unfused mode), hit emission and minimizer gather kernels of libpandora_amd (gfx950).
// Included by pmap_kernels.hip.h only.
//
// The sketch kernel leaves, per wavefront tile g, tile_cnt[g] = {minimizers, entries, hits, 0} and
// the entries themselves in the tile's 512 slots of `ents`:
// {read index, position | strand << 31, first record, record count}.
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

// probe as its own kernel (unfused mode): one wavefront per tile looks up the tile's minimizers
// (add_read_hits' index->minhash.find, src/utils.cpp:175)
__global__ __launch_bounds__(256) void probe_wave_kernel(const uint64_t *__restrict__ mkmer, uint32_t n_tiles,
                                                         const PmapSlot *__restrict__ slots, uint32_t log2cap,
                                                         const uint32_t *__restrict__ filt, uint32_t fbits,
                                                         uint4 *__restrict__ ents, uint4 *__restrict__ tile_cnt) {
    const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_tiles) return;
    const uint32_t n = tile_cnt[g].y;
    const uint64_t slot0 = (uint64_t) g * WAVE_POS;
    uint32_t mine = 0;
    for (uint32_t e = lane; e < n; e += 64) {
        uint32_t cnt, off;
        index_lookup(slots, log2cap, filt, fbits, mkmer[slot0 + e], &cnt, &off);
        uint32_t *ent = (uint32_t *) (ents + slot0 + e);
        ent[2] = off;
        ent[3] = cnt;
        mine += cnt;
    }
    mine = wave_sum(mine);
    if (lane == 0) ((uint32_t *) (tile_cnt + g))[2] = mine;
}

// Tiles are handled in groups of 64 (one wavefront of the scatter kernel per group).
// group_sum[G] = sums of the tile counters over group G: one wavefront per group.
__global__ __launch_bounds__(256) void group_reduce_kernel(const uint4 *__restrict__ tile_cnt, uint32_t n_tiles,
                                                           uint4 *__restrict__ group_sum) {
    const uint32_t lane = threadIdx.x & 63, G = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t t0 = G * 64;
    if (t0 >= n_tiles) return;
    const uint32_t t = t0 + lane;
    uint4 c = make_uint4(0, 0, 0, 0);
    if (t < n_tiles) c = tile_cnt[t];
    const uint32_t x = wave_sum(c.x), y = wave_sum(c.y), z = wave_sum(c.z);
    if (lane == 0) group_sum[G] = make_uint4(x, y, z, 0);
}

// one block: group_hoff = exclusive scan of the groups' hit counts; the three grand totals go to
// the counters (minimizers, entries, hits)
__global__ __launch_bounds__(256) void group_spine_kernel(const uint4 *__restrict__ group_sum, uint32_t n_groups,
                                                          uint32_t *__restrict__ group_hoff,
                                                          unsigned long long *__restrict__ n_mini,
                                                          unsigned long long *__restrict__ n_ent,
                                                          unsigned long long *__restrict__ n_hits) {
    __shared__ uint32_t lds4[4];
    unsigned long long cx = 0, cy = 0, cz = 0;
    for (uint32_t base = 0; base < n_groups; base += 256) {
        const uint32_t i = base + threadIdx.x;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i < n_groups) v = group_sum[i];
        uint32_t tx, ty, tz;
        block_scan_256(v.x, lds4, &tx);
        block_scan_256(v.y, lds4, &ty);
        const uint32_t ex = block_scan_256(v.z, lds4, &tz);
        if (i < n_groups) group_hoff[i] = (uint32_t) (cz + ex);
        cx += tx;
        cy += ty;
        cz += tz;
    }
    if (threadIdx.x == 0) {
        *n_mini = cx;
        *n_ent = cy;
        *n_hits = cz;
    }
}

// scatter: one wavefront per group of 64 tiles.  The group's entries are numbered consecutively
// (tile by tile, so still in (read, position) order); lane e takes entry e, finds its tile by a
// search over the per-tile prefix sums held in the lanes, and writes the entry's hits (MinimizerHit
// ctor, src/minihit.cpp:14-22) at the group's offset + the prefix sum of the record counts.
__global__ __launch_bounds__(256) void scatter_group_kernel(
    const uint4 *__restrict__ tile_cnt, const uint32_t *__restrict__ group_hoff, uint32_t n_tiles,
    const uint4 *__restrict__ ents, const PmapRec *__restrict__ recs, uint32_t hit_cap,
    uint32_t *__restrict__ h_read, uint32_t *__restrict__ h_key, uint32_t *__restrict__ h_pos,
    uint32_t *__restrict__ h_rank, uint32_t *__restrict__ h_rec) {
    const uint32_t lane = threadIdx.x & 63, G = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t t0 = G * 64;
    if (t0 >= n_tiles) return;
    const uint32_t t = t0 + lane;
    const uint32_t n = t < n_tiles ? tile_cnt[t].y : 0;
    uint32_t E;
    const uint32_t estart = wave_scan(n, &E);
    if (E == 0) return;
    uint32_t run = group_hoff[G];
    for (uint32_t base = 0; base < E; base += 64) {
        const uint32_t e = base + lane;
        // the tile of entry e: the largest T with estart[T] <= e
        uint32_t T = 0;
#pragma unroll
        for (uint32_t step = 32; step > 0; step >>= 1) {
            const uint32_t cand = T + step;
            const uint32_t v = __shfl(estart, cand & 63);
            if (v <= e) T = cand;
        }
        const uint32_t first = __shfl(estart, T);
        uint4 ent = make_uint4(0, 0, 0, 0);
        if (e < E) ent = ents[(uint64_t) (t0 + T) * WAVE_POS + (e - first)];
        const uint32_t c = ent.w;
        uint32_t tot;
        const uint32_t ho = run + wave_scan(c, &tot);
        const uint32_t pos = ent.y & 0x7FFFFFFFu, strand = ent.y >> 31;
        for (uint32_t q = 0; q < c; ++q) {
            const uint32_t d = ho + q;
            if (d >= hit_cap) break;
            const uint4 rec = *(const uint4 *) (recs + ent.z + q);   // prg, knode<<1|strand, rank, orig
            const uint32_t fwd = (strand == (rec.y & 1u));
            h_read[d] = ent.x;
            h_key[d] = rec.x << 1 | (fwd ? 0u : 1u);                 // forward hits sort first (src/minihit.cpp:54-55)
            h_pos[d] = pos;
            h_rank[d] = rec.z;
            h_rec[d] = rec.w;
        }
        run += tot;
    }
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
                                                                 uint32_t n_tiles,
                                                                 const uint64_t *__restrict__ mkmer,
                                                                 const uint4 *__restrict__ ents,
                                                                 uint64_t *__restrict__ o_kmer,
                                                                 uint32_t *__restrict__ o_read,
                                                                 uint32_t *__restrict__ o_ps) {
    const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_tiles) return;
    const uint32_t n = tile_nmini[g], o = tile_moff[g];
    const uint64_t slot0 = (uint64_t) g * WAVE_POS;
    for (uint32_t e = lane; e < n; e += 64) {
        const uint4 ent = ents[slot0 + e];
        o_kmer[o + e] = mkmer[slot0 + e];
        o_read[o + e] = ent.x;
        o_ps[o + e] = ent.y;
    }
}

#endif
