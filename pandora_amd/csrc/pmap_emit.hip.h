// Probe (unfused mode), hit emission and minimizer gather kernels of libpandora_amd (gfx950).
// Included by pmap_kernels.hip.h only.
//
// The sketch kernel leaves, per wavefront tile g, `tile_nent[g]` entries in the tile's 256 slots
// of `ents`: {read index, position | strand << 31, first record, record count}.
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
__global__ __launch_bounds__(256) void probe_wave_kernel(const uint64_t *__restrict__ mkmer,
                                                         const uint32_t *__restrict__ tile_nent, uint32_t n_tiles,
                                                         const PmapSlot *__restrict__ slots, uint32_t log2cap,
                                                         uint4 *__restrict__ ents, uint32_t *__restrict__ tile_nhits) {
    const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_tiles) return;
    const uint32_t n = tile_nent[g];
    const uint64_t slot0 = (uint64_t) g * WAVE_POS;
    uint32_t mine = 0;
    for (uint32_t e = lane; e < n; e += 64) {
        uint32_t cnt, off;
        table_lookup(slots, log2cap, mkmer[slot0 + e], &cnt, &off);
        uint32_t *ent = (uint32_t *) (ents + slot0 + e);
        ent[2] = off;
        ent[3] = cnt;
        mine += cnt;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
    if (lane == 0) tile_nhits[g] = mine;
}

// scatter: one wavefront per group of 64 tiles.  The group's entries are numbered consecutively
// (tile by tile, so still in (read, position) order); lane e takes entry e, finds its tile by a
// search over the per-tile prefix sums held in the lanes, and writes the entry's hits (MinimizerHit
// ctor, src/minihit.cpp:14-22) at the group's offset + the prefix sum of the record counts.
__global__ __launch_bounds__(256) void scatter_group_kernel(
    const uint32_t *__restrict__ tile_nent, const uint32_t *__restrict__ tile_hoff, uint32_t n_tiles,
    const uint4 *__restrict__ ents, const PmapRec *__restrict__ recs, uint32_t hit_cap,
    uint32_t *__restrict__ h_read, uint32_t *__restrict__ h_key, uint32_t *__restrict__ h_pos,
    uint32_t *__restrict__ h_rank, uint32_t *__restrict__ h_rec) {
    const uint32_t lane = threadIdx.x & 63, G = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t t0 = G * 64;
    if (t0 >= n_tiles) return;
    const uint32_t t = t0 + lane;
    const uint32_t n = t < n_tiles ? tile_nent[t] : 0;
    uint32_t E;
    const uint32_t estart = wave_scan(n, &E);
    if (E == 0) return;
    uint32_t run = tile_hoff[t0];
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
