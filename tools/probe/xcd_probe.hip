// Does routing index lookups to the XCD whose L2 holds their slice of the key-range bitmap pay?
//
// Model of the lookup phase of a 200 000-locus index (28 M keys below 2^30, exact bitmap of 128 MiB over the key
// range, 2 GiB slot table), without the sketching: tiles of 132 keys drawn like window minima (the smallest of 25
// uniform values: half of them fall into the lowest 3 % of the range), 156 000 tiles per launch.
//
//   direct       one wavefront per tile: bitmap test for every key, table slot (32 bytes) for the present ones,
//                entries of the present ones written compacted in tile order (what the fused sketch kernel does).
//   partitioned  split:  a wavefront per tile writes {key, ordinal} into 8 queues per tile, queue = bits 10..12 of
//                        the key (the 128-byte line of the bitmap, modulo 8);
//                lookup: workgroup b serves queue b % 8 of 8 tiles, so all workgroups of one XCD (blocks are dealt
//                        round-robin over the XCDs) touch one eighth of the bitmap lines; answers of present keys
//                        go densely to (tile, queue);
//                merge:  a wavefront per tile puts the 8 answer lists back in ordinal order, same output as direct.
//   scrambled    the partitioned kernels with workgroup b serving queue (b / 8) % 8: same work, no XCD affinity.
//
// Prints the time of every kernel alone and a checksum of the outputs.
// Build: hipcc --offload-arch=gfx950 -O3 xcd_probe.hip -o xcd_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static const uint32_t TILE_KEYS = 132;      // keys per tile
static const uint32_t QCAP = 32;            // entries per (tile, queue)
static const uint32_t ENT_STRIDE = 136;     // entry slots per tile in the compacted output
static const uint32_t KEY_BITS = 30;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// the smallest of `n` pseudo-random KEY_BITS-bit values
__device__ __forceinline__ uint32_t draw_key(uint32_t seed, uint32_t n) {
    uint32_t m = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; ++i) {
        seed = mix(seed + 0x9e3779b9u * (i + 1));
        const uint32_t v = seed >> (32 - KEY_BITS);
        m = v < m ? v : m;
    }
    return m;
}

__global__ void fill_bitmap(uint32_t *bitmap, uint32_t n_keys, uint32_t n_min) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys) return;
    const uint32_t key = draw_key(i * 2u + 1u, n_min);
    atomicOr(&bitmap[key >> 5], 1u << (key & 31));
}

__global__ void fill_keys(uint32_t *keys, uint32_t n, uint32_t n_min) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = draw_key(i * 2u + 0x40000000u, n_min);
}

__device__ __forceinline__ uint32_t rank_in(unsigned long long b) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0));
}

// ---- direct
template <int VARIANT>      // 0: as the library does; 1: only the first half of a slot is read; 2: and only half an entry written;
                            // 3: a slot is read by a pair of lanes in one instruction; 4: and an entry written by a pair
__global__ __launch_bounds__(256) void direct_kernel(const uint32_t *__restrict__ keys, uint32_t n_tiles,
                                                     const uint32_t *__restrict__ bitmap, const uint4 *__restrict__ table,
                                                     uint32_t slot_mask, uint4 *__restrict__ ents, uint32_t *__restrict__ tile_n) {
    const uint32_t lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    uint32_t nent = 0;
    for (uint32_t base = 0; base < TILE_KEYS; base += 64) {
        const uint32_t e = base + lane;
        bool present = false;
        uint32_t key = 0;
        if (e < TILE_KEYS) {
            key = keys[(size_t) t * TILE_KEYS + e];
            present = (bitmap[key >> 5] >> (key & 31)) & 1u;
        }
        uint4 v = make_uint4(0, 0, 0, 0), w = v;
        const uint32_t slot = mix(key) & slot_mask;
        if (VARIANT < 3) {
            if (present) {
                v = table[2 * (size_t) slot];
                if (VARIANT == 0) w = table[2 * (size_t) slot + 1];
            }
        } else {
            // pass 0 serves the even lane of every pair, pass 1 the odd one: both lanes of the pair read one half each
            // of that lane's slot in a single instruction, then the halves go back through DPP (quad_perm swaps)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                // slot and presence of the pair's lane `pass`
                const int pr = present ? 1 : 0;
                const uint32_t src_slot = pass ? (uint32_t) __builtin_amdgcn_update_dpp(0, (int) slot, 0xF5, 0xf, 0xf, false)
                                               : (uint32_t) __builtin_amdgcn_update_dpp(0, (int) slot, 0xA0, 0xf, 0xf, false);
                const uint32_t src_pres = pass ? (uint32_t) __builtin_amdgcn_update_dpp(0, pr, 0xF5, 0xf, 0xf, false)
                                               : (uint32_t) __builtin_amdgcn_update_dpp(0, pr, 0xA0, 0xf, 0xf, false);
                uint4 x = make_uint4(0, 0, 0, 0);
                if (src_pres) x = table[2 * (size_t) src_slot + (lane & 1u)];
                // the other lane's half
                uint4 y;
                y.x = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x.x, 0xB1, 0xf, 0xf, false);
                y.y = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x.y, 0xB1, 0xf, 0xf, false);
                y.z = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x.z, 0xB1, 0xf, 0xf, false);
                y.w = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x.w, 0xB1, 0xf, 0xf, false);
                if ((lane & 1u) == (uint32_t) pass) {
                    v = pass ? y : x;
                    w = pass ? x : y;
                }
            }
        }
        const unsigned long long b = __builtin_amdgcn_ballot_w64(present);
        if (VARIANT < 4) {
            if (present) {
                const size_t s = (size_t) t * ENT_STRIDE + nent + rank_in(b);
                ents[2 * s] = make_uint4(key, e, v.x, v.w);
                if (VARIANT != 2) ents[2 * s + 1] = w;
            }
        } else {
            // entries through LDS so that consecutive lanes write consecutive 16-byte halves
            __shared__ uint4 stage[4][128];
            uint4 *st = stage[threadIdx.x >> 6];
            const uint32_t r = rank_in(b);
            if (present) {
                st[2 * r] = make_uint4(key, e, v.x, v.w);
                st[2 * r + 1] = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t nh = 2 * (uint32_t) __popcll(b);
            const size_t s0 = 2 * ((size_t) t * ENT_STRIDE + nent);
            if (lane < nh) ents[s0 + lane] = st[lane];
            if (lane + 64 < nh) ents[s0 + lane + 64] = st[lane + 64];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        nent += __popcll(b);
    }
    if (lane == 0) tile_n[t] = nent;
}

// ---- partitioned: split
__global__ __launch_bounds__(256) void split_kernel(const uint32_t *__restrict__ keys, uint32_t n_tiles,
                                                    uint2 *__restrict__ queue, uint32_t *__restrict__ qcnt,
                                                    uint32_t *__restrict__ dropped) {
    const uint32_t lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t base = 0; base < TILE_KEYS; base += 64) {
        const uint32_t e = base + lane;
        const bool have = e < TILE_KEYS;
        const uint32_t key = have ? keys[(size_t) t * TILE_KEYS + e] : 0u;
        const uint32_t q = (key >> 10) & 7u;
        uint32_t my = 0;
#pragma unroll
        for (uint32_t qq = 0; qq < 8; ++qq) {
            const unsigned long long b = __builtin_amdgcn_ballot_w64(have && q == qq);
            if (q == qq) my = cnt[qq] + rank_in(b);
            cnt[qq] += __popcll(b);
        }
        if (have) {
            if (my < QCAP) queue[((size_t) t * 8 + q) * QCAP + my] = make_uint2(key, e);
            else atomicAdd(dropped, 1u);
        }
    }
    if (lane < 8) {
        uint32_t c = 0;
#pragma unroll
        for (uint32_t qq = 0; qq < 8; ++qq)
            if (lane == qq) c = cnt[qq];
        qcnt[(size_t) t * 8 + lane] = c < QCAP ? c : QCAP;
    }
}

// ---- partitioned: lookup.  Workgroup b: queue q of the tiles 8 * (b / 8) .. + 7, two tiles per wavefront.
// One (queue, 8 tiles) unit of the lookup pass, by a workgroup.
__device__ __forceinline__ void lookup_unit(uint32_t q, uint32_t grp, const uint2 *__restrict__ queue,
                                            const uint32_t *__restrict__ qcnt, uint32_t n_tiles,
                                            const uint32_t *__restrict__ bitmap, const uint4 *__restrict__ table,
                                            uint32_t slot_mask, uint4 *__restrict__ ans, uint32_t *__restrict__ acnt,
                                            uint32_t skip = 0) {
    const uint32_t lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
    const uint32_t t = grp * 8 + (threadIdx.x >> 6) * 2 + half;
    bool present = false;
    uint32_t key = 0, e = 0;
    if (t < n_tiles) {
        const uint32_t n = qcnt[(size_t) t * 8 + q];
        if (li < n && li < QCAP) {
            const uint2 x = queue[((size_t) t * 8 + q) * QCAP + li];
            key = x.x & ((1u << KEY_BITS) - 1);
            e = x.y;
            present = (skip & 4u) ? (mix(key) & 3u) == 0u : (bitmap[key >> 5] >> (key & 31)) & 1u;
        }
    }
    uint4 v = make_uint4(0, 0, 0, 0), w = v;
    if (present && !(skip & 1u)) {
        const uint32_t slot = mix(key) & slot_mask;
        v = table[2 * (size_t) slot];
        w = table[2 * (size_t) slot + 1];
    }
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(present);
    const uint32_t mine = half ? (uint32_t) (bal >> 32) : (uint32_t) bal;
    if (present && !(skip & 2u)) {
        const uint32_t r = __popc(mine & ((1u << li) - 1u));
        const size_t s = ((size_t) t * 8 + q) * QCAP + r;
        ans[2 * s] = make_uint4(key, e, v.x, v.w);
        ans[2 * s + 1] = w;
    }
    if (li == 0 && t < n_tiles) acnt[(size_t) t * 8 + q] = __popc(mine);
}

// ---- partitioned: lookup.  Workgroup b: queue q of the tiles 8 * (b / 8) .. + 7, two tiles per wavefront.
// hist[8 * xcc + q] counts the workgroups that served queue q on XCD xcc.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void lookup_kernel(const uint2 *__restrict__ queue, const uint32_t *__restrict__ qcnt,
                                                     uint32_t n_tiles, const uint32_t *__restrict__ bitmap,
                                                     const uint4 *__restrict__ table, uint32_t slot_mask,
                                                     uint4 *__restrict__ ans, uint32_t *__restrict__ acnt,
                                                     uint32_t *__restrict__ hist, uint32_t skip) {
    const uint32_t b = blockIdx.x;
    const uint32_t q = ALIGNED ? (b & 7u) : ((b >> 3) & 7u);
    const uint32_t grp = ALIGNED ? (b >> 3) : ((b & 7u) | ((b >> 6) << 3));
    if (hist && threadIdx.x == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        atomicAdd(&hist[8 * (xcc & 7u) + q], 1u);
    }
    lookup_unit(q, grp, queue, qcnt, n_tiles, bitmap, table, slot_mask, ans, acnt, skip);
}

// The same with a fixed grid (a multiple of 64 workgroups): workgroup b keeps its queue (b % 8, or (b / 8) % 8
// when not ALIGNED) and walks over the tile groups.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void lookup_loop_kernel(const uint2 *__restrict__ queue, const uint32_t *__restrict__ qcnt,
                                                          uint32_t n_tiles, uint32_t n_groups,
                                                          const uint32_t *__restrict__ bitmap, const uint4 *__restrict__ table,
                                                          uint32_t slot_mask, uint4 *__restrict__ ans,
                                                          uint32_t *__restrict__ acnt) {
    const uint32_t b = blockIdx.x;
    const uint32_t q = ALIGNED ? (b & 7u) : ((b >> 3) & 7u);
    const uint32_t first = ALIGNED ? (b >> 3) : ((b & 7u) | ((b >> 6) << 3));
    for (uint32_t grp = first; grp < n_groups; grp += gridDim.x / 8)
        lookup_unit(q, grp, queue, qcnt, n_tiles, bitmap, table, slot_mask, ans, acnt);
}

// ---- partitioned: merge, one wavefront per tile
__global__ __launch_bounds__(256) void merge_kernel(const uint4 *__restrict__ ans, const uint32_t *__restrict__ acnt,
                                                    uint32_t n_tiles, uint4 *__restrict__ ents, uint32_t *__restrict__ tile_n) {
    __shared__ uint32_t s_ord[4][8 * QCAP];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, t = blockIdx.x * 4 + wv;
    if (t >= n_tiles) return;
    uint32_t start[9];
    start[0] = 0;
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
        uint32_t c = acnt[(size_t) t * 8 + q];
        c = c < QCAP ? c : QCAP;
        start[q + 1] = start[q] + c;
    }
    const uint32_t n = start[8];        // <= 256
    // where entry j of the concatenated lists sits
    auto place = [&](uint32_t j) -> size_t {
        uint32_t q = 0;
#pragma unroll
        for (uint32_t qq = 1; qq < 8; ++qq) q += j >= start[qq] ? 1u : 0u;
        uint32_t sq = 0;
#pragma unroll
        for (uint32_t qq = 0; qq < 8; ++qq)
            if (q == qq) sq = start[qq];
        return ((size_t) t * 8 + q) * QCAP + (j - sq);
    };
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t j = base + lane;
        if (j < n) s_ord[wv][j] = ans[2 * place(j)].y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t j = base + lane;
        if (j < n) {
            const size_t s = place(j);
            const uint4 a0 = ans[2 * s], a1 = ans[2 * s + 1];
            uint32_t r = 0;
            for (uint32_t i = 0; i < n; ++i) r += s_ord[wv][i] < a0.y ? 1u : 0u;
            if (r < ENT_STRIDE) {
                const size_t d = (size_t) t * ENT_STRIDE + r;
                ents[2 * d] = a0;
                ents[2 * d + 1] = a1;
            }
        }
    }
    if (lane == 0) tile_n[t] = n < ENT_STRIDE ? n : ENT_STRIDE;
}

// ---- partitioned, second form: a wavefront serves queue q of 8 tiles with its lanes assigned densely to the
// ~132 queued keys (up to 4 rounds of 64, all loads of a stage in flight together); the answers of the 8 tiles
// go densely, in (tile, ordinal) order, into one list per (8 tiles, queue).  Workgroup b: queue b % 8 of the
// tile groups 4 * (b / 8) .. + 3.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void lookup8_kernel(const uint2 *__restrict__ queue, const uint32_t *__restrict__ qcnt,
                                                      uint32_t n_tiles, uint32_t n_groups,
                                                      const uint32_t *__restrict__ bitmap, const uint4 *__restrict__ table,
                                                      uint32_t slot_mask, uint4 *__restrict__ ans,
                                                      uint32_t *__restrict__ acnt, uint32_t skip) {
    const uint32_t b = blockIdx.x;
    const uint32_t q = ALIGNED ? (b & 7u) : ((b >> 3) & 7u);
    const uint32_t g4 = ALIGNED ? (b >> 3) : ((b & 7u) | ((b >> 6) << 3));
    const uint32_t lane = threadIdx.x & 63, g8 = g4 * 4 + (threadIdx.x >> 6);
    if (g8 >= n_groups) return;
    uint32_t c = 0;
    if (lane < 8 && g8 * 8 + lane < n_tiles) {
        c = qcnt[((size_t) g8 * 8 + lane) * 8 + q];
        c = c < QCAP ? c : QCAP;
    }
    uint32_t start[9];
    start[0] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) start[i + 1] = start[i] + (uint32_t) __builtin_amdgcn_readlane((int) c, i);
    const uint32_t n = start[8];            // <= 256, the same in every lane
    uint32_t key[4], e[4], ti[4];
    bool have[4], present[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t j = 64 * r + lane;
        have[r] = j < n;
        key[r] = e[r] = ti[r] = 0;
        if (have[r]) {
            uint32_t t = 0;
#pragma unroll
            for (int i = 1; i < 8; ++i) t += j >= start[i] ? 1u : 0u;
            uint32_t st = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (t == (uint32_t) i) st = start[i];
            ti[r] = t;
            const uint2 x = queue[(((size_t) g8 * 8 + t) * 8 + q) * QCAP + (j - st)];
            key[r] = x.x & ((1u << KEY_BITS) - 1);
            e[r] = x.y;
        }
    }
    uint32_t bw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bw[r] = have[r] ? ((skip & 4u) ? ((mix(key[r]) & 3u) == 0u ? 0xFFFFFFFFu : 0u) : bitmap[key[r] >> 5]) : 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) present[r] = (bw[r] >> (key[r] & 31)) & 1u;
    uint4 v[4], w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = w[r] = make_uint4(0, 0, 0, 0);
        if (present[r] && !(skip & 1u)) {
            const uint32_t slot = mix(key[r]) & slot_mask;
            v[r] = table[2 * (size_t) slot];
            w[r] = table[2 * (size_t) slot + 1];
        }
    }
    uint32_t nans = 0, mine = 0;        // lanes 0..7: answers of tile `lane`
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(present[r]);
        if (present[r] && !(skip & 2u)) {
            const size_t s = ((size_t) g8 * 8 + q) * (8 * QCAP) + nans + rank_in(bal);
            ans[2 * s] = make_uint4(key[r], e[r], v[r].x, v[r].w);
            ans[2 * s + 1] = w[r];
        }
        nans += __popcll(bal);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t k = __popcll(__builtin_amdgcn_ballot_w64(present[r] && ti[r] == (uint32_t) i));
            if (lane == (uint32_t) i) mine += k;
        }
    }
    if (lane < 8 && g8 * 8 + lane < n_tiles) acnt[((size_t) g8 * 8 + lane) * 8 + q] = mine;
}

// merge for the lists of lookup8_kernel, one wavefront per tile
__global__ __launch_bounds__(256) void merge8_kernel(const uint4 *__restrict__ ans, const uint32_t *__restrict__ acnt,
                                                     uint32_t n_tiles, uint4 *__restrict__ ents, uint32_t *__restrict__ tile_n) {
    __shared__ uint32_t s_ord[4][8 * QCAP];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, t = blockIdx.x * 4 + wv;
    if (t >= n_tiles) return;
    const uint32_t g8 = t >> 3, ti = t & 7u;
    // lanes 0..7 (queue = lane): answers of this tile and of the tiles before it in its group
    uint32_t cnt = 0, off = 0;
    if (lane < 8) {
        for (uint32_t i = 0; i <= ti; ++i) {
            uint32_t c = g8 * 8 + i < n_tiles ? acnt[((size_t) g8 * 8 + i) * 8 + lane] : 0u;
            c = c < QCAP ? c : QCAP;
            if (i < ti) off += c;
            else cnt = c;
        }
    }
    uint32_t start[9], offs[8];
    start[0] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        start[i + 1] = start[i] + (uint32_t) __builtin_amdgcn_readlane((int) cnt, i);
        offs[i] = (uint32_t) __builtin_amdgcn_readlane((int) off, i);
    }
    const uint32_t n = start[8];
    auto place = [&](uint32_t j) -> size_t {
        uint32_t q = 0;
#pragma unroll
        for (uint32_t qq = 1; qq < 8; ++qq) q += j >= start[qq] ? 1u : 0u;
        uint32_t sq = 0, oq = 0;
#pragma unroll
        for (uint32_t qq = 0; qq < 8; ++qq)
            if (q == qq) { sq = start[qq]; oq = offs[qq]; }
        uint32_t i = oq + (j - sq);
        i = i < 8 * QCAP ? i : 8 * QCAP - 1;
        return ((size_t) g8 * 8 + q) * (8 * QCAP) + i;
    };
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t j = base + lane;
        if (j < n) s_ord[wv][j] = ans[2 * place(j)].y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t j = base + lane;
        if (j < n) {
            const size_t s = place(j);
            const uint4 a0 = ans[2 * s], a1 = ans[2 * s + 1];
            uint32_t r = 0;
            for (uint32_t i = 0; i < n; ++i) r += s_ord[wv][i] < a0.y ? 1u : 0u;
            if (r < ENT_STRIDE) {
                const size_t d = (size_t) t * ENT_STRIDE + r;
                ents[2 * d] = a0;
                ents[2 * d + 1] = a1;
            }
        }
    }
    if (lane == 0) tile_n[t] = n < ENT_STRIDE ? n : ENT_STRIDE;
}

__global__ void checksum_kernel(const uint4 *__restrict__ ents, const uint32_t *__restrict__ tile_n, uint32_t n_tiles,
                                unsigned long long *__restrict__ sums) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    unsigned long long s = 0;
    const uint32_t n = tile_n[t] < ENT_STRIDE ? tile_n[t] : ENT_STRIDE;
    for (uint32_t i = 0; i < n; ++i) {
        const uint4 a = ents[2 * ((size_t) t * ENT_STRIDE + i)];
        s += (unsigned long long) mix(a.x + 31u * i) + a.y;
    }
    atomicAdd(&sums[0], s);
    atomicAdd(&sums[1], (unsigned long long) n);
}

struct Timer {
    hipEvent_t a, b;
    int init() { return hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess; }
};

#define TIME(label, launch)                                                              \
    do {                                                                                 \
        launch;                                                                          \
        CK(hipDeviceSynchronize());                                                      \
        CK(hipEventRecord(tm.a));                                                        \
        for (int r_ = 0; r_ < reps; ++r_) { launch; }                                    \
        CK(hipEventRecord(tm.b));                                                        \
        CK(hipEventSynchronize(tm.b));                                                   \
        float ms_ = 0;                                                                   \
        CK(hipEventElapsedTime(&ms_, tm.a, tm.b));                                       \
        printf("  %-34s %8.3f ms\n", label, ms_ / reps);                                 \
    } while (0)

int main(int argc, char **argv) {
    const uint32_t n_tiles = argc > 1 ? (uint32_t) atoi(argv[1]) : 156000u;
    const uint32_t n_index_keys = argc > 2 ? (uint32_t) atoi(argv[2]) : 34000000u;   // draws; duplicates collapse
    const uint32_t n_min = 25;
    const int reps = 5;
    const size_t table_bytes = 2ull << 30;
    const uint32_t slot_mask = (uint32_t) (table_bytes / 32 - 1);
    const uint32_t n_keys = n_tiles * TILE_KEYS;
    if (n_tiles == 0 || n_tiles > 400000u) { printf("tiles out of range\n"); return 1; }

    uint32_t *bitmap, *keys, *qcnt, *acnt, *tile_n, *dropped;
    uint2 *queue;
    uint4 *table, *ans, *ents;
    unsigned long long *sums;
    const uint32_t tiles_pad = (n_tiles + 255) & ~255u;     // the scrambled mapping covers whole runs of 64 workgroups
    CK(hipMalloc((void **) &bitmap, (1ull << KEY_BITS) / 8));
    CK(hipMemset(bitmap, 0, (1ull << KEY_BITS) / 8));
    CK(hipMalloc((void **) &keys, (size_t) n_keys * 4));
    CK(hipMalloc((void **) &table, table_bytes));
    CK(hipMemset(table, 1, table_bytes));
    CK(hipMalloc((void **) &queue, (size_t) tiles_pad * 8 * QCAP * sizeof(uint2)));
    CK(hipMalloc((void **) &qcnt, (size_t) tiles_pad * 8 * 4));
    CK(hipMemset(qcnt, 0, (size_t) tiles_pad * 8 * 4));
    CK(hipMalloc((void **) &acnt, (size_t) tiles_pad * 8 * 4));
    CK(hipMemset(acnt, 0, (size_t) tiles_pad * 8 * 4));
    CK(hipMalloc((void **) &ans, (size_t) tiles_pad * 8 * QCAP * 32));
    CK(hipMalloc((void **) &ents, (size_t) tiles_pad * ENT_STRIDE * 32));
    CK(hipMalloc((void **) &tile_n, (size_t) tiles_pad * 4));
    CK(hipMalloc((void **) &dropped, 4));
    CK(hipMemset(dropped, 0, 4));
    CK(hipMalloc((void **) &sums, 16));
    uint32_t *hist;
    CK(hipMalloc((void **) &hist, 256));

    fill_bitmap<<<(n_index_keys + 255) / 256, 256>>>(bitmap, n_index_keys, n_min);
    fill_keys<<<(n_keys + 255) / 256, 256>>>(keys, n_keys, n_min);
    CK(hipDeviceSynchronize());
    Timer tm;
    if (tm.init()) return 1;

    const uint32_t tile_blocks = (n_tiles + 3) / 4, lookup_blocks = tiles_pad;     // (tiles / 8) groups x 8 queues
    unsigned long long h[2];
    printf("%u tiles of %u keys, %u index key draws, 128 MiB bitmap, 2 GiB table\n", n_tiles, TILE_KEYS, n_index_keys);

    printf("direct\n");
    TIME("  half a slot read", (direct_kernel<1><<<tile_blocks, 256>>>(keys, n_tiles, bitmap, table, slot_mask, ents, tile_n)));
    TIME("  and half an entry written", (direct_kernel<2><<<tile_blocks, 256>>>(keys, n_tiles, bitmap, table, slot_mask, ents, tile_n)));
    TIME("  slot read by a pair of lanes", (direct_kernel<3><<<tile_blocks, 256>>>(keys, n_tiles, bitmap, table, slot_mask, ents, tile_n)));
    CK(hipMemset(sums, 0, 16));
    checksum_kernel<<<(n_tiles + 255) / 256, 256>>>(ents, tile_n, n_tiles, sums);
    CK(hipMemcpy(h, sums, 16, hipMemcpyDeviceToHost));
    printf("  present %llu, checksum %llx\n", h[1], h[0]);
    TIME("  and entries written through LDS", (direct_kernel<4><<<tile_blocks, 256>>>(keys, n_tiles, bitmap, table, slot_mask, ents, tile_n)));
    CK(hipMemset(sums, 0, 16));
    checksum_kernel<<<(n_tiles + 255) / 256, 256>>>(ents, tile_n, n_tiles, sums);
    CK(hipMemcpy(h, sums, 16, hipMemcpyDeviceToHost));
    printf("  present %llu, checksum %llx\n", h[1], h[0]);
    TIME("direct_kernel", (direct_kernel<0><<<tile_blocks, 256>>>(keys, n_tiles, bitmap, table, slot_mask, ents, tile_n)));
    CK(hipMemset(sums, 0, 16));
    checksum_kernel<<<(n_tiles + 255) / 256, 256>>>(ents, tile_n, n_tiles, sums);
    CK(hipMemcpy(h, sums, 16, hipMemcpyDeviceToHost));
    printf("  present %llu of %u keys (%.1f %%), checksum %llx\n", h[1], n_keys, 100.0 * h[1] / n_keys, h[0]);

    for (int aligned = 1; aligned >= 0; --aligned) {
        printf(aligned ? "partitioned (workgroup b serves queue b %% 8)\n" : "scrambled (workgroup b serves queue (b / 8) %% 8)\n");
        CK(hipMemset(ents, 0, (size_t) tiles_pad * ENT_STRIDE * 32));
        CK(hipMemset(dropped, 0, 4));
        TIME("split_kernel", (split_kernel<<<tile_blocks, 256>>>(keys, n_tiles, queue, qcnt, dropped)));
        CK(hipMemset(hist, 0, 256));
        if (aligned)
            lookup_kernel<true><<<lookup_blocks, 256>>>(queue, qcnt, n_tiles, bitmap, table, slot_mask, ans, acnt, hist, 0u);
        else
            lookup_kernel<false><<<lookup_blocks, 256>>>(queue, qcnt, n_tiles, bitmap, table, slot_mask, ans, acnt, hist, 0u);
        uint32_t hh[64];
        CK(hipMemcpy(hh, hist, 256, hipMemcpyDeviceToHost));
        printf("  workgroups by (XCD, queue):");
        for (int x = 0; x < 8; ++x) {
            uint32_t best = 0, tot = 0;
            for (int qq = 0; qq < 8; ++qq) { tot += hh[8 * x + qq]; if (hh[8 * x + qq] > hh[8 * x + best]) best = qq; }
            printf("  xcd%d: %.0f%% q%u", x, tot ? 100.0 * hh[8 * x + best] / tot : 0.0, best);
        }
        printf("\n");
        if (aligned)
            TIME("lookup_kernel", (lookup_kernel<true><<<lookup_blocks, 256>>>(queue, qcnt, n_tiles, bitmap, table, slot_mask, ans, acnt, nullptr, 0u)));
        else
            TIME("lookup_kernel", (lookup_kernel<false><<<lookup_blocks, 256>>>(queue, qcnt, n_tiles, bitmap, table, slot_mask, ans, acnt, nullptr, 0u)));
        if (aligned) {
            const char *what[] = {"", "  no table reads", "  no answer writes", "  neither", "  no bitmap reads", "  bitmap reads only... (no bitmap, no table)", "  no bitmap, no writes", "  queue reads only"};
            for (uint32_t skip = 1; skip < 8; ++skip)
                TIME(what[skip], (lookup_kernel<true><<<lookup_blocks, 256>>>(queue, qcnt, n_tiles, bitmap, table, slot_mask, ans, acnt, nullptr, skip)));
        }
        for (uint32_t per_cu = 2; per_cu <= 8; per_cu *= 2) {
            char label[64];
            snprintf(label, sizeof label, "lookup_loop_kernel, %u blocks/CU", per_cu);
            const uint32_t grid = 256 * per_cu;
            if (aligned)
                TIME(label, (lookup_loop_kernel<true><<<grid, 256>>>(queue, qcnt, n_tiles, tiles_pad / 8, bitmap, table, slot_mask, ans, acnt)));
            else
                TIME(label, (lookup_loop_kernel<false><<<grid, 256>>>(queue, qcnt, n_tiles, tiles_pad / 8, bitmap, table, slot_mask, ans, acnt)));
        }
        TIME("merge_kernel", (merge_kernel<<<tile_blocks, 256>>>(ans, acnt, n_tiles, ents, tile_n)));
        CK(hipMemset(sums, 0, 16));
        checksum_kernel<<<(n_tiles + 255) / 256, 256>>>(ents, tile_n, n_tiles, sums);
        CK(hipMemcpy(h, sums, 16, hipMemcpyDeviceToHost));
        uint32_t hd = 0;
        CK(hipMemcpy(&hd, dropped, 4, hipMemcpyDeviceToHost));
        printf("  present %llu, checksum %llx, keys beyond a queue's %u entries: %u (over %d launches)\n", h[1], h[0], QCAP,
               hd, reps + 1);
        // second form
        CK(hipMemset(ents, 0, (size_t) tiles_pad * ENT_STRIDE * 32));
        const uint32_t n_groups = tiles_pad / 8, blocks8 = tiles_pad / 4;      // (groups / 4) x 8 queues, a multiple of 64... of 16
        const char *what8[] = {"lookup8_kernel", "  no table reads", "  no answer writes", "  neither", "  no bitmap reads", "  no bitmap, no table", "  no bitmap, no writes", "  queue reads only"};
        for (uint32_t skip = 7; skip + 1 > 0; --skip) {
            if (!aligned && skip) continue;
            if (aligned)
                TIME(what8[skip], (lookup8_kernel<true><<<blocks8, 256>>>(queue, qcnt, n_tiles, n_groups, bitmap, table, slot_mask, ans, acnt, skip)));
            else
                TIME(what8[skip], (lookup8_kernel<false><<<blocks8, 256>>>(queue, qcnt, n_tiles, n_groups, bitmap, table, slot_mask, ans, acnt, skip)));
        }
        TIME("merge8_kernel", (merge8_kernel<<<tile_blocks, 256>>>(ans, acnt, n_tiles, ents, tile_n)));
        CK(hipMemset(sums, 0, 16));
        checksum_kernel<<<(n_tiles + 255) / 256, 256>>>(ents, tile_n, n_tiles, sums);
        CK(hipMemcpy(h, sums, 16, hipMemcpyDeviceToHost));
        printf("  present %llu, checksum %llx\n", h[1], h[0]);
    }
    return 0;
}
