// Ceiling of random table probes on one MI355X: every lane of every wavefront reads `bytes` (16 or 32) at a
// pseudo-random 32-byte-aligned slot of a table of a given size, `inflight` independent probes per lane
// before the first is used.  Prints probes per second for tables in the L2 (2 MB), the Infinity Cache
// (128 MB) and HBM (2 GB, 8 GB): the denominators for the lookup rates bench.py reports.
// Build: hipcc --offload-arch=gfx950 -O3 random_probe.hip -o random_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int BYTES, int INFLIGHT>
__global__ __launch_bounds__(256) void probe(const uint4 *__restrict__ table, uint32_t slot_mask, uint32_t rounds,
                                             uint32_t *__restrict__ out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0, seed = tid * 2654435761u + 12345u;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint4 v[INFLIGHT], w[INFLIGHT];
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
            seed = mix(seed + i + 1);
            const uint32_t slot = seed & slot_mask;
            v[i] = table[2 * (size_t) slot];
            if (BYTES == 32) w[i] = table[2 * (size_t) slot + 1];
        }
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
            acc += v[i].x ^ v[i].w;
            if (BYTES == 32) acc += w[i].y;
        }
    }
    if (acc == 0x12345678u) out[tid] = acc;      // keeps the loads alive
}

template <int BYTES, int INFLIGHT>
static int run(const uint4 *table, uint32_t slot_mask, uint32_t *out, int blocks, const char *what) {
    const uint32_t rounds = 64 / INFLIGHT * 4;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    probe<BYTES, INFLIGHT><<<blocks, 256>>>(table, slot_mask, rounds, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) probe<BYTES, INFLIGHT><<<blocks, 256>>>(table, slot_mask, rounds, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double probes = (double) reps * blocks * 256.0 * rounds * INFLIGHT;
    printf("  %-28s %2d B x %d in flight: %7.1f G probes/s  (%6.2f TB/s of useful bytes)\n", what, BYTES, INFLIGHT,
           probes / (ms * 1e-3) / 1e9, probes * BYTES / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    const size_t sizes[] = {2ull << 20, 128ull << 20, 2ull << 30, 8ull << 30};
    const char *names[] = {"2 MB table (L2)", "128 MB table (Infinity Cache)", "2 GB table (HBM)", "8 GB table (HBM)"};
    uint32_t *out;
    CK(hipMalloc((void **) &out, 4u << 24));
    for (int s = 0; s < 4; ++s) {
        uint4 *table;
        CK(hipMalloc((void **) &table, sizes[s]));
        CK(hipMemset(table, 1, sizes[s]));
        const uint32_t slot_mask = (uint32_t) (sizes[s] / 32 - 1);
        printf("%s\n", names[s]);
        const int blocks = 256 * 32;        // 8 blocks of 256 threads per CU, 4 waves each, several rounds of them
        if (run<16, 1>(table, slot_mask, out, blocks, "one probe at a time")) return 1;
        if (run<16, 4>(table, slot_mask, out, blocks, "four independent probes")) return 1;
        if (run<32, 1>(table, slot_mask, out, blocks, "one probe at a time")) return 1;
        if (run<32, 4>(table, slot_mask, out, blocks, "four independent probes")) return 1;
        if (run<32, 8>(table, slot_mask, out, blocks, "eight independent probes")) return 1;
        CK(hipFree(table));
    }
    return 0;
}
