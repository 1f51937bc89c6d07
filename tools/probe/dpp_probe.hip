// Probe of DPP whole-wavefront shifts on gfx950: semantics across the 16-lane rows and issue rate
// next to plain v_min_u32.  Build: hipcc --offload-arch=gfx950 -O3 dpp_probe.hip -o dpp_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ uint32_t shr1(uint32_t x, uint32_t old) {
    return __builtin_amdgcn_update_dpp(old, x, 0x138, 0xf, 0xf, false);   // wave_shr:1
}
__device__ __forceinline__ uint32_t shl1(uint32_t x, uint32_t old) {
    return __builtin_amdgcn_update_dpp(old, x, 0x130, 0xf, 0xf, false);   // wave_shl:1
}

__global__ void semantics(const uint32_t *in, uint32_t *out) {
    const uint32_t v = in[threadIdx.x];
    out[threadIdx.x] = shr1(v, 0xAAAAAAAAu);
    out[64 + threadIdx.x] = shl1(v, 0xBBBBBBBBu);
    out[128 + threadIdx.x] = min(v, shr1(v, 0xFFFFFFFFu));
    out[192 + threadIdx.x] = max(v, shl1(v, 0u));
}

template <int MODE>
__global__ __launch_bounds__(256) void rate(uint32_t *out, int iters) {
    uint32_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 2654435761u + j * 40503u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) a[j] = min(a[j] + 1u, a[(j + 3) & 7]);                       // add + plain min
            else if (MODE == 1) a[j] = min(a[j] + 1u, shr1(a[(j + 3) & 7], 0xFFFFFFFFu));   // add + dpp min
            else a[j] = min(a[j] + 1u, __builtin_amdgcn_update_dpp(0xFFFFFFFFu, a[(j + 3) & 7], 0x111, 0xf, 0xf, false)); // row_shr:1
        }
    }
    uint32_t s = 0;
    for (int j = 0; j < 8; ++j) s ^= a[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    uint32_t *din, *dout;
    CK(hipMalloc(&din, 64 * 4));
    CK(hipMalloc(&dout, 256 * 4));
    std::vector<uint32_t> in(64), out(256);
    for (int i = 0; i < 64; ++i) in[i] = 1000 + (i * 37) % 64;
    CK(hipMemcpy(din, in.data(), 256, hipMemcpyHostToDevice));
    semantics<<<1, 64>>>(din, dout);
    CK(hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const uint32_t e0 = i ? in[i - 1] : 0xAAAAAAAAu, e1 = i < 63 ? in[i + 1] : 0xBBBBBBBBu;
        const uint32_t e2 = i ? (in[i] < in[i - 1] ? in[i] : in[i - 1]) : in[i];
        const uint32_t e3 = i < 63 ? (in[i] > in[i + 1] ? in[i] : in[i + 1]) : in[i];
        if (out[i] != e0 || out[64 + i] != e1 || out[128 + i] != e2 || out[192 + i] != e3) {
            if (bad < 8) printf("lane %d: shr %x (want %x) shl %x (want %x) min %u (want %u) max %u (want %u)\n", i, out[i], e0, out[64 + i], e1, out[128 + i], e2, out[192 + i], e3);
            ++bad;
        }
    }
    printf("semantics: %s (%d lanes differ)\n", bad ? "DIFFER" : "as expected", bad);

    const int blocks = 256 * 8, iters = 4096;
    uint32_t *dbig;
    CK(hipMalloc(&dbig, (size_t) blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) rate<0><<<blocks, 256>>>(dbig, iters);
            else if (mode == 1) rate<1><<<blocks, 256>>>(dbig, iters);
            else rate<2><<<blocks, 256>>>(dbig, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double insts = (double) blocks * 4 * iters * 16;   // wavefront instructions (add + min per element)
        printf("mode %d (%s): %.3f ms, %.1f G wavefront-instr/s\n", mode, mode == 0 ? "plain min" : mode == 1 ? "wave_shr:1 min" : "row_shr:1 min", best, insts / best / 1e6);
    }
    return 0;
}
