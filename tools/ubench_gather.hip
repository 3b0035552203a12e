// Micro-benchmark: what a 16-byte-per-lane gather instruction costs a CU, by address pattern, next to
// the same data read from an LDS tile.  Prints cycles per wave-instruction per CU (2.4 GHz assumed).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_gather ubench_gather.hip && ./ubench_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef const f4 __attribute__((address_space(1)))* gf4;

// pattern: lane l reads index (start + l * lane_stride + i * step) & mask at iteration i.
template <int UNROLL>
__global__ void __launch_bounds__(64) k_gather(const f4* tab, const int mask, const int lane_stride, const int step,
                                               const int iters, float* out) {
    gf4 p = (gf4)(const void*)tab;
    const int lane = threadIdx.x;
    unsigned idx = (blockIdx.x * 977u + lane * lane_stride) & mask;
    f4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; i += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const f4 v = p[(idx + (i + u) * step) & mask];
            acc += v;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

// dependent chain: next index comes from the loaded value (like a binary search)
__global__ void __launch_bounds__(64) k_chain(const f4* tab, const int mask, const int lane_stride, const int iters, float* out) {
    gf4 p = (gf4)(const void*)tab;
    const int lane = threadIdx.x;
    unsigned idx = (blockIdx.x * 977u + lane * lane_stride) & mask;
    float acc = 0;
    for (int i = 0; i < iters; ++i) {
        const f4 v = p[idx];
        acc += v.x;
        idx = (idx + (unsigned)__float_as_uint(v.w)) & mask;     // w holds a small integer step
    }
    if (acc == 123.456f) out[0] = acc;
}

// LDS tile: the wave loads `tile` consecutive points once, then every lane reads a 20-point window.
__global__ void __launch_bounds__(64) k_lds(const f4* tab, const int mask, const int reps, float* out) {
    __shared__ f4 tile[128];
    gf4 p = (gf4)(const void*)tab;
    const int lane = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        const unsigned base = (blockIdx.x * 977u + r * 131u) & mask;
        tile[lane] = p[(base + lane) & mask];
        tile[lane + 64] = p[(base + lane + 64) & mask];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int rel = (lane * 3) >> 2;        // windows of neighbouring lanes overlap
#pragma unroll
        for (int i = 0; i < 20; ++i) acc += tile[rel + i];
        __builtin_amdgcn_wave_barrier();
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

int main() {
    const int n = 1 << 17;                      // 2 MB table, L2 resident
    std::vector<float> h(4 * n);
    for (int i = 0; i < n; ++i) { h[4 * i] = 1.0f; h[4 * i + 1] = 2.0f; h[4 * i + 2] = 3.0f; unsigned s = 1 + (i * 7) % 5; memcpy(&h[4 * i + 3], &s, 4); }
    f4* tab; float* out;
    CHECK(hipMalloc(&tab, 16 * n)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemcpy(tab, h.data(), 16 * n, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 32 * 4;            // 32 waves per CU resident, 4 rounds
    const int iters = 2048;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto report = [&](const char* name, double instr_per_wave, float ms) {
        const double per_cu = (double)blocks / cus * instr_per_wave;
        printf("%-44s %8.3f ms  %7.1f cycles per wave-instruction per CU\n", name, ms, ms * 1e-3 * 2.4e9 / per_cu);
    };
    struct Pat { const char* name; int lane_stride, step; };
    const Pat pats[] = {
        {"all lanes one address, step 1", 0, 1},
        {"lanes consecutive 16 B (1 KB), step 1", 1, 1},
        {"lanes consecutive, step 64 (new lines)", 1, 64},
        {"lane stride 2 (32 B), step 1", 2, 1},
        {"lane stride 4 (64 B), step 1", 4, 1},
        {"lane stride 8 (128 B), step 1", 8, 1},
        {"lane stride 8, step 8 (new lines)", 8, 8},
        {"lane stride 1031 (scattered), step 1", 1031, 1},
        {"lane stride 1031, step 517", 1031, 517},
    };
    for (const Pat& P : pats) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            k_gather<8><<<blocks, 64>>>(tab, n - 1, P.lane_stride, P.step, iters, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report(P.name, iters, ms);
        }
    }
    for (int ls : {1, 8, 1031}) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            k_chain<<<blocks, 64>>>(tab, n - 1, ls, 512, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            char nm[64]; snprintf(nm, 64, "dependent chain, lane stride %d", ls);
            if (rep) report(nm, 512, ms);
        }
    }
    for (int rep = 0; rep < 2; ++rep) {
        const int reps = 256;
        CHECK(hipEventRecord(e0));
        k_lds<<<blocks, 64>>>(tab, n - 1, reps, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) report("LDS tile: 2 loads + 2 writes + 20 reads (per 24)", reps * 24.0, ms);
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
