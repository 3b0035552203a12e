// Micro-benchmark: what a 16-byte gather costs a CU by the number of ACTIVE lanes of the wave that issues it
// (the ring passes of the search run with 27 to 41 of 64 lanes active).  Table of 2 MB (L2 resident), 32 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_lanes ubench_lanes.hip && ./ubench_lanes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define GL __attribute__((address_space(1)))
typedef float f4 __attribute__((ext_vector_type(4)));

// lanes whose bit is set in `active` gather; lane l reads record (start + l * lane_stride + i * step)
__global__ void __launch_bounds__(64) k_lanes(const char* tab, const unsigned mask, const unsigned long long active,
                                              const int lane_stride, const int step, const int iters, float* out) {
    const char GL* p = (const char GL*)(const void*)tab;
    const int lane = threadIdx.x;
    if (!((active >> lane) & 1ull)) return;
    const unsigned idx = blockIdx.x * 977u + lane * lane_stride;
    float acc = 0;
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned off = ((idx + (i + u) * step) << 4) & mask;
            const f4 v = *(const f4 GL*)(p + off);
            acc += v.x + v.w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const size_t bytes = 2u << 20;
    std::vector<float> h(bytes / 4, 1.0f);
    char* tab; float* out;
    CHECK(hipMalloc(&tab, bytes + 4096)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 32 * 4, iters = 2048;
    const unsigned mask = (unsigned)bytes - 1u - 15u;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Act { const char* name; unsigned long long bits; };
    const Act acts[] = {{"64 lanes", ~0ull}, {"first 32", 0xffffffffull}, {"first 16", 0xffffull}, {"first 4", 0xfull}, {"first 1", 1ull},
                        {"every 2nd lane (32)", 0x5555555555555555ull}, {"every 4th lane (16)", 0x1111111111111111ull},
                        {"every 16th lane (4)", 0x0001000100010001ull}, {"27 lanes, irregular", 0x9a6d35b2c94e5a31ull & 0x7bffffffdfffffffull}};
    struct Pat { const char* name; int lane_stride, step; };
    const Pat pats[] = {{"lanes 16 B apart", 1, 1}, {"lanes scattered", 1031, 517}};
    for (const Pat& P : pats)
        for (const Act& A : acts) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                k_lanes<<<blocks, 64>>>(tab, mask, A.bits, P.lane_stride, P.step, iters, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double per_cu = (double)blocks / cus * iters;
            printf("%-18s %-22s (%2d active) %8.3f ms  %6.1f cycles per wave-instruction per CU\n", P.name, A.name,
                   __builtin_popcountll(A.bits), ms, ms * 1e-3 * 2.4e9 / per_cu);
        }
    CHECK(hipDeviceSynchronize());
    return 0;
}
