// Micro-benchmark: what a gather instruction costs a CU by its WIDTH (4, 8, 12, 16 bytes per lane), and what a
// lane's window of 20 consecutive points costs when the points are 16-byte records (20 aligned 16-byte loads)
// or packed 12-byte records (15 unaligned 16-byte loads).  Table of 2 MB (L2 resident), 32 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_width ubench_width.hip && ./ubench_width
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define GL __attribute__((address_space(1)))

typedef float f1;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <class T> __device__ __forceinline__ float sum_of(const T v);
template <> __device__ __forceinline__ float sum_of<f1>(const f1 v) { return v; }
template <> __device__ __forceinline__ float sum_of<f2>(const f2 v) { return v.x + v.y; }
template <> __device__ __forceinline__ float sum_of<f3>(const f3 v) { return v.x + v.y + v.z; }
template <> __device__ __forceinline__ float sum_of<f4>(const f4 v) { return v.x + v.y + v.z + v.w; }

// lane l reads the T at byte offset ((start + l * lane_stride + i * step) * 16) & mask: the 16-byte record's first bytes
template <class T>
__global__ void __launch_bounds__(64) k_width(const char* tab, const unsigned mask, const int lane_stride, const int step,
                                              const int iters, float* out) {
    const char GL* p = (const char GL*)(const void*)tab;
    const int lane = threadIdx.x;
    const unsigned idx = blockIdx.x * 977u + lane * lane_stride;
    float acc = 0;
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned off = ((idx + (i + u) * step) << 4) & mask;
            acc += sum_of<T>(*(const T GL*)(p + off));
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

// a lane's window: `loads` consecutive 16-byte loads from byte offset base * stride_bytes
template <int LOADS, int STRIDE>
__global__ void __launch_bounds__(64) k_window(const char* tab, const unsigned mask, const int lane_stride, const int reps, float* out) {
    const char GL* p = (const char GL*)(const void*)tab;
    const int lane = threadIdx.x;
    float acc = 0;
    for (int r = 0; r < reps; ++r) {
        const unsigned first = (blockIdx.x * 977u + r * 131u + lane * lane_stride) & 0xffff;
        const unsigned off = (first * STRIDE) & mask;
        f4 v[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) v[i] = *(const f4 GL*)(p + off + 16 * i);
#pragma unroll
        for (int i = 0; i < LOADS; ++i) acc += v[i].x + v[i].w;
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
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 32 * 4;
    const unsigned mask = (unsigned)bytes - 1u - 15u;       // keeps offsets 16-byte aligned and inside the table
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto report = [&](const char* name, double instr_per_wave, float ms) {
        const double per_cu = (double)blocks / cus * instr_per_wave;
        printf("%-64s %8.3f ms  %7.1f cycles per wave-instruction per CU\n", name, ms, ms * 1e-3 * 2.4e9 / per_cu);
    };
    const int iters = 2048;
    struct Pat { const char* name; int lane_stride, step; };
    const Pat pats[] = {{"lanes 32 B apart, step 1", 2, 1}, {"lanes 16 B apart, step 1", 1, 1}, {"lanes scattered (1031), step 517", 1031, 517}};
    for (const Pat& P : pats) {
        for (int w = 0; w < 4; ++w) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (w == 0) k_width<f1><<<blocks, 64>>>(tab, mask, P.lane_stride, P.step, iters, out);
                if (w == 1) k_width<f2><<<blocks, 64>>>(tab, mask, P.lane_stride, P.step, iters, out);
                if (w == 2) k_width<f3><<<blocks, 64>>>(tab, mask, P.lane_stride, P.step, iters, out);
                if (w == 3) k_width<f4><<<blocks, 64>>>(tab, mask, P.lane_stride, P.step, iters, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            char nm[96]; snprintf(nm, 96, "%2d bytes per lane, %s", 4 * (w + 1), P.name);
            report(nm, iters, ms);
        }
    }
    for (int ls : {1, 3}) {
        float ms = 0;
        const int reps = 128;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            k_window<20, 16><<<blocks, 64>>>(tab, mask & ~0x1ffu, ls, reps, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        char nm[96]; snprintf(nm, 96, "window of 20 points, 16-byte records (20 loads), lanes %d points apart", ls);
        report(nm, reps, ms);
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            k_window<15, 12><<<blocks, 64>>>(tab, mask & ~0x1ffu, ls, reps, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        snprintf(nm, 96, "window of 20 points, packed 12-byte records (15 loads), lanes %d points apart", ls);
        report(nm, reps, ms);
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
