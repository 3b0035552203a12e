// What the record layout of the radix sort's scatter is worth: the copy-out pattern of one pass (every block of 4,096
// elements leaves 256 runs, one per digit, at that digit's place in the output) with
//   (a) keys (8 B) and values (4 B) in two arrays: runs of 128 B and 64 B, 12 B read + 12 B written per element;
//   (b) one array of 16-byte records (key, value, pad): runs of 256 B, 16 B read + 16 B written per element;
//   (c) the histogram pass that goes with each (keys alone: 8 B per element; records: the key of every record, 16 B of lines).
// Digits are exactly uniform and already in digit order inside a block, so there is no ranking and no LDS: the kernels
// are the memory pattern alone, an upper bound for either layout.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_build/ubench_scatter tools/ubench_scatter.hip && tools/_build/ubench_scatter
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int TILE = 4096, THREADS = 256, DIGITS = 256, RUN = TILE / DIGITS;     // 16 elements per digit and block

struct Rec { unsigned long long key; unsigned val; unsigned pad; };

__device__ inline size_t dest_of(int block, int j, size_t per_digit) {
    return (size_t)(j / RUN) * per_digit + (size_t)block * RUN + (j % RUN);
}

__global__ __launch_bounds__(THREADS) void k_split(const unsigned long long* __restrict__ k_in, const unsigned* __restrict__ v_in,
                                                   unsigned long long* __restrict__ k_out, unsigned* __restrict__ v_out, size_t per_digit) {
    const size_t base = (size_t)blockIdx.x * TILE;
    unsigned long long k[TILE / THREADS];
    unsigned v[TILE / THREADS];
#pragma unroll
    for (int r = 0; r < TILE / THREADS; ++r) {
        k[r] = k_in[base + r * THREADS + threadIdx.x];
        v[r] = v_in[base + r * THREADS + threadIdx.x];
    }
#pragma unroll
    for (int r = 0; r < TILE / THREADS; ++r) {
        const size_t d = dest_of(blockIdx.x, r * THREADS + threadIdx.x, per_digit);
        k_out[d] = k[r];
        v_out[d] = v[r];
    }
}

__global__ __launch_bounds__(THREADS) void k_records(const Rec* __restrict__ in, Rec* __restrict__ out, size_t per_digit) {
    const size_t base = (size_t)blockIdx.x * TILE;
    uint4 x[TILE / THREADS];
#pragma unroll
    for (int r = 0; r < TILE / THREADS; ++r) x[r] = reinterpret_cast<const uint4*>(in)[base + r * THREADS + threadIdx.x];
#pragma unroll
    for (int r = 0; r < TILE / THREADS; ++r)
        reinterpret_cast<uint4*>(out)[dest_of(blockIdx.x, r * THREADS + threadIdx.x, per_digit)] = x[r];
}

__global__ __launch_bounds__(THREADS) void k_hist_keys(const unsigned long long* __restrict__ k_in, unsigned* __restrict__ counts) {
    const size_t base = (size_t)blockIdx.x * TILE;
    unsigned acc = 0;
#pragma unroll
    for (int r = 0; r < TILE / THREADS; ++r) acc += (unsigned)(k_in[base + r * THREADS + threadIdx.x] >> 8) & 255u;
    if (acc == 7u) counts[blockIdx.x] = acc;                   // keeps the loads alive (never true for the data used)
}

__global__ __launch_bounds__(THREADS) void k_hist_records(const Rec* __restrict__ in, unsigned* __restrict__ counts) {
    const size_t base = (size_t)blockIdx.x * TILE;
    unsigned acc = 0;
#pragma unroll
    for (int r = 0; r < TILE / THREADS; ++r) acc += (unsigned)(in[base + r * THREADS + threadIdx.x].key >> 8) & 255u;
    if (acc == 7u) counts[blockIdx.x] = acc;
}

// the sort's histogram kernel as it is: one LDS atomic per key, counts written digit-major (stride = number of blocks)
__global__ __launch_bounds__(THREADS) void k_hist_real(const unsigned long long* __restrict__ k_in, unsigned* __restrict__ hist, int nblk,
                                                       int digit_major) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * TILE;
#pragma unroll
    for (int r = 0; r < TILE / THREADS; ++r) atomicAdd(&h[(unsigned)(k_in[base + r * THREADS + threadIdx.x] >> 8) & 255u], 1u);
    __syncthreads();
    if (digit_major) hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
    else hist[(size_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

// G tiles per block, the block's G counts of a digit written next to each other
template <int G>
__global__ __launch_bounds__(THREADS) void k_hist_group(const unsigned long long* __restrict__ k_in, unsigned* __restrict__ hist, int nblk) {
    __shared__ unsigned h[G][256];
    for (int g = 0; g < G; ++g) h[g][threadIdx.x] = 0;
    __syncthreads();
    for (int g = 0; g < G; ++g) {
        const size_t base = ((size_t)blockIdx.x * G + g) * TILE;
#pragma unroll
        for (int r = 0; r < TILE / THREADS; ++r) {
            const unsigned d = (unsigned)(k_in[base + r * THREADS + threadIdx.x] >> 8) & 255u;
            const unsigned first = __builtin_amdgcn_readfirstlane(d);
            if (__all(d == first)) { if ((threadIdx.x & 63) == 0) atomicAdd(&h[g][d], 64u); }
            else atomicAdd(&h[g][d], 1u);
        }
    }
    __syncthreads();
    for (int g = 0; g < G; ++g) hist[(size_t)threadIdx.x * nblk + (size_t)blockIdx.x * G + g] = h[g][threadIdx.x];
}

__global__ void k_fill_keys(unsigned long long* k, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) k[i] = (i * 0x9E3779B97F4A7C15ull) >> 20;                     // digits of neighbouring keys differ
}

int main(int argc, char** argv) {
    const size_t blocks = argc > 1 ? (size_t)atol(argv[1]) : 12288;            // 50.3 M elements
    const size_t n = blocks * TILE, per_digit = n / DIGITS;
    unsigned long long *k0, *k1;
    unsigned *v0, *v1, *counts;
    Rec *r0, *r1;
    CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
    CK(hipMalloc(&r0, n * 16)); CK(hipMalloc(&r1, n * 16)); CK(hipMalloc(&counts, blocks * 4));
    CK(hipMemset(k0, 1, n * 8)); CK(hipMemset(v0, 1, n * 4)); CK(hipMemset(r0, 1, n * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    float ms[4] = {0, 0, 0, 0};
    for (int which = 0; which < 4; ++which) {
        for (int w = 0; w < 2; ++w) {                                         // second round is the timed one
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) {
                if (which == 0) hipLaunchKernelGGL(k_split, dim3(blocks), dim3(THREADS), 0, 0, k0, v0, k1, v1, per_digit);
                if (which == 1) hipLaunchKernelGGL(k_records, dim3(blocks), dim3(THREADS), 0, 0, r0, r1, per_digit);
                if (which == 2) hipLaunchKernelGGL(k_hist_keys, dim3(blocks), dim3(THREADS), 0, 0, k0, counts);
                if (which == 3) hipLaunchKernelGGL(k_hist_records, dim3(blocks), dim3(THREADS), 0, 0, r0, counts);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms[which], e0, e1));
        ms[which] /= reps;
    }
    unsigned* hist;
    CK(hipMalloc(&hist, blocks * 256 * 4));
    float ms_real[2][2];
    for (int data = 0; data < 2; ++data) {
        if (data == 1) hipLaunchKernelGGL(k_fill_keys, dim3((n + 255) / 256), dim3(256), 0, 0, k0, n);
        for (int major = 0; major < 2; ++major) {
            for (int w = 0; w < 2; ++w) {
                CK(hipEventRecord(e0, 0));
                for (int r = 0; r < reps; ++r)
                    hipLaunchKernelGGL(k_hist_real, dim3(blocks), dim3(THREADS), 0, 0, k0, hist, (int)blocks, major);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
            }
            CK(hipEventElapsedTime(&ms_real[data][major], e0, e1));
            ms_real[data][major] /= reps;
        }
    }
    float ms_g[2][2];
    for (int data = 0; data < 2; ++data) {
        if (data == 0) CK(hipMemset(k0, 1, n * 8));
        else hipLaunchKernelGGL(k_fill_keys, dim3((n + 255) / 256), dim3(256), 0, 0, k0, n);
        for (int G = 0; G < 2; ++G) {
            for (int w = 0; w < 2; ++w) {
                CK(hipEventRecord(e0, 0));
                for (int r = 0; r < reps; ++r) {
                    if (G == 0) hipLaunchKernelGGL(k_hist_group<4>, dim3(blocks / 4), dim3(THREADS), 0, 0, k0, hist, (int)blocks);
                    else hipLaunchKernelGGL(k_hist_group<8>, dim3(blocks / 8), dim3(THREADS), 0, 0, k0, hist, (int)blocks);
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
            }
            CK(hipEventElapsedTime(&ms_g[data][G], e0, e1));
            ms_g[data][G] /= reps;
        }
    }
    printf("histogram, 4 / 8 tiles per block, uniform-wave shortcut: equal digits %.3f / %.3f ms, mixed digits %.3f / %.3f ms\n",
           ms_g[0][0], ms_g[0][1], ms_g[1][0], ms_g[1][1]);
    printf("histogram with LDS atomics, equal digits:   block-major counts %.3f ms, digit-major %.3f ms\n", ms_real[0][0], ms_real[0][1]);
    printf("histogram with LDS atomics, mixed digits:   block-major counts %.3f ms, digit-major %.3f ms\n", ms_real[1][0], ms_real[1][1]);
    const double g = 1e-6 * n;                                               // bytes per ms -> GB/s needs 1e-6
    printf("%zu elements, %zu blocks of %d\n", n, blocks, TILE);
    printf("scatter, keys + values apart (128 B + 64 B runs): %.3f ms  (%.0f GB/s moved, 24 B per element)\n", ms[0], 24 * g / ms[0]);
    printf("scatter, 16-byte records (256 B runs):            %.3f ms  (%.0f GB/s moved, 32 B per element)\n", ms[1], 32 * g / ms[1]);
    printf("histogram read, keys alone:                       %.3f ms  (%.0f GB/s, 8 B per element)\n", ms[2], 8 * g / ms[2]);
    printf("histogram read, keys inside records:              %.3f ms  (%.0f GB/s of lines, 16 B per element)\n", ms[3], 16 * g / ms[3]);
    printf("one pass = histogram + scatter: apart %.3f ms, records %.3f ms\n", ms[0] + ms[2], ms[1] + ms[3]);
    return 0;
}
