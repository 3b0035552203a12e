// Ordered stream compaction: element i (in index order) with pred(i) true goes to slot
// rank(i) = number of earlier elements with pred true; emit(i, slot) writes it.  Three launches:
// count per 2048-element tile, single-block scan of the counts, ballot-ranked scatter.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcp {

constexpr int FR_THREADS = 256;
constexpr int FR_ITEMS = 8;
constexpr int FR_TILE = FR_THREADS * FR_ITEMS;

template <class Pred>
static __global__ void __launch_bounds__(FR_THREADS) k_count(const Pred pred, size_t n, unsigned* __restrict__ counts) {
    __shared__ unsigned total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * FR_TILE;
    unsigned mine = 0;
#pragma unroll
    for (int k = 0; k < FR_ITEMS; ++k) {
        const size_t i = base + (size_t)k * FR_THREADS + threadIdx.x;
        if (i < n && pred(i)) ++mine;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) mine += __shfl_down(mine, s, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

// Exclusive prefix of counts[0..nblk) in place; the grand total goes to counts[nblk].
static __global__ void __launch_bounds__(1024) k_scan_counts(unsigned* __restrict__ counts, int nblk) {
    __shared__ unsigned s[1024];
    __shared__ unsigned run;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned v = (i < nblk) ? counts[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned t = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblk) counts[i] = run + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) run += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[nblk] = run;
}

// Ordered scatter: element i (in index order) with pred(i) goes to slot
// offsets[block] + its rank inside the block; emit(i, slot) writes it.
template <class Pred, class Emit>
static __global__ void __launch_bounds__(FR_THREADS)
k_scatter(const Pred pred, const Emit emit, size_t n, const unsigned* __restrict__ offsets) {
    __shared__ unsigned wave_total[FR_THREADS / 64];
    __shared__ unsigned running;
    if (threadIdx.x == 0) running = offsets[blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * FR_TILE;
    for (int k = 0; k < FR_ITEMS; ++k) {
        const size_t i = base + (size_t)k * FR_THREADS + threadIdx.x;
        const bool keep = i < n && pred(i);
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_total[w] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = running, all = 0;
        for (int j = 0; j < FR_THREADS / 64; ++j) { if (j < w) before += wave_total[j]; all += wave_total[j]; }
        if (keep) emit(i, (size_t)before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull)));
        __syncthreads();
        if (threadIdx.x == 0) running += all;
        __syncthreads();
    }
}


// Scratch: compact_count_elems(n) uints; after the launches counts[ntiles] holds the number kept.
inline size_t compact_count_elems(size_t n) { return (n + FR_TILE - 1) / FR_TILE + 2; }
inline int compact_tiles(size_t n) { return (int)((n + FR_TILE - 1) / FR_TILE); }

template <class Pred, class Emit>
inline void compact_ordered(hipStream_t stream, const Pred& pred, const Emit& emit, size_t n, unsigned* counts) {
    const int nblk = compact_tiles(n);
    if (nblk == 0) return;
    hipLaunchKernelGGL(k_count<Pred>, dim3(nblk), dim3(FR_THREADS), 0, stream, pred, n, counts);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, stream, counts, nblk);
    hipLaunchKernelGGL((k_scatter<Pred, Emit>), dim3(nblk), dim3(FR_THREADS), 0, stream, pred, emit, n, counts);
}

}  // namespace pcp
