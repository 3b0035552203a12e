// Stable LSD radix sort of (uint64 key, uint32 value) pairs, 8 bits per pass, shared by the
// cell grid (Morton keys) and the voxel map (lexicographic voxel keys).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

namespace pcp {

constexpr unsigned long long RS_EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;

// LSD radix sort, 8 bits per pass, stable.  A block of 4 waves owns a tile of RS_TILE keys, RS_ROUNDS
// rounds of 64 per wave.  Larger tiles mean fewer per-block histograms to scan.
constexpr int RS_THREADS = 256;
constexpr int RS_ROUNDS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;

static __global__ void __launch_bounds__(RS_THREADS)
k_rs_hist(const unsigned long long* __restrict__ keys, size_t n, int shift, unsigned* __restrict__ hist, int nblk) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_TILE / RS_THREADS; ++i) {
        const size_t j = base + (size_t)i * RS_THREADS + threadIdx.x;
        if (j < n) atomicAdd(&h[(unsigned)(keys[j] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];     // digit-major
}

// Block d: exclusive prefix of hist[d][0..nblk) in place, digit total to totals[d].
static __global__ void __launch_bounds__(256) k_rs_scan(unsigned* __restrict__ hist, int nblk, unsigned* __restrict__ totals) {
    __shared__ unsigned s[256];
    unsigned* h = hist + (size_t)blockIdx.x * nblk;
    unsigned run = 0;
    for (int base = 0; base < nblk; base += 256) {
        const int i = base + threadIdx.x;
        const unsigned v = (i < nblk) ? h[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const unsigned t = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblk) h[i] = run + s[threadIdx.x] - v;
        run += s[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = run;
}

static __global__ void __launch_bounds__(256) k_rs_base(const unsigned* __restrict__ totals, unsigned* __restrict__ base) {
    __shared__ unsigned s[256];
    const unsigned v = totals[threadIdx.x];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned t = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0u;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    base[threadIdx.x] = s[threadIdx.x] - v;
}

// Each wave ranks its own 64 * RS_ROUNDS consecutive keys (rounds of 64, in order), so
// the rank of a key among equal digits is: digit base + blocks before + waves
// before + earlier rounds of this wave + lower lanes of this round.
static __global__ void __launch_bounds__(RS_THREADS)
k_rs_scatter(const unsigned long long* __restrict__ keys_in, const unsigned* __restrict__ vals_in, size_t n, int shift,
             const unsigned* __restrict__ hist, const unsigned* __restrict__ base, int nblk,
             unsigned long long* __restrict__ keys_out, unsigned* __restrict__ vals_out) {
    __shared__ unsigned cnt[4][256];
    __shared__ unsigned goff[256];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (int j = tid; j < 4 * 256; j += RS_THREADS) (&cnt[0][0])[j] = 0;
    goff[tid] = base[tid] + hist[(size_t)tid * nblk + blockIdx.x];
    __syncthreads();
    unsigned long long k[RS_ROUNDS];
    unsigned v[RS_ROUNDS], loc[RS_ROUNDS];
    const size_t seg = (size_t)blockIdx.x * RS_TILE + (size_t)w * (64 * RS_ROUNDS);
#pragma unroll
    for (int i = 0; i < RS_ROUNDS; ++i) {
        const size_t j = seg + (size_t)i * 64 + lane;
        const bool valid = j < n;
        k[i] = valid ? keys_in[j] : RS_EMPTY_KEY;
        v[i] = valid ? vals_in[j] : 0u;
        const unsigned digit = (unsigned)(k[i] >> shift) & 255u;
        unsigned long long mask = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long bm = __ballot(valid && bit);
            mask &= bit ? bm : ~bm;
        }
        int leader = lane;
        unsigned rank = 0, old = 0;
        if (valid) {
            rank = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
            leader = __ffsll((long long)mask) - 1;
            if (lane == leader) old = atomicAdd(&cnt[w][digit], (unsigned)__popcll(mask));
        }
        old = __shfl(old, leader, 64);
        loc[i] = old + rank;
    }
    __syncthreads();
    {
        const unsigned c0 = cnt[0][tid], c1 = cnt[1][tid], c2 = cnt[2][tid];
        cnt[0][tid] = 0;
        cnt[1][tid] = c0;
        cnt[2][tid] = c0 + c1;
        cnt[3][tid] = c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ROUNDS; ++i) {
        const size_t j = seg + (size_t)i * 64 + lane;
        if (j < n) {
            const unsigned digit = (unsigned)(k[i] >> shift) & 255u;
            const size_t pos = (size_t)goff[digit] + cnt[w][digit] + loc[i];
            keys_out[pos] = k[i];
            vals_out[pos] = v[i];
        }
    }
}


// Scratch a sort of n pairs needs besides the two key and two value buffers.
inline size_t radix_hist_elems(size_t n) { return (size_t)256 * ((n + RS_TILE - 1) / RS_TILE); }

// Sorts by the low `bits` bits of the keys.  keys0/vals0 hold the input; the result ends in
// (*keys_out, *vals_out), which is either pair of buffers.  hist: radix_hist_elems(n) uints,
// totals and base: 256 uints each.  Only enqueues work on `stream`.
inline void radix_sort_pairs(hipStream_t stream, unsigned long long* keys0, unsigned long long* keys1, unsigned* vals0,
                             unsigned* vals1, size_t n, int bits, unsigned* hist, unsigned* totals, unsigned* base,
                             unsigned long long** keys_out, unsigned** vals_out) {
    const int nblk = (int)((n + RS_TILE - 1) / RS_TILE);
    const int passes = (bits + 7) / 8;
    unsigned long long *kin = keys0, *kout = keys1;
    unsigned *vin = vals0, *vout = vals1;
    for (int p = 0; p < passes && n > 0; ++p) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(k_rs_hist, dim3(nblk), dim3(RS_THREADS), 0, stream, kin, n, shift, hist, nblk);
        hipLaunchKernelGGL(k_rs_scan, dim3(256), dim3(256), 0, stream, hist, nblk, totals);
        hipLaunchKernelGGL(k_rs_base, dim3(1), dim3(256), 0, stream, totals, base);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(RS_THREADS), 0, stream, kin, vin, n, shift, hist, base, nblk, kout, vout);
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    *keys_out = kin;
    *vals_out = vin;
}

}  // namespace pcp
