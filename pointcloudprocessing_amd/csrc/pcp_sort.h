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
#ifndef PCP_RS_ROUNDS
#define PCP_RS_ROUNDS 16
#endif
constexpr int RS_ROUNDS = PCP_RS_ROUNDS;
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;

static __device__ __forceinline__ void rs_hist_body(const unsigned long long* __restrict__ keys, size_t n, int shift,
                                                    unsigned* __restrict__ hist, int nblk) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_TILE / RS_THREADS; ++i) {
        const size_t j = base + (size_t)i * RS_THREADS + threadIdx.x;
        const bool valid = j < n;
        const unsigned d = valid ? (unsigned)(keys[j] >> shift) & 255u : 256u;
        // a wave whose keys all share the digit (the top digits of keys from one region, input that is already in
        // order) adds its 64 at once instead of through 64 atomics on one address
        if (__all(d == (unsigned)__builtin_amdgcn_readfirstlane((int)d))) {
            if (valid && (threadIdx.x & 63) == 0) atomicAdd(&h[d], 64u);
        } else if (valid) {
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];     // digit-major
}

static __global__ void __launch_bounds__(RS_THREADS)
k_rs_hist(const unsigned long long* __restrict__ keys, size_t n, int shift, unsigned* __restrict__ hist, int nblk) {
    rs_hist_body(keys, n, shift, hist, nblk);
}

// Inclusive prefix sum over the 256 threads of a block (4 waves): DPP-free shuffles inside a wave,
// the wave totals through LDS.  s4: 4 uints of LDS.  Every thread of the block calls it.
static __device__ __forceinline__ unsigned block_scan256(unsigned v, unsigned* s4, unsigned& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    __syncthreads();                       // s4 may still be read from an earlier call
    if (lane == 63) s4[w] = v;
    __syncthreads();
    const unsigned a = s4[0], b = s4[1], c = s4[2], d = s4[3];
    total = a + b + c + d;
    return v + (w > 0 ? a : 0u) + (w > 1 ? b : 0u) + (w > 2 ? c : 0u);
}

// Block d: exclusive prefix of hist[d][0..nblk) in place, digit total to totals[d].  Thread t owns
// a run of consecutive entries: it sums them, the 256 sums are scanned once, and the thread then
// rewrites its run.
static __global__ void __launch_bounds__(256) k_rs_scan(unsigned* __restrict__ hist, int nblk, unsigned* __restrict__ totals) {
    __shared__ unsigned s4[4];
    unsigned* h = hist + (size_t)blockIdx.x * nblk;
    const int per = (nblk + 255) / 256;
    const int lo = min((int)threadIdx.x * per, nblk), hi = min(lo + per, nblk);
    unsigned sum = 0;
    for (int i = lo; i < hi; ++i) sum += h[i];
    unsigned total;
    unsigned run = block_scan256(sum, s4, total) - sum;
    for (int i = lo; i < hi; ++i) {
        const unsigned v = h[i];
        h[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
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

// Each wave ranks its own 64 * RS_ROUNDS consecutive keys (rounds of 64, in order), so the rank of
// a key among the tile's keys of equal digit is: waves before + earlier rounds of this wave + lower
// lanes of this round.  The tile is then put in digit order in LDS and copied out from there: the
// keys of one digit leave as one run of consecutive addresses instead of one 8-byte write per key.
static __device__ __forceinline__ void
rs_scatter_body(const unsigned long long* __restrict__ keys_in, const unsigned* __restrict__ vals_in, size_t n, int shift,
                const unsigned goff,           // thread d: first place in the output of this tile's keys of digit d
                unsigned long long* __restrict__ keys_out, unsigned* __restrict__ vals_out) {
    __shared__ unsigned cnt[4][256];
    __shared__ unsigned tstart[256];       // first place of a digit in the sorted tile
    __shared__ unsigned delta[256];        // place in the tile -> place in the output, per digit (mod 2^32)
    __shared__ unsigned s4[4];
    __shared__ unsigned long long sk[RS_TILE];
    __shared__ unsigned sv[RS_TILE];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (int j = tid; j < 4 * 256; j += RS_THREADS) (&cnt[0][0])[j] = 0;
    __syncthreads();
    unsigned long long k[RS_ROUNDS];
    unsigned v[RS_ROUNDS], loc[RS_ROUNDS];
    const size_t tile0 = (size_t)blockIdx.x * RS_TILE;
    const size_t seg = tile0 + (size_t)w * (64 * RS_ROUNDS);
#pragma unroll
    for (int i = 0; i < RS_ROUNDS; ++i) {
        const size_t j = seg + (size_t)i * 64 + lane;
        const bool valid = j < n;
        k[i] = valid ? keys_in[j] : RS_EMPTY_KEY;
        v[i] = valid ? vals_in[j] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RS_ROUNDS; ++i) {
        const bool valid = seg + (size_t)i * 64 + lane < n;
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
        const unsigned c0 = cnt[0][tid], c1 = cnt[1][tid], c2 = cnt[2][tid], c3 = cnt[3][tid];
        cnt[0][tid] = 0;
        cnt[1][tid] = c0;
        cnt[2][tid] = c0 + c1;
        cnt[3][tid] = c0 + c1 + c2;
        const unsigned mine = c0 + c1 + c2 + c3;
        unsigned total;
        const unsigned first = block_scan256(mine, s4, total) - mine;
        tstart[tid] = first;
        delta[tid] = goff - first;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ROUNDS; ++i) {
        if (seg + (size_t)i * 64 + lane < n) {
            const unsigned digit = (unsigned)(k[i] >> shift) & 255u;
            const unsigned p = tstart[digit] + cnt[w][digit] + loc[i];
            sk[p] = k[i];
            sv[p] = v[i];
        }
    }
    __syncthreads();
    const unsigned in_tile = (unsigned)(n - tile0 < (size_t)RS_TILE ? n - tile0 : (size_t)RS_TILE);
#pragma unroll
    for (int i = 0; i < RS_ROUNDS; ++i) {
        const unsigned p = (unsigned)i * RS_THREADS + tid;
        if (p < in_tile) {
            const unsigned long long key = sk[p];
            const size_t pos = (size_t)(unsigned)(delta[(unsigned)(key >> shift) & 255u] + p);
            keys_out[pos] = key;
            vals_out[pos] = sv[p];
        }
    }
}

static __global__ void __launch_bounds__(RS_THREADS)
k_rs_scatter(const unsigned long long* __restrict__ keys_in, const unsigned* __restrict__ vals_in, size_t n, int shift,
             const unsigned* __restrict__ hist, const unsigned* __restrict__ base, int nblk,
             unsigned long long* __restrict__ keys_out, unsigned* __restrict__ vals_out) {
    rs_scatter_body(keys_in, vals_in, n, shift, base[threadIdx.x] + hist[(size_t)threadIdx.x * nblk + blockIdx.x], keys_out, vals_out);
}

// ---- the same sort driven from device memory ----------------------------------------------------------
// For callers that must not wait for the number of keys or of passes: both sit in device memory (RsCtl, written by an
// earlier kernel of the stream), the launches are sized for a capacity, tiles beyond the keys and passes beyond the
// needed ones leave at once.  Pass p reads buffer p & 1 and writes the other one, so the result is in buffer
// passes & 1.  Meant for at most a few hundred tiles: one block scans all 256 digit rows.
struct RsCtl { unsigned n; int passes; };
#ifndef PCP_RSD_ONE_KERNEL_TILES
#define PCP_RSD_ONE_KERNEL_TILES 64
#endif
constexpr int RSD_ONE_KERNEL_TILES = PCP_RSD_ONE_KERNEL_TILES;   // up to this many tiles a pass is one kernel (k_rsd_pass)

static __global__ void __launch_bounds__(RS_THREADS)
k_rsd_hist(const RsCtl* __restrict__ ctl, int pass, unsigned long long* const k0, unsigned long long* const k1,
           unsigned* __restrict__ hist, int nblk) {
    if (pass >= ctl->passes) return;
    rs_hist_body((pass & 1) ? k1 : k0, (size_t)ctl->n, 8 * pass, hist, nblk);
}

// Thread d: exclusive prefix of hist[d][0..nblk) in place; then the exclusive prefix of the 256 digit totals to base.
static __global__ void __launch_bounds__(256) k_rsd_scan(const RsCtl* __restrict__ ctl, int pass, unsigned* __restrict__ hist,
                                                          int nblk, unsigned* __restrict__ base) {
    __shared__ unsigned s4[4];
    if (pass >= ctl->passes) return;
    unsigned* h = hist + (size_t)threadIdx.x * nblk;
    const int used = min(nblk, (int)((ctl->n + RS_TILE - 1) / RS_TILE));
    unsigned run = 0;
    for (int i = 0; i < used; ++i) { const unsigned v = h[i]; h[i] = run; run += v; }
    unsigned total;
    base[threadIdx.x] = block_scan256(run, s4, total) - run;
}

static __global__ void __launch_bounds__(RS_THREADS)
k_rsd_scatter(const RsCtl* __restrict__ ctl, int pass, unsigned long long* const k0, unsigned long long* const k1,
              unsigned* const v0, unsigned* const v1, const unsigned* __restrict__ hist, const unsigned* __restrict__ base, int nblk) {
    if (pass >= ctl->passes) return;
    const size_t n = (size_t)ctl->n;
    if ((size_t)blockIdx.x * RS_TILE >= n) return;
    const bool odd = pass & 1;
    rs_scatter_body(odd ? k1 : k0, odd ? v1 : v0, n, 8 * pass, base[threadIdx.x] + hist[(size_t)threadIdx.x * nblk + blockIdx.x],
                    odd ? k0 : k1, odd ? v0 : v1);
}

// A whole pass as one kernel, for a few tiles (the unknown points of one sweep): every block counts the digits of ALL keys
// itself -- those of the tiles before its own and those from its own on, one LDS atomic per key, the keys come from L2 --
// which gives it what the histogram and scan kernels would: where each digit starts (sum over all tiles of the smaller
// digits) and how many keys of the digit the earlier tiles hold.  Same ranks, same places, one launch instead of three.
static __global__ void __launch_bounds__(RS_THREADS)
k_rsd_pass(const RsCtl* __restrict__ ctl, int pass, unsigned long long* const k0, unsigned long long* const k1,
           unsigned* const v0, unsigned* const v1) {
    __shared__ unsigned before[256], rest[256], s4[4];
    if (pass >= ctl->passes) return;
    const size_t n = (size_t)ctl->n;
    const size_t tile0 = (size_t)blockIdx.x * RS_TILE;
    if (tile0 >= n) return;
    const bool odd = pass & 1;
    const unsigned long long* __restrict__ kin = odd ? k1 : k0;
    const int shift = 8 * pass;
    before[threadIdx.x] = 0;
    rest[threadIdx.x] = 0;
    __syncthreads();
    for (size_t j0 = 0; j0 < n; j0 += 8 * RS_THREADS) {          // eight loads in flight per thread
        unsigned long long k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t j = j0 + (size_t)u * RS_THREADS + threadIdx.x;
            k[u] = j < n ? kin[j] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t j = j0 + (size_t)u * RS_THREADS + threadIdx.x;
            const unsigned d = (unsigned)(k[u] >> shift) & 255u;
            if (j < n) atomicAdd(j < tile0 ? &before[d] : &rest[d], 1u);
        }
    }
    __syncthreads();
    const unsigned mine = before[threadIdx.x] + rest[threadIdx.x];
    unsigned total;
    const unsigned start = block_scan256(mine, s4, total) - mine;
    rs_scatter_body(kin, odd ? v1 : v0, n, shift, start + before[threadIdx.x], odd ? k0 : k1, odd ? v0 : v1);
}

// Enqueues max_passes passes over up to n_cap keys.
inline void radix_sort_pairs_dev(hipStream_t stream, const RsCtl* ctl, unsigned long long* k0, unsigned long long* k1, unsigned* v0,
                                 unsigned* v1, size_t n_cap, int max_passes, unsigned* hist, unsigned* base) {
    const int nblk = (int)((n_cap + RS_TILE - 1) / RS_TILE);
    if (nblk == 0) return;
    if (nblk <= RSD_ONE_KERNEL_TILES) {
        for (int p = 0; p < max_passes; ++p)
            hipLaunchKernelGGL(k_rsd_pass, dim3(nblk), dim3(RS_THREADS), 0, stream, ctl, p, k0, k1, v0, v1);
        return;
    }
    for (int p = 0; p < max_passes; ++p) {
        hipLaunchKernelGGL(k_rsd_hist, dim3(nblk), dim3(RS_THREADS), 0, stream, ctl, p, k0, k1, hist, nblk);
        hipLaunchKernelGGL(k_rsd_scan, dim3(1), dim3(256), 0, stream, ctl, p, hist, nblk, base);
        hipLaunchKernelGGL(k_rsd_scatter, dim3(nblk), dim3(RS_THREADS), 0, stream, ctl, p, k0, k1, v0, v1, hist, base, nblk);
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

// The same for keys of which only the bits set in `varying` can differ between any two of them:
// an 8-bit digit in which no key differs leaves the order as it is, so its pass is skipped.  Returns
// the number of passes made.
inline int radix_sort_pairs_varying(hipStream_t stream, unsigned long long* keys0, unsigned long long* keys1,
                                    unsigned* vals0, unsigned* vals1, size_t n, unsigned long long varying, unsigned* hist,
                                    unsigned* totals, unsigned* base, unsigned long long** keys_out, unsigned** vals_out) {
    const int nblk = (int)((n + RS_TILE - 1) / RS_TILE);
    unsigned long long *kin = keys0, *kout = keys1;
    unsigned *vin = vals0, *vout = vals1;
    int passes = 0;
    for (int p = 0; p < 8 && n > 0; ++p) {
        const int shift = 8 * p;
        if (((varying >> shift) & 0xFFull) == 0) continue;
        hipLaunchKernelGGL(k_rs_hist, dim3(nblk), dim3(RS_THREADS), 0, stream, kin, n, shift, hist, nblk);
        hipLaunchKernelGGL(k_rs_scan, dim3(256), dim3(256), 0, stream, hist, nblk, totals);
        hipLaunchKernelGGL(k_rs_base, dim3(1), dim3(256), 0, stream, totals, base);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(RS_THREADS), 0, stream, kin, vin, n, shift, hist, base, nblk, kout, vout);
        std::swap(kin, kout);
        std::swap(vin, vout);
        ++passes;
    }
    *keys_out = kin;
    *vals_out = vin;
    return passes;
}

}  // namespace pcp
