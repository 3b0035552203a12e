// Registration kernels and their C ABI (see include/pcp.h).
//
// Data layout in device memory, per context:
//   a, ua, normals  float4[n_a]   target cloud, its unit-sphere projection, ring normals
//   b32             float4[n_b]   source cloud as the kernels see it (stepwise calls)
//   b_in, b64       double[3*n_b] source cloud of the device-resident loop as uploaded, and
//                                 its working copy (written by iteration 0, then moved in place)
//   hint            float[n_b]    corr[:,3] of the reference: ring of the last match, -1 = none
//   partials        float[ceil(n_b/64)][32]  level-1 sums of 64 consecutive points
//                                 (27 used: 21 upper-triangle JtJ + 6 Jte)
//   lvl[2]          ping-pong buffers of the upper tree levels
//   state           IcpState      step to apply next, iteration counters, flags
// The 8x8 per-point tiles of the reference (Hs, 256 B per point) are only
// materialised on request (pcp_ls_tiles) -- the fused kernels reduce in registers.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "../../include/pcp.h"
#include "pcp_compact.h"
#include "pcp_ctx.h"
#include "pcp_device.h"
#include "pcp_solve.h"
#include "pcp_host.h"

using namespace pcp;

namespace {

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

__global__ void k_unit(const float4* __restrict__ a, float4* __restrict__ ua, int n, float4 o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ua[i] = unit_sphere(a[i], o);
}

__global__ void k_fill(float* __restrict__ p, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// LDS image of the ring table: Z float4 elevation bands, then Z int2 (begin, end).
__device__ __forceinline__ void load_rings(char* smem, const int2* __restrict__ scan,
                                           const float4* __restrict__ stat, int Z,
                                           const int2*& s_scan, const float4*& s_stat) {
    float4* st = reinterpret_cast<float4*>(smem);
    int2* sc = reinterpret_cast<int2*>(smem + (size_t)Z * sizeof(float4));
    for (int i = threadIdx.x; i < Z; i += blockDim.x) {
        sc[i] = scan[i];
        st[i] = stat[i];
    }
    __syncthreads();
    s_scan = sc;
    s_stat = st;
}

// Order-preserving map between floats and ints, for integer atomicMin / atomicMax on LDS.
__device__ __forceinline__ int float_key(const float f) {
    const int i = __float_as_int(f);
    return i ^ ((i >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key_float(const int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

// Elevation band of every ring: (sin, cos) of the lowest and of the highest
// unit-vector elevation among its points, for the whole ring (stat[r]) and, per
// azimuth bin b, for the points in bins b-1, b, b+1 (bins[b][r]; cos < 0 = none).
// One wave per ring.
__device__ __forceinline__ void ring_stats_body(const float4* __restrict__ a, const float4 o, const float4* __restrict__ ua,
                                                const int2* __restrict__ scan, int Z, float4* __restrict__ stat,
                                                float4* __restrict__ bins, const int r) {
    __shared__ int blo[PCP_AZ_BINS], bhi[PCP_AZ_BINS], rlo[PCP_AZ_BINS], rhi[PCP_AZ_BINS];
    if (r >= Z) return;
    const int2 se = scan[r];
    for (int b = threadIdx.x; b < PCP_AZ_BINS; b += 64) {
        blo[b] = float_key(2.0f); bhi[b] = float_key(-2.0f);
        rlo[b] = float_key(PCP_BIG); rhi[b] = float_key(0.0f);
    }
    __syncthreads();
    float lo = 2.0f, hi = -2.0f;
    bool bad = false;
    for (int p = se.x + (int)threadIdx.x; p <= se.y; p += 64) {
        const float4 u = ua[p];
        if (u.y != u.y || u.x != u.x || u.z != u.z) { bad = true; continue; }
        lo = fminf(lo, u.y);
        hi = fmaxf(hi, u.y);
        const int b = az_bin(u.x, u.z);
        atomicMin(&blo[b], float_key(u.y));
        atomicMax(&bhi[b], float_key(u.y));
        // the point's range from the sensor, by the expressions a query's range is computed with (find_closest)
        const float4 pa = a[p];
        const float4 v = make_float4(pa.x - o.x, (pa.y - o.y) - PCP_LIDAR_OFF_Y, pa.z - o.z, pa.w - o.w);
        const float rho = sqrtf(dot4(v, v));
        if (rho == rho) {
            atomicMin(&rlo[b], float_key(rho));
            atomicMax(&rhi[b], float_key(rho));
        } else {
            bad = true;
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, s, 64));
        hi = fmaxf(hi, __shfl_xor(hi, s, 64));
    }
    bad = __any(bad);
    __syncthreads();
    const float nan = __int_as_float(0x7fc00000);
    const float clo = (float)sqrt(fmax(0.0, 1.0 - (double)lo * (double)lo));
    const float chi = (float)sqrt(fmax(0.0, 1.0 - (double)hi * (double)hi));
    const bool unusable = bad || !(clo > 0.05f) || !(chi > 0.05f) || lo > hi;
    if (threadIdx.x == 0) stat[r] = unusable ? make_float4(nan, nan, nan, nan) : make_float4(lo, clo, hi, chi);
    for (int b = threadIdx.x; b < PCP_AZ_BINS; b += 64) {
        const int l = (b + PCP_AZ_BINS - 1) % PCP_AZ_BINS, h = (b + 1) % PCP_AZ_BINS;
        const float mlo = key_float(min(blo[b], min(blo[l], blo[h])));
        const float mhi = key_float(max(bhi[b], max(bhi[l], bhi[h])));
        float4 out;
        if (unusable) out = make_float4(nan, nan, nan, nan);
        else if (mlo > mhi) out = make_float4(0.0f, -1.0f, 0.0f, -1.0f);          // no point in the three bins
        else out = make_float4(mlo, (float)sqrt(fmax(0.0, 1.0 - (double)mlo * (double)mlo)),
                               mhi, (float)sqrt(fmax(0.0, 1.0 - (double)mhi * (double)mhi)));
        bins[(size_t)b * Z + r] = out;
        // second half of the table: the range band of the same points (lowest, highest); (0, BIG) rules nothing out
        const float qlo = key_float(min(rlo[b], min(rlo[l], rlo[h])));
        const float qhi = key_float(max(rhi[b], max(rhi[l], rhi[h])));
        const bool none = unusable || mlo > mhi || !(qlo <= qhi);
        bins[(size_t)(PCP_AZ_BINS + b) * Z + r] = none ? make_float4(0.0f, PCP_BIG, 0.0f, 0.0f) : make_float4(qlo, qhi, 0.0f, 0.0f);
    }
}

__global__ void __launch_bounds__(64)
k_ring_stats(const float4* __restrict__ a, const float4 o, const float4* __restrict__ ua, const int2* __restrict__ scan, int Z,
             float4* __restrict__ stat, float4* __restrict__ bins) {
    ring_stats_body(a, o, ua, scan, Z, stat, bins, blockIdx.x);
}

// Prepare-phase kernels for a batch: blockIdx.y / blockIdx.z selects the pair.
struct PrepArgs {
    const float4* a; float4* ua; float4* normals; const int2* scan; float4* stat; float4* bins; float* hint;
    int n_a, Z, max_points; float4 o;
};

__global__ void k_unit_batch(const PrepArgs* __restrict__ all) {
    const PrepArgs P = all[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n_a) P.ua[i] = unit_sphere(P.a[i], P.o);
    if (i < P.max_points) P.hint[i] = -1.0f;                  // setInitialScanGuess
}

__global__ void __launch_bounds__(64) k_ring_stats_batch(const PrepArgs* __restrict__ all) {
    const PrepArgs P = all[blockIdx.y];
    ring_stats_body(P.a, P.o, P.ua, P.scan, P.Z, P.stat, P.bins, blockIdx.x);
}

// NearestNeighbourScan.glsl:1-35.  One wave per block = one shader workgroup.
__global__ void __launch_bounds__(PCP_BLOCK)
k_nn(const float4* __restrict__ a, const float4* __restrict__ ua, const int2* __restrict__ scan,
     const float4* __restrict__ stat, const float4* __restrict__ bins, int n_a, int Z, int Zs, float4 o,
     const float4* __restrict__ b, int n_b, float* __restrict__ hint, uint32_t* __restrict__ idx,
     float* __restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int2* s_scan;
    const float4* s_stat;
    load_rings(smem, scan, stat, Z, s_scan, s_stat);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n_b;
    const Target T{as_global(a), as_global(ua), s_scan, s_stat, as_global(bins), n_a, Z, Zs, o};
    const float4 q = in ? b[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float h = in ? hint[i] : -1.0f;
    if (!in) return;
    const int nn = find_closest(T, q, h);
    hint[i] = h;
    idx[i] = (uint32_t)nn;
    dist[i] = distsq(q, a[nn]);
}

// Which (pair, block) a workgroup of a batched launch works on.  Workgroups are dealt to the eight
// XCDs round-robin by their linear index, and every XCD has an L2 of its own: with the plain
// (blockIdx.y, blockIdx.x) reading all XCDs walk through the same pairs together and each L2 sees the
// targets of every pair in flight.  When the number of pairs is a multiple of 8 the linear index is
// re-read so that pair p runs on the XCD p mod 8 gets, one pair after the other: an L2 then serves one
// or two targets at a time.  A pure renumbering: every (pair, block) is still taken exactly once.
__device__ __forceinline__ void pair_and_block(unsigned& pair, unsigned& block) {
    pair = blockIdx.y;
    block = blockIdx.x;
    if ((gridDim.y & 7u) == 0u) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned slot = lin >> 3;
        const unsigned local = slot / gridDim.x;
        block = slot - local * gridDim.x;
        pair = (lin & 7u) + 8u * local;
    }
}

// NormalShader.glsl:73-101
template <int CHECK>
__device__ __forceinline__ int closest_distinct(const Target& T, const float4 q, const float4 ref,
                                                int pid, int ring) {
    const int2 se = T.scan[ring];
    int best;
    float d;
    ring_window<CHECK, true>(T, q, ref, se.x, se.y, pid, best, d);
    return best;
}

__device__ __forceinline__ void normals_body(const float4* __restrict__ a, const float4* __restrict__ ua,
                                             const int2* __restrict__ scan, const float4* __restrict__ stat,
                                             int n_a, int Z, float4 o, float4* __restrict__ normals, char* smem,
                                             const unsigned block) {
    const int2* s_scan;
    const float4* s_stat;
    load_rings(smem, scan, stat, Z, s_scan, s_stat);
    const int i = block * blockDim.x + threadIdx.x;
    if (i >= n_a) return;
    const Target T{as_global(a), as_global(ua), s_scan, s_stat, nullptr, n_a, Z, Z, o};
    // getScanLineOfPoint, NormalShader.glsl:1-10: first ring that contains the point
    int ring = -1;
    for (int r = 0; r < Z; ++r) {
        const int2 se = s_scan[r];
        if (se.x <= i && i <= se.y) { ring = r; break; }
    }
    if (ring < 0) {  // the shader falls off the end of the function here (undefined)
        normals[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 q = a[i];
    const float4 ref = ua[i];   // == projectSpherePoint(points_a[i])
    // own ring, ring above, ring below: three searches for one query, run together (ring_windows).  An edge ring
    // has one neighbour only (NormalShader.glsl:36-60); the missing one is not searched.
    const bool has_above = ring > 0, has_below = ring < Z - 1;
    const int2 se[3] = {s_scan[ring], s_scan[has_above ? ring - 1 : ring], s_scan[has_below ? ring + 1 : ring]};
    const bool live[3] = {true, has_above, has_below};
    int found[3];
    float dist[3];
    ring_windows<3, 5, true>(T, q, ref, se, live, i, found, dist);
    const int on_line = found[0];
    int nxt;
    if (!has_below) nxt = found[1];
    else if (!has_above) nxt = found[2];
    else nxt = (distsq(q, a[found[2]]) < distsq(q, a[found[1]])) ? found[2] : found[1];
    const float4 pl = a[on_line], pn = a[nxt];
    const float v1x = pl.x - q.x, v1y = pl.y - q.y, v1z = pl.z - q.z;
    const float v2x = pn.x - q.x, v2y = pn.y - q.y, v2z = pn.z - q.z;
    const float cx = v1y * v2z - v2y * v1z;
    const float cy = v1z * v2x - v2z * v1x;
    const float cz = v1x * v2y - v2x * v1y;
    const float cl = sqrtf(dot3(cx, cy, cz, cx, cy, cz));
    float nx = cx / cl, ny = cy / cl, nz = cz / cl;
    // normalTowardsOrigin, NormalShader.glsl:62-71
    float tx = o.x - q.x, ty = o.y - q.y, tz = o.z - q.z;
    const float tl = sqrtf(dot3(tx, ty, tz, tx, ty, tz));
    tx = tx / tl; ty = ty / tl; tz = tz / tl;
    const float sgn = (dot3(nx, ny, nz, tx, ty, tz) > dot3(-nx, -ny, -nz, tx, ty, tz)) ? 1.0f : -1.0f;
    normals[i] = make_float4(nx * sgn, ny * sgn, nz * sgn, 0.0f);
}

__global__ void __launch_bounds__(PCP_BLOCK)
k_normals(const float4* __restrict__ a, const float4* __restrict__ ua, const int2* __restrict__ scan,
          const float4* __restrict__ stat, int n_a, int Z, float4 o, float4* __restrict__ normals) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    normals_body(a, ua, scan, stat, n_a, Z, o, normals, smem, blockIdx.x);
}

// Resident waves per SIMD of the batched normals: with the three searches of a point in flight together the kernel
// wants 68 registers; at 8 waves (64 registers) five of them spill.  Measured per launch of 64 targets: 8 waves 542 us,
// 7 waves 506 us, 6 waves 534 us.
#ifndef PCP_NORMALS_WAVES
#define PCP_NORMALS_WAVES 7
#endif
__global__ void __launch_bounds__(PCP_BLOCK) __attribute__((amdgpu_waves_per_eu(PCP_NORMALS_WAVES, PCP_NORMALS_WAVES))) k_normals_batch(const PrepArgs* __restrict__ all) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned pair, block;
    pair_and_block(pair, block);
    const PrepArgs P = all[pair];
    if ((int)(block * PCP_BLOCK) >= P.n_a) return;
    normals_body(P.a, P.ua, P.scan, P.stat, P.n_a, P.Z, P.o, P.normals, smem, block);
}

// LeastSquaresPoint.glsl / LeastSquaresPlane.glsl fused with the first level of
// ParalellHBSum.glsl: one wave = the 64 consecutive points of one shader
// workgroup; its tree sum goes to partials[wave].
//   F64SRC: source points are the float64 cloud of the device loop; the step
//           left pending by the previous iteration is applied first
//           (apply_iteration_of_ls, PointcloudIcpContainer.py:172-174).
struct LsArgs {
    const float4* a; const float4* ua; const int2* scan; const float4* stat; const float4* bins;
    int n_a, Z, Zs; float4 o;
    const float4* normals; const float4* b32; const double* b_in; double* b64; IcpState* st;
    int n_b, debug;
    float* hint; float* partials; float* tiles;
    int skip_store_iter;      // the dispatch (st->iter) whose moved points nobody will read: not written back (-1 = none)
};

template <int MODE, bool F64SRC, int POOL = 0>
__device__ __forceinline__ void ls_body(const LsArgs& A, char* smem, const unsigned block) {
    const float4* __restrict__ a = A.a;
    const float4* __restrict__ normals = A.normals;
    const double* __restrict__ b_in = A.b_in;
    double* __restrict__ b64 = A.b64;
    IcpState* __restrict__ st = A.st;
    float* __restrict__ hint = A.hint;
    float* __restrict__ partials = A.partials;
    float* __restrict__ tiles = A.tiles;
    const int n_b = A.n_b, debug = A.debug;
    if (F64SRC) {
        if (st->done) return;
    }
    const int2* s_scan;
    const float4* s_stat;
    load_rings(smem, A.scan, A.stat, A.Z, s_scan, s_stat);
    const int i = block * blockDim.x + threadIdx.x;
    const bool in = i < n_b;
    float v[PCP_NVAL];
#pragma unroll
    for (int k = 0; k < PCP_NVAL; ++k) v[k] = 0.0f;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) {
        if (F64SRC) {
            const bool first = st->iter == 0;       // iteration 0 reads the cloud as uploaded
            const double* rd = first ? b_in : b64;
            double px = rd[3 * (size_t)i + 0], py = rd[3 * (size_t)i + 1], pz = rd[3 * (size_t)i + 2];
            if (st->pending) {
                const double nx = ((st->R[0] * px + st->R[1] * py) + st->R[2] * pz) - st->t[0];
                const double ny = ((st->R[3] * px + st->R[4] * py) + st->R[5] * pz) - st->t[1];
                const double nz = ((st->R[6] * px + st->R[7] * py) + st->R[8] * pz) - st->t[2];
                px = nx; py = ny; pz = nz;
            }
            if ((st->pending || first) && st->iter != A.skip_store_iter) {
                b64[3 * (size_t)i + 0] = px; b64[3 * (size_t)i + 1] = py; b64[3 * (size_t)i + 2] = pz;
            }
            q = make_float4((float)px, (float)py, (float)pz, 0.0f);
        } else {
            q = A.b32[i];
        }
    }
    const Target T{as_global(a), as_global(A.ua), s_scan, s_stat, as_global(A.bins), A.n_a, A.Z, A.Zs, A.o};
    int nn = in ? i : 0;      // debug_info.x >= 1: identity correspondences, LeastSquaresPoint.glsl:78-81
    float d = 0.0f;
    if (!debug) {
        float h = in ? hint[i] : -1.0f;
        bool pooled = false;
        if (POOL == 2) {
            // hinted launches: the block shares the ring searches behind every query's hinted ring
            __shared__ BlockPool block_pool;
            if (T.Z <= 256) {              // uniform: every thread of the block takes part
                const bool hinted = in && h != -1.0f;
                const int found = find_closest_block(T, q, hinted, h, block_pool);
                if (hinted) { nn = found; pooled = true; }
            }
        }
        if (POOL == 1) {
            // the wave's ring searches behind the first one run as dense passes
            __shared__ unsigned long long pool_slot[64];
            __shared__ unsigned short pool_items[128];
            if (T.Z <= 256) {
                pooled = true;
                nn = find_closest_pooled(T, q, in, h, pool_slot, pool_items);
                if (!in) nn = 0;
            }
        }
        if (in) {
            if (!pooled) nn = find_closest(T, q, h);
            hint[i] = h;
            d = distsq(q, ld4(T.a, nn));
        }
    }
    if (in) {
        const float4 pa = ld4(T.a, nn);
        if (MODE == PCP_MODE_POINT) ls_point_values(pa, q, d, v);
        else ls_plane_values(pa, q, ld4(as_global(normals), nn), d, v);
        if (tiles) {
            float* t = tiles + (size_t)i * 64;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = 0; c < 6; ++c) t[r * 8 + c] = v[r <= c ? tri(r, c) : tri(c, r)];
                t[r * 8 + 6] = v[21 + r];
                t[r * 8 + 7] = 0.0f;
            }
#pragma unroll
            for (int k = 48; k < 64; ++k) t[k] = 0.0f;
        }
    }
    const int lane = threadIdx.x & 63;
    const int wave = i >> 6;               // == shader workgroup id
    if ((wave << 6) < n_b) {               // wave-uniform
        const float mine = wave_tree_sum_all(v, lane);      // lane k: value k of the wave's tree sum
        if (lane < PCP_PSTRIDE) partials[(size_t)wave * PCP_PSTRIDE + lane] = mine;
    }
}

template <int MODE, bool F64SRC>
__global__ void __launch_bounds__(PCP_BLOCK) k_ls(const LsArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ls_body<MODE, F64SRC>(A, smem, blockIdx.x);
}

// ls_body<MODE, true> for a single pair in flight.  Such a launch is two waves per SIMD, each walking
// 2.4 ring passes of 14 dependent memory round trips: its time is that chain, not the work.  Here a
// block of four waves serves the 64 points of one shader workgroup: wave w searches ring from + w of
// every point's four hinted rings (so a wave's gathers still go to one stretch of one ring), the four
// results meet in LDS, and the first wave merges them in ring order, computes the tile values of the 64
// points and their tree sum exactly as ls_body does.  Points whose hint is still -1 are searched by
// their lane of the first wave over all rings.  Same results; the batched launches keep ls_body (they
// fill the GPU, and the unpruned rings would be wasted work there).
template <int MODE, bool F64SRC>
__device__ __forceinline__ void ls_quad_body(const LsArgs& A, char* smem, const unsigned block) {
    __shared__ float s_d[4][64];
    __shared__ int s_glob[4][64], s_ring[4][64];
    IcpState* __restrict__ st = A.st;
    if (F64SRC) {
        if (st->done) return;
    }
    const int2* s_scan;
    const float4* s_stat;
    load_rings(smem, A.scan, A.stat, A.Z, s_scan, s_stat);
    const int t = threadIdx.x, qi = t & 63, sub = t >> 6;
    const int i = block * 64 + qi;
    const bool in = i < A.n_b;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    double px = 0.0, py = 0.0, pz = 0.0;
    bool moved = false;
    if (in && F64SRC) {
        const bool first = st->iter == 0;       // iteration 0 reads the cloud as uploaded
        const double* rd = first ? A.b_in : A.b64;
        px = rd[3 * (size_t)i + 0]; py = rd[3 * (size_t)i + 1]; pz = rd[3 * (size_t)i + 2];
        if (st->pending) {
            const double nx = ((st->R[0] * px + st->R[1] * py) + st->R[2] * pz) - st->t[0];
            const double ny = ((st->R[3] * px + st->R[4] * py) + st->R[5] * pz) - st->t[1];
            const double nz = ((st->R[6] * px + st->R[7] * py) + st->R[8] * pz) - st->t[2];
            px = nx; py = ny; pz = nz;
        }
        moved = (st->pending || first) && st->iter != A.skip_store_iter;
        q = make_float4((float)px, (float)py, (float)pz, 0.0f);
    } else if (in) {
        q = A.b32[i];
    }
    const Target T{as_global(A.a), as_global(A.ua), s_scan, s_stat, as_global(A.bins), A.n_a, A.Z, A.Zs, A.o};
    float h = in ? A.hint[i] : -1.0f;
    const bool quad = in && h != -1.0f;
    float mind;
    int glob, ring;
    hinted_ring(T, quad, q, sub, h, mind, glob, ring);
    s_d[sub][qi] = mind;
    s_glob[sub][qi] = glob;
    s_ring[sub][qi] = ring;
    __syncthreads();
    if (sub != 0) return;
    float v[PCP_NVAL];
#pragma unroll
    for (int k = 0; k < PCP_NVAL; ++k) v[k] = 0.0f;
    if (in) {
        // all four waves have read the point by the barrier above: only now is the moved one stored
        if (moved) { A.b64[3 * (size_t)i + 0] = px; A.b64[3 * (size_t)i + 1] = py; A.b64[3 * (size_t)i + 2] = pz; }
        int nn = 0;
        if (quad) {
            float best = PCP_BIG;
            int won = -1;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float d = s_d[w][qi];
                if (s_ring[w][qi] >= 0 && d < best) { best = d; nn = s_glob[w][qi]; won = s_ring[w][qi]; }
            }
            if (won >= 0) h = (float)won;
        } else {
            nn = find_closest(T, q, h);
        }
        A.hint[i] = h;
        const float4 pa = ld4(T.a, nn);
        const float d = distsq(q, pa);
        if (MODE == PCP_MODE_POINT) ls_point_values(pa, q, d, v);
        else ls_plane_values(pa, q, ld4(as_global(A.normals), nn), d, v);
    }
    const float mine = wave_tree_sum_all(v, qi);      // lane k: value k of the wave's tree sum
    if (qi < PCP_PSTRIDE) A.partials[(size_t)block * PCP_PSTRIDE + qi] = mine;
}

#ifndef PCP_LS_QUAD_WAVES
#define PCP_LS_QUAD_WAVES 8
#endif
template <int MODE, bool F64SRC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PCP_LS_QUAD_WAVES, PCP_LS_QUAD_WAVES))) k_ls_quad(const LsArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ls_quad_body<MODE, F64SRC>(A, smem, blockIdx.x);
}

// The same with the descriptor in device memory: a loop whose target (ring table, origin, sizes) was made by
// earlier kernels of the same stream, so that the host never learns them.  Launched for a capacity; blocks
// beyond the source's size leave at once.
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PCP_LS_QUAD_WAVES, PCP_LS_QUAD_WAVES)))
k_ls_quad_ind(const LsArgs* __restrict__ desc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LsArgs A = *desc;
    if ((int)(blockIdx.x * 64) >= A.n_b) return;
    ls_quad_body<MODE, true>(A, smem, blockIdx.x);
}

// Iteration 0 of the device loop on a freshly prepared pair searches all rings (hint -1), and with one
// pair in flight that launch is a couple of waves per SIMD whose slowest lanes walk 60 rings one after
// the other: a third to a half of a whole registration.  This pre-pass finds the ring that search ends on
// with K lanes per query and leaves it in the hint; the iteration's own search then covers the four
// rings around it, which hold the all-rings winner (the point of smallest distance on the lowest such
// ring is also the first of any subset of rings that contains its ring), so every result is unchanged.
template <int K, bool F64SRC>
__device__ __forceinline__ void first_rings_body(const LsArgs& A, char* smem, const unsigned block) {
    const int2* s_scan;
    const float4* s_stat;
    load_rings(smem, A.scan, A.stat, A.Z, s_scan, s_stat);
    const unsigned t = block * blockDim.x + threadIdx.x;
    const int i = (int)(t / K), sub = (int)(t % K);
    bool live = i < A.n_b;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        live = A.hint[i] == -1.0f;
        if (F64SRC) q = make_float4((float)A.b_in[3 * (size_t)i + 0], (float)A.b_in[3 * (size_t)i + 1], (float)A.b_in[3 * (size_t)i + 2], 0.0f);
        else q = A.b32[i];
    }
    const Target T{as_global(A.a), as_global(A.ua), s_scan, s_stat, as_global(A.bins), A.n_a, A.Z, A.Zs, A.o};
    const int ring = first_ring<K>(T, live, q, sub);
    if (live && sub == 0 && ring >= 0) A.hint[i] = (float)ring;
}

#ifndef PCP_FIRST_RINGS_WAVES
#define PCP_FIRST_RINGS_WAVES 8
#endif
template <int K, bool F64SRC>
__global__ void __launch_bounds__(PCP_BLOCK) __attribute__((amdgpu_waves_per_eu(PCP_FIRST_RINGS_WAVES, PCP_FIRST_RINGS_WAVES)))
k_first_rings(const LsArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    first_rings_body<K, F64SRC>(A, smem, blockIdx.x);
}

template <int K>
__global__ void __launch_bounds__(PCP_BLOCK) __attribute__((amdgpu_waves_per_eu(PCP_FIRST_RINGS_WAVES, PCP_FIRST_RINGS_WAVES)))
k_first_rings_ind(const LsArgs* __restrict__ desc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LsArgs A = *desc;
    if ((int)((blockIdx.x * PCP_BLOCK) / K) >= A.n_b || A.st->done) return;
    first_rings_body<K, true>(A, smem, blockIdx.x);
}

// The same for a batch of independent pairs.
#ifndef PCP_LS_WAVES
#define PCP_LS_WAVES 8      // resident waves per SIMD the register budget is set for
#endif
#ifndef PCP_LS_BLOCK_POOL
#define PCP_LS_BLOCK_POOL 1 // hinted launches: the ring searches behind the hinted ring as a list the block shares (find_closest_block);
                            // 0 with -DPCP_LS_THREADS=64 restores one wave per block walking its rings in lock step
#endif
template <int MODE>
__global__ void __launch_bounds__(PCP_LS_THREADS) __attribute__((amdgpu_waves_per_eu(PCP_LS_WAVES, PCP_LS_WAVES))) k_ls_batch(const LsArgs* __restrict__ all) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned pair, block;
    pair_and_block(pair, block);
    const LsArgs A = all[pair];
    if ((int)(block * PCP_LS_THREADS) >= A.n_b) return;
    ls_body<MODE, true, PCP_LS_BLOCK_POOL ? 2 : 0>(A, smem, block);
}

// The first dispatch of freshly prepared pairs (no query has a hint yet): the same kernel with the all-rings search
// of a wave run as dense passes (find_closest_pooled); a wave in which some lane has a hint searches as above.
#ifndef PCP_LS_FIRST_WAVES
#define PCP_LS_FIRST_WAVES 8
#endif
template <int MODE>
__global__ void __launch_bounds__(PCP_BLOCK) __attribute__((amdgpu_waves_per_eu(PCP_LS_FIRST_WAVES, PCP_LS_FIRST_WAVES))) k_ls_batch_first(const LsArgs* __restrict__ all) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned pair, block;
    pair_and_block(pair, block);
    const LsArgs A = all[pair];
    if ((int)(block * PCP_BLOCK) >= A.n_b) return;
    ls_body<MODE, true, 1>(A, smem, block);
}


__global__ void __launch_bounds__(1024)
k_finish(const FinishArgs F, int solve, int mode, int max_iter, float tol, int full_iter) {
    finish_body(F, solve, mode, max_iter, tol, full_iter);
}

__global__ void __launch_bounds__(1024)
k_finish_batch(const FinishArgs* __restrict__ all, int mode, int max_iter, float tol, int full_iter) {
    const FinishArgs F = all[blockIdx.x];
    finish_body(F, 1, mode, max_iter, tol, full_iter);
}

// uniform_optimal_icp of every pair of a batch / of one context: one block per pair.
__global__ void __launch_bounds__(PCP_POSE_THREADS) k_pose_batch(const FinishArgs* __restrict__ all) {
    const FinishArgs F = all[blockIdx.x];
    pose_body(F);
}

__global__ void __launch_bounds__(PCP_POSE_THREADS) k_pose(const FinishArgs F) { pose_body(F); }

__global__ void __launch_bounds__(PCP_POSE_THREADS) k_pose_steps(const float* __restrict__ dx, int n, double* __restrict__ out) {
    pose_from_steps(dx, n, out);
}

__global__ void k_apply_pending_batch(const LsArgs* __restrict__ all) {
    const LsArgs A = all[blockIdx.y];
    if (!A.st->pending) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_b) return;
    double* b64 = A.b64;
    const IcpState* st = A.st;
    const double px = b64[3 * (size_t)i + 0], py = b64[3 * (size_t)i + 1], pz = b64[3 * (size_t)i + 2];
    b64[3 * (size_t)i + 0] = ((st->R[0] * px + st->R[1] * py) + st->R[2] * pz) - st->t[0];
    b64[3 * (size_t)i + 1] = ((st->R[3] * px + st->R[4] * py) + st->R[5] * pz) - st->t[1];
    b64[3 * (size_t)i + 2] = ((st->R[6] * px + st->R[7] * py) + st->R[8] * pz) - st->t[2];
}

__global__ void k_state_batch(const LsArgs* __restrict__ all, int n, int reset) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    IcpState* st = all[p].st;
    if (reset) {
        st->pending = 0; st->iter = 0; st->done = 0; st->converged = 0; st->singular = 0;
    } else {
        st->pending = 0;
    }
}

// rows x[n], y[n], z[n] -> points (x, y, z)[n]
__global__ void k_interleave_xyz(const double* __restrict__ rows, int n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[3 * (size_t)i + 0] = rows[i];
    out[3 * (size_t)i + 1] = rows[(size_t)n + i];
    out[3 * (size_t)i + 2] = rows[2 * (size_t)n + i];
}

// Applies the step still pending after the last dispatch.
__global__ void k_apply_pending(double* __restrict__ b64, int n_b, IcpState* __restrict__ st) {
    if (!st->pending) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_b) return;
    const double px = b64[3 * (size_t)i + 0], py = b64[3 * (size_t)i + 1], pz = b64[3 * (size_t)i + 2];
    b64[3 * (size_t)i + 0] = ((st->R[0] * px + st->R[1] * py) + st->R[2] * pz) - st->t[0];
    b64[3 * (size_t)i + 1] = ((st->R[3] * px + st->R[4] * py) + st->R[5] * pz) - st->t[1];
    b64[3 * (size_t)i + 2] = ((st->R[6] * px + st->R[7] * py) + st->R[8] * pz) - st->t[2];
}

__global__ void k_clear_pending(IcpState* __restrict__ st) { st->pending = 0; }

// ParalellHBSum.glsl:127-171, one level.  A wave owns one block of 64 tiles and
// keeps element `lane` of each of them in a register, so every load is one
// contiguous 256-byte tile row across the wave and the pairwise tree
// (s = 32..1: a[t] += a[t+s]) runs in registers without any exchange.
// Rows 6-7 and column 7 carry no data in the reference; they are written as 0.
__global__ void __launch_bounds__(256)
k_hb_level(const float* __restrict__ in, int n, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int groups = (n + 63) >> 6;
    if (g >= groups) return;
    const bool live = lane < 48 && (lane & 7) != 7;
    float v[64];
    const size_t base = (size_t)g * 64;
#pragma unroll
    for (int t = 0; t < 64; ++t)
        v[t] = (live && base + t < (size_t)n) ? in[(base + t) * 64 + lane] : 0.0f;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
#pragma unroll
        for (int t = 0; t < s; ++t) v[t] = v[t] + v[t + s];
    }
    out[(size_t)g * 64 + lane] = live ? v[0] : 0.0f;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

// ---- fixed scratch of pcp_hb_sum_static ----------------------------------------------------
// The intermediate tree levels of up to PCP_HB_STATIC_MAX_TILES tiles: ceil(n/64) tiles of level 1 and
// ceil(n/4096) of level 2 (level 3 reuses the front).  Part of the code object: every GPU the library
// runs on holds its own copy (1.04 MB) at its own address until the process ends.  In host code the
// array's NAME is only a handle for hipGetSymbolAddress; kernels receive the address that call
// returned for the device they run on.
#define PCP_HB_STATIC_MAX_TILES 262144
#define PCP_HB_STATIC_FLOATS ((PCP_HB_STATIC_MAX_TILES / 64 + PCP_HB_STATIC_MAX_TILES / 4096) * 64)
static __device__ __align__(256) float scratch[PCP_HB_STATIC_FLOATS];

namespace {

const int kMaxDevices = 64;
float* g_hb_scratch[kMaxDevices];        // per device: device address of `scratch`, resolved once
std::mutex g_hb_scratch_mutex;

// Resolves the device address of `scratch` on `device` (which must be the current device).
// Called from pcp_create, i.e. before any context of that device exists.
int hb_scratch_bind(int device) {
    if (device < 0 || device >= kMaxDevices) return pcp_fail("device %d beyond the %d the scratch table covers", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_hb_scratch_mutex);
    if (g_hb_scratch[device]) return 0;
    float* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) return pcp_fail("hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
    g_hb_scratch[device] = ptr;
    return 0;
}

int hb_scratch_of(int device, float** out) {
    std::lock_guard<std::mutex> lock(g_hb_scratch_mutex);
    if (device < 0 || device >= kMaxDevices || !g_hb_scratch[device])
        return pcp_fail("no scratch address for device %d (contexts come from pcp_create, which resolves it)", device);
    *out = g_hb_scratch[device];
    return 0;
}

}  // namespace

namespace {

int use(pcp_ctx* c) {
    if (!c) return pcp_fail("null context");
    PCP_CHECK(hipSetDevice(c->device));
    return 0;
}

int ensure_tiles(pcp_ctx* c) {
    if (c->tiles) return 0;
    PCP_CHECK(hipMalloc(&c->tiles, (size_t)c->max_points * 64 * sizeof(float)));
    const int g1 = cdiv(c->max_points, 64);
    PCP_CHECK(hipMalloc(&c->tiles2, (size_t)(g1 + cdiv(g1, 64)) * 64 * sizeof(float)));
    return 0;
}

size_t scan_smem(const pcp_ctx* c) { return (size_t)c->n_scans * (sizeof(int2) + sizeof(float4)); }

int upload_b32(pcp_ctx* c, const float* b_xyzw, int n_b) {
    if (!c->have_target) return pcp_fail("no target: call pcp_set_target first");
    if (!b_xyzw) return pcp_fail("null source points");
    if (n_b < 1 || n_b > c->max_points) return pcp_fail("n_b = %d outside [1, %d]", n_b, c->max_points);
    c->n_b = n_b;
    PCP_CHECK(hipMemcpyAsync(c->b32, b_xyzw, (size_t)n_b * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    return 0;
}

// The source cloud of a stepwise call from the caller's (pageable) array into b32 -- the float32 rows of 4 that
// ComputeShader.pad_points (:956-961) makes -- through a page-locked buffer of the context: the rows are converted /
// padded chunk by chunk on the host and every finished chunk is on its way up while the next one is converted, instead
// of one blocking copy out of pageable memory behind a full NumPy pass.  The casts are C's float64 -> float32
// conversions (round to nearest even, NumPy's astype).
int stage_source(pcp_ctx* c, const void* pts, int layout, int n_b) {
    if (!c->have_target) return pcp_fail("no target: call pcp_set_target first");
    if (!pts) return pcp_fail("null source points");
    if (n_b < 1 || n_b > c->max_points) return pcp_fail("n_b = %d outside [1, %d]", n_b, c->max_points);
    if (layout != PCP_PTS_F32_XYZW && layout != PCP_PTS_F32_XYZ && layout != PCP_PTS_F64_XYZ && layout != PCP_PTS_F64_ROWS)
        return pcp_fail("unknown point layout %d", layout);
    if (!c->stage_in) PCP_CHECK(hipHostMalloc((void**)&c->stage_in, (size_t)c->max_points * sizeof(float4)));
    c->n_b = n_b;
    const int chunk = 32768;
    for (int first = 0; first < n_b; first += chunk) {
        const int m = (n_b - first < chunk) ? n_b - first : chunk;
        float4* dst = c->stage_in + first;
        if (layout == PCP_PTS_F32_XYZW) {
            memcpy(dst, (const float4*)pts + first, (size_t)m * sizeof(float4));
        } else if (layout == PCP_PTS_F32_XYZ) {
            const float* src = (const float*)pts + 3 * (size_t)first;
            for (int i = 0; i < m; ++i) dst[i] = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], 0.0f);
        } else if (layout == PCP_PTS_F64_ROWS) {       // three coordinate rows of n_b: a transposed (3, n) cloud as it lies
            const double* x = (const double*)pts + first;
            const double* y = x + n_b;
            const double* z = y + n_b;
            for (int i = 0; i < m; ++i) dst[i] = make_float4((float)x[i], (float)y[i], (float)z[i], 0.0f);
        } else {
            const double* src = (const double*)pts + 3 * (size_t)first;
            for (int i = 0; i < m; ++i) dst[i] = make_float4((float)src[3 * i], (float)src[3 * i + 1], (float)src[3 * i + 2], 0.0f);
        }
        PCP_CHECK(hipMemcpyAsync(c->b32 + first, dst, (size_t)m * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    }
    return 0;
}

LsArgs ls_args(const pcp_ctx* c, int debug, float* tiles) {
    LsArgs A;
    memset(&A, 0, sizeof(A));          // padding bytes too: descriptors are compared with memcmp
    A.a = c->a; A.ua = c->ua; A.scan = c->scan; A.stat = c->ring_stat; A.bins = c->ring_bins;
    A.n_a = c->n_a; A.Z = c->n_scans; A.Zs = c->n_scans_search; A.o = c->origin;
    A.normals = c->normals; A.b32 = c->b32; A.b_in = c->b_in; A.b64 = c->b64; A.st = c->state;
    A.n_b = c->n_b; A.debug = debug;
    A.hint = c->hint; A.partials = c->partials; A.tiles = tiles;
    A.skip_store_iter = -1;
    return A;
}

FinishArgs finish_args(const pcp_ctx* c) {
    FinishArgs F;
    memset(&F, 0, sizeof(F));
    F.partials = c->partials; F.n1 = cdiv(c->n_b, 64); F.lvl0 = c->lvl0; F.lvl1 = c->lvl1;
    F.st = c->state; F.hb_out = c->hb_trace; F.dx_out = c->dx_trace; F.pose = c->pose;
    return F;
}

int check_ls(const pcp_ctx* c, int mode, int debug) {
    if (mode != PCP_MODE_POINT && mode != PCP_MODE_PLANE) return pcp_fail("unknown mode %d", mode);
    if (mode == PCP_MODE_PLANE && !c->have_normals) return pcp_fail("plane mode needs pcp_normals after pcp_set_target");
    if (debug && c->n_b > c->n_a) return pcp_fail("identity correspondences need n_b <= n_a");
    return 0;
}

int launch_ls(pcp_ctx* c, int mode, bool f64src, int debug, float* tiles) {
    if (int e = check_ls(c, mode, debug)) return e;
    const dim3 grid(cdiv(c->n_b, PCP_BLOCK)), block(PCP_BLOCK);
    const size_t sm = scan_smem(c);
    const LsArgs A = ls_args(c, debug, tiles);
    if (mode == PCP_MODE_POINT) {
        if (f64src) hipLaunchKernelGGL((k_ls<PCP_MODE_POINT, true>), grid, block, sm, c->stream, A);
        else hipLaunchKernelGGL((k_ls<PCP_MODE_POINT, false>), grid, block, sm, c->stream, A);
    } else {
        if (f64src) hipLaunchKernelGGL((k_ls<PCP_MODE_PLANE, true>), grid, block, sm, c->stream, A);
        else hipLaunchKernelGGL((k_ls<PCP_MODE_PLANE, false>), grid, block, sm, c->stream, A);
    }
    PCP_CHECK(hipGetLastError());
    return 0;
}

int launch_finish(pcp_ctx* c, int solve, int mode, int max_iter, float tol, int full_iter) {
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(1024), 0, c->stream, finish_args(c), solve, mode, max_iter, tol,
                       full_iter);
    PCP_CHECK(hipGetLastError());
    return 0;
}

// The two kernels that serve a single pair in flight (the device loop, and the stepwise dispatches of
// prepareLS / dispatchLS and friends): the all-rings pre-pass and the four-wave search.  Both leave every
// result as the one-lane kernel does.  PCP_FIRST_LANES=0 / PCP_LS_QUAD=0 in the environment switch them off.
template <bool F64SRC>
int launch_first_rings(pcp_ctx* c) {
    static const int lanes = [] {
        const char* e = getenv("PCP_FIRST_LANES");
        const int v = e ? atoi(e) : 8;
        return (v == 4 || v == 8 || v == 16 || v == 32) ? v : (v == 0 ? 0 : 8);
    }();
    if (!lanes) return 0;
    // every source point up to hinted_upto has been through a search since the hints were last reset: nothing for the
    // pre-pass to do (a point that came out of it without a ring keeps -1 and walks all rings in the search kernel)
    if (c->n_b <= c->hinted_upto) return 0;
    c->hinted_upto = c->n_b;
    const LsArgs A = ls_args(c, 0, nullptr);
    const dim3 grid(cdiv(c->n_b * lanes, PCP_BLOCK)), block(PCP_BLOCK);
    const size_t sm = scan_smem(c);
    if (lanes == 4) hipLaunchKernelGGL((k_first_rings<4, F64SRC>), grid, block, sm, c->stream, A);
    else if (lanes == 8) hipLaunchKernelGGL((k_first_rings<8, F64SRC>), grid, block, sm, c->stream, A);
    else if (lanes == 16) hipLaunchKernelGGL((k_first_rings<16, F64SRC>), grid, block, sm, c->stream, A);
    else hipLaunchKernelGGL((k_first_rings<32, F64SRC>), grid, block, sm, c->stream, A);
    PCP_CHECK(hipGetLastError());
    return 0;
}

int use_pool() {       // PCP_LS_POOL: 0 = every dispatch of a batch walks its rings in lock step, 1 = the first one pools, 2 = all do
    static const int pool = [] { const char* e = getenv("PCP_LS_POOL"); return e ? atoi(e) : 1; }();
    return pool;
}

bool use_quad() {
    static const bool quad = [] { const char* e = getenv("PCP_LS_QUAD"); return !(e && e[0] == '0'); }();
    return quad;
}

template <bool F64SRC>
int launch_ls_quad(pcp_ctx* c, int mode) {
    if (int e = check_ls(c, mode, 0)) return e;
    const LsArgs A = ls_args(c, 0, nullptr);
    const dim3 grid(cdiv(c->n_b, 64)), block(256);
    if (mode == PCP_MODE_POINT) hipLaunchKernelGGL((k_ls_quad<PCP_MODE_POINT, F64SRC>), grid, block, scan_smem(c), c->stream, A);
    else hipLaunchKernelGGL((k_ls_quad<PCP_MODE_PLANE, F64SRC>), grid, block, scan_smem(c), c->stream, A);
    PCP_CHECK(hipGetLastError());
    return 0;
}

int ls_stepwise(pcp_ctx* c, const void* pts, int layout, int n_b, int mode, int debug, float* hb, float* tiles_host) {
    if (int e = use(c)) return e;
    if (layout == 0) {                      // the caller's padded float32 rows, copied as they are
        if (int e = upload_b32(c, (const float*)pts, n_b)) return e;
    } else if (int e = stage_source(c, pts, layout, n_b)) return e;
    float* tiles = nullptr;
    if (tiles_host) {
        if (int e = ensure_tiles(c)) return e;
        tiles = c->tiles;
    }
    if (!debug && !tiles && use_quad()) {
        if (int e = check_ls(c, mode, 0)) return e;
        if (int e = launch_first_rings<false>(c)) return e;
        if (int e = launch_ls_quad<false>(c, mode)) return e;
    } else if (int e = launch_ls(c, mode, false, debug, tiles)) return e;
    if (hb) {
        if (int e = launch_finish(c, 0, mode, 0, 0.f, 0)) return e;
        PCP_CHECK(hipMemcpyAsync(hb, c->hb_trace, 64 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    if (tiles_host)
        PCP_CHECK(hipMemcpyAsync(tiles_host, tiles, (size_t)n_b * 64 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

int pcp_version(void) { return 1; }

int pcp_device_count(int* count) {
    if (!count) return pcp_fail("null count");
    PCP_CHECK(hipGetDeviceCount(count));
    return 0;
}

int pcp_create(int device, int max_points, int max_scans, pcp_ctx** out) {
    if (!out) return pcp_fail("null out pointer");
    *out = nullptr;
    if (max_points < 1 || max_points > (1 << 24))
        return pcp_fail("max_points = %d outside [1, 2^24] (indices travel as float32 in the reference)", max_points);
    // every search block keeps the ring table (24 bytes per ring) in its 64 KB of dynamic LDS
    if (max_scans < 1 || max_scans > 2048) return pcp_fail("max_scans = %d outside [1, 2048]", max_scans);
    int ndev = 0;
    PCP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return pcp_fail("device %d not present (%d visible)", device, ndev);
    PCP_CHECK(hipSetDevice(device));
    if (int e = hb_scratch_bind(device)) return e;      // device address of the fixed scratch array, once per device
    pcp_ctx* c = new pcp_ctx();
    c->device = device;
    c->max_points = max_points;
    c->max_scans = max_scans;
    const size_t n = (size_t)max_points;
    const int n1 = cdiv(max_points, 64), n2 = cdiv(n1, 64);
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    A((void**)&c->a, n * sizeof(float4));
    A((void**)&c->ua, n * sizeof(float4));
    A((void**)&c->normals, n * sizeof(float4));
    A((void**)&c->b32, n * sizeof(float4));
    A((void**)&c->b_in, n * 3 * sizeof(double));
    A((void**)&c->b64, n * 3 * sizeof(double));
    A((void**)&c->hint, n * sizeof(float));
    A((void**)&c->nn_idx, 2 * n * sizeof(uint32_t));      // indices, and behind them the squared distances (one copy fetches both)
    A((void**)&c->scan, (size_t)max_scans * sizeof(int2));
    A((void**)&c->ring_stat, (size_t)max_scans * sizeof(float4));
    A((void**)&c->ring_bins, (size_t)max_scans * PCP_AZ_BINS * 2 * sizeof(float4));      // elevation bands, then range bands
    A((void**)&c->partials, (size_t)n1 * PCP_PSTRIDE * sizeof(float));
    A((void**)&c->lvl0, (size_t)n2 * PCP_PSTRIDE * sizeof(float));
    A((void**)&c->lvl1, (size_t)n2 * PCP_PSTRIDE * sizeof(float));
    A((void**)&c->hb_trace, (size_t)pcp_ctx::max_iter_cap * 64 * sizeof(float));
    A((void**)&c->dx_trace, (size_t)pcp_ctx::max_iter_cap * 6 * sizeof(float));
    A((void**)&c->state, sizeof(IcpState));
    A((void**)&c->pose, 16 * sizeof(double));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->scan_host, (size_t)max_scans * sizeof(int2));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->scan_copied, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        pcp_fail("pcp_create: %s", hipGetErrorString(e));
        pcp_destroy(c);
        return 1;
    }
    c->stream = c->own_stream;
    *out = c;
    return 0;
}

static void batch_forget(pcp_batch* b, pcp_ctx* c);

int pcp_destroy(pcp_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->batch) {               // destroyed before the batch that adopted it: drain that stream, leave the batch
        (void)hipStreamSynchronize(c->stream);
        batch_forget(c->batch, c);
    }
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    void* bufs[] = {c->a, c->ua, c->normals, c->b32, c->b_in, c->b64, c->hint, c->nn_idx, c->scan, c->ring_stat, c->ring_bins,
                    c->partials, c->lvl0, c->lvl1, c->hb_trace, c->dx_trace, c->state, c->pose, c->pose_steps, c->tiles, c->tiles2};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (c->vox) pcp_voxel_state_free(c->vox);
    for (hipEvent_t ev : c->prof_events) (void)hipEventDestroy(ev);
    if (c->scan_host) (void)hipHostFree(c->scan_host);
    for (void* p : {(void*)c->stage_in, (void*)c->stage_idx, (void*)c->stage_hint})
        if (p) (void)hipHostFree(p);
    if (c->scan_copied) (void)hipEventDestroy(c->scan_copied);
    if (c->nn_ev0) (void)hipEventDestroy(c->nn_ev0);
    if (c->nn_ev1) (void)hipEventDestroy(c->nn_ev1);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return 0;
}

int pcp_set_stream(pcp_ctx* c, void* hip_stream) {
    if (!c) return pcp_fail("null context");
    if (c->batch) return pcp_fail("the context runs on its batch's stream until pcp_batch_destroy");
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return 0;
}

int pcp_synchronize(pcp_ctx* c) {
    if (int e = use(c)) return e;
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int pcp_reset_hints(pcp_ctx* c) {
    if (int e = use(c)) return e;
    hipLaunchKernelGGL(k_fill, dim3(cdiv(c->max_points, 256)), dim3(256), 0, c->stream, c->hint, c->max_points, -1.0f);
    c->hinted_upto = 0;
    PCP_CHECK(hipGetLastError());
    return 0;
}

int pcp_prepare_resident(pcp_ctx* c, int with_normals) {
    if (int e = use(c)) return e;
    if (!c->have_target) return pcp_fail("no target: call pcp_set_target first");
    hipLaunchKernelGGL(k_unit, dim3(cdiv(c->n_a, 256)), dim3(256), 0, c->stream, c->a, c->ua, c->n_a, c->origin);
    PCP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_ring_stats, dim3(c->n_scans), dim3(64), 0, c->stream, c->a, c->origin, c->ua, c->scan, c->n_scans, c->ring_stat, c->ring_bins);
    PCP_CHECK(hipGetLastError());
    if (int e = pcp_reset_hints(c)) return e;
    if (with_normals) {
        if (c->n_scans < 2) return pcp_fail("normals need at least 2 scan lines");
        hipLaunchKernelGGL(k_normals, dim3(cdiv(c->n_a, PCP_BLOCK)), dim3(PCP_BLOCK), scan_smem(c), c->stream, c->a, c->ua,
                           c->scan, c->ring_stat, c->n_a, c->n_scans, c->origin, c->normals);
        PCP_CHECK(hipGetLastError());
        c->have_normals = true;
    }
    return 0;
}

static int set_target_impl(pcp_ctx* c, const float* a_xyzw, int n_a, const uint32_t* scan_lines, int n_scans,
                           const float* origin4, hipMemcpyKind kind);

int pcp_set_target(pcp_ctx* c, const float* a_xyzw, int n_a, const uint32_t* scan_lines, int n_scans,
                   const float* origin4) {
    return set_target_impl(c, a_xyzw, n_a, scan_lines, n_scans, origin4, hipMemcpyHostToDevice);
}

int pcp_set_target_dev(pcp_ctx* c, const float* a_xyzw_dev, int n_a, const uint32_t* scan_lines, int n_scans,
                       const float* origin4) {
    return set_target_impl(c, a_xyzw_dev, n_a, scan_lines, n_scans, origin4, hipMemcpyDeviceToDevice);
}

static int set_target_impl(pcp_ctx* c, const float* a_xyzw, int n_a, const uint32_t* scan_lines, int n_scans,
                           const float* origin4, hipMemcpyKind kind) {
    if (int e = use(c)) return e;
    if (!a_xyzw || !scan_lines || !origin4) return pcp_fail("null argument");
    if (n_a < 1 || n_a > c->max_points) return pcp_fail("n_a = %d outside [1, %d]", n_a, c->max_points);
    // the reference overflows its scan-line SSBO here and re-raises the GL error (ComputeShader.py:506-512)
    if (n_scans < 1 || n_scans > c->max_scans)
        return pcp_fail("%d scan lines do not fit maxScans = %d", n_scans, c->max_scans);
    for (int i = 0; i < n_scans; ++i) {
        const uint32_t b = scan_lines[4 * i + 0], e = scan_lines[4 * i + 1];
        if (b > e || e >= (uint32_t)n_a)
            return pcp_fail("scan line %d = (%u, %u) outside a cloud of %d points", i, b, e, n_a);
    }
    // The table goes through a page-locked buffer the context owns, so the call need not wait for the copy;
    // only a second upload while the first is still in flight waits (for that copy alone).
    PCP_CHECK(hipEventSynchronize(c->scan_copied));
    int2* lines = c->scan_host;
    for (int i = 0; i < n_scans; ++i) lines[i] = make_int2((int)scan_lines[4 * i + 0], (int)scan_lines[4 * i + 1]);
    c->have_target = false;
    c->have_normals = false;
    PCP_CHECK(hipMemcpyAsync(c->a, a_xyzw, (size_t)n_a * sizeof(float4), kind, c->stream));
    PCP_CHECK(hipMemcpyAsync(c->scan, lines, (size_t)n_scans * sizeof(int2), hipMemcpyHostToDevice, c->stream));
    PCP_CHECK(hipEventRecord(c->scan_copied, c->stream));
    c->origin = make_float4(origin4[0], origin4[1], origin4[2], origin4[3]);
    c->n_a = n_a;
    c->n_scans = n_scans;
    int zs = n_scans;              // trailing rows equal to the row before them are padding copies
    while (zs > 1 && lines[zs - 1].x == lines[zs - 2].x && lines[zs - 1].y == lines[zs - 2].y) --zs;
    c->n_scans_search = zs;
    c->have_target = true;
    return pcp_prepare_resident(c, 0);
}

int pcp_normals(pcp_ctx* c, float* out_normals) {
    if (int e = use(c)) return e;
    if (!c->have_target) return pcp_fail("no target: call pcp_set_target first");
    if (c->n_scans < 2) return pcp_fail("normals need at least 2 scan lines");
    hipLaunchKernelGGL(k_normals, dim3(cdiv(c->n_a, PCP_BLOCK)), dim3(PCP_BLOCK), scan_smem(c), c->stream, c->a, c->ua, c->scan,
                       c->ring_stat, c->n_a, c->n_scans, c->origin, c->normals);
    PCP_CHECK(hipGetLastError());
    c->have_normals = true;
    if (out_normals) {
        PCP_CHECK(hipMemcpyAsync(out_normals, c->normals, (size_t)c->n_a * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
        PCP_CHECK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int pcp_nn(pcp_ctx* c, const float* b_xyzw, int n_b, uint32_t* idx, float* dist2, float* hint_out) {
    if (int e = use(c)) return e;
    if (!idx || !dist2) return pcp_fail("null output");
    if (int e = upload_b32(c, b_xyzw, n_b)) return e;
    if (!c->nn_ev0) {
        PCP_CHECK(hipEventCreate(&c->nn_ev0));
        PCP_CHECK(hipEventCreate(&c->nn_ev1));
    }
    PCP_CHECK(hipEventRecord(c->nn_ev0, c->stream));
    if (int e = launch_first_rings<false>(c)) return e;          // queries still without a hint: their ring, found by 8 lanes each
    hipLaunchKernelGGL(k_nn, dim3(cdiv(n_b, PCP_BLOCK)), dim3(PCP_BLOCK), scan_smem(c), c->stream, c->a, c->ua, c->scan, c->ring_stat, c->ring_bins,
                       c->n_a, c->n_scans, c->n_scans_search, c->origin, c->b32, n_b, c->hint, c->nn_idx, (float*)(c->nn_idx + n_b));
    PCP_CHECK(hipGetLastError());
    PCP_CHECK(hipEventRecord(c->nn_ev1, c->stream));
    c->nn_timed = true;
    PCP_CHECK(hipMemcpyAsync(idx, c->nn_idx, (size_t)n_b * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipMemcpyAsync(dist2, c->nn_idx + n_b, (size_t)n_b * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (hint_out)
        PCP_CHECK(hipMemcpyAsync(hint_out, c->hint, (size_t)n_b * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int pcp_nn_staged(pcp_ctx* c, const void* pts, int layout, int n_b, const uint32_t** idx, const float** dist2, const float** hint) {
    if (int e = use(c)) return e;
    if (!idx || !dist2) return pcp_fail("null output");
    if (int e = stage_source(c, pts, layout, n_b)) return e;
    if (!c->stage_idx) {
        PCP_CHECK(hipHostMalloc((void**)&c->stage_idx, 2 * (size_t)c->max_points * sizeof(uint32_t)));
        PCP_CHECK(hipHostMalloc((void**)&c->stage_hint, (size_t)c->max_points * sizeof(float)));
    }
    if (!c->nn_ev0) {
        PCP_CHECK(hipEventCreate(&c->nn_ev0));
        PCP_CHECK(hipEventCreate(&c->nn_ev1));
    }
    PCP_CHECK(hipEventRecord(c->nn_ev0, c->stream));
    if (int e = launch_first_rings<false>(c)) return e;
    hipLaunchKernelGGL(k_nn, dim3(cdiv(n_b, PCP_BLOCK)), dim3(PCP_BLOCK), scan_smem(c), c->stream, c->a, c->ua, c->scan, c->ring_stat, c->ring_bins,
                       c->n_a, c->n_scans, c->n_scans_search, c->origin, c->b32, n_b, c->hint, c->nn_idx, (float*)(c->nn_idx + n_b));
    PCP_CHECK(hipGetLastError());
    PCP_CHECK(hipEventRecord(c->nn_ev1, c->stream));
    c->nn_timed = true;
    // indices and distances lie behind one another on both sides: one copy
    PCP_CHECK(hipMemcpyAsync(c->stage_idx, c->nn_idx, 2 * (size_t)n_b * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (hint) PCP_CHECK(hipMemcpyAsync(c->stage_hint, c->hint, (size_t)n_b * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    *idx = c->stage_idx;
    *dist2 = (const float*)(c->stage_idx + n_b);
    if (hint) *hint = c->stage_hint;
    return 0;
}

int pcp_ls_staged(pcp_ctx* c, const void* pts, int layout, int n_b, int mode, int debug, float* hb) {
    if (!hb) return pcp_fail("null output");
    if (layout != PCP_PTS_F32_XYZW && layout != PCP_PTS_F32_XYZ && layout != PCP_PTS_F64_XYZ && layout != PCP_PTS_F64_ROWS)
        return pcp_fail("unknown point layout %d", layout);
    return ls_stepwise(c, pts, layout, n_b, mode, debug, hb, nullptr);
}

int pcp_nn_kernel_ms(pcp_ctx* c, float* ms) {
    if (int e = use(c)) return e;
    if (!ms) return pcp_fail("null output");
    if (!c->nn_timed) return pcp_fail("no pcp_nn call to report on");
    PCP_CHECK(hipEventSynchronize(c->nn_ev1));
    PCP_CHECK(hipEventElapsedTime(ms, c->nn_ev0, c->nn_ev1));
    return 0;
}

int pcp_ls_point(pcp_ctx* c, const float* b_xyzw, int n_b, int debug, float* hb) {
    if (!hb) return pcp_fail("null output");
    return ls_stepwise(c, b_xyzw, 0, n_b, PCP_MODE_POINT, debug, hb, nullptr);
}

int pcp_ls_plane(pcp_ctx* c, const float* b_xyzw, int n_b, float* hb) {
    if (!hb) return pcp_fail("null output");
    return ls_stepwise(c, b_xyzw, 0, n_b, PCP_MODE_PLANE, 0, hb, nullptr);
}

int pcp_ls_tiles(pcp_ctx* c, const float* b_xyzw, int n_b, int mode, int debug, float* tiles) {
    if (!tiles) return pcp_fail("null output");
    return ls_stepwise(c, b_xyzw, 0, n_b, mode, debug, nullptr, tiles);
}

// ComputeShader.py:682-707: levels ping-pong between two buffers until one block is left.
// Level 1 writes g1 = ceil(n/64) tiles to scratch[0, g1), level 2 writes ceil(g1/64) tiles
// behind them, level 3 reuses the front, and so on.  Every level writes all 64 floats of each
// of its output tiles and reads only the tiles the level before wrote, so the scratch needs no
// particular contents on entry.
static int hb_levels(pcp_ctx* c, const float* tiles_dev, int n, float* scratch_dev, float* out_dev) {
    const int g1 = cdiv(n, 64);
    float* region[2] = {scratch_dev, scratch_dev + (size_t)g1 * 64};
    const float* in = tiles_dev;
    int cur = n;
    for (int level = 0;; ++level) {
        const int groups = cdiv(cur, 64);
        float* dst = (groups == 1) ? out_dev : region[level & 1];
        hipLaunchKernelGGL(k_hb_level, dim3(cdiv(groups, 4)), dim3(256), 0, c->stream, in, cur, dst);
        PCP_CHECK(hipGetLastError());
        if (groups == 1) break;
        in = dst;
        cur = groups;
    }
    return 0;
}

int pcp_hb_sum_dev(pcp_ctx* c, const float* tiles_dev, int n, float* scratch_dev, float* out_dev) {
    if (int e = use(c)) return e;
    if (!tiles_dev || !scratch_dev || !out_dev) return pcp_fail("null argument");
    if (n < 1) return pcp_fail("n = %d: nothing to sum", n);
    return hb_levels(c, tiles_dev, n, scratch_dev, out_dev);
}

int pcp_hb_sum_static(pcp_ctx* c, const float* tiles_dev, int n, float* out_dev) {
    if (int e = use(c)) return e;
    if (!tiles_dev || !out_dev) return pcp_fail("null argument");
    if (n < 1) return pcp_fail("n = %d: nothing to sum", n);
    if (n > PCP_HB_STATIC_MAX_TILES)
        return pcp_fail("n = %d tiles exceed the %d the library's fixed scratch array holds; use pcp_hb_sum_dev with a "
                        "workspace of your own", n, PCP_HB_STATIC_MAX_TILES);
    float* ptr = nullptr;
    if (int e = hb_scratch_of(c->device, &ptr)) return e;
    return hb_levels(c, tiles_dev, n, ptr, out_dev);
}

int pcp_hb_sum(pcp_ctx* c, const float* tiles, int n, float* out) {
    if (int e = use(c)) return e;
    if (!tiles || !out) return pcp_fail("null argument");
    if (n < 1 || n > c->max_points) return pcp_fail("n = %d outside [1, %d]", n, c->max_points);
    if (int e = ensure_tiles(c)) return e;
    PCP_CHECK(hipMemcpyAsync(c->tiles, tiles, (size_t)n * 64 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (int e = pcp_hb_sum_dev(c, c->tiles, n, c->tiles2, c->hb_trace)) return e;
    PCP_CHECK(hipMemcpyAsync(out, c->hb_trace, 64 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int pcp_upload_source(pcp_ctx* c, const double* b_xyz, int n_b) {
    if (int e = use(c)) return e;
    if (!b_xyz) return pcp_fail("null source points");
    if (n_b < 1 || n_b > c->max_points) return pcp_fail("n_b = %d outside [1, %d]", n_b, c->max_points);
    c->n_b = n_b;
    PCP_CHECK(hipMemcpyAsync(c->b_in, b_xyz, (size_t)n_b * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
}

int pcp_upload_source_soa(pcp_ctx* c, const double* x, const double* y, const double* z, int n_b) {
    if (int e = use(c)) return e;
    if (!x || !y || !z) return pcp_fail("null source points");
    if (n_b < 1 || n_b > c->max_points) return pcp_fail("n_b = %d outside [1, %d]", n_b, c->max_points);
    c->n_b = n_b;
    // the three rows land in the working copy (rewritten by the loop's first iteration anyway) and are
    // interleaved from there
    const size_t row = (size_t)n_b * sizeof(double);
    PCP_CHECK(hipMemcpyAsync(c->b64, x, row, hipMemcpyHostToDevice, c->stream));
    PCP_CHECK(hipMemcpyAsync(c->b64 + n_b, y, row, hipMemcpyHostToDevice, c->stream));
    PCP_CHECK(hipMemcpyAsync(c->b64 + 2 * (size_t)n_b, z, row, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_interleave_xyz, dim3(cdiv(n_b, 256)), dim3(256), 0, c->stream, c->b64, n_b, c->b_in);
    PCP_CHECK(hipGetLastError());
    return 0;
}

int pcp_upload_source_dev(pcp_ctx* c, const double* b_xyz_dev, int n_b) {
    if (int e = use(c)) return e;
    if (!b_xyz_dev) return pcp_fail("null source points");
    if (n_b < 1 || n_b > c->max_points) return pcp_fail("n_b = %d outside [1, %d]", n_b, c->max_points);
    c->n_b = n_b;
    PCP_CHECK(hipMemcpyAsync(c->b_in, b_xyz_dev, (size_t)n_b * 3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int pcp_icp_launch(pcp_ctx* c, int mode, int max_iter, float tol, int full_iter) {
    if (int e = use(c)) return e;
    if (!c->have_target) return pcp_fail("no target: call pcp_set_target first");
    if (c->n_b < 1) return pcp_fail("no source: call pcp_upload_source first");
    if (max_iter < 0 || max_iter > pcp_ctx::max_iter_cap)
        return pcp_fail("max_iter = %d outside [0, %d]", max_iter, pcp_ctx::max_iter_cap);
    if (mode != PCP_MODE_POINT && mode != PCP_MODE_PLANE) return pcp_fail("unknown mode %d", mode);
    PCP_CHECK(hipMemsetAsync(c->state, 0, sizeof(IcpState), c->stream));
    c->run_mode = mode;
    c->run_max_iter = max_iter;
    if (max_iter == 0)
        PCP_CHECK(hipMemcpyAsync(c->b64, c->b_in, (size_t)c->n_b * 3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    c->prof_used = 0;
    if (max_iter > 0)
        if (int e = launch_first_rings<true>(c)) return e;
    for (int it = 0; it < max_iter; ++it) {
        if (c->profile) {
            while ((int)c->prof_events.size() < 2 * (it + 1)) {
                hipEvent_t ev;
                PCP_CHECK(hipEventCreate(&ev));
                c->prof_events.push_back(ev);
            }
            PCP_CHECK(hipEventRecord(c->prof_events[2 * it], c->stream));
        }
        if (use_quad()) {
            if (int e = launch_ls_quad<true>(c, mode)) return e;
        } else if (int e = launch_ls(c, mode, true, 0, nullptr)) return e;
        if (c->profile) {
            PCP_CHECK(hipEventRecord(c->prof_events[2 * it + 1], c->stream));
            c->prof_used = it + 1;
        }
        if (int e = launch_finish(c, 1, mode, max_iter, tol, full_iter)) return e;
    }
    hipLaunchKernelGGL(k_apply_pending, dim3(cdiv(c->n_b, 256)), dim3(256), 0, c->stream, c->b64, c->n_b, c->state);
    PCP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_clear_pending, dim3(1), dim3(1), 0, c->stream, c->state);
    PCP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_pose, dim3(1), dim3(PCP_POSE_THREADS), 0, c->stream, finish_args(c));
    PCP_CHECK(hipGetLastError());
    c->have_pose = true;
    return 0;
}

int pcp_pose_from_steps(pcp_ctx* c, const float* dx_steps, int n_steps, double* t_opt3, double* R_opt9) {
    if (int e = use(c)) return e;
    if (!t_opt3 || !R_opt9 || (n_steps > 0 && !dx_steps)) return pcp_fail("null argument");
    if (n_steps < 0 || n_steps > pcp_ctx::max_iter_cap) return pcp_fail("n_steps = %d outside [0, %d]", n_steps, pcp_ctx::max_iter_cap);
    if (!c->pose_steps) {
        PCP_CHECK(hipMalloc(&c->pose_steps, (size_t)pcp_ctx::max_iter_cap * 6 * sizeof(float) + 16 * sizeof(double)));
    }
    double* out = reinterpret_cast<double*>(c->pose_steps);                 // 16 doubles in front, the steps behind them
    float* steps = reinterpret_cast<float*>(out + 16);
    if (n_steps) PCP_CHECK(hipMemcpyAsync(steps, dx_steps, (size_t)n_steps * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_pose_steps, dim3(1), dim3(PCP_POSE_THREADS), 0, c->stream, steps, n_steps, out);
    PCP_CHECK(hipGetLastError());
    double pose[13];
    PCP_CHECK(hipMemcpyAsync(pose, out, sizeof(pose), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; ++k) t_opt3[k] = pose[k];
    for (int k = 0; k < 9; ++k) R_opt9[k] = pose[3 + k];
    if (pose[12] != 0.0) return pcp_fail("the tracking grid's cross-covariance is rank-deficient");
    return 0;
}

int pcp_icp_fetch_pose(pcp_ctx* c, double* t_opt3, double* R_opt9) {
    if (int e = use(c)) return e;
    if (!t_opt3 || !R_opt9) return pcp_fail("null output");
    if (!c->have_pose) return pcp_fail("no finished loop to take a pose from (pcp_icp_launch or a batch launch first)");
    double pose[16];
    PCP_CHECK(hipMemcpyAsync(pose, c->pose, sizeof(pose), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; ++k) t_opt3[k] = pose[k];
    for (int k = 0; k < 9; ++k) R_opt9[k] = pose[3 + k];
    if (pose[0] != pose[0]) return pcp_fail("no pose: the loop met singular normal equations");
    if (pose[12] != 0.0) return pcp_fail("no pose: the tracking grid's cross-covariance is rank-deficient");
    return 0;
}

int pcp_icp_fetch(pcp_ctx* c, float* dx_trace, float* hb_trace, int* iters, int* converged, double* b_out) {
    if (int e = use(c)) return e;
    IcpState st;
    PCP_CHECK(hipMemcpyAsync(&st, c->state, sizeof(IcpState), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    const int n = st.iter;
    if (dx_trace && n > 0)
        PCP_CHECK(hipMemcpyAsync(dx_trace, c->dx_trace, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (hb_trace && n > 0)
        PCP_CHECK(hipMemcpyAsync(hb_trace, c->hb_trace, (size_t)n * 64 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (b_out)
        PCP_CHECK(hipMemcpyAsync(b_out, c->b64, (size_t)c->n_b * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    if (iters) *iters = n;
    if (converged) *converged = st.converged;
    if (st.singular) return pcp_fail("singular normal equations at iteration %d", n - 1);
    return 0;
}

int pcp_icp_run(pcp_ctx* c, const double* b_xyz, int n_b, int mode, int max_iter, float tol, int full_iter,
                float* dx_trace, float* hb_trace, int* iters, int* converged, double* b_out) {
    if (int e = pcp_upload_source(c, b_xyz, n_b)) return e;
    if (int e = pcp_reset_hints(c)) return e;
    if (int e = pcp_icp_launch(c, mode, max_iter, tol, full_iter)) return e;
    return pcp_icp_fetch(c, dx_trace, hb_trace, iters, converged, b_out);
}

int pcp_profile_enable(pcp_ctx* c, int on) {
    if (!c) return pcp_fail("null context");
    c->profile = on != 0;
    return 0;
}

int pcp_profile_read(pcp_ctx* c, float* ls_ms, int* launches) {
    if (int e = use(c)) return e;
    if (!ls_ms || !launches) return pcp_fail("null output");
    PCP_CHECK(hipStreamSynchronize(c->stream));
    IcpState st;
    PCP_CHECK(hipMemcpy(&st, c->state, sizeof(IcpState), hipMemcpyDeviceToHost));
    // launches past the loop's end return at once (done flag); only the dispatches that ran count
    const int n = st.iter < c->prof_used ? st.iter : c->prof_used;
    float total = 0.f;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        PCP_CHECK(hipEventElapsedTime(&ms, c->prof_events[2 * i], c->prof_events[2 * i + 1]));
        total += ms;
    }
    *ls_ms = total;
    *launches = n;
    return 0;
}

int pcp_ring_stats(pcp_ctx* c, float* out) {
    if (int e = use(c)) return e;
    if (!out) return pcp_fail("null output");
    if (!c->have_target) return pcp_fail("no target: call pcp_set_target first");
    PCP_CHECK(hipMemcpyAsync(out, c->ring_stat, (size_t)c->n_scans * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- batches of independent pairs ---------------------------------------------------------

struct pcp_batch {
    int device = 0;
    std::vector<pcp_ctx*> ctxs;
    hipStream_t stream = nullptr;
    LsArgs *ls_host = nullptr, *ls_dev = nullptr;            // pinned host mirror / device array
    FinishArgs *fin_host = nullptr, *fin_dev = nullptr;
    PrepArgs *prep_host = nullptr, *prep_dev = nullptr;
    std::vector<LsArgs> ls_sent;            // what the device arrays hold (skip the copy when unchanged)
    std::vector<FinishArgs> fin_sent;
    std::vector<PrepArgs> prep_sent;
    bool profile = false;
    std::vector<hipEvent_t> prof_events;
    int prof_used = 0;
    int run_max_iter = 0;
};

static void batch_forget(pcp_batch* b, pcp_ctx* c) {
    for (size_t i = 0; i < b->ctxs.size(); ++i)
        if (b->ctxs[i] == c) { b->ctxs.erase(b->ctxs.begin() + (long)i); break; }
    b->ls_sent.clear(); b->fin_sent.clear(); b->prep_sent.clear();      // the descriptor arrays are rebuilt on the next launch
    c->batch = nullptr;
    c->stream = c->own_stream;
}

int pcp_batch_destroy(pcp_batch* b) {
    if (!b) return 0;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (pcp_ctx* c : b->ctxs) { c->stream = c->own_stream; c->batch = nullptr; }
    for (hipEvent_t ev : b->prof_events) (void)hipEventDestroy(ev);
    if (b->ls_host) (void)hipHostFree(b->ls_host);
    if (b->fin_host) (void)hipHostFree(b->fin_host);
    if (b->prep_host) (void)hipHostFree(b->prep_host);
    if (b->prep_dev) (void)hipFree(b->prep_dev);
    if (b->ls_dev) (void)hipFree(b->ls_dev);
    if (b->fin_dev) (void)hipFree(b->fin_dev);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return 0;
}

int pcp_batch_create(pcp_ctx** ctxs, int n, pcp_batch** out) {
    if (!out) return pcp_fail("null out pointer");
    *out = nullptr;
    if (!ctxs || n < 1 || n > 65535) return pcp_fail("a batch holds 1..65535 contexts");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return pcp_fail("null context in batch");
        if (ctxs[i]->device != ctxs[0]->device) return pcp_fail("all contexts of a batch must live on one device");
        if (ctxs[i]->batch) return pcp_fail("context %d already belongs to a batch", i);
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return pcp_fail("context listed twice in a batch");
    }
    PCP_CHECK(hipSetDevice(ctxs[0]->device));
    pcp_batch* b = new pcp_batch();
    b->device = ctxs[0]->device;
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void**)&b->ls_host, (size_t)n * sizeof(LsArgs));
    if (e == hipSuccess) e = hipHostMalloc((void**)&b->fin_host, (size_t)n * sizeof(FinishArgs));
    if (e == hipSuccess) e = hipMalloc((void**)&b->ls_dev, (size_t)n * sizeof(LsArgs));
    if (e == hipSuccess) e = hipMalloc((void**)&b->fin_dev, (size_t)n * sizeof(FinishArgs));
    if (e == hipSuccess) e = hipHostMalloc((void**)&b->prep_host, (size_t)n * sizeof(PrepArgs));
    if (e == hipSuccess) e = hipMalloc((void**)&b->prep_dev, (size_t)n * sizeof(PrepArgs));
    if (e != hipSuccess) {
        pcp_fail("pcp_batch_create: %s", hipGetErrorString(e));
        pcp_batch_destroy(b);
        return 1;
    }
    for (int i = 0; i < n; ++i) {
        e = hipStreamSynchronize(ctxs[i]->stream);             // work queued on the old stream comes first
        if (e != hipSuccess) {
            pcp_fail("pcp_batch_create: %s", hipGetErrorString(e));
            pcp_batch_destroy(b);                              // hands the adopted contexts their own streams back
            return 1;
        }
        ctxs[i]->stream = b->stream;
        ctxs[i]->batch = b;
        b->ctxs.push_back(ctxs[i]);
    }
    *out = b;
    return 0;
}

int pcp_batch_prepare(pcp_batch* b, int with_normals) {
    if (!b) return pcp_fail("null batch");
    PCP_CHECK(hipSetDevice(b->device));
    const int n = (int)b->ctxs.size();
    if (n == 0) return pcp_fail("the batch holds no context any more");
    int max_na = 0, max_pts = 0, max_scans = 0;
    std::vector<PrepArgs> want((size_t)n);
    for (int i = 0; i < n; ++i) {
        pcp_ctx* c = b->ctxs[i];
        if (!c->have_target) return pcp_fail("pair %d: no target", i);
        if (with_normals && c->n_scans < 2) return pcp_fail("pair %d: normals need at least 2 scan lines", i);
        PrepArgs P;
        memset(&P, 0, sizeof(P));
        P.a = c->a; P.ua = c->ua; P.normals = c->normals; P.scan = c->scan; P.stat = c->ring_stat; P.bins = c->ring_bins; P.hint = c->hint;
        c->hinted_upto = 0;                  // k_unit_batch resets the hints
        P.n_a = c->n_a; P.Z = c->n_scans; P.max_points = c->max_points; P.o = c->origin;
        want[i] = P;
        max_na = c->n_a > max_na ? c->n_a : max_na;
        max_pts = c->max_points > max_pts ? c->max_points : max_pts;
        max_scans = c->n_scans > max_scans ? c->n_scans : max_scans;
    }
    if (b->prep_sent.size() != want.size() || memcmp(b->prep_sent.data(), want.data(), want.size() * sizeof(PrepArgs)) != 0) {
        PCP_CHECK(hipStreamSynchronize(b->stream));      // the pinned mirror may still be in flight
        memcpy(b->prep_host, want.data(), want.size() * sizeof(PrepArgs));
        PCP_CHECK(hipMemcpyAsync(b->prep_dev, b->prep_host, (size_t)n * sizeof(PrepArgs), hipMemcpyHostToDevice, b->stream));
        b->prep_sent = want;
    }
    hipLaunchKernelGGL(k_unit_batch, dim3(cdiv(max_pts, 256), n), dim3(256), 0, b->stream, b->prep_dev);
    PCP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_ring_stats_batch, dim3(max_scans, n), dim3(64), 0, b->stream, b->prep_dev);
    PCP_CHECK(hipGetLastError());
    if (with_normals) {
        const size_t sm = (size_t)max_scans * (sizeof(int2) + sizeof(float4));
        hipLaunchKernelGGL(k_normals_batch, dim3(cdiv(max_na, PCP_BLOCK), n), dim3(PCP_BLOCK), sm, b->stream, b->prep_dev);
        PCP_CHECK(hipGetLastError());
        for (pcp_ctx* c : b->ctxs) c->have_normals = true;
    }
    return 0;
}

int pcp_batch_icp_launch(pcp_batch* b, int mode, int max_iter, float tol, int full_iter) {
    if (!b) return pcp_fail("null batch");
    PCP_CHECK(hipSetDevice(b->device));
    if (max_iter < 0 || max_iter > pcp_ctx::max_iter_cap)
        return pcp_fail("max_iter = %d outside [0, %d]", max_iter, pcp_ctx::max_iter_cap);
    const int n = (int)b->ctxs.size();
    if (n == 0) return pcp_fail("the batch holds no context any more");
    int max_nb = 0, max_scans = 0;
    std::vector<LsArgs> want_ls((size_t)n);
    std::vector<FinishArgs> want_fin((size_t)n);
    for (int i = 0; i < n; ++i) {
        pcp_ctx* c = b->ctxs[i];
        if (!c->have_target) return pcp_fail("pair %d: no target", i);
        if (c->n_b < 1) return pcp_fail("pair %d: no source", i);
        if (int e = check_ls(c, mode, 0)) return e;
        memset(&want_ls[i], 0, sizeof(LsArgs));
        memset(&want_fin[i], 0, sizeof(FinishArgs));
        want_ls[i] = ls_args(c, 0, nullptr);
        want_fin[i] = finish_args(c);
        max_nb = c->n_b > max_nb ? c->n_b : max_nb;
        max_scans = c->n_scans > max_scans ? c->n_scans : max_scans;
        c->run_mode = mode;
        c->run_max_iter = max_iter;
    }
    if (b->ls_sent.size() != want_ls.size() ||
        memcmp(b->ls_sent.data(), want_ls.data(), want_ls.size() * sizeof(LsArgs)) != 0 ||
        memcmp(b->fin_sent.data(), want_fin.data(), want_fin.size() * sizeof(FinishArgs)) != 0) {
        PCP_CHECK(hipStreamSynchronize(b->stream));      // the pinned mirrors may still be in flight
        memcpy(b->ls_host, want_ls.data(), want_ls.size() * sizeof(LsArgs));
        memcpy(b->fin_host, want_fin.data(), want_fin.size() * sizeof(FinishArgs));
        PCP_CHECK(hipMemcpyAsync(b->ls_dev, b->ls_host, (size_t)n * sizeof(LsArgs), hipMemcpyHostToDevice, b->stream));
        PCP_CHECK(hipMemcpyAsync(b->fin_dev, b->fin_host, (size_t)n * sizeof(FinishArgs), hipMemcpyHostToDevice, b->stream));
        b->ls_sent = want_ls;
        b->fin_sent = want_fin;
    }
    hipLaunchKernelGGL(k_state_batch, dim3(cdiv(n, 64)), dim3(64), 0, b->stream, b->ls_dev, n, 1);
    PCP_CHECK(hipGetLastError());
    if (max_iter == 0)
        for (pcp_ctx* c : b->ctxs)
            PCP_CHECK(hipMemcpyAsync(c->b64, c->b_in, (size_t)c->n_b * 3 * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
    const dim3 grid(cdiv(max_nb, PCP_BLOCK), n), block(PCP_BLOCK);
    const size_t sm = (size_t)max_scans * (sizeof(int2) + sizeof(float4));
    b->prof_used = 0;
    b->run_max_iter = max_iter;
    for (int it = 0; it < max_iter; ++it) {
        if (b->profile) {
            while ((int)b->prof_events.size() < 2 * (it + 1)) {
                hipEvent_t ev;
                PCP_CHECK(hipEventCreate(&ev));
                b->prof_events.push_back(ev);
            }
            PCP_CHECK(hipEventRecord(b->prof_events[2 * it], b->stream));
        }
        if ((it == 0 && use_pool() >= 1) || use_pool() >= 2) {
            if (mode == PCP_MODE_POINT) hipLaunchKernelGGL((k_ls_batch_first<PCP_MODE_POINT>), grid, block, sm, b->stream, b->ls_dev);
            else hipLaunchKernelGGL((k_ls_batch_first<PCP_MODE_PLANE>), grid, block, sm, b->stream, b->ls_dev);
        } else if (mode == PCP_MODE_POINT) hipLaunchKernelGGL((k_ls_batch<PCP_MODE_POINT>), dim3(cdiv(max_nb, PCP_LS_THREADS), n), dim3(PCP_LS_THREADS), sm, b->stream, b->ls_dev);
        else hipLaunchKernelGGL((k_ls_batch<PCP_MODE_PLANE>), dim3(cdiv(max_nb, PCP_LS_THREADS), n), dim3(PCP_LS_THREADS), sm, b->stream, b->ls_dev);
        PCP_CHECK(hipGetLastError());
        if (b->profile) {
            PCP_CHECK(hipEventRecord(b->prof_events[2 * it + 1], b->stream));
            b->prof_used = it + 1;
        }
        hipLaunchKernelGGL(k_finish_batch, dim3(n), dim3(1024), 0, b->stream, b->fin_dev, mode, max_iter, tol, full_iter);
        PCP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_apply_pending_batch, dim3(cdiv(max_nb, 256), n), dim3(256), 0, b->stream, b->ls_dev);
    PCP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_state_batch, dim3(cdiv(n, 64)), dim3(64), 0, b->stream, b->ls_dev, n, 0);
    PCP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_pose_batch, dim3(n), dim3(PCP_POSE_THREADS), 0, b->stream, b->fin_dev);
    PCP_CHECK(hipGetLastError());
    for (pcp_ctx* c : b->ctxs) c->have_pose = true;
    return 0;
}

int pcp_batch_synchronize(pcp_batch* b) {
    if (!b) return pcp_fail("null batch");
    PCP_CHECK(hipSetDevice(b->device));
    PCP_CHECK(hipStreamSynchronize(b->stream));
    return 0;
}

int pcp_batch_profile_enable(pcp_batch* b, int on) {
    if (!b) return pcp_fail("null batch");
    b->profile = on != 0;
    return 0;
}

int pcp_batch_profile_read(pcp_batch* b, float* ls_ms, int* launches, long long* points) {
    if (!b) return pcp_fail("null batch");
    if (!ls_ms || !launches || !points) return pcp_fail("null output");
    PCP_CHECK(hipSetDevice(b->device));
    PCP_CHECK(hipStreamSynchronize(b->stream));
    // a launch did work for the pairs that had not finished before it
    int longest = 0;
    long long pts = 0;
    for (pcp_ctx* c : b->ctxs) {
        IcpState st;
        PCP_CHECK(hipMemcpy(&st, c->state, sizeof(IcpState), hipMemcpyDeviceToHost));
        longest = st.iter > longest ? st.iter : longest;
        pts += (long long)st.iter * c->n_b;
    }
    const int n = longest < b->prof_used ? longest : b->prof_used;
    float total = 0.f;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        PCP_CHECK(hipEventElapsedTime(&ms, b->prof_events[2 * i], b->prof_events[2 * i + 1]));
        total += ms;
    }
    *ls_ms = total;
    *launches = n;
    *points = pts;
    return 0;
}

#ifdef PCP_STATS
int pcp_stats_read(unsigned long long* out8, int reset) {
    PCP_CHECK(hipDeviceSynchronize());
    PCP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(pcp_stats), 64));
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        PCP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(pcp_stats), z, 64));
    }
    return 0;
}
#endif

#ifdef PCP_STATS
int pcp_stats2_read(unsigned long long* out16, int reset) {
    PCP_CHECK(hipDeviceSynchronize());
    PCP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(pcp_stats2), 128));
    if (reset) {
        unsigned long long z[16] = {0};
        PCP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(pcp_stats2), z, 128));
    }
    return 0;
}
#endif

int pcp_timer_start(pcp_ctx* c) {
    if (int e = use(c)) return e;
    PCP_CHECK(hipEventRecord(c->ev0, c->stream));
    return 0;
}

int pcp_timer_stop(pcp_ctx* c, float* ms) {
    if (int e = use(c)) return e;
    if (!ms) return pcp_fail("null output");
    PCP_CHECK(hipEventRecord(c->ev1, c->stream));
    PCP_CHECK(hipEventSynchronize(c->ev1));
    PCP_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
}

}  // extern "C"

#include "pcp_sequence.inl"
#include "pcp_chain.inl"
