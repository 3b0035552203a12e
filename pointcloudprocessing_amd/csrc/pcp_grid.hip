// Exact nearest neighbours for ring-less clouds and maps, and the registration
// steps built on them.
//
// The reference's exact search is a scipy KD-tree on float64 points
// (PointcloudIcpContainer.getKDTree / getNearestNeighbours, :467-472) feeding the
// trimmed Kabsch step icp_step_KD (:474-498).  Here the target ("map") is binned
// into cubic cells, sorted by the Morton code of its cell with a hand-written LSD
// radix sort, and indexed by an open-addressing hash table of the occupied cells.
// A query walks the cell shells around its own cell outwards until the best
// distance found is smaller than the distance to the unvisited space, so the
// answer is the exact float64 minimum (lowest point index on ties), as the
// KD-tree's.
//
// Device buffers (all float64 unless noted):
//   pts      {x, y, z, id} per map point, sorted by cell key            32 B / point
//   normals  float4 per map point in the same order (optional)          16 B / point
//   table    {key, start, count} per hash slot, capacity >= 2 x cells   16 B / slot
//   src_in / src   source cloud as uploaded (for rewinding) / working copy  24 B / point
//   nn_d, nn_idx, nn_pos, sel    per source point: distance, map point id,
//                                position in pts, membership of the trimmed set
//   partials, lvl0, lvl1, state, hb_trace, dx_trace: as in the ring registration.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include <cmath>

#include <algorithm>
#include <vector>

#include "../../include/pcp.h"
#include "pcp_ctx.h"
#include "pcp_device.h"
#include "pcp_host.h"
#include "pcp_compact.h"
#include "pcp_solve.h"
#include "pcp_sort.h"

using namespace pcp;

namespace {

struct __align__(32) GridPoint { double x, y, z; long long id; };
struct __align__(16) Cell { unsigned long long key; unsigned int start, count; };
constexpr unsigned long long EMPTY_KEY = RS_EMPTY_KEY;
constexpr int COORD_BIAS = 1 << 20;          // absolute cell coordinates live in [-2^20, 2^20)
constexpr int COORD_CLAMP = 1 << 29;

struct GridView {
    const GridPoint* pts;
    const Cell* table;
    int log_cap;
    size_t n;
    double cell, eps;
    int cminx, cminy, cminz;     // absolute cell coordinates of the grid's corner
    int ex, ey, ez;              // grid extent in cells
};

__host__ __device__ inline unsigned long long spread21(unsigned long long v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__host__ __device__ inline unsigned long long morton3(unsigned x, unsigned y, unsigned z) {
    return spread21(x) | (spread21(y) << 1) | (spread21(z) << 2);
}

// Key of a cell from its ABSOLUTE coordinates: zig-zag per axis (0, -1, 1, -2, ... -> 0, 1, 2, 3, ...), then
// bit interleaving.  A function of the cell alone, not of the map's bounding box, so points appended
// later sort into the very order a build of the whole would give them; and the cells of a map around the
// origin still have short keys (few radix passes).
__host__ __device__ inline unsigned zigzag(const int c) {
    return c >= 0 ? ((unsigned)c << 1) : ((((unsigned)(-(c + 1))) << 1) | 1u);
}
__host__ __device__ inline unsigned long long cell_key(const int ax, const int ay, const int az) {
    return morton3(zigzag(ax), zigzag(ay), zigzag(az));
}

__device__ __forceinline__ int cell_of(const double v, const double cell) {
    double f = floor(v / cell);
    f = fmin(fmax(f, -(double)COORD_CLAMP), (double)COORD_CLAMP);
    return (f == f) ? (int)f : COORD_CLAMP;
}

// Slot t = 0, 1, 2, ... of a key's probe sequence: multiplicative hash, linear probing.  (Hashing
// 2 x 2 x 2 blocks of cells to one 128-byte line of the table each, so that the 27 cells of a query
// lie in at most 8 lines, was measured: builds 10-20 % faster, queries up to 1.8x slower -- the cells of
// dense blocks fill their lines and every miss or collision walks a whole line.)
__device__ __forceinline__ unsigned probe_slot(const unsigned long long key, const int log_cap, const unsigned t) {
    return ((unsigned)((key * 0x9E3779B97F4A7C15ull) >> (64 - log_cap)) + t) & ((1u << log_cap) - 1u);
}

// ------------------------------------------------------------------------------------------
// build: cell range, keys, radix sort, hash table, gather
// ------------------------------------------------------------------------------------------

// Grid-stride: every thread folds many points before the wave reduction, so the seven
// global atomics are issued once per wave of a small fixed grid, not once per 64 points.
__global__ void __launch_bounds__(256)
k_cell_range(const double* __restrict__ xyz, size_t n, double cell, int* __restrict__ mm) {
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
    int bad = 0;
    // four points per trip, all twelve loads issued before the first division: a thread's trips are
    // serial, so the loads in flight per trip set the rate
    constexpr int U = 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += U * stride) {
        double v[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + u * stride;
#pragma unroll
            for (int a = 0; a < 3; ++a) v[u][a] = (i < n) ? xyz[3 * i + a] : xyz[3 * i0 + a];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (!(fabs(v[u][a] / cell) < (double)(COORD_BIAS - 2))) bad = 1;
                const int c = cell_of(v[u][a], cell);
                lo[a] = min(lo[a], c);
                hi[a] = max(hi[a], c);
            }
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = min(lo[a], __shfl_down(lo[a], s, 64));
            hi[a] = max(hi[a], __shfl_down(hi[a], s, 64));
        }
        bad |= __shfl_down(bad, s, 64);
    }
    // the block's four waves meet in LDS, so that one set of same-address atomics leaves per block
    __shared__ int part[4][7];
    if ((threadIdx.x & 63) == 0) {
        int* p = part[threadIdx.x >> 6];
#pragma unroll
        for (int a = 0; a < 3; ++a) { p[a] = lo[a]; p[3 + a] = hi[a]; }
        p[6] = bad;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int t = threadIdx.x;
        const int w0 = part[0][t], w1 = part[1][t], w2 = part[2][t], w3 = part[3][t];
        if (t < 3) atomicMin(&mm[t], min(min(w0, w1), min(w2, w3)));
        else if (t < 6) atomicMax(&mm[t], max(max(w0, w1), max(w2, w3)));
        else if (w0 | w1 | w2 | w3) atomicOr(&mm[6], 1);
    }
}

__global__ void k_cell_keys(const double* __restrict__ xyz, size_t n, double cell,
                            unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = cell_key(cell_of(xyz[3 * i + 0], cell), cell_of(xyz[3 * i + 1], cell), cell_of(xyz[3 * i + 2], cell));
    vals[i] = (unsigned)i;
}

__global__ void __launch_bounds__(256)
k_count_cells(const unsigned long long* __restrict__ keys, size_t n, unsigned long long* __restrict__ count) {
    unsigned mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        mine += (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) mine += __shfl_down(mine, s, 64);
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned all = part[0] + part[1] + part[2] + part[3];
        if (all) atomicAdd(count, (unsigned long long)all);
    }
}

// One table entry per run of equal keys, written by the thread of the run's last element: it finds
// the run's first element by doubling steps backwards and a bisection (the keys are sorted), claims
// a slot and stores (start, count) with one 8-byte write.
__global__ void k_insert_cells(const unsigned long long* __restrict__ keys, size_t n, Cell* __restrict__ table, int log_cap) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (i + 1 != n && keys[i] == keys[i + 1])) return;
    const unsigned long long key = keys[i];
    long long hi = (long long)i, lo = -1, step = 1;          // keys[hi] == key, keys[lo] != key (lo = -1: before the array)
    while (hi - step >= 0) {
        if (keys[hi - step] != key) { lo = hi - step; break; }
        hi -= step;
        step <<= 1;
    }
    while (hi - lo > 1) {
        const long long mid = lo + (hi - lo) / 2;
        if (keys[mid] == key) hi = mid; else lo = mid;
    }
    const unsigned long long entry = (unsigned long long)(unsigned)hi | ((unsigned long long)(unsigned)(i + 1 - (size_t)hi) << 32);
    for (unsigned t = 0;; ++t) {
        const unsigned slot = probe_slot(key, log_cap, t);
        if (atomicCAS(&table[slot].key, EMPTY_KEY, key) == EMPTY_KEY) {
            *reinterpret_cast<unsigned long long*>(&table[slot].start) = entry;
            return;
        }
    }
}

__device__ __forceinline__ bool find_cell(const Cell* __restrict__ table, const int log_cap, const unsigned long long key,
                                          unsigned& start, unsigned& count) {
    for (unsigned t = 0;; ++t) {
        const ulonglong2 c = *reinterpret_cast<const ulonglong2*>(&table[probe_slot(key, log_cap, t)]);
        if (c.x == key) { start = (unsigned)c.y; count = (unsigned)(c.y >> 32); return true; }
        if (c.x == EMPTY_KEY) return false;
    }
}

__global__ void k_gather_points(const double* __restrict__ xyz, const unsigned* __restrict__ vals, size_t n,
                                const float4* __restrict__ normals_in, const unsigned* __restrict__ orig, long long first_id,
                                const long long* __restrict__ ids, GridPoint* __restrict__ pts, float4* __restrict__ normals) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned i = vals[j];
    GridPoint p;
    p.x = xyz[3 * (size_t)i + 0]; p.y = xyz[3 * (size_t)i + 1]; p.z = xyz[3 * (size_t)i + 2];
    // index in the caller's array (+ what came before it); collected appends bring every point's id along
    p.id = ids ? ids[i] : first_id + (orig ? (long long)orig[i] : (long long)i);
    pts[j] = p;
    if (normals_in) normals[j] = normals_in[i];
}

// ids of a batch that waits to be merged: first_id + (index in the caller's array)
__global__ void k_pending_ids(const unsigned* __restrict__ orig, long long first_id, size_t n, long long* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = first_id + (orig ? (long long)orig[i] : (long long)i);
}

// Merge of two runs sorted by key into one: every element finds its place by one binary search in the
// OTHER run.  Equal keys keep the old run's elements in front, so ingesting batches one after the other
// leaves a cell's points in the order of their ids, as one stable sort of everything would.
__global__ void __launch_bounds__(256)
k_merge_old(const unsigned long long* __restrict__ ko, size_t n, const unsigned long long* __restrict__ kn, size_t m,
            const GridPoint* __restrict__ po, const float4* __restrict__ no, GridPoint* __restrict__ pout,
            unsigned long long* __restrict__ kout, float4* __restrict__ nout) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = ko[i];
    size_t lo = 0, hi = m;                       // first new element whose key is not below this one's
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (kn[mid] < key) lo = mid + 1; else hi = mid;
    }
    const size_t pos = i + lo;
    pout[pos] = po[i];
    kout[pos] = key;
    if (no) nout[pos] = no[i];
}

__global__ void __launch_bounds__(256)
k_merge_new(const unsigned long long* __restrict__ ko, size_t n, const unsigned long long* __restrict__ kn, size_t m,
            const GridPoint* __restrict__ pn, const float4* __restrict__ nn, GridPoint* __restrict__ pout,
            unsigned long long* __restrict__ kout, float4* __restrict__ nout) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const unsigned long long key = kn[j];
    size_t lo = 0, hi = n;                       // first old element whose key is above this one's
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (ko[mid] <= key) lo = mid + 1; else hi = mid;
    }
    const size_t pos = j + lo;
    pout[pos] = pn[j];
    kout[pos] = key;
    if (nn) nout[pos] = nn[j];
}

// ------------------------------------------------------------------------------------------
// exact nearest neighbour
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ double axis_gap(const double q, const double lo, const double hi, const double eps) {
    const double g = fmax(fmax(lo - q, q - hi), 0.0) - eps;
    return fmax(g, 0.0);
}

// Cells with more points than this are not scanned by the lane that meets them but handed to the
// whole wave (grid_nn): a lane's serial walk through a crowded cell would otherwise set the time of
// its wave, and with it of the launch.
constexpr unsigned CROWDED = 96;
constexpr int DEFER_SLOTS = 4;

struct Deferred {              // crowded cells a lane met in the current shell
    unsigned start[DEFER_SLOTS], count[DEFER_SLOTS];
    double boxd2[DEFER_SLOTS];
    int n;
};

__device__ __forceinline__ void take_point(const GridPoint& p, const unsigned j, const double qx, const double qy,
                                           const double qz, double& best, long long& bid, int& bpos) {
    const double dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    const double d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 < best || (d2 == best && p.id < bid)) { best = d2; bid = p.id; bpos = (int)j; }
}

__device__ __forceinline__ void visit_cell(const GridView& G, const int x, const int y, const int z,
                                           const double qx, const double qy, const double qz,
                                           double& best, long long& bid, int& bpos, Deferred& D) {
    // no point of the cell can be nearer than the cell's box
    const double gx = axis_gap(qx, (double)(x + G.cminx) * G.cell, (double)(x + G.cminx + 1) * G.cell, G.eps);
    const double gy = axis_gap(qy, (double)(y + G.cminy) * G.cell, (double)(y + G.cminy + 1) * G.cell, G.eps);
    const double gz = axis_gap(qz, (double)(z + G.cminz) * G.cell, (double)(z + G.cminz + 1) * G.cell, G.eps);
    const double boxd2 = (gx * gx + gy * gy) + gz * gz;
    if (boxd2 > best) return;
    unsigned start, count;
    if (!find_cell(G.table, G.log_cap, cell_key(x + G.cminx, y + G.cminy, z + G.cminz), start, count)) return;
    if (count > CROWDED && D.n < DEFER_SLOTS) {
#pragma unroll
        for (int e = 0; e < DEFER_SLOTS; ++e)
            if (e == D.n) { D.start[e] = start; D.count[e] = count; D.boxd2[e] = boxd2; }
        ++D.n;
        return;
    }
    for (unsigned j = start; j < start + count; ++j) take_point(G.pts[j], j, qx, qy, qz, best, bid, bpos);
}

// Exact nearest map point of (qx, qy, qz).  EVERY lane of the wave must call this together (has =
// this lane holds a query): the shells of cells are walked lane by lane, but the crowded cells a
// shell turns up are scanned by all 64 lanes for one owner at a time.  best must hold the squared
// search radius (or +inf) on entry; only strictly nearer points are accepted; bid = -1 when none is.
__device__ __forceinline__ void grid_nn(const GridView& G, const bool has, const double qx, const double qy,
                                        const double qz, double& best, long long& bid, int& bpos) {
    const int lane = threadIdx.x & 63;
    bid = -1;
    bpos = -1;
    const int ax = cell_of(qx, G.cell), ay = cell_of(qy, G.cell), az = cell_of(qz, G.cell);
    const int cx = ax - G.cminx, cy = ay - G.cminy, cz = az - G.cminz;
    auto outside = [](int c, int e) { return c < 0 ? -c : (c > e - 1 ? c - (e - 1) : 0); };
    auto reach = [](int c, int e) { return max(c, e - 1 - c); };
    int k = max(outside(cx, G.ex), max(outside(cy, G.ey), outside(cz, G.ez)));
    const int k1 = max(reach(cx, G.ex), max(reach(cy, G.ey), reach(cz, G.ez)));
    bool live = has && G.n != 0 && k <= k1;
    while (__any(live)) {
        Deferred D;
        D.n = 0;
        if (live) {
            const int z0 = max(cz - k, 0), z1 = min(cz + k, G.ez - 1);
            const int y0 = max(cy - k, 0), y1 = min(cy + k, G.ey - 1);
            const int x0 = max(cx - k, 0), x1 = min(cx + k, G.ex - 1);
            for (int z = z0; z <= z1; ++z) {
                const bool zedge = (z == cz - k) || (z == cz + k);
                for (int y = y0; y <= y1; ++y) {
                    if (zedge || y == cy - k || y == cy + k) {
                        for (int x = x0; x <= x1; ++x) visit_cell(G, x, y, z, qx, qy, qz, best, bid, bpos, D);
                    } else {
                        if (cx - k >= 0 && cx - k < G.ex) visit_cell(G, cx - k, y, z, qx, qy, qz, best, bid, bpos, D);
                        if (cx + k >= 0 && cx + k < G.ex) visit_cell(G, cx + k, y, z, qx, qy, qz, best, bid, bpos, D);
                    }
                }
            }
        }
        // crowded cells: one owner at a time, all lanes scan, the owner keeps the result
        unsigned long long pending = __ballot(D.n > 0);
        while (pending) {
            const int owner = __ffsll((long long)pending) - 1;
            pending &= pending - 1;
            const int on = __shfl(D.n, owner, 64);
            const double ox = __shfl(qx, owner, 64), oy = __shfl(qy, owner, 64), oz = __shfl(qz, owner, 64);
#pragma unroll
            for (int e = 0; e < DEFER_SLOTS; ++e) {
                if (e >= on) break;                                        // wave-uniform
                const double obest = __shfl(best, owner, 64);              // tightened by the entries before
                if (__shfl(D.boxd2[e], owner, 64) > obest) continue;
                const unsigned start = __shfl(D.start[e], owner, 64), count = __shfl(D.count[e], owner, 64);
                double lb = obest;
                long long lid = __shfl(bid, owner, 64);
                int lpos = -1;
                for (unsigned j = start + lane; j < start + count; j += 64)
                    take_point(G.pts[j], j, ox, oy, oz, lb, lid, lpos);
                if (lpos < 0) lb = INFINITY;                               // nothing better in this lane's share
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) {
                    const double ob = __shfl_xor(lb, sft, 64);
                    const long long oi = __shfl_xor(lid, sft, 64);
                    const int op = __shfl_xor(lpos, sft, 64);
                    if (ob < lb || (ob == lb && op >= 0 && (lpos < 0 || oi < lid))) { lb = ob; lid = oi; lpos = op; }
                }
                if (lane == owner && lpos >= 0) { best = lb; bid = lid; bpos = lpos; }
            }
        }
        if (live) {
            // everything not yet visited lies outside the cube of cells [c-k, c+k]^3
            const double mx = fmin(qx - (double)(ax - k) * G.cell, (double)(ax + k + 1) * G.cell - qx);
            const double my = fmin(qy - (double)(ay - k) * G.cell, (double)(ay + k + 1) * G.cell - qy);
            const double mz = fmin(qz - (double)(az - k) * G.cell, (double)(az + k + 1) * G.cell - qz);
            const double m = fmin(mx, fmin(my, mz)) - G.eps;
            ++k;
            if ((m > 0.0 && best < m * m) || k > k1) live = false;
        }
    }
}

__global__ void __launch_bounds__(256)
k_grid_nn(const GridView G, const double* __restrict__ src, size_t m, double max_d2,
          double* __restrict__ nn_d, long long* __restrict__ nn_idx, int* __restrict__ nn_pos) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool has = i < m;
    double best = max_d2;
    long long bid;
    int bpos;
    const size_t at = has ? i : 0;
    grid_nn(G, has, src[3 * at + 0], src[3 * at + 1], src[3 * at + 2], best, bid, bpos);
    if (!has) return;
    nn_d[i] = (bid >= 0) ? sqrt(best) : INFINITY;
    nn_idx[i] = bid;
    nn_pos[i] = bpos;
}


// ------------------------------------------------------------------------------------------
// The search with 32 lanes per query.  One lane per query leaves a scan of ~10^5 queries with a few
// waves per SIMD, each walking a chain of dependent look-ups; here the cells of a shell are dealt out
// to the 32 lanes of a half-wave, every lane keeps its own best, and a butterfly merges them (lowest
// index on ties) before the shell's termination test.  Same result, 32 times the waves.
// ------------------------------------------------------------------------------------------

struct Nn32Args {
    GridView G;
    double* src;               // m x 3
    size_t m;
    double max_d2;
    IcpState* st;              // non-null: registration loop (apply the pending step, honour done)
    double own_cell;
    unsigned long long own_lo, own_hi;
    double* nn_d;
    long long* nn_idx;
    int* nn_pos;
};

#ifndef PCP_GRID_NN_WAVES
#define PCP_GRID_NN_WAVES 6
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PCP_GRID_NN_WAVES, PCP_GRID_NN_WAVES))) k_grid_nn32(const Nn32Args A) {
    const GridView& G = A.G;
    __shared__ uint2 s_seg[8][32];          // per half-wave: (first point, offset in the dealt list) of 32 cells
    if (A.st && A.st->done) return;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t qi = gid >> 5;
    const int sub = (int)(gid & 31);
    const int lane = threadIdx.x & 63;
    const int leader = lane & 32;                       // lane of the half-wave holding the query
    const bool has = qi < A.m;
    double qx = 0.0, qy = 0.0, qz = 0.0;
    int owned = 0;
    if (has && sub == 0) {
        qx = A.src[3 * qi + 0]; qy = A.src[3 * qi + 1]; qz = A.src[3 * qi + 2];
        if (A.st && A.st->pending) {
            const IcpState* st = A.st;
            const double nx = ((st->R[0] * qx + st->R[1] * qy) + st->R[2] * qz) - st->t[0];
            const double ny = ((st->R[3] * qx + st->R[4] * qy) + st->R[5] * qz) - st->t[1];
            const double nz = ((st->R[6] * qx + st->R[7] * qy) + st->R[8] * qz) - st->t[2];
            qx = nx; qy = ny; qz = nz;
            A.src[3 * qi + 0] = qx; A.src[3 * qi + 1] = qy; A.src[3 * qi + 2] = qz;
        }
        const unsigned long long key = morton3((unsigned)(cell_of(qx, A.own_cell) + COORD_BIAS),
                                               (unsigned)(cell_of(qy, A.own_cell) + COORD_BIAS),
                                               (unsigned)(cell_of(qz, A.own_cell) + COORD_BIAS));
        owned = (key >= A.own_lo && key < A.own_hi) ? 1 : 0;
    }
    qx = __shfl(qx, leader, 64); qy = __shfl(qy, leader, 64); qz = __shfl(qz, leader, 64);
    owned = __shfl(owned, leader, 64);
    double best = A.max_d2;
    long long bid = -1;
    int bpos = -1;
    const int ax = cell_of(qx, G.cell), ay = cell_of(qy, G.cell), az = cell_of(qz, G.cell);
    const int cx = ax - G.cminx, cy = ay - G.cminy, cz = az - G.cminz;
    auto outside = [](int c, int e) { return c < 0 ? -c : (c > e - 1 ? c - (e - 1) : 0); };
    auto reach = [](int c, int e) { return max(c, e - 1 - c); };
    int k = max(outside(cx, G.ex), max(outside(cy, G.ey), outside(cz, G.ez)));
    const int k1 = max(reach(cx, G.ex), max(reach(cy, G.ey), reach(cz, G.ez)));
    bool live = has && owned && G.n != 0 && k <= k1;     // uniform over the half-wave
    int walked = k;                // cells nearer than this many cells (Chebyshev) were walked by earlier passes
    while (live) {
        const int z0 = max(cz - k, 0), z1 = min(cz + k, G.ez - 1);
        const int y0 = max(cy - k, 0), y1 = min(cy + k, G.ey - 1);
        const int x0 = max(cx - k, 0), x1 = min(cx + k, G.ex - 1);
        const long long nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
        const long long total = (nx > 0 && ny > 0 && nz > 0) ? nx * ny * nz : 0;
        if (total == 1) {
            // a shell of one cell (the query's own): its points, not the cells, are dealt out to the lanes,
            // so the shells after it start from a tight best distance
            unsigned start, count;
            if (find_cell(G.table, G.log_cap, cell_key(x0 + G.cminx, y0 + G.cminy, z0 + G.cminz), start, count))
                for (unsigned j = start + sub; j < start + count; j += 32) take_point(G.pts[j], j, qx, qy, qz, best, bid, bpos);
        } else {
            // 32 cells at a time: every lane looks ONE cell up, then the points of the cells found are
            // dealt out evenly over the 32 lanes (an exclusive scan of the counts and a search of that
            // scan in LDS tell a lane which cell its e-th point belongs to).  A lane walking its own
            // cell's points one after the other would make the fullest cell set the half-wave's time.
            for (long long p0 = 0; p0 < total; p0 += 32) {                  // uniform over the half-wave
                const long long p = p0 + sub;
                unsigned start = 0, count = 0;
                if (p < total) {
                    int x, y, z;
                    if (total <= 0x7fffffffll) {        // 32-bit division where it suffices (64-bit is emulated)
                        const unsigned pu = (unsigned)p, ux = (unsigned)nx, uxy = (unsigned)(nx * ny);
                        const unsigned zz = pu / uxy, r = pu - zz * uxy, yy = r / ux;
                        x = x0 + (int)(r - yy * ux); y = y0 + (int)yy; z = z0 + (int)zz;
                    } else {
                        x = x0 + (int)(p % nx); y = y0 + (int)((p / nx) % ny); z = z0 + (int)(p / (nx * ny));
                    }
                    if (max(abs(x - cx), max(abs(y - cy), abs(z - cz))) >= walked) {      // not an earlier shell
                        const double gx = axis_gap(qx, (double)(x + G.cminx) * G.cell, (double)(x + G.cminx + 1) * G.cell, G.eps);
                        const double gy = axis_gap(qy, (double)(y + G.cminy) * G.cell, (double)(y + G.cminy + 1) * G.cell, G.eps);
                        const double gz = axis_gap(qz, (double)(z + G.cminz) * G.cell, (double)(z + G.cminz + 1) * G.cell, G.eps);
                        if (!((gx * gx + gy * gy) + gz * gz > best))
                            if (!find_cell(G.table, G.log_cap, cell_key(x + G.cminx, y + G.cminy, z + G.cminz), start, count)) count = 0;
                    }
                }
                unsigned inc = count;
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    const unsigned o = __shfl_up(inc, d, 32);
                    if (sub >= d) inc += o;
                }
                const unsigned tot = __shfl(inc, 31, 32);
                if (tot == 0) continue;
                uint2* const seg = s_seg[threadIdx.x >> 5];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the previous round's reads are done
                seg[sub] = make_uint2(start, inc - count);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                auto locate = [&](const unsigned e) {                       // point e of the dealt list
                    int lo = 0, hi = 31;                                     // last cell whose offset is <= e
                    uint2 sg = seg[0];
#pragma unroll
                    for (int it = 0; it < 5; ++it) {
                        const int mid = (lo + hi + 1) >> 1;
                        const uint2 c = seg[mid];
                        if (c.y <= e) { lo = mid; sg = c; } else hi = mid - 1;
                    }
                    return sg.x + (e - sg.y);
                };
                for (unsigned e = sub; e < tot; e += 64) {                   // two points in flight per lane
                    const bool two = e + 32 < tot;
                    const unsigned j1 = locate(e), j2 = two ? locate(e + 32) : j1;
                    const GridPoint p1 = G.pts[j1], p2 = G.pts[j2];
                    take_point(p1, j1, qx, qy, qz, best, bid, bpos);
                    if (two) take_point(p2, j2, qx, qy, qz, best, bid, bpos);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#pragma unroll
        for (int sft = 16; sft > 0; sft >>= 1) {
            const double ob = __shfl_xor(best, sft, 64);
            const long long oi = __shfl_xor(bid, sft, 64);
            const int op = __shfl_xor(bpos, sft, 64);
            if (oi >= 0 && (bid < 0 || ob < best || (ob == best && oi < bid))) { best = ob; bid = oi; bpos = op; }
        }
        const double mx = fmin(qx - (double)(ax - k) * G.cell, (double)(ax + k + 1) * G.cell - qx);
        const double my = fmin(qy - (double)(ay - k) * G.cell, (double)(ay + k + 1) * G.cell - qy);
        const double mz = fmin(qz - (double)(az - k) * G.cell, (double)(az + k + 1) * G.cell - qz);
        const double mrg = fmin(mx, fmin(my, mz)) - G.eps;
        ++k;
        walked = k;
        if ((mrg > 0.0 && best < mrg * mrg) || k > k1) live = false;
    }
    if (has && sub == 0) {
        A.nn_d[qi] = (bid >= 0) ? sqrt(best) : INFINITY;
        A.nn_idx[qi] = bid;
        A.nn_pos[qi] = bpos;
    }
}

// ------------------------------------------------------------------------------------------
// trimmed Kabsch step (icp_step_KD): selection of the `taken` smallest distances, sums
// ------------------------------------------------------------------------------------------

struct SelState { unsigned long long prefix, remaining, thr, take_equal; };

__global__ void k_sel_init(SelState* s, unsigned long long taken) {
    s->prefix = 0; s->remaining = taken; s->thr = 0; s->take_equal = 0;
}

__global__ void __launch_bounds__(256)
k_sel_hist(const double* __restrict__ d, size_t m, int shift, const SelState* __restrict__ s, unsigned* __restrict__ hist) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = s->prefix;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(d[i]);
        const bool match = (shift == 56) || ((b >> (shift + 8)) == (prefix >> (shift + 8)));
        if (match) atomicAdd(&h[(unsigned)(b >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void k_sel_pick(unsigned* __restrict__ hist, SelState* __restrict__ s, int shift) {
    if (threadIdx.x == 0 && s->remaining > 0) {
        unsigned long long cum = 0;
        for (int dgt = 0; dgt < 256; ++dgt) {
            const unsigned long long c = hist[dgt];
            if (cum + c >= s->remaining) {
                s->remaining -= cum;
                s->prefix |= (unsigned long long)dgt << shift;
                break;
            }
            cum += c;
        }
        if (shift == 0) { s->thr = s->prefix; s->take_equal = s->remaining; }
    }
    __syncthreads();
    hist[threadIdx.x] = 0;
}

// sel[i] = 1 for the `taken` smallest distances; equal distances at the threshold are
// taken in index order.  One block walks the array in order.
__global__ void __launch_bounds__(1024)
k_sel_mark(const double* __restrict__ d, size_t m, const SelState* __restrict__ s, unsigned long long taken,
           unsigned char* __restrict__ sel) {
    __shared__ unsigned wsum[16];
    __shared__ unsigned long long running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const unsigned long long thr = s->thr, take_equal = s->take_equal;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (size_t base = 0; base < m; base += 1024) {
        const size_t i = base + threadIdx.x;
        const unsigned long long b = (i < m) ? (unsigned long long)__double_as_longlong(d[i]) : ~0ull;
        const bool eq = (i < m) && b == thr;
        const unsigned long long bal = __ballot(eq);
        const unsigned in_wave = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = 0, total = 0;
        for (int j = 0; j < 16; ++j) { if (j < w) before += wsum[j]; total += wsum[j]; }
        const unsigned long long rank = running + before + in_wave;
        if (i < m) sel[i] = (taken > 0 && (b < thr || (eq && rank < take_equal))) ? 1 : 0;
        __syncthreads();
        if (threadIdx.x == 0) running += total;
        __syncthreads();
    }
}

// PASS 0: sums of the matched map points, of the source points, and the count (7 values).
// PASS 1: W[i][j] = sum (ref_i - mean_ref_i)(other_j - mean_other_j) (9 values); means at mean[0..6).
constexpr int KD_BLOCKS = 512;
template <int PASS>
__global__ void __launch_bounds__(256)
k_kd_sum(const GridPoint* __restrict__ pts, const int* __restrict__ nn_pos, const double* __restrict__ src,
         const unsigned char* __restrict__ sel, size_t m, const double* __restrict__ mean, double* __restrict__ partial) {
    constexpr int V = PASS == 0 ? 7 : 9;
    __shared__ double red[256];
    double acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.0;
    double mr[3] = {0, 0, 0}, mo[3] = {0, 0, 0};
    if (PASS == 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { mr[a] = mean[a]; mo[a] = mean[3 + a]; }
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
        const int p = nn_pos[i];
        if (!sel[i] || p < 0) continue;
        const double r[3] = {pts[p].x, pts[p].y, pts[p].z};
        const double o[3] = {src[3 * i + 0], src[3 * i + 1], src[3 * i + 2]};
        if (PASS == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { acc[a] += r[a]; acc[3 + a] += o[a]; }
            acc[6] += 1.0;
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) acc[a * 3 + b] += (r[a] - mr[a]) * (o[b] - mo[b]);
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        red[threadIdx.x] = acc[v];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < (unsigned)s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(size_t)blockIdx.x * 16 + v] = red[0];
        __syncthreads();
    }
}

// out[0..V) = sums over the blocks in block order; PASS 0 also writes the six means to out[16..22).
template <int PASS>
__global__ void k_kd_final(const double* __restrict__ partial, int blocks, double* __restrict__ out) {
    constexpr int V = PASS == 0 ? 7 : 9;
    __shared__ double s[16];
    const int v = threadIdx.x;
    if (v < V) {
        double acc = 0.0;
        for (int b = 0; b < blocks; ++b) acc += partial[(size_t)b * 16 + v];
        s[v] = acc;
        out[PASS * 32 + v] = acc;
    }
    __syncthreads();
    if (PASS == 0 && v < 6) out[16 + v] = s[v] / s[6];
}

// p <- R p + sign * t on the working copy (applyIcpStep, PointcloudIcpContainer.py:315-317, sign = +1;
// apply_iteration_of_ls, :172-174, sign = -1).
__global__ void k_src_apply(double* __restrict__ src, size_t m, const double* __restrict__ Rt, double sign) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double px = src[3 * i + 0], py = src[3 * i + 1], pz = src[3 * i + 2];
    src[3 * i + 0] = ((Rt[0] * px + Rt[1] * py) + Rt[2] * pz) + sign * Rt[9];
    src[3 * i + 1] = ((Rt[3] * px + Rt[4] * py) + Rt[5] * pz) + sign * Rt[10];
    src[3 * i + 2] = ((Rt[6] * px + Rt[7] * py) + Rt[8] * pz) + sign * Rt[11];
}

// ------------------------------------------------------------------------------------------
// Normals of a map that has no scan rings (a PLY cloud, an accumulated map): the reference's
// NormalShader.glsl needs ring neighbours, so for such clouds the normal of a map point is the
// direction of least spread of the map points strictly within `radius` of it (the point itself
// included): smallest eigenvector of their scatter matrix, float64, turned towards a viewpoint like
// normalTowardsOrigin (NormalShader.glsl:62-71) when one is given.  A half-wave serves one map point:
// its lanes look up the cells the ball can touch (32 per round), the points of the cells found are
// read 32 at a time, ten sums per lane meet in a butterfly.
// ------------------------------------------------------------------------------------------

struct PcaArgs {
    GridView G;
    double radius2;
    int reach;                 // cells to either side of the own cell that the ball can touch
    int min_points;
    int towards;
    double vx, vy, vz;
    float4* normals;           // in the map's sorted order
    unsigned* counts;          // points within the radius, or null
};

// One Jacobi rotation of the symmetric 3x3 matrix in the (p, q) plane; r is the third index.
__device__ __forceinline__ void jacobi_rotate(double& app, double& aqq, double& apq, double& arp, double& arq,
                                              double (&vp)[3], double (&vq)[3]) {
    if (apq == 0.0) return;
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = copysign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    app -= t * apq;
    aqq += t * apq;
    apq = 0.0;
    const double rp = c * arp - s * arq, rq = s * arp + c * arq;
    arp = rp;
    arq = rq;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double a = c * vp[k] - s * vq[k], b = s * vp[k] + c * vq[k];
        vp[k] = a;
        vq[k] = b;
    }
}

__global__ void __launch_bounds__(256) k_grid_pca(const PcaArgs A) {
    const GridView& G = A.G;
    const int lane = threadIdx.x & 31;
    const size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;       // one map point per half-wave
    if (j >= G.n) return;
    const GridPoint q = G.pts[j];
    const int cx = cell_of(q.x, G.cell) - G.cminx, cy = cell_of(q.y, G.cell) - G.cminy, cz = cell_of(q.z, G.cell) - G.cminz;
    const int w = 2 * A.reach + 1, ncell = w * w * w;
    double s[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) s[k] = 0.0;
    for (int base = 0; base < ncell; base += 32) {
        const int ci = base + lane;
        unsigned start = 0, count = 0;
        if (ci < ncell) {
            const int x = cx + ci % w - A.reach, y = cy + (ci / w) % w - A.reach, z = cz + ci / (w * w) - A.reach;
            if (x >= 0 && x < G.ex && y >= 0 && y < G.ey && z >= 0 && z < G.ez) {
                const double gx = axis_gap(q.x, (double)(x + G.cminx) * G.cell, (double)(x + G.cminx + 1) * G.cell, G.eps);
                const double gy = axis_gap(q.y, (double)(y + G.cminy) * G.cell, (double)(y + G.cminy + 1) * G.cell, G.eps);
                const double gz = axis_gap(q.z, (double)(z + G.cminz) * G.cell, (double)(z + G.cminz + 1) * G.cell, G.eps);
                if ((gx * gx + gy * gy) + gz * gz < A.radius2)
                    if (!find_cell(G.table, G.log_cap, cell_key(x + G.cminx, y + G.cminy, z + G.cminz), start, count)) count = 0;
            }
        }
        const int in_round = min(32, ncell - base);
        for (int c = 0; c < in_round; ++c) {
            const unsigned cs = __shfl(start, c, 32), cn = __shfl(count, c, 32);
            for (unsigned jj = cs + lane; jj < cs + cn; jj += 32) {
                const GridPoint p = G.pts[jj];
                const double dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                if ((dx * dx + dy * dy) + dz * dz < A.radius2) {
                    s[0] += 1.0;
                    s[1] += dx; s[2] += dy; s[3] += dz;
                    s[4] += dx * dx; s[5] += dx * dy; s[6] += dx * dz;
                    s[7] += dy * dy; s[8] += dy * dz; s[9] += dz * dz;
                }
            }
        }
    }
#pragma unroll
    for (int sft = 16; sft > 0; sft >>= 1) {
#pragma unroll
        for (int k = 0; k < 10; ++k) s[k] += __shfl_xor(s[k], sft, 32);
    }
    if (lane != 0) return;
    if (A.counts) A.counts[j] = (unsigned)s[0];
    float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (s[0] >= (double)A.min_points) {
        const double inv = 1.0 / s[0];
        double a00 = s[4] - s[1] * s[1] * inv, a01 = s[5] - s[1] * s[2] * inv, a02 = s[6] - s[1] * s[3] * inv;
        double a11 = s[7] - s[2] * s[2] * inv, a12 = s[8] - s[2] * s[3] * inv, a22 = s[9] - s[3] * s[3] * inv;
        double v0[3] = {1.0, 0.0, 0.0}, v1[3] = {0.0, 1.0, 0.0}, v2[3] = {0.0, 0.0, 1.0};
        for (int sweep = 0; sweep < 12; ++sweep) {
            if (a01 == 0.0 && a02 == 0.0 && a12 == 0.0) break;
            jacobi_rotate(a00, a11, a01, a02, a12, v0, v1);
            jacobi_rotate(a00, a22, a02, a01, a12, v0, v2);
            jacobi_rotate(a11, a22, a12, a01, a02, v1, v2);
        }
        double nx = v0[0], ny = v0[1], nz = v0[2], lam = a00;
        if (a11 < lam) { nx = v1[0]; ny = v1[1]; nz = v1[2]; lam = a11; }
        if (a22 < lam) { nx = v2[0]; ny = v2[1]; nz = v2[2]; lam = a22; }
        const double len = sqrt((nx * nx + ny * ny) + nz * nz);
        nx /= len; ny /= len; nz /= len;
        bool flip;
        if (A.towards) flip = ((A.vx - q.x) * nx + (A.vy - q.y) * ny) + (A.vz - q.z) * nz < 0.0;
        else {                     // no viewpoint: the largest component is made positive
            const double ax = fabs(nx), ay = fabs(ny), az = fabs(nz);
            flip = (ax >= ay && ax >= az) ? nx < 0.0 : ((ay >= az) ? ny < 0.0 : nz < 0.0);
        }
        if (flip) { nx = -nx; ny = -ny; nz = -nz; }
        if (len > 0.0 && len == len) out = make_float4((float)nx, (float)ny, (float)nz, 0.0f);
    }
    A.normals[j] = out;
}

// ------------------------------------------------------------------------------------------
// scan-to-map least squares: exact neighbour within the search radius, then the
// tile arithmetic of LeastSquaresPoint.glsl / LeastSquaresPlane.glsl on the float32
// casts, summed by the ParalellHBSum tree (one wave = one shader workgroup).
// Only source points whose cell key lies in [own_lo, own_hi) contribute: with the
// key space split over ranks and a one-cell halo of map points on every rank, the
// rank partials add up to the whole sum.
// ------------------------------------------------------------------------------------------

struct MapLsArgs {
    GridView G;
    const float4* normals;
    double* src;
    IcpState* st;
    size_t m;
    double radius2, own_cell;
    unsigned long long own_lo, own_hi;
    float* partials;
    long long* nn_idx;        // optional per-point outputs
    double* nn_d;
};

template <int MODE>
__global__ void __launch_bounds__(PCP_BLOCK) k_map_ls(const MapLsArgs A) {
    IcpState* __restrict__ st = A.st;
    if (st->done) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < A.m;
    float v[PCP_NVAL];
#pragma unroll
    for (int k = 0; k < PCP_NVAL; ++k) v[k] = 0.0f;
    double px = 0.0, py = 0.0, pz = 0.0;
    bool owned = false;
    if (in) {
        px = A.src[3 * i + 0]; py = A.src[3 * i + 1]; pz = A.src[3 * i + 2];
        if (st->pending) {
            const double nx = ((st->R[0] * px + st->R[1] * py) + st->R[2] * pz) - st->t[0];
            const double ny = ((st->R[3] * px + st->R[4] * py) + st->R[5] * pz) - st->t[1];
            const double nz = ((st->R[6] * px + st->R[7] * py) + st->R[8] * pz) - st->t[2];
            px = nx; py = ny; pz = nz;
            A.src[3 * i + 0] = px; A.src[3 * i + 1] = py; A.src[3 * i + 2] = pz;
        }
        const unsigned long long key = morton3((unsigned)(cell_of(px, A.own_cell) + COORD_BIAS),
                                               (unsigned)(cell_of(py, A.own_cell) + COORD_BIAS),
                                               (unsigned)(cell_of(pz, A.own_cell) + COORD_BIAS));
        owned = key >= A.own_lo && key < A.own_hi;
    }
    long long bid = -1;
    int bpos = -1;
    double best = A.radius2;
    grid_nn(A.G, owned, px, py, pz, best, bid, bpos);          // the whole wave takes part
    if (in) {
        if (A.nn_idx) { A.nn_idx[i] = bid; A.nn_d[i] = (bid >= 0) ? sqrt(best) : INFINITY; }
        if (bid >= 0) {
            const GridPoint p = A.G.pts[bpos];
            const float4 pa = make_float4((float)p.x, (float)p.y, (float)p.z, 0.0f);
            const float4 q = make_float4((float)px, (float)py, (float)pz, 0.0f);
            const float d = distsq(q, pa);
            if (MODE == PCP_MODE_POINT) ls_point_values(pa, q, d, v);
            else ls_plane_values(pa, q, A.normals[bpos], d, v);
        }
    }
    const int lane = threadIdx.x & 63;
    const size_t wave = i >> 6;
    if ((wave << 6) < A.m) {
        const float mine = wave_tree_sum_all(v, lane);      // lane k: value k of the wave's tree sum
        if (lane < PCP_PSTRIDE) A.partials[wave * PCP_PSTRIDE + lane] = mine;
    }
}

// Second stage of a scan-to-map iteration: tile values from the neighbours k_grid_nn32 found.
template <int MODE>
__global__ void __launch_bounds__(PCP_BLOCK)
k_map_tiles(const GridView G, const float4* __restrict__ normals, const double* __restrict__ src,
            const int* __restrict__ nn_pos, const IcpState* __restrict__ st, size_t m, float* __restrict__ partials) {
    if (st->done) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float v[PCP_NVAL];
#pragma unroll
    for (int k = 0; k < PCP_NVAL; ++k) v[k] = 0.0f;
    if (i < m) {
        const int bpos = nn_pos[i];
        if (bpos >= 0) {
            const GridPoint p = G.pts[bpos];
            const float4 pa = make_float4((float)p.x, (float)p.y, (float)p.z, 0.0f);
            const float4 q = make_float4((float)src[3 * i + 0], (float)src[3 * i + 1], (float)src[3 * i + 2], 0.0f);
            const float d = distsq(q, pa);
            if (MODE == PCP_MODE_POINT) ls_point_values(pa, q, d, v);
            else ls_plane_values(pa, q, normals[bpos], d, v);
        }
    }
    const int lane = threadIdx.x & 63;
    const size_t wave = i >> 6;
    if ((wave << 6) < m) {
        const float mine = wave_tree_sum_all(v, lane);      // lane k: value k of the wave's tree sum
        if (lane < PCP_PSTRIDE) partials[wave * PCP_PSTRIDE + lane] = mine;
    }
}

__global__ void __launch_bounds__(1024)
k_map_finish(const FinishArgs F, int solve, int mode, int max_iter, float tol, int full_iter) {
    finish_body(F, solve, mode, max_iter, tol, full_iter);
}

// The host half of one iteration from an already summed (all-reduced) 8x8 tile.
__global__ void k_map_pose(const float* __restrict__ tile, IcpState* __restrict__ st, float* __restrict__ hb_trace,
                           float* __restrict__ dx_trace, int mode, int max_iter, float tol, int full_iter) {
    if (threadIdx.x >= 64 || st->done) return;       // one wave: pose_step spreads the step's trigonometry over its lanes
    const int it = st->iter;
    float v[PCP_NVAL];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = r; c < 6; ++c) v[tri(r, c)] = tile[r * 8 + c];
        v[21 + r] = tile[r * 8 + 6];
    }
    hb_trace[(size_t)it * 64 + threadIdx.x] = tile[threadIdx.x];
    pose_step(v, st, dx_trace, it, mode, max_iter, tol, full_iter);
}

__global__ void k_map_apply_pending(double* __restrict__ src, size_t m, IcpState* __restrict__ st) {
    if (!st->pending) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double px = src[3 * i + 0], py = src[3 * i + 1], pz = src[3 * i + 2];
    src[3 * i + 0] = ((st->R[0] * px + st->R[1] * py) + st->R[2] * pz) - st->t[0];
    src[3 * i + 1] = ((st->R[3] * px + st->R[4] * py) + st->R[5] * pz) - st->t[1];
    src[3 * i + 2] = ((st->R[6] * px + st->R[7] * py) + st->R[8] * pz) - st->t[2];
}

__global__ void k_map_state(IcpState* __restrict__ st, int reset) {
    if (reset) { st->iter = 0; st->done = 0; st->converged = 0; st->singular = 0; }
    st->pending = 0;
}


// Map split over ranks: a point is kept by the rank owning keys [lo, hi) when its partition cell
// (edge own_cell) or any of the 26 cells around it has its key in the range (one-cell halo).
struct InSlab {
    const double* xyz;
    double own_cell;
    unsigned long long lo, hi;
    int* bad;
    __device__ bool operator()(size_t i) const {
        int c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double v = xyz[3 * i + a];
            if (!(fabs(v / own_cell) < (double)(COORD_BIAS - 2))) { atomicOr(bad, 1); return false; }
            c[a] = cell_of(v, own_cell) + COORD_BIAS;
        }
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const unsigned long long k = morton3((unsigned)(c[0] + dx), (unsigned)(c[1] + dy), (unsigned)(c[2] + dz));
                    if (k >= lo && k < hi) return true;
                }
        return false;
    }
};

struct EmitKept {
    const double* xyz;
    const float4* nrm;
    double* xyz_out;
    float4* nrm_out;
    unsigned* ids_out;
    __device__ void operator()(size_t i, size_t slot) const {
        xyz_out[3 * slot + 0] = xyz[3 * i + 0];
        xyz_out[3 * slot + 1] = xyz[3 * i + 1];
        xyz_out[3 * slot + 2] = xyz[3 * i + 2];
        if (nrm) nrm_out[slot] = nrm[i];
        ids_out[slot] = (unsigned)i;
    }
};

__global__ void k_partition_keys(const double* __restrict__ xyz, size_t n, double own_cell,
                                 unsigned long long* __restrict__ keys, unsigned* __restrict__ vals, int* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double v = xyz[3 * i + a];
        if (!(fabs(v / own_cell) < (double)(COORD_BIAS - 2))) atomicOr(bad, 1);
        c[a] = (unsigned)(cell_of(v, own_cell) + COORD_BIAS);
    }
    keys[i] = morton3(c[0], c[1], c[2]);
    vals[i] = (unsigned)i;
}

inline unsigned blocks_for(size_t n, int threads) { return (unsigned)((n + threads - 1) / threads); }

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

struct pcp_grid {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    float* comm_tile = nullptr;      // 64 floats: the partial tile of pcp_grid_icp_run_comm, made on first use
    long long id_end = 0;            // one past the highest id any build or append has brought: appends must not go below it
    // scratch of the appends, kept between them (grow only): sort buffers and the sorted batch; a map fed frame by
    // frame would otherwise allocate and free nine buffers per append
    // appends collected between pcp_grid_collect_begin and pcp_grid_collect_end: coordinates and ids of the points waiting
    bool collecting = false;
    bool deferred = false;           // appends wait (as between collect_begin and collect_end) until somebody reads the map
    bool implicit = false;           // the collection under way was begun by a deferred append, not by the caller
    double* pend_xyz = nullptr; long long* pend_id = nullptr; size_t pend_n = 0, pend_cap = 0;
    static constexpr int n_scr = 16;          // 0-8: a sorted run (sort_run); 9-15: staging of a build's input
    void* scr[n_scr] = {};
    size_t scr_bytes[n_scr] = {};
    // map
    size_t n = 0, cap = 0;
    double cell = 0.0;
    int cmin[3] = {0, 0, 0}, ext[3] = {0, 0, 0};
    GridPoint* pts = nullptr;
    float4* normals = nullptr;
    unsigned long long* keys = nullptr;     // cell key of every entry of pts (sorted): what an append merges by
    GridPoint* pts_alt = nullptr;           // second set of buffers an append merges into (then the sets swap)
    float4* normals_alt = nullptr;
    unsigned long long* keys_alt = nullptr;
    size_t cap_alt = 0;
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};   // bounding cells of the map (cmin = lo, ext = hi - lo + 1)
    bool have_normals = false;
    bool normals_estimated = false;         // the normals came from pcp_grid_estimate_normals, not from the caller
    Cell* table = nullptr;
    int log_cap = 0;
    size_t table_cap = 0;
    size_t table_alloc = 0;                  // entries the table's allocation holds (>= table_cap)
    unsigned long long n_cells = 0;
    int sort_passes = 0;
    float build_ms = 0.0f, sort_ms = 0.0f;
    // source
    size_t m = 0, m_cap = 0;
    double *src_in = nullptr, *src = nullptr, *nn_d = nullptr;
    long long* nn_idx = nullptr;
    int* nn_pos = nullptr;
    unsigned char* sel = nullptr;
    float *partials = nullptr, *lvl0 = nullptr, *lvl1 = nullptr;
    // small state
    IcpState* state = nullptr;
    float *hb_trace = nullptr, *dx_trace = nullptr, *tile = nullptr;
    SelState* sel_state = nullptr;
    unsigned* sel_hist = nullptr;
    double *kd_partial = nullptr, *kd_out = nullptr, *rt = nullptr;
    int* mm = nullptr;
    unsigned long long* cell_count = nullptr;
    unsigned long long own_lo = 0, own_hi = 1ull << 63;
    double own_cell = 1.0;
    bool nn_valid = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    static const int max_iter_cap = 1024;
};

namespace {

GridView view_of(const pcp_grid* g) {
    GridView G;
    G.pts = g->pts; G.table = g->table; G.log_cap = g->log_cap; G.n = g->n;
    G.cell = g->cell; G.eps = g->cell * 1e-7;
    G.cminx = g->cmin[0]; G.cminy = g->cmin[1]; G.cminz = g->cmin[2];
    G.ex = g->ext[0]; G.ey = g->ext[1]; G.ez = g->ext[2];
    return G;
}

int free_map(pcp_grid* g) {
    void* p[] = {g->pts, g->normals, g->keys, g->pts_alt, g->normals_alt, g->keys_alt, g->table};
    for (void* q : p) if (q) PCP_CHECK(hipFree(q));
    g->pts = g->pts_alt = nullptr; g->normals = g->normals_alt = nullptr; g->keys = g->keys_alt = nullptr; g->table = nullptr;
    g->cap = g->cap_alt = 0; g->table_cap = 0; g->table_alloc = 0; g->n = 0; g->log_cap = 0;
    return 0;
}

int free_source(pcp_grid* g) {
    void* p[] = {g->src_in, g->src, g->nn_d, g->nn_idx, g->nn_pos, g->sel, g->partials, g->lvl0, g->lvl1};
    for (void* q : p) if (q) PCP_CHECK(hipFree(q));
    g->src_in = g->src = g->nn_d = nullptr; g->nn_idx = nullptr; g->nn_pos = nullptr; g->sel = nullptr;
    g->partials = g->lvl0 = g->lvl1 = nullptr;
    g->m_cap = 0; g->m = 0;
    return 0;
}

int reserve_source(pcp_grid* g, size_t m) {
    if (m <= g->m_cap) return 0;
    if (free_source(g)) return 1;
    const size_t cap = std::max<size_t>(m, 1024);
    const size_t waves = (cap + 63) / 64, l1 = (waves + 63) / 64 + 1;
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    A((void**)&g->src_in, cap * 24);
    A((void**)&g->src, cap * 24);
    A((void**)&g->nn_d, cap * 8);
    A((void**)&g->nn_idx, cap * 8);
    A((void**)&g->nn_pos, cap * 4);
    A((void**)&g->sel, cap);
    A((void**)&g->partials, waves * PCP_PSTRIDE * 4);
    A((void**)&g->lvl0, l1 * PCP_PSTRIDE * 4);
    A((void**)&g->lvl1, l1 * PCP_PSTRIDE * 4);
    if (e != hipSuccess) {          // leave the grid without a source rather than with half of one
        pcp_fail("source buffers for %zu points: %s", cap, hipGetErrorString(e));
        (void)free_source(g);
        g->m = 0;
        return 1;
    }
    g->m_cap = cap;
    return 0;
}

int flush_collected(pcp_grid* g);

// Every call that reads or rebuilds the map starts here: batches that deferred appends left waiting join the map now;
// a collection the caller began must be ended by the caller.
int check_collecting(pcp_grid* g) {
    if (g->collecting && g->implicit) return flush_collected(g);
    if (g->collecting) return pcp_fail("appends are being collected: the map is not complete before pcp_grid_collect_end");
    return 0;
}

int check_source(pcp_grid* g) {
    if (!g) return pcp_fail("null grid");
    if (check_collecting(g)) return 1;
    if (g->m == 0) return pcp_fail("no source cloud uploaded (pcp_grid_source)");
    return 0;
}

MapLsArgs map_args(pcp_grid* g, double radius, bool per_point) {
    MapLsArgs A;
    A.G = view_of(g);
    A.normals = g->normals;
    A.src = g->src; A.st = g->state; A.m = g->m;
    A.radius2 = radius * radius;
    A.own_lo = g->own_lo; A.own_hi = g->own_hi; A.own_cell = g->own_cell;
    A.partials = g->partials;
    A.nn_idx = per_point ? g->nn_idx : nullptr;
    A.nn_d = per_point ? g->nn_d : nullptr;
    return A;
}

// Which search kernel: with few queries a launch is a handful of waves per SIMD walking dependent
// look-ups, and 32 lanes per query win (12 M-point map, 122 k queries: 1.7 -> 0.65 ms).  With many
// queries the GPU is full either way: one lane per query does less redundant work on sparse cells
// (1 M uniform points, 4 per cell: 0.30 against 0.71 ms), 32 lanes share the scan of crowded ones
// (828 k LiDAR points, 83 per cell: 2.49 against 1.28 ms); the two meet at about 8 points per cell.
// PCP_GRID_LANES=1 or =32 in the environment forces one of them.
constexpr size_t LANES32_MAX_QUERIES = 262144;
constexpr size_t LANES32_MIN_PER_CELL = 8;
bool lane_per_query(const pcp_grid* g) {
    static const int forced = [] {
        const char* e = getenv("PCP_GRID_LANES");
        return !e ? 0 : (e[0] == '1' && e[1] == 0 ? 1 : (e[0] == '3' && e[1] == '2' && e[2] == 0 ? 32 : 0));
    }();
    if (forced) return forced == 1;
    return g->m > LANES32_MAX_QUERIES && g->n < LANES32_MIN_PER_CELL * g->n_cells;
}

int launch_nn32(pcp_grid* g, double max_d2, bool in_loop) {
    Nn32Args A;
    A.G = view_of(g);
    A.src = g->src; A.m = g->m; A.max_d2 = max_d2;
    A.st = in_loop ? g->state : nullptr;
    A.own_cell = in_loop ? g->own_cell : 1.0;
    A.own_lo = in_loop ? g->own_lo : 0ull;
    A.own_hi = in_loop ? g->own_hi : ~0ull;
    A.nn_d = g->nn_d; A.nn_idx = g->nn_idx; A.nn_pos = g->nn_pos;
    hipLaunchKernelGGL(k_grid_nn32, dim3(blocks_for(g->m * 32, 256)), dim3(256), 0, g->stream, A);
    PCP_CHECK(hipGetLastError());
    return 0;
}

int launch_map_ls(pcp_grid* g, int mode, double radius, bool per_point) {
    if (mode != PCP_MODE_POINT && mode != PCP_MODE_PLANE) return pcp_fail("unknown mode %d", mode);
    if (mode == PCP_MODE_PLANE && !g->have_normals) return pcp_fail("plane mode needs map normals (pcp_grid_build)");
    if (!(radius > 0.0)) return pcp_fail("search radius must be positive");
    if (!lane_per_query(g)) {
        if (launch_nn32(g, radius * radius, true)) return 1;
        const unsigned nb = blocks_for(g->m, PCP_BLOCK);
        if (mode == PCP_MODE_POINT)
            hipLaunchKernelGGL(k_map_tiles<PCP_MODE_POINT>, dim3(nb), dim3(PCP_BLOCK), 0, g->stream, view_of(g), g->normals,
                               g->src, g->nn_pos, g->state, g->m, g->partials);
        else
            hipLaunchKernelGGL(k_map_tiles<PCP_MODE_PLANE>, dim3(nb), dim3(PCP_BLOCK), 0, g->stream, view_of(g), g->normals,
                               g->src, g->nn_pos, g->state, g->m, g->partials);
        PCP_CHECK(hipGetLastError());
        return 0;
    }
    const MapLsArgs A = map_args(g, radius, per_point);
    const unsigned blocks = blocks_for(g->m, PCP_BLOCK);
    if (mode == PCP_MODE_POINT) hipLaunchKernelGGL(k_map_ls<PCP_MODE_POINT>, dim3(blocks), dim3(PCP_BLOCK), 0, g->stream, A);
    else hipLaunchKernelGGL(k_map_ls<PCP_MODE_PLANE>, dim3(blocks), dim3(PCP_BLOCK), 0, g->stream, A);
    PCP_CHECK(hipGetLastError());
    return 0;
}

FinishArgs map_finish_args(pcp_grid* g, float* hb_out) {
    FinishArgs F;
    F.partials = g->partials; F.n1 = (int)((g->m + 63) / 64); F.lvl0 = g->lvl0; F.lvl1 = g->lvl1;
    F.st = g->state; F.hb_out = hb_out; F.dx_out = g->dx_trace;
    return F;
}

}  // namespace

extern "C" {

int pcp_grid_create(int device, pcp_grid** out) {
    if (!out) return pcp_fail("out is null");
    *out = nullptr;
    int count = 0;
    PCP_CHECK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return pcp_fail("device %d out of range (%d devices)", device, count);
    PCP_CHECK(hipSetDevice(device));
    pcp_grid* g = new pcp_grid();
    g->device = device;
    PCP_CHECK(hipStreamCreateWithFlags(&g->own_stream, hipStreamNonBlocking));
    g->stream = g->own_stream;
    PCP_CHECK(hipMalloc(&g->state, sizeof(IcpState)));
    PCP_CHECK(hipMemset(g->state, 0, sizeof(IcpState)));
    PCP_CHECK(hipMalloc(&g->hb_trace, (size_t)pcp_grid::max_iter_cap * 64 * 4));
    PCP_CHECK(hipMalloc(&g->dx_trace, (size_t)pcp_grid::max_iter_cap * 6 * 4));
    PCP_CHECK(hipMalloc(&g->tile, 64 * 4));
    PCP_CHECK(hipMalloc(&g->sel_state, sizeof(SelState)));
    PCP_CHECK(hipMalloc(&g->sel_hist, 256 * 4));
    PCP_CHECK(hipMemset(g->sel_hist, 0, 256 * 4));
    PCP_CHECK(hipMalloc(&g->kd_partial, (size_t)KD_BLOCKS * 16 * 8));
    PCP_CHECK(hipMalloc(&g->kd_out, 64 * 8));
    PCP_CHECK(hipMalloc(&g->rt, 12 * 8));
    PCP_CHECK(hipMalloc(&g->mm, 8 * 4));
    PCP_CHECK(hipMalloc(&g->cell_count, 8));
    PCP_CHECK(hipEventCreate(&g->ev0));
    PCP_CHECK(hipEventCreate(&g->ev1));
    PCP_CHECK(hipEventCreate(&g->ev2));
    PCP_CHECK(hipEventCreate(&g->ev3));
    *out = g;
    return 0;
}

int pcp_grid_destroy(pcp_grid* g) {
    if (!g) return 0;
    PCP_CHECK(hipSetDevice(g->device));
    PCP_CHECK(hipStreamSynchronize(g->stream));
    if (free_map(g) || free_source(g)) return 1;
    void* p[] = {g->state, g->hb_trace, g->dx_trace, g->tile, g->sel_state, g->sel_hist, g->kd_partial, g->kd_out,
                 g->rt, g->mm, g->cell_count, g->comm_tile};
    for (void* q : p) if (q) PCP_CHECK(hipFree(q));
    for (void* q : g->scr) if (q) PCP_CHECK(hipFree(q));
    for (void* q : {(void*)g->pend_xyz, (void*)g->pend_id}) if (q) PCP_CHECK(hipFree(q));
    for (hipEvent_t e : {g->ev0, g->ev1, g->ev2, g->ev3}) if (e) PCP_CHECK(hipEventDestroy(e));
    PCP_CHECK(hipStreamDestroy(g->own_stream));
    delete g;
    return 0;
}

int pcp_grid_set_stream(pcp_grid* g, void* hip_stream) {
    if (!g) return pcp_fail("null grid");
    g->stream = hip_stream ? (hipStream_t)hip_stream : g->own_stream;
    return 0;
}

int pcp_grid_synchronize(pcp_grid* g) {
    if (!g) return pcp_fail("null grid");
    PCP_CHECK(hipSetDevice(g->device));
    PCP_CHECK(hipStreamSynchronize(g->stream));
    return 0;
}

// Builds the grid from points already on the device (d_ids: original index of every point, or null).
// Bits of a cell key that can differ between two cells of the box [lo, hi]: per axis the zig-zag values of
// the box's cells lie between two numbers, so they share those numbers' common binary prefix.
static unsigned long long varying_key_bits(const int lo[3], const int hi[3]) {
    unsigned long long mask = 0;
    for (int a = 0; a < 3; ++a) {
        unsigned zlo, zhi;
        if (lo[a] < 0 && hi[a] >= 0) { zlo = 0; zhi = std::max(zigzag(lo[a]), zigzag(hi[a])); }
        else if (hi[a] < 0) { zlo = zigzag(hi[a]); zhi = zigzag(lo[a]); }
        else { zlo = zigzag(lo[a]); zhi = zigzag(hi[a]); }
        int top = 0;
        while (top < 32 && ((zlo ^ zhi) >> top) != 0) ++top;
        const unsigned long long m = (top >= 21) ? 0x1FFFFFull : ((1ull << top) - 1ull);
        mask |= spread21(m) << a;
    }
    return mask;
}

// A batch of points brought into key order on the device: what a build installs as the map and an
// append merges into it.
struct SortedRun {
    GridPoint* pts = nullptr;
    unsigned long long* keys = nullptr;
    float4* normals = nullptr;
    size_t n = 0;
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    int passes = 0;
    bool owned = true;        // false: the buffers are the grid's append scratch and stay with it
    void release() {
        if (owned)
            for (void* q : {(void*)pts, (void*)keys, (void*)normals}) if (q) (void)hipFree(q);
        pts = nullptr; keys = nullptr; normals = nullptr; n = 0;
    }
};

// Slot `k` of the grid's append scratch with room for `bytes` (contents are not kept across a growth).
static hipError_t grid_scratch(pcp_grid* g, int k, size_t bytes, void** out) {
    if (g->scr_bytes[k] < bytes) {
        if (g->scr[k]) (void)hipFree(g->scr[k]);
        g->scr[k] = nullptr;
        g->scr_bytes[k] = 0;
        size_t want = bytes + bytes / 2;
        if (want < 4096) want = 4096;
        const hipError_t e = hipMalloc(&g->scr[k], want);
        if (e != hipSuccess) return e;
        g->scr_bytes[k] = want;
    }
    *out = g->scr[k];
    return hipSuccess;
}

// Keys, stable radix sort by key, points gathered into that order.  d_ids (or null) maps the rows of
// d_xyz to indices in the caller's array; ids are first_id + that index.  Events ev0..ev2 bracket the
// whole and the sort when `timed`.
static int sort_run(pcp_grid* g, const double* d_xyz, const float4* d_nrm, const unsigned* d_ids, long long first_id,
                    size_t n, double cell, bool timed, SortedRun* out, bool pooled = false, const long long* abs_ids = nullptr) {
    hipStream_t s = g->stream;
    unsigned long long *k0 = nullptr, *k1 = nullptr;
    unsigned *v0 = nullptr, *v1 = nullptr, *hist = nullptr, *totals = nullptr, *base = nullptr;
    const int nblk = (int)((n + RS_TILE - 1) / RS_TILE);
    SortedRun run;
    run.owned = !pooled;
    auto cleanup = [&]() {
        if (pooled) return;
        for (void* q : {(void*)k0, (void*)k1, (void*)v0, (void*)v1, (void*)hist, (void*)totals, (void*)base})
            if (q) (void)hipFree(q);
    };
    int slot = 0;
    auto get = [&](void** p, size_t bytes) { return pooled ? grid_scratch(g, slot++, bytes, p) : hipMalloc(p, bytes); };
#define GRID_TRY(expr)                                                                                      \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            cleanup();                                                                                      \
            run.release();                                                                                  \
            return pcp_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);     \
        }                                                                                                   \
    } while (0)
    if (timed) GRID_TRY(hipEventRecord(g->ev0, s));
    // bounding cells
    const int init[8] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN, 0, 0};
    GRID_TRY(hipMemcpyAsync(g->mm, init, sizeof(init), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_cell_range, dim3(std::min<unsigned>(blocks_for(n, 256), 2048)), dim3(256), 0, s, d_xyz, n, cell, g->mm);
    int mm[8];
    GRID_TRY(hipMemcpyAsync(mm, g->mm, sizeof(mm), hipMemcpyDeviceToHost, s));
    GRID_TRY(hipStreamSynchronize(s));
    if (mm[6]) {
        cleanup();
        return pcp_fail("map coordinates must be finite and within 2^20 cells of the origin");
    }
    for (int a = 0; a < 3; ++a) { run.lo[a] = mm[a]; run.hi[a] = mm[3 + a]; }
    // keys + sort
    GRID_TRY(get((void**)&k0, n * 8));
    GRID_TRY(get((void**)&k1, n * 8));
    GRID_TRY(get((void**)&v0, n * 4));
    GRID_TRY(get((void**)&v1, n * 4));
    GRID_TRY(get((void**)&hist, (size_t)256 * nblk * 4));
    GRID_TRY(get((void**)&totals, 256 * 4));
    GRID_TRY(get((void**)&base, 256 * 4));
    hipLaunchKernelGGL(k_cell_keys, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_xyz, n, cell, k0, v0);
    if (timed) GRID_TRY(hipEventRecord(g->ev1, s));
    unsigned long long* kin = nullptr;
    unsigned* vin = nullptr;
    run.passes = radix_sort_pairs_varying(s, k0, k1, v0, v1, n, varying_key_bits(run.lo, run.hi), hist, totals, base, &kin, &vin);
    if (timed) GRID_TRY(hipEventRecord(g->ev2, s));
    GRID_TRY(hipGetLastError());
    GRID_TRY(get((void**)&run.pts, n * sizeof(GridPoint)));
    if (d_nrm) GRID_TRY(get((void**)&run.normals, n * 16));
    hipLaunchKernelGGL(k_gather_points, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_xyz, vin, n, d_nrm, d_ids, first_id, abs_ids,
                       run.pts, run.normals);
    GRID_TRY(hipGetLastError());
    // the sorted keys stay (in whichever of the two sort buffers they ended); everything else goes
    run.keys = kin;
    if (kin == k0) k0 = nullptr; else k1 = nullptr;
    GRID_TRY(hipStreamSynchronize(s));
#undef GRID_TRY
    cleanup();
    run.n = n;
    *out = run;
    return 0;
}

// Occupied cells of the resident sorted keys -> hash table (key -> first entry, count), sized by the
// rule capacity = smallest power of two >= 2 x cells; reallocated only when that size changes.
static int install_table(pcp_grid* g) {
    hipStream_t s = g->stream;
    PCP_CHECK(hipMemsetAsync(g->cell_count, 0, 8, s));
    hipLaunchKernelGGL(k_count_cells, dim3(std::min<unsigned>(blocks_for(g->n, 256), 2048)), dim3(256), 0, s, g->keys, g->n, g->cell_count);
    unsigned long long cells = 0;
    PCP_CHECK(hipMemcpyAsync(&cells, g->cell_count, 8, hipMemcpyDeviceToHost, s));
    PCP_CHECK(hipStreamSynchronize(s));
    g->n_cells = cells;
    int log_cap = 4;
    while ((1ull << log_cap) < 2 * cells) ++log_cap;
    // the table's size follows the rule; its allocation only grows (a smaller table lives in the front of a larger one)
    g->log_cap = log_cap;
    g->table_cap = (size_t)1 << log_cap;
    if (!g->table || g->table_cap > g->table_alloc) {
        if (g->table) PCP_CHECK(hipFree(g->table));
        g->table = nullptr;
        g->table_alloc = 0;
        PCP_CHECK(hipMalloc(&g->table, g->table_cap * sizeof(Cell)));
        g->table_alloc = g->table_cap;
    }
    PCP_CHECK(hipMemsetAsync(g->table, 0xFF, g->table_cap * sizeof(Cell), s));
    hipLaunchKernelGGL(k_insert_cells, dim3(blocks_for(g->n, 256)), dim3(256), 0, s, g->keys, g->n, g->table, g->log_cap);
    PCP_CHECK(hipGetLastError());
    return 0;
}

static void set_bounds(pcp_grid* g, const int lo[3], const int hi[3]) {
    for (int a = 0; a < 3; ++a) {
        g->lo[a] = lo[a]; g->hi[a] = hi[a];
        g->cmin[a] = lo[a];
        g->ext[a] = hi[a] - lo[a] + 1;
    }
}

// Up to this many points a build sorts in the grid's scratch and copies the result into map buffers it keeps from
// build to build (a build then allocates nothing once the buffers are large enough: allocation and release were most
// of a 1 M-point build's time); larger maps hand the sorted run's own buffers over and keep no scratch.
constexpr size_t GRID_POOLED_BUILD = (size_t)1 << 24;

static int build_device_pooled(pcp_grid* g, const double* d_xyz, const float4* d_nrm, const unsigned* d_ids, size_t n, double cell) {
    hipStream_t s = g->stream;
    g->n = 0;
    g->cell = cell;
    g->have_normals = d_nrm != nullptr;
    g->normals_estimated = false;
    g->n_cells = 0;
    g->sort_passes = 0;
    g->ext[0] = g->ext[1] = g->ext[2] = 0;
    g->nn_valid = false;
    if (!d_nrm)                                     // a map without normals holds no buffers for them
        for (float4** q : {&g->normals, &g->normals_alt}) { if (*q) PCP_CHECK(hipFree(*q)); *q = nullptr; }
    if (n == 0) return 0;
    SortedRun run;
    if (sort_run(g, d_xyz, d_nrm, d_ids, 0, n, cell, true, &run, true)) return 1;
    if (g->cap < n || (d_nrm && !g->normals)) {
        for (void* q : {(void*)g->pts, (void*)g->keys, (void*)g->normals}) if (q) PCP_CHECK(hipFree(q));
        g->pts = nullptr; g->keys = nullptr; g->normals = nullptr; g->cap = 0;
        hipError_t e = hipMalloc(&g->pts, n * sizeof(GridPoint));
        if (e == hipSuccess) e = hipMalloc(&g->keys, n * 8);
        if (e == hipSuccess && d_nrm) e = hipMalloc(&g->normals, n * 16);
        if (e != hipSuccess) return pcp_fail("buffers for a map of %zu points: %s", n, hipGetErrorString(e));
        g->cap = n;
    }
    PCP_CHECK(hipMemcpyAsync(g->pts, run.pts, n * sizeof(GridPoint), hipMemcpyDeviceToDevice, s));
    PCP_CHECK(hipMemcpyAsync(g->keys, run.keys, n * 8, hipMemcpyDeviceToDevice, s));
    if (d_nrm) PCP_CHECK(hipMemcpyAsync(g->normals, run.normals, n * 16, hipMemcpyDeviceToDevice, s));
    g->n = n;
    g->sort_passes = run.passes;
    set_bounds(g, run.lo, run.hi);
    if (install_table(g)) { g->n = 0; return 1; }
    PCP_CHECK(hipEventRecord(g->ev3, s));
    PCP_CHECK(hipStreamSynchronize(s));
    PCP_CHECK(hipEventElapsedTime(&g->build_ms, g->ev0, g->ev3));
    PCP_CHECK(hipEventElapsedTime(&g->sort_ms, g->ev1, g->ev2));
    return 0;
}

static int build_device(pcp_grid* g, const double* d_xyz, const float4* d_nrm, const unsigned* d_ids, size_t n, double cell) {
    if (n <= GRID_POOLED_BUILD) return build_device_pooled(g, d_xyz, d_nrm, d_ids, n, cell);
    hipStream_t s = g->stream;
    if (free_map(g)) return 1;
    for (int k = 0; k < pcp_grid::n_scr; ++k) {       // a map this large keeps no scratch of its own size around
        if (g->scr[k]) PCP_CHECK(hipFree(g->scr[k]));
        g->scr[k] = nullptr; g->scr_bytes[k] = 0;
    }
    g->cell = cell;
    g->have_normals = d_nrm != nullptr;
    g->normals_estimated = false;
    g->n_cells = 0;
    g->sort_passes = 0;
    g->ext[0] = g->ext[1] = g->ext[2] = 0;
    g->nn_valid = false;
    if (n == 0) return 0;
    SortedRun run;
    if (sort_run(g, d_xyz, d_nrm, d_ids, 0, n, cell, true, &run)) return 1;
    // the run's buffers become the map
    g->pts = run.pts; g->keys = run.keys; g->normals = run.normals;
    g->n = n;
    g->cap = n;
    g->sort_passes = run.passes;
    set_bounds(g, run.lo, run.hi);
    if (install_table(g)) return 1;
    PCP_CHECK(hipEventRecord(g->ev3, s));
    PCP_CHECK(hipStreamSynchronize(s));
    PCP_CHECK(hipEventElapsedTime(&g->build_ms, g->ev0, g->ev3));
    PCP_CHECK(hipEventElapsedTime(&g->sort_ms, g->ev1, g->ev2));
    return 0;
}

// Points already on the device join the resident map: sorted among themselves, merged into the second
// set of buffers (which then becomes the first), table rebuilt from the merged keys.  The result is the
// map a build of all points in order of their ids would give: same order, same cells.
static int append_device(pcp_grid* g, const double* d_xyz, const float4* d_nrm, const unsigned* d_ids, long long first_id, size_t m,
                         const long long* abs_ids = nullptr) {
    hipStream_t s = g->stream;
    if (m == 0) return 0;
    if (g->collecting && g->implicit && d_nrm && !abs_ids) {       // normals cannot wait: what waits goes in first
        if (flush_collected(g)) return 1;
    }
    if (g->deferred && !g->collecting && !d_nrm && !abs_ids) {
        g->collecting = true;
        g->implicit = true;
        g->pend_n = 0;
    }
    if (g->collecting && !abs_ids) {
        // the batch waits: coordinates and ids go behind what is waiting already (this stream; the caller's buffer is free
        // again once the stream has passed the copy, which the caller's next wait on the grid covers)
        if (d_nrm) return pcp_fail("collected appends carry no normals");
        if (g->pend_n + m > g->pend_cap) {
            size_t cap = std::max<size_t>(g->pend_cap * 2, std::max<size_t>(g->pend_n + m, 1u << 20));
            double* nx = nullptr; long long* ni = nullptr;
            hipError_t e = hipMalloc(&nx, cap * 24);
            if (e == hipSuccess) e = hipMalloc(&ni, cap * 8);
            if (e == hipSuccess && g->pend_n) e = hipMemcpyAsync(nx, g->pend_xyz, g->pend_n * 24, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess && g->pend_n) e = hipMemcpyAsync(ni, g->pend_id, g->pend_n * 8, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { if (nx) (void)hipFree(nx); if (ni) (void)hipFree(ni); return pcp_fail("room for %zu waiting points: %s", cap, hipGetErrorString(e)); }
            if (g->pend_xyz) (void)hipFree(g->pend_xyz);
            if (g->pend_id) (void)hipFree(g->pend_id);
            g->pend_xyz = nx; g->pend_id = ni; g->pend_cap = cap;
        }
        PCP_CHECK(hipMemcpyAsync(g->pend_xyz + 3 * g->pend_n, d_xyz, m * 24, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_pending_ids, dim3(blocks_for(m, 256)), dim3(256), 0, s, d_ids, first_id, m, g->pend_id + g->pend_n);
        PCP_CHECK(hipGetLastError());
        PCP_CHECK(hipStreamSynchronize(s));        // the caller may reuse its buffer when the call returns
        g->pend_n += m;
        return 0;
    }
    if (g->n + m >= (1ull << 31)) return pcp_fail("%zu map points exceed the 2^31 limit of one grid", g->n + m);
    if (g->n && g->have_normals && !d_nrm) {
        if (!g->normals_estimated) return pcp_fail("the map carries the caller's normals: appended points need them too");
        g->have_normals = false;              // estimated from the map as it was: stale now, estimate again
        g->normals_estimated = false;
        for (float4** q : {&g->normals, &g->normals_alt}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    }
    if (g->n && !g->have_normals && d_nrm) return pcp_fail("the map carries no normals");
    g->nn_valid = false;
    SortedRun run;
    // a batch that joins a map is sorted in the grid's scratch; the first batch of an empty map becomes the map itself
    if (sort_run(g, d_xyz, d_nrm, d_ids, first_id, m, g->cell, false, &run, g->n != 0, abs_ids)) return 1;
    g->sort_passes = run.passes;
    if (g->n == 0) {
        if (free_map(g)) { run.release(); return 1; }
        g->pts = run.pts; g->keys = run.keys; g->normals = run.normals;
        g->n = g->cap = m;
        g->have_normals = d_nrm != nullptr;
        set_bounds(g, run.lo, run.hi);
        return install_table(g);
    }
    const size_t total = g->n + m;
    if (g->cap_alt < total || (g->have_normals && !g->normals_alt)) {      // grow by half at least: a map fed frame by frame reallocates rarely
        for (void* q : {(void*)g->pts_alt, (void*)g->keys_alt, (void*)g->normals_alt}) if (q) (void)hipFree(q);
        g->pts_alt = nullptr; g->keys_alt = nullptr; g->normals_alt = nullptr; g->cap_alt = 0;
        const size_t cap = std::max(total, g->cap + g->cap / 2);
        hipError_t e = hipMalloc(&g->pts_alt, cap * sizeof(GridPoint));
        if (e == hipSuccess) e = hipMalloc(&g->keys_alt, cap * 8);
        if (e == hipSuccess && g->have_normals) e = hipMalloc(&g->normals_alt, cap * 16);
        if (e != hipSuccess) {
            run.release();
            return pcp_fail("buffers for a map of %zu points: %s", cap, hipGetErrorString(e));
        }
        g->cap_alt = cap;
    }
    hipLaunchKernelGGL(k_merge_old, dim3(blocks_for(g->n, 256)), dim3(256), 0, s, g->keys, g->n, run.keys, m, g->pts, g->normals,
                       g->pts_alt, g->keys_alt, g->normals_alt);
    hipLaunchKernelGGL(k_merge_new, dim3(blocks_for(m, 256)), dim3(256), 0, s, g->keys, g->n, run.keys, m, run.pts, run.normals,
                       g->pts_alt, g->keys_alt, g->normals_alt);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    run.release();
    if (e != hipSuccess) return pcp_fail("merging %zu points into the map: %s", m, hipGetErrorString(e));
    std::swap(g->pts, g->pts_alt);
    std::swap(g->keys, g->keys_alt);
    std::swap(g->normals, g->normals_alt);
    std::swap(g->cap, g->cap_alt);
    g->n = total;
    int lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(g->lo[a], run.lo[a]); hi[a] = std::max(g->hi[a], run.hi[a]); }
    set_bounds(g, lo, hi);
    if (install_table(g)) {
        // merged entries without a cell table cannot be searched: the map is given up rather than left half made
        g->n = 0;
        g->n_cells = 0;
        g->nn_valid = false;
        return 1;
    }
    return 0;
}

namespace {
int flush_collected(pcp_grid* g) {
    PCP_CHECK(hipSetDevice(g->device));
    g->collecting = false;
    g->implicit = false;
    const size_t m = g->pend_n;
    g->pend_n = 0;
    if (m == 0) return 0;
    // one sort, one merge and one cell table for everything that waited: the map an append per batch would have left
    // (a cell's points in the order of their ids) as long as the batches' ids ascend, as they do for appends in general
    return append_device(g, g->pend_xyz, nullptr, nullptr, 0, m, g->pend_id);
}
}  // namespace

static int check_build_args(pcp_grid* g, const double* xyz, size_t n, double cell) {
    if (!g) return pcp_fail("null grid");
    if (!(cell > 0.0) || !std::isfinite(cell)) return pcp_fail("cell size must be positive and finite");
    if (n >= (1ull << 31)) return pcp_fail("%zu map points exceed the 2^31 limit of one grid", n);
    if (n && !xyz) return pcp_fail("xyz is null");
    return 0;
}

int pcp_grid_build(pcp_grid* g, const double* xyz, size_t n, double cell, const float* normals_xyzw) {
    if (check_build_args(g, xyz, n, cell)) return 1;
    if (g->collecting && g->implicit) { g->collecting = false; g->implicit = false; g->pend_n = 0; }   // replaced with the map
    if (check_collecting(g)) return 1;
    PCP_CHECK(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    PCP_CHECK(hipStreamSynchronize(s));
    double* d_xyz = nullptr;
    float4* d_nrm = nullptr;
    int rc = 0;
    const bool pooled = n <= GRID_POOLED_BUILD;      // the input waits in the grid's scratch, which stays
    if (n) {
        hipError_t e = pooled ? grid_scratch(g, 9, n * 24, (void**)&d_xyz) : hipMalloc(&d_xyz, n * 24);
        if (e == hipSuccess) e = hipMemcpyAsync(d_xyz, xyz, n * 24, hipMemcpyHostToDevice, s);
        if (e == hipSuccess && normals_xyzw) {
            e = pooled ? grid_scratch(g, 10, n * 16, (void**)&d_nrm) : hipMalloc(&d_nrm, n * 16);
            if (e == hipSuccess) e = hipMemcpyAsync(d_nrm, normals_xyzw, n * 16, hipMemcpyHostToDevice, s);
        }
        if (e != hipSuccess) rc = pcp_fail("uploading the map failed: %s", hipGetErrorString(e));
    }
    if (!rc) rc = build_device(g, d_xyz, d_nrm, nullptr, n, cell);
    if (!rc) g->id_end = (long long)n;
    if (!pooled) {
        if (d_xyz) (void)hipFree(d_xyz);
        if (d_nrm) (void)hipFree(d_nrm);
    }
    return rc;
}

int pcp_grid_build_slab(pcp_grid* g, const double* xyz, size_t n, double cell, const float* normals_xyzw,
                        double own_cell, unsigned long long lo, unsigned long long hi, size_t* n_kept) {
    if (check_build_args(g, xyz, n, cell)) return 1;
    if (g->collecting && g->implicit) { g->collecting = false; g->implicit = false; g->pend_n = 0; }   // replaced with the map
    if (check_collecting(g)) return 1;
    if (!(own_cell > 0.0)) return pcp_fail("partition cell size must be positive");
    if (lo > hi) return pcp_fail("owned key range is empty");
    PCP_CHECK(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    PCP_CHECK(hipStreamSynchronize(s));
    if (n_kept) *n_kept = 0;
    double *d_xyz = nullptr, *d_keep = nullptr;
    float4 *d_nrm = nullptr, *d_nkeep = nullptr;
    unsigned *d_ids = nullptr, *d_counts = nullptr;
    int* d_bad = nullptr;
    int rc = 0;
    auto cleanup = [&]() {
        for (void* q : {(void*)d_xyz, (void*)d_keep, (void*)d_nrm, (void*)d_nkeep, (void*)d_ids, (void*)d_counts, (void*)d_bad})
            if (q) (void)hipFree(q);
    };
    if (n == 0) { g->id_end = 0; return build_device(g, nullptr, nullptr, nullptr, 0, cell); }
    hipError_t e = hipMalloc(&d_xyz, n * 24);
    if (e == hipSuccess) e = hipMemcpyAsync(d_xyz, xyz, n * 24, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMalloc(&d_keep, n * 24);
    if (e == hipSuccess) e = hipMalloc(&d_ids, n * 4);
    if (e == hipSuccess) e = hipMalloc(&d_counts, compact_count_elems(n) * 4);
    if (e == hipSuccess) e = hipMalloc(&d_bad, 4);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 4, s);
    if (e == hipSuccess && normals_xyzw) {
        e = hipMalloc(&d_nrm, n * 16);
        if (e == hipSuccess) e = hipMemcpyAsync(d_nrm, normals_xyzw, n * 16, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMalloc(&d_nkeep, n * 16);
    }
    unsigned kept = 0;
    int bad = 0;
    if (e == hipSuccess) {
        compact_ordered(s, InSlab{d_xyz, own_cell, lo, hi, d_bad}, EmitKept{d_xyz, d_nrm, d_keep, d_nkeep, d_ids}, n, d_counts);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&kept, d_counts + compact_tiles(n), 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = pcp_fail("selecting the slab failed: %s", hipGetErrorString(e));
    else if (bad) rc = pcp_fail("map coordinates must be finite and within 2^20 partition cells of the origin");
    if (!rc) rc = build_device(g, d_keep, d_nkeep, d_ids, (size_t)kept, cell);
    if (!rc) g->id_end = (long long)n;          // ids are places in the caller's whole array
    cleanup();
    if (!rc && n_kept) *n_kept = kept;
    return rc;
}

static int append_host(pcp_grid* g, const double* xyz, size_t n, const float* normals_xyzw, long long first_id,
                       bool slab, double own_cell, unsigned long long lo, unsigned long long hi, size_t* n_kept,
                       bool xyz_on_device = false) {
    if (!g) return pcp_fail("null grid");
    if (!(g->cell > 0.0)) return pcp_fail("the grid has no cell size yet: call pcp_grid_build first (an empty map will do)");
    if (n && !xyz) return pcp_fail("xyz is null");
    if (n >= (1ull << 31)) return pcp_fail("%zu points exceed the 2^31 limit of one grid", n);
    if (first_id < 0) return pcp_fail("first_id must not be negative");
    // equal keys keep resident entries in front of new ones: only with ids that go up from batch to batch is that the
    // order a build of everything gives
    if (first_id < g->id_end)
        return pcp_fail("first_id %lld lies below the ids the map already holds (next free id: %lld)", first_id, g->id_end);
    if (n_kept) *n_kept = 0;
    if (n == 0) return 0;
    PCP_CHECK(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    PCP_CHECK(hipStreamSynchronize(s));
    double *d_xyz = nullptr, *d_keep = nullptr;
    float4 *d_nrm = nullptr, *d_nkeep = nullptr;
    unsigned *d_ids = nullptr, *d_counts = nullptr;
    int* d_bad = nullptr;
    auto cleanup = [&]() {
        for (void* q : {(void*)(xyz_on_device ? nullptr : d_xyz), (void*)d_keep, (void*)d_nrm, (void*)d_nkeep, (void*)d_ids,
                        (void*)d_counts, (void*)d_bad})
            if (q) (void)hipFree(q);
    };
    hipError_t e = hipSuccess;
    if (xyz_on_device) {
        d_xyz = const_cast<double*>(xyz);            // read only; the caller's work on it has completed (see pcp.h)
    } else {
        e = hipMalloc(&d_xyz, n * 24);
        if (e == hipSuccess) e = hipMemcpyAsync(d_xyz, xyz, n * 24, hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess && normals_xyzw) {
        e = hipMalloc(&d_nrm, n * 16);
        if (e == hipSuccess) e = hipMemcpyAsync(d_nrm, normals_xyzw, n * 16, hipMemcpyHostToDevice, s);
    }
    int rc = 0;
    size_t kept = n;
    const double* use_xyz = d_xyz;
    const float4* use_nrm = d_nrm;
    const unsigned* use_ids = nullptr;
    if (e == hipSuccess && slab) {          // this rank's share: own cells plus the one-cell halo, as pcp_grid_build_slab
        e = hipMalloc(&d_keep, n * 24);
        if (e == hipSuccess) e = hipMalloc(&d_ids, n * 4);
        if (e == hipSuccess) e = hipMalloc(&d_counts, compact_count_elems(n) * 4);
        if (e == hipSuccess) e = hipMalloc(&d_bad, 4);
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 4, s);
        if (e == hipSuccess && d_nrm) e = hipMalloc(&d_nkeep, n * 16);
        unsigned k = 0;
        int bad = 0;
        if (e == hipSuccess) {
            compact_ordered(s, InSlab{d_xyz, own_cell, lo, hi, d_bad}, EmitKept{d_xyz, d_nrm, d_keep, d_nkeep, d_ids}, n, d_counts);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&k, d_counts + compact_tiles(n), 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess && bad) rc = pcp_fail("map coordinates must be finite and within 2^20 partition cells of the origin");
        kept = k;
        use_xyz = d_keep; use_nrm = d_nkeep; use_ids = d_ids;
    }
    if (e != hipSuccess) rc = pcp_fail("uploading the points failed: %s", hipGetErrorString(e));
    if (!rc) rc = append_device(g, use_xyz, use_nrm, use_ids, first_id, kept);
    if (!rc) g->id_end = first_id + (long long)n;
    cleanup();
    if (!rc && n_kept) *n_kept = kept;
    return rc;
}

int pcp_grid_collect_begin(pcp_grid* g) {
    if (!g) return pcp_fail("null grid");
    if (!(g->cell > 0.0)) return pcp_fail("the grid has no cell size yet: call pcp_grid_build first (an empty map will do)");
    if (g->collecting && !g->implicit) return pcp_fail("appends are being collected already");
    if (!g->collecting) g->pend_n = 0;
    g->collecting = true;            // batches deferred so far stay in the collection, which is the caller's to end now
    g->implicit = false;
    return 0;
}

int pcp_grid_collect_end(pcp_grid* g) {
    if (!g) return pcp_fail("null grid");
    if (!g->collecting) return pcp_fail("no appends are being collected");
    return flush_collected(g);
}

int pcp_grid_defer_appends(pcp_grid* g, int on) {
    if (!g) return pcp_fail("null grid");
    if (!on && g->collecting && g->implicit && flush_collected(g)) return 1;
    g->deferred = on != 0;
    return 0;
}

int pcp_grid_append(pcp_grid* g, const double* xyz, size_t n, const float* normals_xyzw, long long first_id) {
    return append_host(g, xyz, n, normals_xyzw, first_id, false, 1.0, 0, 0, nullptr);
}

int pcp_grid_append_slab(pcp_grid* g, const double* xyz, size_t n, const float* normals_xyzw, double own_cell,
                         unsigned long long lo, unsigned long long hi, long long first_id, size_t* n_kept) {
    if (!(own_cell > 0.0)) return pcp_fail("partition cell size must be positive");
    if (lo > hi) return pcp_fail("owned key range is empty");
    return append_host(g, xyz, n, normals_xyzw, first_id, true, own_cell, lo, hi, n_kept);
}

int pcp_grid_append_dev(pcp_grid* g, const double* xyz_dev, size_t n, long long first_id) {
    return append_host(g, xyz_dev, n, nullptr, first_id, false, 1.0, 0, 0, nullptr, true);
}

int pcp_grid_append_slab_dev(pcp_grid* g, const double* xyz_dev, size_t n, double own_cell, unsigned long long lo,
                             unsigned long long hi, long long first_id, size_t* n_kept) {
    if (!(own_cell > 0.0)) return pcp_fail("partition cell size must be positive");
    if (lo > hi) return pcp_fail("owned key range is empty");
    return append_host(g, xyz_dev, n, nullptr, first_id, true, own_cell, lo, hi, n_kept, true);
}

int pcp_grid_partition(int device, const double* xyz, size_t n, double own_cell, int world, unsigned long long* cuts) {
    if (!cuts || world < 1) return pcp_fail("cuts is null or world < 1");
    if (!(own_cell > 0.0)) return pcp_fail("partition cell size must be positive");
    if (n && !xyz) return pcp_fail("xyz is null");
    if (n >= (1ull << 31)) return pcp_fail("%zu map points exceed the 2^31 limit", n);
    cuts[0] = 0;
    cuts[world] = 1ull << 63;
    for (int r = 1; r < world; ++r) cuts[r] = 0;
    if (world == 1 || n == 0) return 0;
    int count = 0;
    PCP_CHECK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return pcp_fail("device %d out of range (%d devices)", device, count);
    PCP_CHECK(hipSetDevice(device));
    double* d_xyz = nullptr;
    unsigned long long *k0 = nullptr, *k1 = nullptr, *keys = nullptr;
    unsigned *v0 = nullptr, *v1 = nullptr, *hist = nullptr, *totals = nullptr, *base = nullptr, *vals = nullptr;
    int* d_bad = nullptr;
    auto cleanup = [&]() {
        for (void* q : {(void*)d_xyz, (void*)k0, (void*)k1, (void*)v0, (void*)v1, (void*)hist, (void*)totals, (void*)base, (void*)d_bad})
            if (q) (void)hipFree(q);
    };
    hipError_t e = hipMalloc(&d_xyz, n * 24);
    if (e == hipSuccess) e = hipMemcpy(d_xyz, xyz, n * 24, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&k0, n * 8);
    if (e == hipSuccess) e = hipMalloc(&k1, n * 8);
    if (e == hipSuccess) e = hipMalloc(&v0, n * 4);
    if (e == hipSuccess) e = hipMalloc(&v1, n * 4);
    if (e == hipSuccess) e = hipMalloc(&hist, radix_hist_elems(n) * 4);
    if (e == hipSuccess) e = hipMalloc(&totals, 256 * 4);
    if (e == hipSuccess) e = hipMalloc(&base, 256 * 4);
    if (e == hipSuccess) e = hipMalloc(&d_bad, 4);
    if (e == hipSuccess) e = hipMemset(d_bad, 0, 4);
    int bad = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_partition_keys, dim3(blocks_for(n, 256)), dim3(256), 0, 0, d_xyz, n, own_cell, k0, v0, d_bad);
        radix_sort_pairs(0, k0, k1, v0, v1, n, 63, hist, totals, base, &keys, &vals);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
    for (int r = 1; r < world && e == hipSuccess; ++r) {
        const size_t at = std::min(n - 1, (n * (size_t)r) / (size_t)world);
        e = hipMemcpy(&cuts[r], keys + at, 8, hipMemcpyDeviceToHost);
    }
    cleanup();
    if (e != hipSuccess) return pcp_fail("partitioning failed: %s", hipGetErrorString(e));
    if (bad) return pcp_fail("map coordinates must be finite and within 2^20 partition cells of the origin");
    for (int r = 1; r <= world; ++r) cuts[r] = std::max(cuts[r], cuts[r - 1]);
    return 0;
}

int pcp_grid_info(pcp_grid* g, unsigned long long* n_points, unsigned long long* n_cells, int* extent3,
                  int* sort_passes, float* build_ms, float* sort_ms) {
    if (!g) return pcp_fail("null grid");
    if (check_collecting(g)) return 1;
    if (n_points) *n_points = g->n;
    if (n_cells) *n_cells = g->n_cells;
    if (extent3) for (int a = 0; a < 3; ++a) extent3[a] = g->ext[a];
    if (sort_passes) *sort_passes = g->sort_passes;
    if (build_ms) *build_ms = g->build_ms;
    if (sort_ms) *sort_ms = g->sort_ms;
    return 0;
}

int pcp_grid_sorted_ids(pcp_grid* g, long long* ids, double* xyz) {
    if (!g) return pcp_fail("null grid");
    if (check_collecting(g)) return 1;
    PCP_CHECK(hipSetDevice(g->device));
    std::vector<GridPoint> h(g->n);
    if (g->n) PCP_CHECK(hipMemcpy(h.data(), g->pts, g->n * sizeof(GridPoint), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < g->n; ++i) {
        if (ids) ids[i] = h[i].id;
        if (xyz) { xyz[3 * i] = h[i].x; xyz[3 * i + 1] = h[i].y; xyz[3 * i + 2] = h[i].z; }
    }
    return 0;
}

int pcp_grid_estimate_normals(pcp_grid* g, double radius, int min_points, const double* viewpoint3, float* normals_xyzw,
                              unsigned* counts) {
    if (!g) return pcp_fail("null grid");
    if (check_collecting(g)) return 1;
    if (!(radius > 0.0) || !std::isfinite(radius)) return pcp_fail("radius must be positive and finite");
    if (g->n == 0) return pcp_fail("the grid holds no map (pcp_grid_build)");
    const double cells = std::ceil(radius / g->cell);
    if (cells > 3.0)
        return pcp_fail("radius %g spans more than 3 cells of edge %g: build the grid with larger cells", radius, g->cell);
    PCP_CHECK(hipSetDevice(g->device));
    if (!g->normals) PCP_CHECK(hipMalloc(&g->normals, std::max(g->cap, g->n) * 16));
    unsigned* d_counts = nullptr;
    if (counts) PCP_CHECK(hipMalloc(&d_counts, g->n * 4));
    PcaArgs A;
    A.G = view_of(g);
    A.radius2 = radius * radius;
    A.reach = (int)cells;
    A.min_points = std::max(min_points, 3);
    A.towards = viewpoint3 ? 1 : 0;
    A.vx = viewpoint3 ? viewpoint3[0] : 0.0; A.vy = viewpoint3 ? viewpoint3[1] : 0.0; A.vz = viewpoint3 ? viewpoint3[2] : 0.0;
    A.normals = g->normals;
    A.counts = d_counts;
    hipLaunchKernelGGL(k_grid_pca, dim3(blocks_for(g->n * 32, 256)), dim3(256), 0, g->stream, A);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e == hipSuccess && normals_xyzw) e = hipMemcpy(normals_xyzw, g->normals, g->n * 16, hipMemcpyDeviceToHost);
    if (e == hipSuccess && counts) e = hipMemcpy(counts, d_counts, g->n * 4, hipMemcpyDeviceToHost);
    if (d_counts) (void)hipFree(d_counts);
    if (e != hipSuccess) return pcp_fail("estimating normals failed: %s", hipGetErrorString(e));
    g->have_normals = true;
    g->normals_estimated = true;
    return 0;
}

int pcp_grid_source(pcp_grid* g, const double* xyz, size_t m) {
    if (!g) return pcp_fail("null grid");
    if (m == 0 || !xyz) return pcp_fail("empty source cloud");
    if (m >= (1ull << 31)) return pcp_fail("%zu source points exceed the 2^31 limit", m);
    PCP_CHECK(hipSetDevice(g->device));
    PCP_CHECK(hipStreamSynchronize(g->stream));
    if (reserve_source(g, m)) return 1;
    g->m = m;
    g->nn_valid = false;
    PCP_CHECK(hipMemcpyAsync(g->src_in, xyz, m * 24, hipMemcpyHostToDevice, g->stream));
    PCP_CHECK(hipMemcpyAsync(g->src, g->src_in, m * 24, hipMemcpyDeviceToDevice, g->stream));
    hipLaunchKernelGGL(k_map_state, dim3(1), dim3(1), 0, g->stream, g->state, 1);
    PCP_CHECK(hipStreamSynchronize(g->stream));
    return 0;
}

int pcp_grid_source_rewind(pcp_grid* g) {
    if (check_source(g)) return 1;
    PCP_CHECK(hipSetDevice(g->device));
    PCP_CHECK(hipMemcpyAsync(g->src, g->src_in, g->m * 24, hipMemcpyDeviceToDevice, g->stream));
    hipLaunchKernelGGL(k_map_state, dim3(1), dim3(1), 0, g->stream, g->state, 1);
    PCP_CHECK(hipGetLastError());
    g->nn_valid = false;
    return 0;
}

int pcp_grid_source_read(pcp_grid* g, double* xyz) {
    if (check_source(g)) return 1;
    if (!xyz) return pcp_fail("xyz is null");
    PCP_CHECK(hipSetDevice(g->device));
    PCP_CHECK(hipMemcpyAsync(xyz, g->src, g->m * 24, hipMemcpyDeviceToHost, g->stream));
    PCP_CHECK(hipStreamSynchronize(g->stream));
    return 0;
}

int pcp_grid_source_apply(pcp_grid* g, const double* R9, const double* t3, int minus_t) {
    if (check_source(g)) return 1;
    if (!R9 || !t3) return pcp_fail("R or t is null");
    PCP_CHECK(hipSetDevice(g->device));
    double rt[12];
    memcpy(rt, R9, 72);
    memcpy(rt + 9, t3, 24);
    PCP_CHECK(hipStreamSynchronize(g->stream));
    PCP_CHECK(hipMemcpyAsync(g->rt, rt, sizeof(rt), hipMemcpyHostToDevice, g->stream));
    hipLaunchKernelGGL(k_src_apply, dim3(blocks_for(g->m, 256)), dim3(256), 0, g->stream, g->src, g->m, g->rt,
                       minus_t ? -1.0 : 1.0);
    PCP_CHECK(hipGetLastError());
    PCP_CHECK(hipStreamSynchronize(g->stream));
    g->nn_valid = false;
    return 0;
}

int pcp_grid_query_resident(pcp_grid* g, double max_dist, long long* idx, double* dist, float* kernel_ms) {
    if (check_source(g)) return 1;
    PCP_CHECK(hipSetDevice(g->device));
    const double max_d2 = (max_dist > 0.0 && std::isfinite(max_dist)) ? max_dist * max_dist : INFINITY;
    PCP_CHECK(hipEventRecord(g->ev0, g->stream));
    if (lane_per_query(g)) {
        hipLaunchKernelGGL(k_grid_nn, dim3(blocks_for(g->m, 256)), dim3(256), 0, g->stream, view_of(g), g->src, g->m, max_d2,
                           g->nn_d, g->nn_idx, g->nn_pos);
        PCP_CHECK(hipGetLastError());
    } else if (launch_nn32(g, max_d2, false)) {
        return 1;
    }
    PCP_CHECK(hipEventRecord(g->ev1, g->stream));
    if (idx) PCP_CHECK(hipMemcpyAsync(idx, g->nn_idx, g->m * 8, hipMemcpyDeviceToHost, g->stream));
    if (dist) PCP_CHECK(hipMemcpyAsync(dist, g->nn_d, g->m * 8, hipMemcpyDeviceToHost, g->stream));
    PCP_CHECK(hipStreamSynchronize(g->stream));
    if (kernel_ms) PCP_CHECK(hipEventElapsedTime(kernel_ms, g->ev0, g->ev1));
    g->nn_valid = true;
    return 0;
}

int pcp_grid_query(pcp_grid* g, const double* q_xyz, size_t m, double max_dist, long long* idx, double* dist) {
    if (pcp_grid_source(g, q_xyz, m)) return 1;
    return pcp_grid_query_resident(g, max_dist, idx, dist, nullptr);
}

int pcp_grid_kd_step(pcp_grid* g, double keep_fraction, double* mean_ref, double* mean_other, double* W9,
                     unsigned long long* taken_out) {
    if (check_source(g)) return 1;
    if (g->n == 0) return pcp_fail("the map is empty");
    if (!(keep_fraction > 0.0 && keep_fraction <= 1.0)) return pcp_fail("keep_fraction must be in (0, 1]");
    PCP_CHECK(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    const size_t m = g->m;
    const unsigned long long taken = (unsigned long long)((double)m * keep_fraction);    // int(len * percent)
    if (taken == 0) return pcp_fail("keep_fraction selects no point");
    if (lane_per_query(g)) {
        hipLaunchKernelGGL(k_grid_nn, dim3(blocks_for(m, 256)), dim3(256), 0, s, view_of(g), g->src, m, (double)INFINITY,
                           g->nn_d, g->nn_idx, g->nn_pos);
    } else if (launch_nn32(g, (double)INFINITY, false)) {
        return 1;
    }
    hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(1), 0, s, g->sel_state, taken);
    const unsigned hb = std::min<unsigned>(blocks_for(m, 256), 1024);
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_sel_hist, dim3(hb), dim3(256), 0, s, g->nn_d, m, shift, g->sel_state, g->sel_hist);
        hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(256), 0, s, g->sel_hist, g->sel_state, shift);
    }
    hipLaunchKernelGGL(k_sel_mark, dim3(1), dim3(1024), 0, s, g->nn_d, m, g->sel_state, taken, g->sel);
    const int kb = (int)std::min<size_t>(KD_BLOCKS, blocks_for(m, 256));
    hipLaunchKernelGGL(k_kd_sum<0>, dim3(kb), dim3(256), 0, s, g->pts, g->nn_pos, g->src, g->sel, m, g->kd_out + 16, g->kd_partial);
    hipLaunchKernelGGL(k_kd_final<0>, dim3(1), dim3(16), 0, s, g->kd_partial, kb, g->kd_out);
    hipLaunchKernelGGL(k_kd_sum<1>, dim3(kb), dim3(256), 0, s, g->pts, g->nn_pos, g->src, g->sel, m, g->kd_out + 16, g->kd_partial);
    hipLaunchKernelGGL(k_kd_final<1>, dim3(1), dim3(16), 0, s, g->kd_partial, kb, g->kd_out);
    PCP_CHECK(hipGetLastError());
    double out[64];
    PCP_CHECK(hipMemcpyAsync(out, g->kd_out, sizeof(out), hipMemcpyDeviceToHost, s));
    PCP_CHECK(hipStreamSynchronize(s));
    g->nn_valid = true;
    if ((unsigned long long)out[6] != taken)
        return pcp_fail("trimmed set holds %.0f points, expected %llu", out[6], taken);
    if (mean_ref) memcpy(mean_ref, out + 16, 24);
    if (mean_other) memcpy(mean_other, out + 19, 24);
    if (W9) memcpy(W9, out + 32, 72);
    if (taken_out) *taken_out = taken;
    return 0;
}

int pcp_grid_nn_read(pcp_grid* g, long long* idx, double* dist, unsigned char* selected) {
    if (check_source(g)) return 1;
    if (!g->nn_valid) return pcp_fail("no neighbour results resident (query or kd step first)");
    PCP_CHECK(hipSetDevice(g->device));
    if (idx) PCP_CHECK(hipMemcpyAsync(idx, g->nn_idx, g->m * 8, hipMemcpyDeviceToHost, g->stream));
    if (dist) PCP_CHECK(hipMemcpyAsync(dist, g->nn_d, g->m * 8, hipMemcpyDeviceToHost, g->stream));
    if (selected) PCP_CHECK(hipMemcpyAsync(selected, g->sel, g->m, hipMemcpyDeviceToHost, g->stream));
    PCP_CHECK(hipStreamSynchronize(g->stream));
    return 0;
}

int pcp_grid_set_owned_keys(pcp_grid* g, double own_cell, unsigned long long lo, unsigned long long hi) {
    if (!g) return pcp_fail("null grid");
    if (lo > hi) return pcp_fail("owned key range is empty");
    if (!(own_cell > 0.0)) return pcp_fail("partition cell size must be positive");
    g->own_cell = own_cell;
    g->own_lo = lo;
    g->own_hi = hi;
    return 0;
}

int pcp_grid_cell_keys(double cell, const double* xyz, size_t n, unsigned long long* keys) {
    if (!(cell > 0.0)) return pcp_fail("cell size must be positive");
    for (size_t i = 0; i < n; ++i) {
        unsigned c[3];
        for (int a = 0; a < 3; ++a) {
            double f = floor(xyz[3 * i + a] / cell);
            f = std::min(std::max(f, -(double)COORD_CLAMP), (double)COORD_CLAMP);
            c[a] = (unsigned)((f == f ? (int)f : COORD_CLAMP) + COORD_BIAS);
        }
        keys[i] = morton3(c[0], c[1], c[2]);
    }
    return 0;
}

int pcp_grid_icp_reset(pcp_grid* g) {
    if (check_source(g)) return 1;
    PCP_CHECK(hipSetDevice(g->device));
    hipLaunchKernelGGL(k_map_state, dim3(1), dim3(1), 0, g->stream, g->state, 1);
    PCP_CHECK(hipGetLastError());
    return 0;
}

int pcp_grid_ls_dev(pcp_grid* g, int mode, double radius, float* tile_dev) {
    if (check_source(g)) return 1;
    if (!tile_dev) return pcp_fail("tile_dev is null");
    PCP_CHECK(hipSetDevice(g->device));
    if (launch_map_ls(g, mode, radius, false)) return 1;
    hipLaunchKernelGGL(k_map_finish, dim3(1), dim3(1024), 0, g->stream, map_finish_args(g, tile_dev), 0, mode, 0, 0.0f, 0);
    PCP_CHECK(hipGetLastError());
    return 0;
}

int pcp_grid_pose_dev(pcp_grid* g, const float* tile_dev, int mode, int max_iter, float tol, int full_iter) {
    if (check_source(g)) return 1;
    if (!tile_dev) return pcp_fail("tile_dev is null");
    if (max_iter < 1 || max_iter > pcp_grid::max_iter_cap) return pcp_fail("max_iter must be in [1, %d]", pcp_grid::max_iter_cap);
    PCP_CHECK(hipSetDevice(g->device));
    hipLaunchKernelGGL(k_map_pose, dim3(1), dim3(64), 0, g->stream, tile_dev, g->state, g->hb_trace, g->dx_trace, mode,
                       max_iter, tol, full_iter);
    PCP_CHECK(hipGetLastError());
    return 0;
}

int pcp_grid_ls(pcp_grid* g, int mode, double radius, float* hb) {
    if (check_source(g)) return 1;
    if (!hb) return pcp_fail("hb is null");
    PCP_CHECK(hipSetDevice(g->device));
    hipLaunchKernelGGL(k_map_state, dim3(1), dim3(1), 0, g->stream, g->state, 1);
    if (launch_map_ls(g, mode, radius, true)) return 1;
    hipLaunchKernelGGL(k_map_finish, dim3(1), dim3(1024), 0, g->stream, map_finish_args(g, g->tile), 0, mode, 0, 0.0f, 0);
    PCP_CHECK(hipGetLastError());
    PCP_CHECK(hipMemcpyAsync(hb, g->tile, 256, hipMemcpyDeviceToHost, g->stream));
    PCP_CHECK(hipStreamSynchronize(g->stream));
    g->nn_valid = true;
    return 0;
}

int pcp_grid_icp_finish(pcp_grid* g) {
    if (check_source(g)) return 1;
    PCP_CHECK(hipSetDevice(g->device));
    hipLaunchKernelGGL(k_map_apply_pending, dim3(blocks_for(g->m, 256)), dim3(256), 0, g->stream, g->src, g->m, g->state);
    hipLaunchKernelGGL(k_map_state, dim3(1), dim3(1), 0, g->stream, g->state, 0);
    PCP_CHECK(hipGetLastError());
    return 0;
}

int pcp_grid_icp_fetch(pcp_grid* g, float* dx_trace, float* hb_trace, int* iters, int* converged, int* done) {
    if (check_source(g)) return 1;
    PCP_CHECK(hipSetDevice(g->device));
    IcpState st;
    PCP_CHECK(hipMemcpyAsync(&st, g->state, sizeof(st), hipMemcpyDeviceToHost, g->stream));
    PCP_CHECK(hipStreamSynchronize(g->stream));
    if (st.singular) return pcp_fail("singular normal equations at iteration %d", st.iter - 1);
    const int it = std::min(st.iter, (int)pcp_grid::max_iter_cap);
    if (dx_trace && it) PCP_CHECK(hipMemcpy(dx_trace, g->dx_trace, (size_t)it * 24, hipMemcpyDeviceToHost));
    if (hb_trace && it) PCP_CHECK(hipMemcpy(hb_trace, g->hb_trace, (size_t)it * 256, hipMemcpyDeviceToHost));
    if (iters) *iters = it;
    if (converged) *converged = st.converged;
    if (done) *done = st.done;
    return 0;
}

int pcp_grid_icp_run(pcp_grid* g, int mode, double radius, int max_iter, float tol, int full_iter, float* dx_trace,
                     float* hb_trace, int* iters, int* converged) {
    if (check_source(g)) return 1;
    if (max_iter < 1 || max_iter > pcp_grid::max_iter_cap) return pcp_fail("max_iter must be in [1, %d]", pcp_grid::max_iter_cap);
    PCP_CHECK(hipSetDevice(g->device));
    hipLaunchKernelGGL(k_map_state, dim3(1), dim3(1), 0, g->stream, g->state, 1);
    for (int it = 0; it < max_iter; ++it) {
        if (launch_map_ls(g, mode, radius, false)) return 1;
        hipLaunchKernelGGL(k_map_finish, dim3(1), dim3(1024), 0, g->stream, map_finish_args(g, g->hb_trace), 1, mode,
                           max_iter, tol, full_iter);
    }
    PCP_CHECK(hipGetLastError());
    if (pcp_grid_icp_finish(g)) return 1;
    return pcp_grid_icp_fetch(g, dx_trace, hb_trace, iters, converged, nullptr);
}

int pcp_grid_allreduce_tile(pcp_grid* g, pcp_comm* comm, float* tile_dev) {
    if (!g) return pcp_fail("null grid");
    return pcp_comm_allreduce_tile(comm, tile_dev, (void*)g->stream);
}

int pcp_grid_icp_run_comm(pcp_grid* g, pcp_comm* comm, int mode, double radius, int max_iter, float tol, int full_iter,
                          float* dx_trace, float* hb_trace, int* iters, int* converged) {
    if (check_source(g)) return 1;
    if (!comm) return pcp_fail("null communicator");
    if (max_iter < 1 || max_iter > pcp_grid::max_iter_cap) return pcp_fail("max_iter must be in [1, %d]", pcp_grid::max_iter_cap);
    PCP_CHECK(hipSetDevice(g->device));
    if (!g->comm_tile) PCP_CHECK(hipMalloc(&g->comm_tile, 64 * sizeof(float)));
    if (pcp_grid_icp_reset(g)) return 1;
    // Every rank enqueues the same max_iter exchanges: the summed tile, hence the solved step and the
    // convergence flag, are identical on all ranks, and once a rank's loop is done its launches return at
    // once while the (256-byte) all-reduces keep the ranks in step.  The partial tile is rewritten from the rank's
    // level-1 sums before every exchange (k_map_finish with solve = 0 does not look at the done flag), so the
    // exchanges behind the loop's end sum the same partial tiles again and again: nothing accumulates in it.
    // A rank that fails in here leaves its peers inside their next all-reduce: the caller of a failed rank has to
    // abort the communicator (pcp_comm_abort), which releases them with an error.
    for (int it = 0; it < max_iter; ++it) {
        if (pcp_grid_ls_dev(g, mode, radius, g->comm_tile)) return 1;
        if (pcp_comm_allreduce_tile(comm, g->comm_tile, (void*)g->stream)) return 1;
        if (pcp_grid_pose_dev(g, g->comm_tile, mode, max_iter, tol, full_iter)) return 1;
    }
    if (pcp_grid_icp_finish(g)) return 1;
    return pcp_grid_icp_fetch(g, dx_trace, hb_trace, iters, converged, nullptr);
}

}  // extern "C"
