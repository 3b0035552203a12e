// Voxelizer / voxel stager kernels and their C ABI (see include/pcp.h).
//
// Device layout (the reference's SSBO bindings 6-8 and 10-12, ComputeShader.py:144-203):
//   table   int4[M]            (vx, vy, vz, data_row), lexicographically sorted, voxel_index
//   data    int[M][max_points] [0] = count (may exceed max_points-1), [1..] = global point ids
//   stat    float4[M]          (prob, count + max prob, stage status, staged), by data_row
//   stage   int[max_staging][max_points]   staged rows, never cleared between calls
//   unknown int[1 + n]         [0] = count, then ids of points whose voxel is not in the table
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/pcp.h"
#include "pcp_compact.h"
#include "pcp_ctx.h"
#include "pcp_host.h"
#include "pcp_sort.h"

struct pcp_voxel_state {
    int4* table = nullptr;   int table_rows = 0, table_valid = 0;   // capacity / rows of the last upload
    int* data = nullptr;     int data_rows = 0, data_width = 0;
    bool data_dirty = false; // rows hold the ids of a map that has been reset
    float4 *stat = nullptr, *stat_in = nullptr;   int stat_rows = 0;
    int stat_valid = 0;      // rows of `stat` that hold statistics (resident mode); rows behind them are made (0.5, 0, 0, 0)
    int* stage = nullptr;    int stage_rows = 0, stage_width = 0;
    int* unknown = nullptr;  int unknown_cap = 0;
    float4* pts = nullptr;   float4* debug = nullptr;   int pts_cap = 0;
    int* counter = nullptr;  // 16 ints
    // state of the last pcp_voxelize call, consumed by pcp_voxel_insert_unknown
    int last_n = 0, last_begin = 0, last_unknown = -1, last_stored = 0;
    float last_size = 0.0f;
    // scratch of the insertion
    int4* table2 = nullptr;  int table2_rows = 0;
    unsigned char* flag = nullptr;
    unsigned long long *k0 = nullptr, *k1 = nullptr;
    unsigned *v0 = nullptr, *v1 = nullptr, *heads = nullptr, *hist = nullptr, *totals = nullptr, *base = nullptr,
             *counts = nullptr;
    int ins_cap = 0;
    int* mm = nullptr;       // 8 ints: coordinate range of the unknown points
    // device copy of the Voxelizer's global point store (added_points, Voxelizer.py:398-425), kept on request:
    // point id = begin_index + i of the dispatch that brought it
    bool keep_store = false;
    float4* store = nullptr; size_t store_cap = 0, store_n = 0;
    // ---- frame feed (pcp_voxel_add*): everything a frame needs to know sits in device memory
    struct VoxFeed* feed = nullptr;          // device control block
    struct pcp_voxel_frame* feed_rec = nullptr;   // pinned ring of per-frame records
    struct pcp_voxel_frame* feed_rec_dev = nullptr;
    hipEvent_t feed_done[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t feed_after = nullptr;          // orders the context's stream behind the stream that produced a device input
    int feed_serial = 0;                     // frames enqueued so far
    int feed_waited = 0;                     // frames whose record has been taken
    std::vector<pcp_voxel_frame> feed_history;    // the last PCP_FEED_HISTORY records taken, by serial % PCP_FEED_HISTORY
    int feed_known_stored = -1;              // voxels in the table after the last frame waited for (-1: not in feed mode)
    long long feed_pending_points = 0;       // points of the frames enqueued since
    unsigned* tile_counts = nullptr; int tile_cap = 0;   // per 256-point tile: unknown points / group heads
    unsigned* rs_base = nullptr;
    void* feed_in = nullptr; size_t feed_in_bytes = 0;   // staging of host input
    int last_staged = 0;     // rows the last pcp_voxel_stage call staged
    unsigned* staged_off = nullptr;          // staged_off_rows + 2: exclusive scan of the staged rows' point counts, total, error flag
    int staged_off_rows = 0;
    double* staged_xyz = nullptr; size_t staged_cap = 0;
    // ---- log of staged points (pcp_voxel_log_*): every fed frame appends the points of the rows it staged to the open slot
    double* log_xyz[2] = {nullptr, nullptr}; size_t log_cap[2] = {0, 0};      // capacity in points
    unsigned* log_ctl = nullptr;             // device, 4 words per slot: points, overflow flag, bad-id flag, rows
    unsigned* log_host = nullptr;            // page-locked copy, written by pcp_voxel_log_end
    unsigned* log_off = nullptr; int log_off_rows = 0;    // per staged row: where its points start in the log; [rows + 1] = rows to copy
    hipEvent_t log_done[2] = {nullptr, nullptr};
    int log_state[2] = {0, 0};               // 0 idle, 1 open (frames append), 2 closed (count on its way), per slot
    int log_slot = -1;                       // the open slot
};

// Control block of the frame feed (pcp_voxel_add*), in device memory.
struct VoxFeed {
    int stored;              // voxels in the table (the device's count is the authority while frames are fed)
    int mm[6];               // min x, y, z / max x, y, z of the voxel coordinates of the frame's unknown points
    pcp::RsCtl sort;         // unknown points, radix passes their keys need
    int bx, by, bz;          // key layout: x in the top bx bits, then y, then z, each relative to mm[0..3)
    int n_new, flags, frame;
    int pad[16];
};

void pcp_voxel_state_free(pcp_voxel_state* v) {
    if (!v) return;
    void* bufs[] = {v->table, v->data, v->stat, v->stat_in, v->stage, v->unknown, v->pts, v->debug, v->counter,
                    v->table2, v->flag, v->k0, v->k1, v->v0, v->v1, v->heads, v->hist, v->totals, v->base, v->counts, v->mm,
                    v->store, v->staged_off, v->staged_xyz, v->feed, v->feed_rec_dev, v->tile_counts, v->rs_base, v->feed_in,
                    v->log_xyz[0], v->log_xyz[1], v->log_ctl, v->log_off};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (v->feed_rec) (void)hipHostFree(v->feed_rec);
    if (v->log_host) (void)hipHostFree(v->log_host);
    for (hipEvent_t ev : v->log_done) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : v->feed_done) if (ev) (void)hipEventDestroy(ev);
    if (v->feed_after) (void)hipEventDestroy(v->feed_after);
    delete v;
}

namespace {

// binSearch, VoxelShader.glsl:38-62 (identical copy in VoxelStage.glsl:65-89)
__device__ __forceinline__ int comp(const int4 v, const int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

__device__ int bin_search(const int4* __restrict__ table, const int axis, const int value, const bool first,
                          int l, int h) {
    int occ = -1;
    while (l <= h) {
        const int mid = (l + h) / 2;
        const int v = comp(table[mid], axis);
        if (v == value) {
            occ = mid;
            if (first) h = mid - 1; else l = mid + 1;
        } else if (v < value) {
            l = mid + 1;
        } else {
            h = mid - 1;
        }
    }
    return occ;
}

// findVoxelId, VoxelShader.glsl:64-98
__device__ int find_voxel(const int4* __restrict__ table, const int stored, const int x, const int y, const int z) {
    if (stored == 0) return -1;
    const int fx = bin_search(table, 0, x, true, 0, stored - 1);
    if (fx == -1) return -1;
    const int lx = bin_search(table, 0, x, false, 0, stored - 1);
    const int fy = bin_search(table, 1, y, true, fx, lx);
    if (fy == -1) return -1;
    const int ly = bin_search(table, 1, y, false, fx, lx);
    return bin_search(table, 2, z, true, fy, ly);
}

// findVoxelId on a table that IS in lexicographic order (the invariant Voxelizer.storeUnknownPoints keeps, Voxelizer.py:170-306,
// and the device merge with it): the first entry not below (x, y, z) in one bisection.  On such a table the five bisections
// of find_voxel end at this very entry -- the x range, the y range inside it, the first z inside that -- so the answer is the
// same, after a fifth of the dependent loads.  Used where the library itself maintains the table (the frame feed).
__device__ int find_voxel_sorted(const int4* __restrict__ table, const int stored, const int x, const int y, const int z) {
    int l = 0, h = stored;
    while (l < h) {
        const int mid = (l + h) >> 1;
        const int4 v = table[mid];
        const bool below = v.x < x || (v.x == x && (v.y < y || (v.y == y && v.z < z)));
        if (below) l = mid + 1; else h = mid;
    }
    if (l >= stored) return -1;
    const int4 v = table[l];
    return (v.x == x && v.y == y && v.z == z) ? l : -1;
}

// (x, y, z) records of 12 bytes -> the (x, y, z, 0) records the kernels read (pad_points, ComputeShader.py:956-961)
__global__ void k_pad_xyz(const float* __restrict__ xyz, int n, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float4(xyz[3 * (size_t)i + 0], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], 0.0f);
}

// three coordinate rows of n floats each (a transposed cloud as it lies in memory) -> the same records
__global__ void k_pad_rows(const float* __restrict__ rows, int n, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float4(rows[i], rows[(size_t)n + i], rows[2 * (size_t)n + i], 0.0f);
}

// VoxelShader.glsl:100-146
__global__ void __launch_bounds__(256)
k_voxelize(const float4* __restrict__ pts, int n, const int4* __restrict__ table, int stored,
           int* __restrict__ data, int data_rows, int max_points, int begin_index, float voxel_size,
           int* __restrict__ unknown, float4* __restrict__ debug) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const int vx = (int)ceilf(p.x / voxel_size);       // getVoxelCoord, :33-36
    const int vy = (int)ceilf(p.y / voxel_size);
    const int vz = (int)ceilf(p.z / voxel_size);
    const int entry = find_voxel(table, stored, vx, vy, vz);
    if (debug) debug[i] = make_float4((float)vx, (float)vy, (float)vz, (float)entry);
    if (entry == -1) {
        const int k = atomicAdd(&unknown[0], 1);
        unknown[1 + k] = begin_index + i;
    } else {
        const int row = table[entry].w;
        if ((unsigned)row < (unsigned)data_rows) {      // a malformed table must not fault the device
            const int c = atomicAdd(&data[(size_t)row * max_points], 1);
            if (c < max_points - 1) data[(size_t)row * max_points + 1 + c] = begin_index + i;
        }
    }
}

// BayesFilter, VoxelStage.glsl:37-39
__device__ __forceinline__ float bayes(const float p, const float alpha, const float beta) {
    return (alpha * p) / (alpha * p + beta * (1.0f - p));
}

// recoverVoxel + groundRule1/2, VoxelStage.glsl:133-197.  Neighbours are read from
// the statistics as uploaded (stat_in): the shader reads them while other threads
// write, this is its all-reads-first outcome.
__device__ bool recover_voxel(const int4* __restrict__ table, const int stored, const float4* __restrict__ stat_in,
                              const int stat_rows, const int4 vid) {
    const int sx[6] = {0, 0, 1, -1, 0, 0}, sy[6] = {1, -1, 0, 0, 0, 0}, sz[6] = {0, 0, 0, 0, 1, -1};
    bool below_found = false;
    int horizontal = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int e = find_voxel(table, stored, vid.x + sx[k], vid.y + sy[k], vid.z + sz[k]);
        bool is_static = false;
        if (e != -1) {
            const int row = table[e].w;
            if ((unsigned)row < (unsigned)stat_rows) {
                const float4 s = stat_in[row];
                is_static = s.w > 0.5f || s.z > 0.1f;
            }
            if (k == 1) below_found = true;
        }
        if (k >= 2 && is_static) ++horizontal;
    }
    return horizontal >= 3 || (horizontal >= 2 && !below_found);
}

// VoxelStage.glsl:201-315.  One lane per table entry; the row copy of
// tryStageVoxel (:41-60) is done by the whole wave for each staging lane in turn.
__global__ void __launch_bounds__(256)
k_voxel_stage(const int4* __restrict__ table, int stored, const int* __restrict__ data, int data_rows,
              int max_points, float4* __restrict__ stat, const float4* __restrict__ stat_in, int stat_rows,
              int* __restrict__ stage, int max_staging, int stage_everything, int no_filter,
              int* __restrict__ counter, const VoxFeed* __restrict__ feed) {
    if (feed) stored = feed->stored;                          // frame feed: the device's own count
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool live = pos < stored;
    int idx = 0, cnt = 0, slot = -1;
    int4 vid = make_int4(0, 0, 0, 0);
    if (live) {
        vid = table[pos];
        idx = vid.w;
        live = (unsigned)idx < (unsigned)stat_rows && (unsigned)idx < (unsigned)data_rows;
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        s = stat[idx];
        live = !(s.w > 0.5f);                              // staged in an earlier run: nothing to do
    }
    bool want_stage = false;
    if (live) {
        cnt = data[(size_t)idx * max_points];
        const float status = s.z;                          // stays stale below, as in the shader (:238-264)
        if (stage_everything) {
            if (status > 0.1f) s.z = 0.8f;
            else if (!no_filter && recover_voxel(table, stored, stat_in, stat_rows, vid)) s.z = 0.8f;
        }
        if (status > 0.5f) {
            want_stage = true;
        } else {
            const float mpp = s.y;
            float max_prob = mpp - (float)(int)mpp;
            const int prev = (int)mpp;
            const float alpha = 0.55f, beta = 0.45f;
            const float nw = (cnt == prev) ? bayes(s.x, 1.0f - alpha, 1.0f - beta) : bayes(s.x, alpha, beta);
            if (nw > max_prob) max_prob = nw;
            s.x = nw;
            if (max_prob > 0.999f) max_prob = 0.999f;
            if (max_prob < 0.001f) max_prob = 0.001f;
            s.y = (float)cnt + max_prob;
            if (nw > 0.9f || status > 0.1f || no_filter) {
                if (status < 0.1f) s.z = 0.2f;
                if (cnt + 1 >= max_points) s.z = 0.8f;
            }
        }
    }
    if (want_stage) {                                      // tryStageVoxel's counter protocol
        slot = atomicAdd(&counter[0], 1);
        if (slot >= max_staging) {
            atomicAdd(&counter[0], -1);
            slot = -1;
        } else {
            s.w = 1.0f;
        }
    }
    if (live) stat[idx] = s;
    unsigned long long todo = __ballot(slot >= 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int r_idx = __shfl(idx, src, 64), r_cnt = __shfl(cnt, src, 64), r_slot = __shfl(slot, src, 64);
        const int* from = data + (size_t)r_idx * max_points;
        int* to = stage + (size_t)r_slot * max_points;
        if (lane == 0) to[0] = r_cnt;
        const int end = min(r_cnt + 1, max_points);
        for (int i = 1 + lane; i < end; i += 64) to[i] = from[i];
    }
}


// ------------------------------------------------------------------------------------------
// Creation of new voxels on the device (the host half of Voxelizer.storeUnknownPoints,
// Voxelizer.py:170-306): the points the last dispatch found no voxel for are grouped by
// voxel coordinate in lexicographic order, every group becomes a voxel whose data row is
// numbered stored + rank, holds the count of ALL its points and the ids of the first
// max_points - 1 of them (ascending ids), and the new entries are merged into the sorted table.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int3 voxel_coord(const float4 p, const float voxel_size) {
    return make_int3((int)ceilf(p.x / voxel_size), (int)ceilf(p.y / voxel_size), (int)ceilf(p.z / voxel_size));
}

__global__ void k_mark_unknown(const int* __restrict__ unknown, int count, int begin_index, unsigned char* __restrict__ flag) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) flag[unknown[1 + j] - begin_index] = 1;
}

__global__ void __launch_bounds__(256)
k_unknown_range(const float4* __restrict__ pts, const unsigned char* __restrict__ flag, int n, float voxel_size,
                int* __restrict__ mm) {
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!flag[i]) continue;
        const int3 c = voxel_coord(pts[i], voxel_size);
        lo[0] = min(lo[0], c.x); hi[0] = max(hi[0], c.x);
        lo[1] = min(lo[1], c.y); hi[1] = max(hi[1], c.y);
        lo[2] = min(lo[2], c.z); hi[2] = max(hi[2], c.z);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = min(lo[a], __shfl_down(lo[a], s, 64));
            hi[a] = max(hi[a], __shfl_down(hi[a], s, 64));
        }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&mm[a], lo[a]);
            atomicMax(&mm[3 + a], hi[a]);
        }
}

struct IsUnknown {
    const unsigned char* flag;
    __device__ bool operator()(size_t i) const { return flag[i] != 0; }
};

// Emits (lexicographic key relative to the range's corner, global point id) in point order.
struct EmitKey {
    const float4* pts;
    float voxel_size;
    int mx, my, mz, bits, begin_index;
    unsigned long long* keys;
    unsigned* vals;
    __device__ void operator()(size_t i, size_t slot) const {
        const int3 c = voxel_coord(pts[i], voxel_size);
        keys[slot] = ((unsigned long long)(unsigned)(c.x - mx) << (2 * bits)) |
                     ((unsigned long long)(unsigned)(c.y - my) << bits) | (unsigned long long)(unsigned)(c.z - mz);
        vals[slot] = (unsigned)(begin_index + (int)i);
    }
};

struct IsHead {
    const unsigned long long* keys;
    __device__ bool operator()(size_t j) const { return j == 0 || keys[j] != keys[j - 1]; }
};
struct EmitHead {
    unsigned* heads;
    __device__ void operator()(size_t j, size_t slot) const { heads[slot] = (unsigned)j; }
};

__device__ __forceinline__ int lex_cmp(const int x0, const int y0, const int z0, const int4 b) {
    if (x0 != b.x) return x0 < b.x ? -1 : 1;
    if (y0 != b.y) return y0 < b.y ? -1 : 1;
    if (z0 != b.z) return z0 < b.z ? -1 : 1;
    return 0;
}

__device__ __forceinline__ int3 key_coord(const unsigned long long key, const int bits, const int mx, const int my, const int mz) {
    const unsigned long long mask = (1ull << bits) - 1ull;
    return make_int3((int)((key >> (2 * bits)) & mask) + mx, (int)((key >> bits) & mask) + my, (int)(key & mask) + mz);
}

// One wave per new voxel: data row stored + k = [count, first max_points-1 ids]; lane 0 also
// places the voxel's table entry at k + (number of old entries below it).
__global__ void __launch_bounds__(256)
k_new_voxels(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals, const unsigned* __restrict__ heads,
             int n_new, int n_unknown, int bits, int mx, int my, int mz, const int4* __restrict__ table, int stored,
             int4* __restrict__ table_out, int* __restrict__ data, int max_points, int* __restrict__ clash) {
    const int k = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (k >= n_new) return;
    const unsigned h = heads[k];
    const unsigned end = (k + 1 < n_new) ? heads[k + 1] : (unsigned)n_unknown;
    const int count = (int)(end - h);
    const int row = stored + k;
    int* out = data + (size_t)row * max_points;
    if (lane == 0) out[0] = count;
    const int keep = min(count, max_points - 1);
    for (int j = lane; j < keep; j += 64) out[1 + j] = (int)vals[h + j];
    for (int j = keep + lane; j < max_points - 1; j += 64) out[1 + j] = 0;
    if (lane == 0) {
        const int3 c = key_coord(keys[h], bits, mx, my, mz);
        int lo = 0, hi = stored;                       // first old entry not below c
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const int cmp = lex_cmp(c.x, c.y, c.z, table[mid]);
            if (cmp == 0) *clash = 1;
            if (cmp > 0) lo = mid + 1; else hi = mid;
        }
        table_out[k + lo] = make_int4(c.x, c.y, c.z, row);
    }
}

// Old entry i moves to i + (number of new voxels below it).
__global__ void __launch_bounds__(256)
k_shift_old(const int4* __restrict__ table, int stored, const unsigned long long* __restrict__ keys,
            const unsigned* __restrict__ heads, int n_new, int bits, int mx, int my, int mz, int4* __restrict__ table_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= stored) return;
    const int4 e = table[i];
    int lo = 0, hi = n_new;                            // first new voxel not below e
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int3 c = key_coord(keys[heads[mid]], bits, mx, my, mz);
        if (lex_cmp(c.x, c.y, c.z, e) < 0) lo = mid + 1; else hi = mid;
    }
    table_out[i + lo] = e;
}

__global__ void k_gather_counts(const int* __restrict__ data, int first_row, int n_rows, int max_points, int* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_rows) out[k] = data[(size_t)(first_row + k) * max_points];
}

// Points of the staged rows, row after row in staging order, each row's ids in row order: what
// Voxelizer.getStoredPointsFromVoxelDataSlice (:396-399) looks up per row, ids = row[1 : min(row[0] + 1, width)].
// One block: exclusive scan of the rows' point counts (rows <= 1024 per pass, carried over passes).
__global__ void __launch_bounds__(1024) k_staged_offsets(const int* __restrict__ stage, int rows, int width,
                                                          unsigned* __restrict__ off) {
    __shared__ unsigned sc[1024];
    __shared__ unsigned run;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    for (int base = 0; base < rows; base += 1024) {
        const int r = base + threadIdx.x;
        unsigned n = 0;
        if (r < rows) {
            const int cnt = stage[(size_t)r * width];
            n = (unsigned)max(0, min(cnt, width - 1));
        }
        sc[threadIdx.x] = n;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const unsigned t = threadIdx.x >= (unsigned)d ? sc[threadIdx.x - d] : 0u;
            __syncthreads();
            sc[threadIdx.x] += t;
            __syncthreads();
        }
        if (r < rows) off[r] = run + sc[threadIdx.x] - n;
        __syncthreads();
        if (threadIdx.x == 0) run += sc[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) { off[rows] = run; off[rows + 1] = 0; }
}

__global__ void __launch_bounds__(256) k_staged_gather(const int* __restrict__ stage, int width, const float4* __restrict__ store,
                                                        size_t store_n, unsigned* __restrict__ off, int rows,
                                                        double* __restrict__ out) {
    const int r = blockIdx.x;
    const unsigned first = off[r], n = off[r + 1] - first;        // off[rows] is the total: the last row's end
    const int* ids = stage + (size_t)r * width + 1;
    for (unsigned j = threadIdx.x; j < n; j += blockDim.x) {
        const int id = ids[j];
        double x = 0.0, y = 0.0, z = 0.0;
        if (id < 0 || (size_t)id >= store_n) {
            off[rows + 1] = 1;                                     // an id the store does not hold
        } else {
            const float4 p = store[id];
            x = (double)p.x; y = (double)p.y; z = (double)p.z;
        }
        double* o = out + 3 * (size_t)(first + j);
        o[0] = x; o[1] = y; o[2] = z;
    }
}

// The same for the rows a fed frame staged, appended to a log other frames have written to before: the number of rows is
// the device's (counter[0] of the stager that just ran), the points go behind the ctl[0] points the log holds.  A frame whose
// points do not fit leaves the log as it was and raises ctl[1].  off[max_rows + 1] = rows k_log_gather copies.
__global__ void __launch_bounds__(1024) k_log_offsets(const int* __restrict__ stage, const int* __restrict__ counter, int max_rows,
                                                       int width, unsigned* __restrict__ off, unsigned* __restrict__ ctl,
                                                       unsigned capacity) {
    __shared__ unsigned sc[1024];
    __shared__ unsigned run;
    const int rows = max(0, min(counter[0], max_rows));
    const unsigned base = ctl[0];
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    for (int first = 0; first < rows; first += 1024) {
        const int r = first + threadIdx.x;
        unsigned n = 0;
        if (r < rows) n = (unsigned)max(0, min(stage[(size_t)r * width], width - 1));
        sc[threadIdx.x] = n;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const unsigned t = threadIdx.x >= (unsigned)d ? sc[threadIdx.x - d] : 0u;
            __syncthreads();
            sc[threadIdx.x] += t;
            __syncthreads();
        }
        if (r < rows) off[r] = base + run + sc[threadIdx.x] - n;
        __syncthreads();
        if (threadIdx.x == 0) run += sc[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const bool fits = (unsigned long long)base + run <= capacity;
        off[rows] = base + run;
        off[max_rows + 1] = fits ? (unsigned)rows : 0u;
        if (fits) { ctl[0] = base + run; ctl[3] += (unsigned)rows; }
        else ctl[1] = 1;
    }
}

__global__ void __launch_bounds__(256) k_log_gather(const int* __restrict__ stage, int width, const float4* __restrict__ store,
                                                     size_t store_n, const unsigned* __restrict__ off, int max_rows,
                                                     double* __restrict__ out, unsigned* __restrict__ ctl) {
    const int r = blockIdx.x;
    if ((unsigned)r >= off[max_rows + 1]) return;
    const unsigned first = off[r], n = off[r + 1] - first;
    const int* ids = stage + (size_t)r * width + 1;
    for (unsigned j = threadIdx.x; j < n; j += blockDim.x) {
        const int id = ids[j];
        double x = 0.0, y = 0.0, z = 0.0;
        if (id < 0 || (size_t)id >= store_n) {
            ctl[2] = 1;                                            // an id the store does not hold
        } else {
            const float4 p = store[id];
            x = (double)p.x; y = (double)p.y; z = (double)p.z;
        }
        double* o = out + 3 * (size_t)(first + j);
        o[0] = x; o[1] = y; o[2] = z;
    }
}

// Statistics of voxels that do not exist yet: probability 0.5, nothing else (Voxelizer.initVoxels, Voxelizer.py:382-388).
__global__ void k_stat_init(float4* __restrict__ stat, int first, int rows) {
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) stat[i] = make_float4(0.5f, 0.0f, 0.0f, 0.0f);
}


// ------------------------------------------------------------------------------------------
// Frame feed: Voxelizer.addPoints (Voxelizer.py:81-133) as one stream of kernels.  The counts the
// step-by-step entry points hand to the host between their steps live in this control block.
// ------------------------------------------------------------------------------------------
#define PCP_FEED_MAX_PASSES 6                // 48 key bits: 2^16 voxels per axis among a frame's unknown points
#define PCP_FEED_RING 8
#define PCP_FEED_HISTORY 1024                // records of frames already waited for stay readable this long

__device__ __forceinline__ void feed_reset(VoxFeed* F) {
    F->mm[0] = F->mm[1] = F->mm[2] = INT_MAX;
    F->mm[3] = F->mm[4] = F->mm[5] = INT_MIN;
    F->sort.n = 0; F->sort.passes = 0;
    F->bx = F->by = F->bz = 0;
    F->n_new = 0; F->flags = 0;
}

__global__ void k_feed_init(VoxFeed* F, int stored, int* counter) {
    feed_reset(F);
    F->stored = stored;
    F->frame = 0;
    counter[0] = 0;
}

template <int LAYOUT>
__device__ __forceinline__ float4 feed_point(const void* __restrict__ src, const int n, const int i) {
    if (LAYOUT == PCP_PTS_F32_XYZW) return reinterpret_cast<const float4*>(src)[i];
    if (LAYOUT == PCP_PTS_F32_XYZ) {
        const float* p = reinterpret_cast<const float*>(src) + 3 * (size_t)i;
        return make_float4(p[0], p[1], p[2], 0.0f);
    }
    if (LAYOUT == PCP_PTS_F32_ROWS) {
        const float* p = reinterpret_cast<const float*>(src);
        return make_float4(p[i], p[(size_t)n + i], p[2 * (size_t)n + i], 0.0f);
    }
    const double* p = reinterpret_cast<const double*>(src) + 3 * (size_t)i;      // handleAddingPoints' float32 cast
    return make_float4((float)p[0], (float)p[1], (float)p[2], 0.0f);
}

// VoxelShader.glsl:100-146 for one frame, with what the creation of new voxels needs noted on the way: a point
// whose voxel is not in the table is flagged (not appended to a list: the list is made in point order by
// k_feed_emit), its tile's count and the coordinate range of all such points are kept.
template <int LAYOUT>
__global__ void __launch_bounds__(256)
k_feed_voxelize(const void* __restrict__ src, int n, const int4* __restrict__ table, VoxFeed* __restrict__ F,
                int* __restrict__ data, int data_rows, int max_points, int begin_index, float voxel_size,
                float4* __restrict__ pts, float4* __restrict__ store, unsigned char* __restrict__ flag,
                unsigned* __restrict__ tile_counts) {
    __shared__ int s_cnt[4], s_lo[4][3], s_hi[4][3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int stored = F->stored;
    bool unknown = false;
    int3 c = make_int3(0, 0, 0);
    if (i < n) {
        const float4 p = feed_point<LAYOUT>(src, n, i);
        pts[i] = p;
        if (store) store[i] = p;
        c = voxel_coord(p, voxel_size);
        const int entry = find_voxel_sorted(table, stored, c.x, c.y, c.z);
        if (entry == -1) {
            unknown = true;
        } else {
            const int row = table[entry].w;
            if ((unsigned)row < (unsigned)data_rows) {
                const int k = atomicAdd(&data[(size_t)row * max_points], 1);
                if (k < max_points - 1) data[(size_t)row * max_points + 1 + k] = begin_index + i;
            }
        }
        flag[i] = unknown ? 1 : 0;
    }
    const unsigned long long bal = __ballot(unknown);
    int lo[3] = {unknown ? c.x : INT_MAX, unknown ? c.y : INT_MAX, unknown ? c.z : INT_MAX};
    int hi[3] = {unknown ? c.x : INT_MIN, unknown ? c.y : INT_MIN, unknown ? c.z : INT_MIN};
    if (bal) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = min(lo[a], __shfl_xor(lo[a], s, 64));
                hi[a] = max(hi[a], __shfl_xor(hi[a], s, 64));
            }
    }
    if (lane == 0) {
        s_cnt[w] = (int)__popcll(bal);
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_lo[w][a] = lo[a]; s_hi[w][a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        tile_counts[blockIdx.x] = (unsigned)total;
        if (total) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                atomicMin(&F->mm[a], min(min(s_lo[0][a], s_lo[1][a]), min(s_lo[2][a], s_lo[3][a])));
                atomicMax(&F->mm[3 + a], max(max(s_hi[0][a], s_hi[1][a]), max(s_hi[2][a], s_hi[3][a])));
            }
        }
    }
}

__device__ __forceinline__ int bits_for(const long long span) {      // smallest b with 2^b >= span (span >= 1)
    int b = 0;
    while ((1ll << b) < span) ++b;
    return b;
}

// Sum of counts[0 .. upto) and of counts[0 .. all) by the whole block (256 threads).
__device__ __forceinline__ void tile_prefix(const unsigned* __restrict__ counts, const int upto, const int all,
                                            unsigned& before, unsigned& total) {
    __shared__ unsigned s_b[4], s_t[4];
    unsigned b = 0, t = 0;
    for (int k = threadIdx.x; k < all; k += 256) {
        const unsigned v = counts[k];
        t += v;
        if (k < upto) b += v;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { b += __shfl_xor(b, s, 64); t += __shfl_xor(t, s, 64); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_b[threadIdx.x >> 6] = b; s_t[threadIdx.x >> 6] = t; }
    __syncthreads();
    before = s_b[0] + s_b[1] + s_b[2] + s_b[3];
    total = s_t[0] + s_t[1] + s_t[2] + s_t[3];
}

// (voxel key, point id) of every flagged point in point order (ids ascending): slot = flagged points before it.
// Block 0 also fixes the key layout from the coordinate range and the number of radix passes it needs.
__global__ void __launch_bounds__(256)
k_feed_emit(const float4* __restrict__ pts, const unsigned char* __restrict__ flag, int n, const unsigned* __restrict__ tile_counts,
            int tiles, VoxFeed* __restrict__ F, int begin_index, float voxel_size, unsigned long long* __restrict__ keys,
            unsigned* __restrict__ vals) {
    __shared__ unsigned s_w[4];
    unsigned before, total;
    tile_prefix(tile_counts, (int)blockIdx.x, tiles, before, total);
    if (total == 0) return;                                   // nothing unknown in this frame (block-uniform)
    const int mx = F->mm[0], my = F->mm[1], mz = F->mm[2];
    const int bx = bits_for((long long)F->mm[3] - mx + 1), by = bits_for((long long)F->mm[4] - my + 1),
              bz = bits_for((long long)F->mm[5] - mz + 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int flags = 0;
        if (bx + by + bz > 8 * PCP_FEED_MAX_PASSES) flags |= 1;
        for (int a = 0; a < 3; ++a)
            if (F->mm[a] <= -(1 << 20) || F->mm[3 + a] >= (1 << 20)) flags |= 4;     // the host statement's own limit
        F->bx = bx; F->by = by; F->bz = bz;
        F->flags = flags;
        F->sort.n = flags ? 0u : total;
        F->sort.passes = flags ? 0 : (bx + by + bz + 7) / 8;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool keep = i < n && flag[i] != 0;
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = (unsigned)__popcll(bal);
    __syncthreads();
    if (!keep) return;
    unsigned slot = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    for (int j = 0; j < w; ++j) slot += s_w[j];
    const int3 c = voxel_coord(pts[i], voxel_size);
    keys[slot] = ((unsigned long long)(unsigned)(c.x - mx) << (by + bz)) | ((unsigned long long)(unsigned)(c.y - my) << bz) |
                 (unsigned long long)(unsigned)(c.z - mz);
    vals[slot] = (unsigned)(begin_index + i);
}

__device__ __forceinline__ int3 feed_key_coord(const unsigned long long key, const VoxFeed* F) {
    const int by = F->by, bz = F->bz, bx = F->bx;
    const unsigned long long kx = (bx ? (key >> (by + bz)) & ((1ull << bx) - 1ull) : 0ull);
    const unsigned long long ky = (by ? (key >> bz) & ((1ull << by) - 1ull) : 0ull);
    const unsigned long long kz = (bz ? key & ((1ull << bz) - 1ull) : 0ull);
    return make_int3((int)kx + F->mm[0], (int)ky + F->mm[1], (int)kz + F->mm[2]);
}

// Group heads of the sorted keys: count per 256-key tile, then their positions in order.
__global__ void __launch_bounds__(256)
k_feed_heads_count(const VoxFeed* __restrict__ F, const unsigned long long* k0, const unsigned long long* k1,
                   unsigned* __restrict__ tile_counts) {
    __shared__ unsigned s_w[4];
    const unsigned U = F->sort.n;
    if ((unsigned)blockIdx.x * 256u >= U) return;
    const unsigned long long* keys = (F->sort.passes & 1) ? k1 : k0;
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    const bool head = j < U && (j == 0 || keys[j] != keys[j - 1]);
    const unsigned long long bal = __ballot(head);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (unsigned)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ void __launch_bounds__(256)
k_feed_heads_emit(VoxFeed* __restrict__ F, const unsigned long long* k0, const unsigned long long* k1,
                  const unsigned* __restrict__ tile_counts, unsigned* __restrict__ heads) {
    __shared__ unsigned s_w[4];
    const unsigned U = F->sort.n;
    if ((unsigned)blockIdx.x * 256u >= U) return;             // block 0 is here whenever U > 0
    const int tiles = (int)((U + 255u) / 256u);
    unsigned before, total;
    tile_prefix(tile_counts, (int)blockIdx.x, tiles, before, total);
    if (blockIdx.x == 0 && threadIdx.x == 0) F->n_new = (int)total;
    const unsigned long long* keys = (F->sort.passes & 1) ? k1 : k0;
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    const bool head = j < U && (j == 0 || keys[j] != keys[j - 1]);
    const unsigned long long bal = __ballot(head);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = (unsigned)__popcll(bal);
    __syncthreads();
    if (!head) return;
    unsigned slot = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    for (int k = 0; k < w; ++k) slot += s_w[k];
    heads[slot] = j;
}

// k_new_voxels with every size read from the control block; waves stride over the new voxels.
__global__ void __launch_bounds__(256)
k_feed_new_voxels(VoxFeed* __restrict__ F, const unsigned long long* k0, const unsigned long long* k1, const unsigned* v0,
                  const unsigned* v1, const unsigned* __restrict__ heads, const int4* __restrict__ table,
                  int4* __restrict__ table_out, int table_rows, int* __restrict__ data, int data_rows, int max_points) {
    const int n_new = F->n_new, stored = F->stored, U = (int)F->sort.n;
    if (n_new == 0) return;
    const bool odd = F->sort.passes & 1;
    const unsigned long long* keys = odd ? k1 : k0;
    const unsigned* vals = odd ? v1 : v0;
    const int lane = threadIdx.x & 63;
    const int waves = (gridDim.x * blockDim.x) >> 6;
    if (stored + n_new > table_rows || stored + n_new > data_rows) {          // capacity is reserved ahead: cannot happen
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&F->flags, 8);
        return;
    }
    for (int k = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; k < n_new; k += waves) {
        const unsigned h = heads[k];
        const unsigned end = (k + 1 < n_new) ? heads[k + 1] : (unsigned)U;
        const int count = (int)(end - h);
        const int row = stored + k;
        int* out = data + (size_t)row * max_points;
        if (lane == 0) out[0] = count;
        const int keep = min(count, max_points - 1);
        for (int j = lane; j < keep; j += 64) out[1 + j] = (int)vals[h + j];
        for (int j = keep + lane; j < max_points - 1; j += 64) out[1 + j] = 0;
        if (lane == 0) {
            const int3 c = feed_key_coord(keys[h], F);
            int lo = 0, hi = stored;                       // first old entry not below c
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int cmp = lex_cmp(c.x, c.y, c.z, table[mid]);
                if (cmp == 0) atomicOr(&F->flags, 2);
                if (cmp > 0) lo = mid + 1; else hi = mid;
            }
            table_out[k + lo] = make_int4(c.x, c.y, c.z, row);
        }
    }
}

__global__ void __launch_bounds__(256)
k_feed_shift_old(const VoxFeed* __restrict__ F, const int4* __restrict__ table, const unsigned long long* k0,
                 const unsigned long long* k1, const unsigned* __restrict__ heads, int4* __restrict__ table_out, int table_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int stored = F->stored;
    if (i >= stored) return;
    const int n_new = (stored + F->n_new > table_rows) ? 0 : F->n_new;
    const unsigned long long* keys = (F->sort.passes & 1) ? k1 : k0;
    const int4 e = table[i];
    int lo = 0, hi = n_new;                            // first new voxel not below e
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int3 c = feed_key_coord(keys[heads[mid]], F);
        if (lex_cmp(c.x, c.y, c.z, e) < 0) lo = mid + 1; else hi = mid;
    }
    table_out[i + lo] = e;
}

// The frame's record; the control block is made ready for the next frame.
__global__ void k_feed_finish(VoxFeed* __restrict__ F, int* __restrict__ counter, int serial, int table_rows,
                              pcp_voxel_frame* __restrict__ rec) {
    pcp_voxel_frame r;
    r.serial = serial;
    r.stored_before = F->stored;
    const int created = (F->flags & (1 | 4 | 8)) ? 0 : F->n_new;
    r.created = created;
    r.stored = F->stored + created;
    r.unknown = (int)F->sort.n;
    r.staged = counter[0];
    r.key_bits = F->bx + F->by + F->bz;
    r.flags = F->flags;
    *rec = r;
    F->stored = r.stored;
    F->frame = serial + 1;
    feed_reset(F);
    counter[0] = 0;
}

int feed_take(pcp_ctx* c, int serial, pcp_voxel_frame* out);

// Every entry point starts here.  The step-by-step ones (drain = true) first take the records of frames still in flight
// from pcp_voxel_add*, so that the host's notion of the table is the device's.
int use(pcp_ctx* c, bool drain = true) {
    if (!c) return pcp_fail("null context");
    PCP_CHECK(hipSetDevice(c->device));
    if (!c->vox) {
        c->vox = new pcp_voxel_state();
        PCP_CHECK(hipMalloc(&c->vox->counter, 16 * sizeof(int)));
    }
    if (drain && c->vox->feed_waited < c->vox->feed_serial) return feed_take(c, c->vox->feed_serial - 1, nullptr);
    return 0;
}

template <typename T>
int grow(T** p, int* cap, size_t want_elems, bool zero, hipStream_t stream) {
    if ((size_t)*cap >= want_elems && *p) return 0;
    if (*p) PCP_CHECK(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    PCP_CHECK(hipMalloc(p, want_elems * sizeof(T)));
    if (zero) PCP_CHECK(hipMemsetAsync(*p, 0, want_elems * sizeof(T), stream));
    *cap = (int)want_elems;
    return 0;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

extern "C" {

int pcp_voxel_index_upload(pcp_ctx* c, const int32_t* voxel_index, int rows) {
    if (int e = use(c)) return e;
    if (!voxel_index || rows < 1) return pcp_fail("empty voxel index");
    pcp_voxel_state* v = c->vox;
    if (rows > v->table_rows) {
        PCP_CHECK(hipStreamSynchronize(c->stream));
        if (v->table) PCP_CHECK(hipFree(v->table));
        v->table = nullptr;
        v->table_rows = 0;
        PCP_CHECK(hipMalloc(&v->table, (size_t)rows * sizeof(int4)));
        v->table_rows = rows;
    }
    PCP_CHECK(hipMemcpyAsync(v->table, voxel_index, (size_t)rows * sizeof(int4), hipMemcpyHostToDevice, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    v->table_valid = rows;
    v->feed_known_stored = -1;             // the device's own voxel count (frame feed) is void
    return 0;
}

int pcp_voxel_data_upload(pcp_ctx* c, const int32_t* voxel_data, int rows, int max_points) {
    if (int e = use(c)) return e;
    if (!voxel_data || rows < 1 || max_points < 2) return pcp_fail("empty voxel data");
    pcp_voxel_state* v = c->vox;
    if (rows != v->data_rows || max_points != v->data_width) {
        PCP_CHECK(hipStreamSynchronize(c->stream));
        if (v->data) PCP_CHECK(hipFree(v->data));
        v->data = nullptr;
        v->data_rows = v->data_width = 0;
        PCP_CHECK(hipMalloc(&v->data, (size_t)rows * max_points * sizeof(int)));
        v->data_rows = rows;
        v->data_width = max_points;
    }
    PCP_CHECK(hipMemcpyAsync(v->data, voxel_data, (size_t)rows * max_points * sizeof(int), hipMemcpyHostToDevice, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int pcp_voxel_data_upload_rows(pcp_ctx* c, const int32_t* rows_data, int first_row, int n_rows, int max_points) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (n_rows == 0) return 0;
    if (!rows_data || first_row < 0 || n_rows < 0) return pcp_fail("bad row range");
    if (!v->data || max_points != v->data_width || first_row + n_rows > v->data_rows)
        return pcp_fail("rows [%d, %d) x %d do not fit the resident voxel data (%d x %d): upload it whole first",
                        first_row, first_row + n_rows, max_points, v->data_rows, v->data_width);
    PCP_CHECK(hipMemcpyAsync(v->data + (size_t)first_row * max_points, rows_data,
                             (size_t)n_rows * max_points * sizeof(int), hipMemcpyHostToDevice, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int pcp_voxel_data_read(pcp_ctx* c, int rows, int max_points, int32_t* out) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (rows == 0) return 0;
    if (!out || rows < 0) return pcp_fail("bad arguments");
    if (!v->data || max_points != v->data_width || rows > v->data_rows)
        return pcp_fail("%d x %d rows requested, %d x %d resident", rows, max_points, v->data_rows, v->data_width);
    PCP_CHECK(hipMemcpyAsync(out, v->data, (size_t)rows * max_points * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

static int voxelize_impl(pcp_ctx* c, const float* pts, int floats_per_point, int n, float voxel_size, int stored,
                         int begin_index, int max_points, int32_t* unknown, int unknown_len, float* debug);

int pcp_voxelize(pcp_ctx* c, const float* pts_xyzw, int n, float voxel_size, int stored, int begin_index,
                 int max_points, int32_t* unknown, int unknown_len, float* debug) {
    return voxelize_impl(c, pts_xyzw, 4, n, voxel_size, stored, begin_index, max_points, unknown, unknown_len, debug);
}

int pcp_voxelize_xyz(pcp_ctx* c, const float* pts_xyz, int n, float voxel_size, int stored, int begin_index,
                     int max_points, int32_t* unknown, int unknown_len, float* debug) {
    return voxelize_impl(c, pts_xyz, 3, n, voxel_size, stored, begin_index, max_points, unknown, unknown_len, debug);
}

int pcp_voxelize_rows(pcp_ctx* c, const float* xyz_rows, int n, float voxel_size, int stored, int begin_index,
                      int max_points, int32_t* unknown, int unknown_len, float* debug) {
    return voxelize_impl(c, xyz_rows, -3, n, voxel_size, stored, begin_index, max_points, unknown, unknown_len, debug);
}

static int voxelize_impl(pcp_ctx* c, const float* pts_xyzw, int floats_per_point, int n, float voxel_size, int stored,
                         int begin_index, int max_points, int32_t* unknown, int unknown_len, float* debug) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (!pts_xyzw || !unknown) return pcp_fail("null argument");
    if (n < 1) return pcp_fail("no points to voxelize");
    if (unknown_len < n + 1) return pcp_fail("unknown list of %d ints cannot hold 1 + %d entries", unknown_len, n);
    if (!(voxel_size > 0.0f)) return pcp_fail("voxel_size must be positive");
    if (stored < 0 || stored > v->table_valid) return pcp_fail("stored = %d, resident table has %d rows", stored, v->table_valid);
    if (stored > 0 && (!v->data || max_points != v->data_width))
        return pcp_fail("voxel data not resident with row width %d", max_points);
    if (int e = grow(&v->unknown, &v->unknown_cap, (size_t)n + 1, false, c->stream)) return e;
    if (n > v->pts_cap) {
        int cap = v->pts_cap, cap2 = v->pts_cap;
        if (int e = grow(&v->pts, &cap, (size_t)n, false, c->stream)) return e;
        if (int e = grow(&v->debug, &cap2, (size_t)n, false, c->stream)) return e;
        v->pts_cap = cap;
    }
    PCP_CHECK(hipMemsetAsync(v->unknown, 0, sizeof(int), c->stream));
    if (floats_per_point == 4) {
        PCP_CHECK(hipMemcpyAsync(v->pts, pts_xyzw, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    } else {
        // the 12-byte records land in the debug buffer (n float4, overwritten by the kernel only after this
        // expansion has read them: one stream) and are widened on the device
        float* staging = reinterpret_cast<float*>(v->debug);
        PCP_CHECK(hipMemcpyAsync(staging, pts_xyzw, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        if (floats_per_point == 3) hipLaunchKernelGGL(k_pad_xyz, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, staging, n, v->pts);
        else hipLaunchKernelGGL(k_pad_rows, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, staging, n, v->pts);
        PCP_CHECK(hipGetLastError());
    }
    if (v->keep_store) {
        if (begin_index < 0) return pcp_fail("begin_index must not be negative");
        const size_t end = (size_t)begin_index + (size_t)n;
        if (end > v->store_cap) {                 // doubling, like the host store (Voxelizer.py:402-416)
            size_t cap = v->store_cap ? v->store_cap : (size_t)1 << 20;
            while (cap < end) cap *= 2;
            float4* bigger = nullptr;
            PCP_CHECK(hipMalloc(&bigger, cap * sizeof(float4)));
            hipError_t e = hipMemsetAsync(bigger, 0, cap * sizeof(float4), c->stream);
            if (e == hipSuccess && v->store_n)
                e = hipMemcpyAsync(bigger, v->store, v->store_n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { (void)hipFree(bigger); return pcp_fail("growing the point store: %s", hipGetErrorString(e)); }
            if (v->store) (void)hipFree(v->store);
            v->store = bigger;
            v->store_cap = cap;
        }
        PCP_CHECK(hipMemcpyAsync(v->store + begin_index, v->pts, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
        if (end > v->store_n) v->store_n = end;
    }
    hipLaunchKernelGGL(k_voxelize, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, v->pts, n, v->table, stored, v->data,
                       v->data_rows, max_points, begin_index, voxel_size, v->unknown, debug ? v->debug : nullptr);
    PCP_CHECK(hipGetLastError());
    int count = 0;
    PCP_CHECK(hipMemcpyAsync(&count, v->unknown, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    if (count < 0 || count > n) return pcp_fail("corrupt unknown count %d", count);
    v->last_n = n; v->last_begin = begin_index; v->last_unknown = count; v->last_stored = stored; v->last_size = voxel_size;
    for (int i = 0; i < unknown_len; ++i) unknown[i] = 0;       // the reference uploads a zeroed list (:855, :881-883)
    PCP_CHECK(hipMemcpyAsync(unknown, v->unknown, (size_t)(1 + count) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (debug) PCP_CHECK(hipMemcpyAsync(debug, v->debug, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int pcp_voxel_reset(pcp_ctx* c) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    PCP_CHECK(hipStreamSynchronize(c->stream));
    // The map is forgotten, its buffers are kept for the next one (a map of 100 k voxels holds 400 MB of rows: freeing
    // and allocating them again per map costs more than a frame).  Rows the next map has not written hold old ids: they
    // belong to no voxel until the creation of a voxel rewrites its row whole, and pcp_voxel_reserve clears what it hands out.
    v->store_n = 0; v->last_staged = 0;
    v->stat_valid = 0;
    v->table_valid = 0;
    v->data_dirty = v->data != nullptr;
    v->last_unknown = -1;
    v->last_n = 0;
    v->feed_known_stored = -1;
    v->feed_pending_points = 0;
    return 0;
}

int pcp_voxel_reserve(pcp_ctx* c, int rows, int max_points) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (rows < 1 || max_points < 2) return pcp_fail("bad sizes");
    PCP_CHECK(hipStreamSynchronize(c->stream));
    if (v->data && max_points != v->data_width) {
        if (v->table_valid > 0) return pcp_fail("resident voxel data has row width %d", v->data_width);
        PCP_CHECK(hipFree(v->data));              // buffers kept from a map of another row width that has been reset
        v->data = nullptr; v->data_rows = v->data_width = 0; v->data_dirty = false;
    }
    if (rows > v->table_rows) {
        int4* t = nullptr;
        PCP_CHECK(hipMalloc(&t, (size_t)rows * sizeof(int4)));
        if (v->table && v->table_valid)
            PCP_CHECK(hipMemcpy(t, v->table, (size_t)v->table_valid * sizeof(int4), hipMemcpyDeviceToDevice));
        if (v->table) PCP_CHECK(hipFree(v->table));
        v->table = t;
        v->table_rows = rows;
    }
    if (v->data_dirty && v->data) {            // buffers kept from a map that has been reset: the rows handed out start empty
        PCP_CHECK(hipMemsetAsync(v->data, 0, (size_t)std::min(rows, v->data_rows) * max_points * sizeof(int), c->stream));
        v->data_dirty = false;
    }
    if (rows > v->data_rows) {
        int* d = nullptr;
        PCP_CHECK(hipMalloc(&d, (size_t)rows * max_points * sizeof(int)));
        if (v->data && v->data_rows)
            PCP_CHECK(hipMemcpy(d, v->data, (size_t)v->data_rows * max_points * sizeof(int), hipMemcpyDeviceToDevice));
        PCP_CHECK(hipMemset(d + (size_t)v->data_rows * max_points, 0,
                            (size_t)(rows - v->data_rows) * max_points * sizeof(int)));
        if (v->data) PCP_CHECK(hipFree(v->data));
        v->data = d;
        v->data_rows = rows;
        v->data_width = max_points;
    }
    return 0;
}

int pcp_voxel_insert_unknown(pcp_ctx* c, int max_points, int* new_stored, int* rows_capacity) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (!new_stored) return pcp_fail("new_stored is null");
    if (v->last_unknown < 0) return pcp_fail("no dispatch to take unknown points from (pcp_voxelize first)");
    const int n = v->last_n, U = v->last_unknown, stored = v->last_stored, begin = v->last_begin;
    v->last_unknown = -1;                                // one insertion per dispatch
    *new_stored = stored;
    if (rows_capacity) *rows_capacity = v->data_rows;
    if (U == 0) return 0;
    if (stored > 0 && max_points != v->data_width) return pcp_fail("resident voxel data has row width %d", v->data_width);
    hipStream_t s = c->stream;
    // scratch
    if (n > v->ins_cap) {
        PCP_CHECK(hipStreamSynchronize(s));
        for (void* p : {(void*)v->flag, (void*)v->k0, (void*)v->k1, (void*)v->v0, (void*)v->v1, (void*)v->heads,
                        (void*)v->hist, (void*)v->counts})
            if (p) PCP_CHECK(hipFree(p));
        v->flag = nullptr; v->k0 = v->k1 = nullptr; v->v0 = v->v1 = v->heads = v->hist = v->counts = nullptr;
        v->ins_cap = 0;
        const size_t cap = (size_t)n;
        PCP_CHECK(hipMalloc(&v->flag, cap));
        PCP_CHECK(hipMalloc(&v->k0, cap * 8));
        PCP_CHECK(hipMalloc(&v->k1, cap * 8));
        PCP_CHECK(hipMalloc(&v->v0, cap * 4));
        PCP_CHECK(hipMalloc(&v->v1, cap * 4));
        PCP_CHECK(hipMalloc(&v->heads, cap * 4));
        PCP_CHECK(hipMalloc(&v->hist, pcp::radix_hist_elems(cap) * 4));
        PCP_CHECK(hipMalloc(&v->counts, pcp::compact_count_elems(cap) * 4));
        v->ins_cap = n;
    }
    if (!v->totals) {
        PCP_CHECK(hipMalloc(&v->totals, 256 * 4));
        PCP_CHECK(hipMalloc(&v->base, 256 * 4));
        PCP_CHECK(hipMalloc(&v->mm, 8 * 4));
    }
    // which points, and the range of their voxel coordinates
    PCP_CHECK(hipMemsetAsync(v->flag, 0, (size_t)n, s));
    hipLaunchKernelGGL(k_mark_unknown, dim3(cdiv(U, 256)), dim3(256), 0, s, v->unknown, U, begin, v->flag);
    const int init[8] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN, 0, 0};
    PCP_CHECK(hipMemcpyAsync(v->mm, init, sizeof(init), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_unknown_range, dim3(std::min(cdiv(n, 256), 1024)), dim3(256), 0, s, v->pts, v->flag, n,
                       v->last_size, v->mm);
    int mm[8];
    PCP_CHECK(hipMemcpyAsync(mm, v->mm, sizeof(mm), hipMemcpyDeviceToHost, s));
    PCP_CHECK(hipStreamSynchronize(s));
    int bits = 1;
    for (int a = 0; a < 3; ++a) {
        // 21 bits per axis at most: three of them fill the 63 key bits (|c| < 2^20, the host statement's own limit)
        if (mm[a] <= -(1 << 20) || mm[3 + a] >= (1 << 20)) return pcp_fail("voxel coordinate beyond +-2^20");
        while ((1ll << bits) < (long long)mm[3 + a] - mm[a] + 1) ++bits;
    }
    // (key, id) in point order, stable sort by key, group heads
    const EmitKey emit{v->pts, v->last_size, mm[0], mm[1], mm[2], bits, begin, v->k0, v->v0};
    pcp::compact_ordered(s, IsUnknown{v->flag}, emit, (size_t)n, v->counts);
    unsigned long long* keys = nullptr;
    unsigned* vals = nullptr;
    pcp::radix_sort_pairs(s, v->k0, v->k1, v->v0, v->v1, (size_t)U, 3 * bits, v->hist, v->totals, v->base, &keys, &vals);
    pcp::compact_ordered(s, IsHead{keys}, EmitHead{v->heads}, (size_t)U, v->counts);
    PCP_CHECK(hipGetLastError());
    unsigned n_new = 0;
    PCP_CHECK(hipMemcpyAsync(&n_new, v->counts + pcp::compact_tiles((size_t)U), 4, hipMemcpyDeviceToHost, s));
    PCP_CHECK(hipStreamSynchronize(s));
    if (n_new < 1 || n_new > (unsigned)U) return pcp_fail("corrupt voxel group count %u", n_new);
    const int total = stored + (int)n_new;
    // capacity: the host rule (double while new_stored >= rows, Voxelizer.py:238-252)
    int rows = std::max(v->data_rows, 4096);
    while (total >= rows) rows *= 2;
    if (int e = pcp_voxel_reserve(c, rows, max_points)) return e;
    if (rows > v->table2_rows) {
        PCP_CHECK(hipStreamSynchronize(s));
        if (v->table2) PCP_CHECK(hipFree(v->table2));
        v->table2 = nullptr;
        PCP_CHECK(hipMalloc(&v->table2, (size_t)rows * sizeof(int4)));
        v->table2_rows = rows;
    }
    PCP_CHECK(hipMemsetAsync(v->mm + 7, 0, 4, s));
    hipLaunchKernelGGL(k_new_voxels, dim3(cdiv((int)n_new * 64, 256)), dim3(256), 0, s, keys, vals, v->heads, (int)n_new, U,
                       bits, mm[0], mm[1], mm[2], v->table, stored, v->table2, v->data, max_points, v->mm + 7);
    if (stored > 0)
        hipLaunchKernelGGL(k_shift_old, dim3(cdiv(stored, 256)), dim3(256), 0, s, v->table, stored, keys, v->heads,
                           (int)n_new, bits, mm[0], mm[1], mm[2], v->table2);
    PCP_CHECK(hipGetLastError());
    int clash = 0;
    PCP_CHECK(hipMemcpyAsync(&clash, v->mm + 7, 4, hipMemcpyDeviceToHost, s));
    PCP_CHECK(hipStreamSynchronize(s));
    if (clash) return pcp_fail("an unknown point's voxel is already in the table");      // Voxelizer.py:228-229
    std::swap(v->table, v->table2);
    std::swap(v->table_rows, v->table2_rows);
    v->table_valid = total;
    v->feed_known_stored = -1;
    *new_stored = total;
    if (rows_capacity) *rows_capacity = v->data_rows;
    return 0;
}

int pcp_voxel_index_read(pcp_ctx* c, int rows, int32_t* voxel_index) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (rows == 0) return 0;
    if (!voxel_index || rows < 0 || rows > v->table_valid) return pcp_fail("%d rows requested, %d resident", rows, v->table_valid);
    PCP_CHECK(hipMemcpyAsync(voxel_index, v->table, (size_t)rows * sizeof(int4), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int pcp_voxel_counts_read(pcp_ctx* c, int first_row, int n_rows, int32_t* counts) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (n_rows == 0) return 0;
    if (!counts || first_row < 0 || n_rows < 0 || first_row + n_rows > v->data_rows) return pcp_fail("bad row range");
    int* tmp = nullptr;
    PCP_CHECK(hipMalloc(&tmp, (size_t)n_rows * 4));
    hipLaunchKernelGGL(k_gather_counts, dim3(cdiv(n_rows, 256)), dim3(256), 0, c->stream, v->data, first_row, n_rows,
                       v->data_width, tmp);
    hipError_t e1 = hipMemcpyAsync(counts, tmp, (size_t)n_rows * 4, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    PCP_CHECK(e1);
    PCP_CHECK(e2);
    return 0;
}

// voxel_stat == nullptr: the statistics stay on the device between calls (resident mode).
static int stage_impl(pcp_ctx* c, float* voxel_stat, int stat_rows, int stored, int max_points, int filter_outliers,
                      int stage_everything, int max_staging, int32_t* counter16, int32_t* stage_out) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (!counter16) return pcp_fail("null argument");
    if (stat_rows < 1 || max_staging < 1 || max_points < 2) return pcp_fail("bad sizes");
    if (stored < 0 || stored > v->table_valid) return pcp_fail("stored = %d, resident table has %d rows", stored, v->table_valid);
    if (stored > 0 && (!v->data || max_points != v->data_width))
        return pcp_fail("voxel data not resident with row width %d", max_points);
    if (stat_rows > v->stat_rows) {
        // the resident statistics survive the growth
        float4 *bigger = nullptr, *bigger_in = nullptr;
        PCP_CHECK(hipMalloc(&bigger, (size_t)stat_rows * sizeof(float4)));
        hipError_t e = hipMalloc(&bigger_in, (size_t)stat_rows * sizeof(float4));
        if (e == hipSuccess && v->stat_valid > 0)
            e = hipMemcpyAsync(bigger, v->stat, (size_t)v->stat_valid * sizeof(float4), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            (void)hipFree(bigger);
            if (bigger_in) (void)hipFree(bigger_in);
            return pcp_fail("growing the voxel statistics: %s", hipGetErrorString(e));
        }
        if (v->stat) (void)hipFree(v->stat);
        if (v->stat_in) (void)hipFree(v->stat_in);
        v->stat = bigger; v->stat_in = bigger_in;
        v->stat_rows = stat_rows;
    }
    if (max_staging != v->stage_rows || max_points != v->stage_width) {   // allocated once, zeroed (:765-770)
        int cap = 0;
        PCP_CHECK(hipStreamSynchronize(c->stream));
        if (v->stage) PCP_CHECK(hipFree(v->stage));
        v->stage = nullptr;
        if (int e = grow(&v->stage, &cap, (size_t)max_staging * max_points, true, c->stream)) return e;
        v->stage_rows = max_staging;
        v->stage_width = max_points;
    }
    int host_counter[16] = {0, max_staging, max_points, stored, stage_everything ? 1 : 0, filter_outliers ? 0 : 1};
    PCP_CHECK(hipMemcpyAsync(v->counter, host_counter, sizeof(host_counter), hipMemcpyHostToDevice, c->stream));
    if (voxel_stat) {
        PCP_CHECK(hipMemcpyAsync(v->stat, voxel_stat, (size_t)stat_rows * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        v->stat_valid = stat_rows;
    } else if (v->stat_valid < stat_rows) {
        hipLaunchKernelGGL(k_stat_init, dim3(cdiv(stat_rows - v->stat_valid, 256)), dim3(256), 0, c->stream, v->stat, v->stat_valid, stat_rows);
        PCP_CHECK(hipGetLastError());
        v->stat_valid = stat_rows;
    }
    PCP_CHECK(hipMemcpyAsync(v->stat_in, v->stat, (size_t)stat_rows * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    if (stored > 0) {
        hipLaunchKernelGGL(k_voxel_stage, dim3(cdiv(stored, 256)), dim3(256), 0, c->stream, v->table, stored, v->data,
                           v->data_rows, max_points, v->stat, v->stat_in, v->stat_rows, v->stage, max_staging,
                           stage_everything ? 1 : 0, filter_outliers ? 0 : 1, v->counter, (const VoxFeed*)nullptr);
        PCP_CHECK(hipGetLastError());
    }
    PCP_CHECK(hipMemcpyAsync(host_counter, v->counter, sizeof(host_counter), hipMemcpyDeviceToHost, c->stream));
    if (voxel_stat) PCP_CHECK(hipMemcpyAsync(voxel_stat, v->stat, (size_t)stat_rows * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 16; ++i) counter16[i] = host_counter[i];
    const int staged = host_counter[0];
    v->last_staged = 0;
    if (staged < 0 || staged > max_staging) return pcp_fail("corrupt staging counter %d", staged);
    v->last_staged = staged;
    if (stage_out && staged > 0) {
        PCP_CHECK(hipMemcpyAsync(stage_out, v->stage, (size_t)staged * max_points * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        PCP_CHECK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int pcp_voxel_stage(pcp_ctx* c, float* voxel_stat, int stat_rows, int stored, int max_points, int filter_outliers,
                    int stage_everything, int max_staging, int32_t* counter16, int32_t* stage_out) {
    if (!voxel_stat) return pcp_fail("null argument");
    return stage_impl(c, voxel_stat, stat_rows, stored, max_points, filter_outliers, stage_everything, max_staging, counter16, stage_out);
}

int pcp_voxel_stage_resident(pcp_ctx* c, int stat_rows, int stored, int max_points, int filter_outliers, int stage_everything,
                             int max_staging, int32_t* counter16, int32_t* stage_out) {
    return stage_impl(c, nullptr, stat_rows, stored, max_points, filter_outliers, stage_everything, max_staging, counter16, stage_out);
}

int pcp_voxel_stat_read(pcp_ctx* c, int rows, float* voxel_stat) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (!voxel_stat || rows < 0) return pcp_fail("bad argument");
    // rows the device has not seen yet read as the statistics of a voxel that does not exist: (0.5, 0, 0, 0)
    const int have = rows < v->stat_valid ? rows : v->stat_valid;
    if (have > 0) PCP_CHECK(hipMemcpyAsync(voxel_stat, v->stat, (size_t)have * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    for (int i = have; i < rows; ++i) { voxel_stat[4 * i] = 0.5f; voxel_stat[4 * i + 1] = voxel_stat[4 * i + 2] = voxel_stat[4 * i + 3] = 0.0f; }
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int pcp_voxel_store_enable(pcp_ctx* c, int on) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (on && !v->keep_store && v->last_n > 0 && v->store_n == 0)
        return pcp_fail("the point store must be switched on before the first dispatch of a map (ids count from there)");
    v->keep_store = on != 0;
    if (!on) {
        PCP_CHECK(hipStreamSynchronize(c->stream));
        if (v->store) PCP_CHECK(hipFree(v->store));
        v->store = nullptr; v->store_cap = v->store_n = 0;
    }
    return 0;
}

int pcp_voxel_staged_points(pcp_ctx* c, const double** xyz_dev, size_t* n_points, int* n_rows) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (!xyz_dev || !n_points) return pcp_fail("null output");
    *xyz_dev = nullptr; *n_points = 0;
    if (n_rows) *n_rows = v->last_staged;
    if (!v->keep_store) return pcp_fail("the device point store is off (pcp_voxel_store_enable)");
    const int rows = v->last_staged;
    if (rows == 0) return 0;
    if (!v->stage || rows > v->stage_rows) return pcp_fail("no staging buffer for %d rows", rows);
    if (!v->staged_off || v->staged_off_rows < v->stage_rows) {      // follows the staging buffer (max_staging may change)
        if (v->staged_off) PCP_CHECK(hipFree(v->staged_off));
        v->staged_off = nullptr; v->staged_off_rows = 0;
        PCP_CHECK(hipMalloc(&v->staged_off, ((size_t)v->stage_rows + 2) * sizeof(unsigned)));
        v->staged_off_rows = v->stage_rows;
    }
    hipLaunchKernelGGL(k_staged_offsets, dim3(1), dim3(1024), 0, c->stream, v->stage, rows, v->stage_width, v->staged_off);
    PCP_CHECK(hipGetLastError());
    unsigned total = 0;
    PCP_CHECK(hipMemcpyAsync(&total, v->staged_off + rows, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    if (total == 0) return 0;
    if ((size_t)total > v->staged_cap) {
        if (v->staged_xyz) PCP_CHECK(hipFree(v->staged_xyz));
        v->staged_xyz = nullptr; v->staged_cap = 0;
        size_t cap = (size_t)1 << 16;
        while (cap < total) cap *= 2;
        PCP_CHECK(hipMalloc(&v->staged_xyz, cap * 3 * sizeof(double)));
        v->staged_cap = cap;
    }
    hipLaunchKernelGGL(k_staged_gather, dim3(rows), dim3(256), 0, c->stream, v->stage, v->stage_width, v->store, v->store_n,
                       v->staged_off, rows, v->staged_xyz);
    PCP_CHECK(hipGetLastError());
    unsigned bad = 0;
    PCP_CHECK(hipMemcpyAsync(&bad, v->staged_off + rows + 1, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    if (bad) return pcp_fail("a staged row names a point the device store does not hold (%zu points stored)", v->store_n);
    *xyz_dev = v->staged_xyz;
    *n_points = total;
    return 0;
}

}  // extern "C"

// ---- frame feed ---------------------------------------------------------------------------------------

namespace {

// Device memory for one more frame of n points on a map of at most `stored_ub` voxels, reserved ahead: table and data rows
// for stored_ub + n voxels (every point could found a voxel), statistics initialised for as many, the creation's scratch.
// Only grows (doubling); a growth waits for the stream and copies what is in use.
int feed_reserve(pcp_ctx* c, int stored_ub, int n, int max_points, int max_staging, size_t store_end) {
    pcp_voxel_state* v = c->vox;
    hipStream_t s = c->stream;
    const long long need = (long long)stored_ub + n;
    if (need > (1ll << 30) / max_points * 2) return pcp_fail("the voxel map cannot hold %lld voxels", need);
    if (v->data && max_points != v->data_width) return pcp_fail("resident voxel data has row width %d", v->data_width);
    if (need > v->table_rows || need > v->data_rows) {
        long long rows = std::max(std::max(v->table_rows, v->data_rows), 4096);
        while (rows < need) rows *= 2;
        PCP_CHECK(hipStreamSynchronize(s));
        if (rows > v->table_rows) {
            int4* t = nullptr;
            PCP_CHECK(hipMalloc(&t, (size_t)rows * sizeof(int4)));
            if (v->table && stored_ub > 0)
                PCP_CHECK(hipMemcpy(t, v->table, (size_t)std::min(stored_ub, v->table_rows) * sizeof(int4), hipMemcpyDeviceToDevice));
            if (v->table) PCP_CHECK(hipFree(v->table));
            v->table = t;
            v->table_rows = (int)rows;
        }
        if (rows > v->data_rows) {
            // rows of voxels that do not exist yet need no contents: the creation writes a new voxel's row whole
            int* d = nullptr;
            PCP_CHECK(hipMalloc(&d, (size_t)rows * max_points * sizeof(int)));
            const int used = std::min(stored_ub, v->data_rows);
            if (v->data && used > 0) PCP_CHECK(hipMemcpy(d, v->data, (size_t)used * max_points * sizeof(int), hipMemcpyDeviceToDevice));
            if (v->data) PCP_CHECK(hipFree(v->data));
            v->data = d;
            v->data_rows = (int)rows;
            v->data_width = max_points;
        }
    }
    if (need > v->table2_rows) {
        PCP_CHECK(hipStreamSynchronize(s));
        if (v->table2) PCP_CHECK(hipFree(v->table2));
        v->table2 = nullptr; v->table2_rows = 0;
        PCP_CHECK(hipMalloc(&v->table2, (size_t)v->table_rows * sizeof(int4)));
        v->table2_rows = v->table_rows;
    }
    if (v->table_rows > v->stat_rows) {                       // statistics for every row the table can hold, initialised once
        float4 *bigger = nullptr, *bigger_in = nullptr;
        PCP_CHECK(hipStreamSynchronize(s));
        PCP_CHECK(hipMalloc(&bigger, (size_t)v->table_rows * sizeof(float4)));
        PCP_CHECK(hipMalloc(&bigger_in, (size_t)v->table_rows * sizeof(float4)));
        if (v->stat_valid > 0) PCP_CHECK(hipMemcpy(bigger, v->stat, (size_t)v->stat_valid * sizeof(float4), hipMemcpyDeviceToDevice));
        if (v->stat) (void)hipFree(v->stat);
        if (v->stat_in) (void)hipFree(v->stat_in);
        v->stat = bigger; v->stat_in = bigger_in;
        v->stat_rows = v->table_rows;
    }
    if (v->stat_valid < v->stat_rows) {
        hipLaunchKernelGGL(k_stat_init, dim3(cdiv(v->stat_rows - v->stat_valid, 256)), dim3(256), 0, s, v->stat, v->stat_valid, v->stat_rows);
        PCP_CHECK(hipGetLastError());
        v->stat_valid = v->stat_rows;
    }
    if (max_staging != v->stage_rows || max_points != v->stage_width) {
        int cap = 0;
        PCP_CHECK(hipStreamSynchronize(s));
        if (v->stage) PCP_CHECK(hipFree(v->stage));
        v->stage = nullptr;
        if (int e = grow(&v->stage, &cap, (size_t)max_staging * max_points, true, s)) return e;
        v->stage_rows = max_staging;
        v->stage_width = max_points;
    }
    if (n > v->pts_cap) {
        PCP_CHECK(hipStreamSynchronize(s));
        int cap = v->pts_cap, cap2 = v->pts_cap;
        if (int e = grow(&v->pts, &cap, (size_t)n, false, s)) return e;
        if (int e = grow(&v->debug, &cap2, (size_t)n, false, s)) return e;
        v->pts_cap = cap;
    }
    if (n > v->ins_cap) {
        PCP_CHECK(hipStreamSynchronize(s));
        for (void* p : {(void*)v->flag, (void*)v->k0, (void*)v->k1, (void*)v->v0, (void*)v->v1, (void*)v->heads,
                        (void*)v->hist, (void*)v->counts})
            if (p) PCP_CHECK(hipFree(p));
        v->flag = nullptr; v->k0 = v->k1 = nullptr; v->v0 = v->v1 = v->heads = v->hist = v->counts = nullptr;
        v->ins_cap = 0;
        const size_t cap = (size_t)n;
        PCP_CHECK(hipMalloc(&v->flag, cap));
        PCP_CHECK(hipMalloc(&v->k0, cap * 8));
        PCP_CHECK(hipMalloc(&v->k1, cap * 8));
        PCP_CHECK(hipMalloc(&v->v0, cap * 4));
        PCP_CHECK(hipMalloc(&v->v1, cap * 4));
        PCP_CHECK(hipMalloc(&v->heads, cap * 4));
        PCP_CHECK(hipMalloc(&v->hist, pcp::radix_hist_elems(cap) * 4));
        PCP_CHECK(hipMalloc(&v->counts, pcp::compact_count_elems(cap) * 4));
        v->ins_cap = n;
    }
    if (cdiv(n, 256) + 2 > v->tile_cap) {
        PCP_CHECK(hipStreamSynchronize(s));
        if (v->tile_counts) PCP_CHECK(hipFree(v->tile_counts));
        v->tile_counts = nullptr; v->tile_cap = 0;
        PCP_CHECK(hipMalloc(&v->tile_counts, (size_t)(cdiv(n, 256) + 2) * sizeof(unsigned)));
        v->tile_cap = cdiv(n, 256) + 2;
    }
    if (!v->rs_base) PCP_CHECK(hipMalloc(&v->rs_base, 256 * sizeof(unsigned)));
    if (v->keep_store && store_end > v->store_cap) {          // doubling, like the host store (Voxelizer.py:402-416)
        size_t cap = v->store_cap ? v->store_cap : (size_t)1 << 20;
        while (cap < store_end) cap *= 2;
        float4* bigger = nullptr;
        PCP_CHECK(hipStreamSynchronize(s));
        PCP_CHECK(hipMalloc(&bigger, cap * sizeof(float4)));
        if (v->store_n) PCP_CHECK(hipMemcpy(bigger, v->store, v->store_n * sizeof(float4), hipMemcpyDeviceToDevice));
        if (v->store) (void)hipFree(v->store);
        v->store = bigger;
        v->store_cap = cap;
    }
    return 0;
}

int feed_enqueue(pcp_ctx* c, const void* pts_dev, int layout, int n, float voxel_size, int stored, int begin_index, int max_points,
                 int filter_outliers, int max_staging, int* serial) {
    pcp_voxel_state* v = c->vox;
    hipStream_t s = c->stream;
    if (!v->feed) {
        PCP_CHECK(hipMalloc(&v->feed, sizeof(VoxFeed)));
        PCP_CHECK(hipMalloc(&v->feed_rec_dev, PCP_FEED_RING * sizeof(pcp_voxel_frame)));
        PCP_CHECK(hipHostMalloc((void**)&v->feed_rec, PCP_FEED_RING * sizeof(pcp_voxel_frame)));
        for (int k = 0; k < PCP_FEED_RING; ++k) PCP_CHECK(hipEventCreateWithFlags(&v->feed_done[k], hipEventDisableTiming));
    }
    if (v->feed_serial - v->feed_waited >= PCP_FEED_RING)          // the ring is full: the oldest frame is taken first
        if (int e = feed_take(c, v->feed_waited, nullptr)) return e;
    if (stored >= 0) {
        if (v->feed_serial != v->feed_waited)
            return pcp_fail("stored = %d given while %d fed frames have not been waited for: pass -1", stored, v->feed_serial - v->feed_waited);
        if (stored > v->table_valid) return pcp_fail("stored = %d, resident table has %d rows", stored, v->table_valid);
        // step-by-step calls may have run since the last fed frame: the control block starts afresh
        hipLaunchKernelGGL(k_feed_init, dim3(1), dim3(1), 0, s, v->feed, stored, v->counter);
        PCP_CHECK(hipGetLastError());
        v->feed_known_stored = stored;
        v->feed_pending_points = 0;
    } else if (v->feed_known_stored < 0) {
        return pcp_fail("stored = -1 needs a frame fed before (the device does not hold the voxel count yet)");
    }
    const long long ub = (long long)v->feed_known_stored + v->feed_pending_points;
    if (ub + n > INT_MAX / 2) return pcp_fail("too many voxels");
    const size_t store_end = (size_t)begin_index + (size_t)n;
    if (int e = feed_reserve(c, (int)ub, n, max_points, max_staging, store_end)) return e;
    const int tiles = cdiv(n, 256);
    float4* store = v->keep_store ? v->store + begin_index : nullptr;
    const dim3 grid(tiles), block(256);
#define PCP_FEED_LAUNCH(L) hipLaunchKernelGGL((k_feed_voxelize<L>), grid, block, 0, s, pts_dev, n, v->table, v->feed, v->data, v->data_rows, \
                                              max_points, begin_index, voxel_size, v->pts, store, v->flag, v->tile_counts)
    if (layout == PCP_PTS_F32_XYZW) PCP_FEED_LAUNCH(PCP_PTS_F32_XYZW);
    else if (layout == PCP_PTS_F32_XYZ) PCP_FEED_LAUNCH(PCP_PTS_F32_XYZ);
    else if (layout == PCP_PTS_F32_ROWS) PCP_FEED_LAUNCH(PCP_PTS_F32_ROWS);
    else PCP_FEED_LAUNCH(PCP_PTS_F64_XYZ);
#undef PCP_FEED_LAUNCH
    PCP_CHECK(hipGetLastError());
    if (v->keep_store && store_end > v->store_n) v->store_n = store_end;
    // the stager on the table as the frame found it (per-frame mode: the neighbour snapshot is not read)
    const int stage_blocks = cdiv((int)std::max(ub, 1ll), 256);
    hipLaunchKernelGGL(k_voxel_stage, dim3(stage_blocks), dim3(256), 0, s, v->table, 0, v->data, v->data_rows, max_points, v->stat,
                       v->stat_in, v->stat_rows, v->stage, max_staging, 0, filter_outliers ? 0 : 1, v->counter, (const VoxFeed*)v->feed);
    if (v->log_slot >= 0) {                             // the staged rows' points join the open log (this frame's are in the store)
        const int k = v->log_slot;
        if (max_staging > v->log_off_rows) {
            PCP_CHECK(hipStreamSynchronize(s));
            if (v->log_off) PCP_CHECK(hipFree(v->log_off));
            v->log_off = nullptr; v->log_off_rows = 0;
            PCP_CHECK(hipMalloc(&v->log_off, ((size_t)max_staging + 2) * sizeof(unsigned)));
            v->log_off_rows = max_staging;
        }
        hipLaunchKernelGGL(k_log_offsets, dim3(1), dim3(1024), 0, s, v->stage, v->counter, max_staging, max_points, v->log_off,
                           v->log_ctl + 4 * k, (unsigned)std::min<size_t>(v->log_cap[k], 0xffffffffu));
        hipLaunchKernelGGL(k_log_gather, dim3(max_staging), dim3(256), 0, s, v->stage, max_points, v->store, v->store_n, v->log_off,
                           max_staging, v->log_xyz[k], v->log_ctl + 4 * k);
        PCP_CHECK(hipGetLastError());
    }
    // the new voxels
    hipLaunchKernelGGL(k_feed_emit, grid, block, 0, s, v->pts, v->flag, n, v->tile_counts, tiles, v->feed, begin_index, voxel_size,
                       v->k0, v->v0);
    PCP_CHECK(hipGetLastError());
    pcp::radix_sort_pairs_dev(s, &v->feed->sort, v->k0, v->k1, v->v0, v->v1, (size_t)n, PCP_FEED_MAX_PASSES, v->hist, v->rs_base);
    hipLaunchKernelGGL(k_feed_heads_count, grid, block, 0, s, v->feed, v->k0, v->k1, v->tile_counts);
    hipLaunchKernelGGL(k_feed_heads_emit, grid, block, 0, s, v->feed, v->k0, v->k1, v->tile_counts, v->heads);
    hipLaunchKernelGGL(k_feed_new_voxels, dim3(std::min(cdiv(n, 4), 2048)), dim3(256), 0, s, v->feed, v->k0, v->k1, v->v0, v->v1,
                       v->heads, v->table, v->table2, std::min(v->table_rows, v->table2_rows), v->data, v->data_rows, max_points);
    hipLaunchKernelGGL(k_feed_shift_old, dim3(stage_blocks), dim3(256), 0, s, v->feed, v->table, v->k0, v->k1, v->heads, v->table2,
                       std::min(v->table_rows, v->table2_rows));
    PCP_CHECK(hipGetLastError());
    std::swap(v->table, v->table2);
    std::swap(v->table_rows, v->table2_rows);
    const int slot = v->feed_serial % PCP_FEED_RING;
    hipLaunchKernelGGL(k_feed_finish, dim3(1), dim3(1), 0, s, v->feed, v->counter, v->feed_serial, v->table_rows, v->feed_rec_dev + slot);
    PCP_CHECK(hipGetLastError());
    PCP_CHECK(hipMemcpyAsync(v->feed_rec + slot, v->feed_rec_dev + slot, sizeof(pcp_voxel_frame), hipMemcpyDeviceToHost, s));
    PCP_CHECK(hipEventRecord(v->feed_done[slot], s));
    v->feed_pending_points += n;
    v->last_unknown = -1;                               // the step-by-step insertion has nothing to take from this dispatch
    v->last_n = n; v->last_begin = begin_index; v->last_size = voxel_size;
    if (serial) *serial = v->feed_serial;
    v->feed_serial++;
    return 0;
}

// Takes the records of every frame up to `serial`; fails on the first refused one.
int feed_take(pcp_ctx* c, int serial, pcp_voxel_frame* out) {
    pcp_voxel_state* v = c->vox;
    if (serial < 0 || serial >= v->feed_serial) return pcp_fail("frame %d has not been fed", serial);
    if (serial < v->feed_waited - PCP_FEED_HISTORY) return pcp_fail("the record of frame %d has been overwritten", serial);
    if (v->feed_history.empty()) v->feed_history.resize(PCP_FEED_HISTORY);
    int bad = -1, bad_flags = 0;
    if (serial < v->feed_waited) {           // taken before (by an earlier wait, or by the ring filling up)
        const pcp_voxel_frame r = v->feed_history[serial % PCP_FEED_HISTORY];
        if (r.flags) { bad = r.serial; bad_flags = r.flags; }
    }
    while (v->feed_waited <= serial) {
        const int slot = v->feed_waited % PCP_FEED_RING;
        PCP_CHECK(hipEventSynchronize(v->feed_done[slot]));
        const pcp_voxel_frame r = v->feed_rec[slot];
        v->feed_history[v->feed_waited % PCP_FEED_HISTORY] = r;
        v->feed_known_stored = r.stored;
        v->table_valid = r.stored;
        v->last_staged = r.staged;
        v->feed_waited++;
        if (r.flags && bad < 0) { bad = r.serial; bad_flags = r.flags; }
    }
    // points of frames still in flight may yet found voxels
    v->feed_pending_points = 0;
    for (int k = v->feed_waited; k < v->feed_serial; ++k) v->feed_pending_points += v->ins_cap;   // an upper bound per frame
    if (out) *out = v->feed_history[serial % PCP_FEED_HISTORY];
    if (bad >= 0) {
        if (bad_flags & 4) return pcp_fail("frame %d: voxel coordinate beyond +-2^20", bad);
        if (bad_flags & 1) return pcp_fail("frame %d: its unknown points span more than 2^16 voxels along an axis (use pcp_voxelize + "
                                           "pcp_voxel_insert_unknown for such a frame); no voxel was created for them", bad);
        if (bad_flags & 2) return pcp_fail("frame %d: an unknown point's voxel is already in the table", bad);      // Voxelizer.py:228-229
        return pcp_fail("frame %d: the reserved capacity did not hold its new voxels", bad);
    }
    return 0;
}

}  // namespace

extern "C" {

int pcp_voxel_add_dev(pcp_ctx* c, const void* pts_dev, int layout, int n, float voxel_size, int stored, int begin_index,
                      int max_points, int filter_outliers, int max_staging, void* after_stream, int* serial) {
    if (int e = use(c, false)) return e;
    if (!pts_dev) return pcp_fail("null points");
    if (layout != PCP_PTS_F32_XYZW && layout != PCP_PTS_F32_XYZ && layout != PCP_PTS_F32_ROWS && layout != PCP_PTS_F64_XYZ)
        return pcp_fail("unknown point layout %d", layout);
    if (n < 1) return pcp_fail("no points to voxelize");
    if (!(voxel_size > 0.0f)) return pcp_fail("voxel_size must be positive");
    if (max_points < 2 || max_staging < 1) return pcp_fail("bad sizes");
    if (begin_index < 0) return pcp_fail("begin_index must not be negative");
    pcp_voxel_state* v = c->vox;
    if (after_stream && (hipStream_t)after_stream != c->stream) {
        if (!v->feed_after) PCP_CHECK(hipEventCreateWithFlags(&v->feed_after, hipEventDisableTiming));
        PCP_CHECK(hipEventRecord(v->feed_after, (hipStream_t)after_stream));
        PCP_CHECK(hipStreamWaitEvent(c->stream, v->feed_after, 0));
    }
    return feed_enqueue(c, pts_dev, layout, n, voxel_size, stored, begin_index, max_points, filter_outliers, max_staging, serial);
}

int pcp_voxel_add(pcp_ctx* c, const void* pts_host, int layout, int n, float voxel_size, int stored, int begin_index,
                  int max_points, int filter_outliers, int max_staging, int* serial) {
    if (int e = use(c, false)) return e;
    if (!pts_host) return pcp_fail("null points");
    if (layout != PCP_PTS_F32_XYZW && layout != PCP_PTS_F32_XYZ && layout != PCP_PTS_F32_ROWS && layout != PCP_PTS_F64_XYZ)
        return pcp_fail("unknown point layout %d", layout);
    if (n < 1) return pcp_fail("no points to voxelize");
    pcp_voxel_state* v = c->vox;
    const size_t bytes = (size_t)n * (layout == PCP_PTS_F64_XYZ ? 24 : (layout == PCP_PTS_F32_XYZW ? 16 : 12));
    if (bytes > v->feed_in_bytes) {
        PCP_CHECK(hipStreamSynchronize(c->stream));
        if (v->feed_in) PCP_CHECK(hipFree(v->feed_in));
        v->feed_in = nullptr; v->feed_in_bytes = 0;
        size_t cap = (size_t)1 << 22;
        while (cap < bytes) cap *= 2;
        PCP_CHECK(hipMalloc(&v->feed_in, cap));
        v->feed_in_bytes = cap;
    }
    PCP_CHECK(hipMemcpyAsync(v->feed_in, pts_host, bytes, hipMemcpyHostToDevice, c->stream));
    return pcp_voxel_add_dev(c, v->feed_in, layout, n, voxel_size, stored, begin_index, max_points, filter_outliers, max_staging,
                             nullptr, serial);
}

int pcp_voxel_add_wait(pcp_ctx* c, int serial, pcp_voxel_frame* out) {
    if (int e = use(c, false)) return e;
    if (!c->vox->feed) return pcp_fail("no frame has been fed");
    return feed_take(c, serial, out);
}

int pcp_voxel_add_unknown_read(pcp_ctx* c, int32_t* ids, int capacity, int* count) {
    if (int e = use(c, false)) return e;
    pcp_voxel_state* v = c->vox;
    if (!count) return pcp_fail("null count");
    if (!v->feed || v->feed_serial == 0) return pcp_fail("no frame has been fed");
    pcp_voxel_frame r;
    if (int e = feed_take(c, v->feed_serial - 1, &r)) return e;
    *count = r.unknown;
    if (r.unknown == 0 || !ids) return 0;               // ids == NULL asks for the count alone
    if (capacity < r.unknown) return pcp_fail("%d unknown points do not fit a list of %d", r.unknown, capacity);
    const int passes = (r.key_bits + 7) / 8;
    PCP_CHECK(hipMemcpyAsync(ids, (passes & 1) ? v->v1 : v->v0, (size_t)r.unknown * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int pcp_voxel_store_read(pcp_ctx* c, long long first, int n, float* xyzw) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (n == 0) return 0;
    if (!xyzw || n < 0 || first < 0) return pcp_fail("bad arguments");
    if (!v->keep_store) return pcp_fail("the device point store is off (pcp_voxel_store_enable)");
    if ((size_t)first + (size_t)n > v->store_n) return pcp_fail("points [%lld, %lld) requested, %zu stored", first, first + n, v->store_n);
    PCP_CHECK(hipMemcpyAsync(xyzw, v->store + first, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int pcp_voxel_log_begin(pcp_ctx* c, int slot, size_t capacity_points, void* after_stream) {
    if (int e = use(c, false)) return e;
    pcp_voxel_state* v = c->vox;
    if (slot < 0 || slot > 1) return pcp_fail("slot %d: there are two logs, 0 and 1", slot);
    if (capacity_points < 1 || capacity_points > 0xffffffffu) return pcp_fail("capacity of %zu points outside [1, 2^32)", capacity_points);
    if (!v->keep_store) return pcp_fail("the log resolves staged rows through the device point store (pcp_voxel_store_enable)");
    if (v->log_slot >= 0) return pcp_fail("log %d is open: pcp_voxel_log_end first", v->log_slot);
    if (v->log_state[slot] == 2) return pcp_fail("log %d has not been taken (pcp_voxel_log_take)", slot);
    hipStream_t s = c->stream;
    if (!v->log_ctl) {
        PCP_CHECK(hipMalloc(&v->log_ctl, 8 * sizeof(unsigned)));
        PCP_CHECK(hipHostMalloc((void**)&v->log_host, 8 * sizeof(unsigned)));
        for (int k = 0; k < 2; ++k) PCP_CHECK(hipEventCreateWithFlags(&v->log_done[k], hipEventDisableTiming));
    }
    if (capacity_points > v->log_cap[slot]) {           // the slot is idle: nothing reads its old buffer on this stream
        PCP_CHECK(hipStreamSynchronize(s));
        if (v->log_xyz[slot]) PCP_CHECK(hipFree(v->log_xyz[slot]));
        v->log_xyz[slot] = nullptr; v->log_cap[slot] = 0;
        PCP_CHECK(hipMalloc(&v->log_xyz[slot], capacity_points * 3 * sizeof(double)));
        v->log_cap[slot] = capacity_points;
    }
    if (after_stream && (hipStream_t)after_stream != s) {          // whoever read the slot's last contents
        if (!v->feed_after) PCP_CHECK(hipEventCreateWithFlags(&v->feed_after, hipEventDisableTiming));
        PCP_CHECK(hipEventRecord(v->feed_after, (hipStream_t)after_stream));
        PCP_CHECK(hipStreamWaitEvent(s, v->feed_after, 0));
    }
    PCP_CHECK(hipMemsetAsync(v->log_ctl + 4 * slot, 0, 4 * sizeof(unsigned), s));
    v->log_state[slot] = 1;
    v->log_slot = slot;
    return 0;
}

int pcp_voxel_log_end(pcp_ctx* c, int slot) {
    if (int e = use(c, false)) return e;
    pcp_voxel_state* v = c->vox;
    if (slot < 0 || slot > 1 || v->log_slot != slot) return pcp_fail("log %d is not open", slot);
    PCP_CHECK(hipMemcpyAsync(v->log_host + 4 * slot, v->log_ctl + 4 * slot, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipEventRecord(v->log_done[slot], c->stream));
    v->log_state[slot] = 2;
    v->log_slot = -1;
    return 0;
}

int pcp_voxel_log_take(pcp_ctx* c, int slot, const double** xyz_dev, size_t* n_points, int* n_rows) {
    if (int e = use(c, false)) return e;
    pcp_voxel_state* v = c->vox;
    if (!xyz_dev || !n_points) return pcp_fail("null output");
    *xyz_dev = nullptr; *n_points = 0;
    if (n_rows) *n_rows = 0;
    if (slot < 0 || slot > 1 || v->log_state[slot] != 2) return pcp_fail("log %d has not been ended (pcp_voxel_log_end)", slot);
    PCP_CHECK(hipEventSynchronize(v->log_done[slot]));
    v->log_state[slot] = 0;
    const unsigned* h = v->log_host + 4 * slot;
    if (h[1]) return pcp_fail("the log of %zu points overflowed after %u points: the staged points of at least one frame are lost "
                              "(open it with a larger capacity)", v->log_cap[slot], h[0]);
    if (h[2]) return pcp_fail("a staged row names a point the device store does not hold (%zu points stored)", v->store_n);
    *xyz_dev = v->log_xyz[slot];
    *n_points = h[0];
    if (n_rows) *n_rows = (int)h[3];
    return 0;
}

int pcp_voxel_staged_read(pcp_ctx* c, int rows, int max_points, int32_t* stage_out) {
    if (int e = use(c)) return e;
    pcp_voxel_state* v = c->vox;
    if (rows == 0) return 0;
    if (!stage_out || rows < 0 || !v->stage || rows > v->stage_rows || max_points != v->stage_width) return pcp_fail("bad staged row range");
    PCP_CHECK(hipMemcpyAsync(stage_out, v->stage, (size_t)rows * max_points * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PCP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
