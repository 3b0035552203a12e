// Frame preparation on the device: the numeric steps between a KITTI velodyne
// file and the registration kernels.
//   decode   LidarDataReader.eatBytes (LidarDataReader.py:144-150), the ego filter
//            |p| > 2.5 m in float32 (:59-61) and the axis-aligning rotation in
//            float64 (:66-67), as one ordered stream compaction
//   move     (R @ p.T).T + t, the IMU pre-alignment of a frame
//            (EnvironmentConstructor.py:169, :209)
//   rings    PointcloudIcpContainer.getScanLines (:362-392): azimuth of the
//            unit-sphere projection about origin + (0, 0.127, 0), a ring ends where
//            |azimuth[i+1] - azimuth[i]| > 6; points after the last wrap belong to no ring
// Buffers: raw float4[cap] (x, y, z, intensity), xyz double[3 cap], intensity
// float[cap], ends uint32[cap], block counts uint32[cap / 2048 + 1].
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/pcp.h"
#include "pcp_compact.h"
#include "pcp_host.h"

using namespace pcp;

namespace {

struct KeepFar {          // ego filter: np.linalg.norm(float32 rows) > min_range
    const float4* raw;
    float min_range;
    __device__ bool operator()(size_t i) const {
        const float4 p = raw[i];
        const float s = (p.x * p.x + p.y * p.y) + p.z * p.z;
        return sqrtf(s) > min_range;
    }
};

struct Wrap {             // ring boundary after point i
    const double* xyz;
    double ox, oy, oz;
    __device__ double azimuth(size_t i) const {
        const double vx = xyz[3 * i + 0] - ox, vy = (xyz[3 * i + 1] - oy) - 0.127, vz = xyz[3 * i + 2] - oz;
        const double mag = sqrt((vx * vx + vy * vy) + vz * vz);
        double a = atan2(vz / mag, vx / mag);
        if (a < 0.0) a += 6.283185307179586;        // np.pi * 2
        return a;
    }
    __device__ bool operator()(size_t i) const { return fabs(azimuth(i + 1) - azimuth(i)) > 6.0; }
};

struct EmitDecoded {      // rotated float64 point + intensity
    const float4* raw;
    double R[9];
    double* xyz;
    float* intensity;
    __device__ void operator()(size_t i, size_t slot) const {
        const float4 p = raw[i];
        const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
        // one fused chain per row, the order of the float64 GEMM kernel NumPy's matmul runs on x86-64
        // (checked against the reference's frames: two points of KITTI frame 0 tell the orders apart)
        xyz[3 * slot + 0] = fma(R[2], z, fma(R[1], y, R[0] * x));
        xyz[3 * slot + 1] = fma(R[5], z, fma(R[4], y, R[3] * x));
        xyz[3 * slot + 2] = fma(R[8], z, fma(R[7], y, R[6] * x));
        intensity[slot] = p.w;
    }
};

struct EmitIndex {
    unsigned* out;
    __device__ void operator()(size_t i, size_t slot) const { out[slot] = (unsigned)i; }
};

__global__ void k_move(double* __restrict__ xyz, size_t n, const double* __restrict__ Rt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    xyz[3 * i + 0] = fma(Rt[2], z, fma(Rt[1], y, Rt[0] * x)) + Rt[9];
    xyz[3 * i + 1] = fma(Rt[5], z, fma(Rt[4], y, Rt[3] * x)) + Rt[10];
    xyz[3 * i + 2] = fma(Rt[8], z, fma(Rt[7], y, Rt[6] * x)) + Rt[11];
}

// ComputeShader.pad_points (:956-961) + the float32 cast of the upload (:301-316).
__global__ void k_pad32(const double* __restrict__ xyz, size_t n, float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = make_float4((float)xyz[3 * i + 0], (float)xyz[3 * i + 1], (float)xyz[3 * i + 2], 0.0f);
}

inline unsigned blocks_for(size_t n, int threads) { return (unsigned)((n + threads - 1) / threads); }

}  // namespace

__global__ void k_interleave_rows(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                                  size_t n, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[3 * i + 0] = x[i]; out[3 * i + 1] = y[i]; out[3 * i + 2] = z[i];
}

struct pcp_frame {
    int device = 0;
    hipStream_t stream = nullptr;
    size_t cap = 0, n = 0;
    float4* raw = nullptr;
    double* xyz = nullptr;
    float* intensity = nullptr;
    float4* xyzw32 = nullptr;
    unsigned *ends = nullptr, *counts = nullptr;
    double* rt = nullptr;
};

extern "C" {

int pcp_frame_destroy(pcp_frame* f);

int pcp_frame_create(int device, size_t max_points, pcp_frame** out) {
    if (!out) return pcp_fail("out is null");
    *out = nullptr;
    if (max_points == 0 || max_points >= (1ull << 31)) return pcp_fail("max_points must be in [1, 2^31)");
    int count = 0;
    PCP_CHECK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return pcp_fail("device %d out of range (%d devices)", device, count);
    PCP_CHECK(hipSetDevice(device));
    pcp_frame* f = new pcp_frame();
    f->device = device;
    f->cap = max_points;
    hipError_t e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    A((void**)&f->raw, max_points * 16);
    A((void**)&f->xyz, max_points * 24);
    A((void**)&f->intensity, max_points * 4);
    A((void**)&f->xyzw32, max_points * 16);
    A((void**)&f->ends, max_points * 4);
    A((void**)&f->counts, (max_points / FR_TILE + 2) * 4);
    A((void**)&f->rt, 12 * 8);
    if (e != hipSuccess) {
        pcp_fail("pcp_frame_create: %s", hipGetErrorString(e));
        pcp_frame_destroy(f);          // frees what was allocated
        return 1;
    }
    *out = f;
    return 0;
}

int pcp_frame_destroy(pcp_frame* f) {
    if (!f) return 0;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    for (void* p : {(void*)f->raw, (void*)f->xyz, (void*)f->intensity, (void*)f->xyzw32, (void*)f->ends,
                    (void*)f->counts, (void*)f->rt})
        if (p) (void)hipFree(p);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
    return 0;
}

int pcp_frame_decode(pcp_frame* f, const float* raw_xyzi, size_t n, float min_range, const double* R9, size_t* n_out) {
    if (!f) return pcp_fail("null frame");
    if (n > f->cap) return pcp_fail("%zu raw points exceed the frame capacity %zu", n, f->cap);
    if (n && !raw_xyzi) return pcp_fail("raw_xyzi is null");
    if (!R9) return pcp_fail("R9 is null");
    PCP_CHECK(hipSetDevice(f->device));
    f->n = 0;
    if (n_out) *n_out = 0;
    if (n == 0) return 0;
    PCP_CHECK(hipMemcpyAsync(f->raw, raw_xyzi, n * 16, hipMemcpyHostToDevice, f->stream));
    const int nblk = (int)((n + FR_TILE - 1) / FR_TILE);
    const KeepFar pred{f->raw, min_range};
    EmitDecoded emit;
    emit.raw = f->raw;
    memcpy(emit.R, R9, sizeof(emit.R));
    emit.xyz = f->xyz;
    emit.intensity = f->intensity;
    compact_ordered(f->stream, pred, emit, n, f->counts);
    PCP_CHECK(hipGetLastError());
    unsigned kept = 0;
    PCP_CHECK(hipMemcpyAsync(&kept, f->counts + nblk, 4, hipMemcpyDeviceToHost, f->stream));
    PCP_CHECK(hipStreamSynchronize(f->stream));
    f->n = kept;
    if (n_out) *n_out = kept;
    return 0;
}

int pcp_frame_upload(pcp_frame* f, const double* xyz, size_t n) {
    if (!f) return pcp_fail("null frame");
    if (n > f->cap) return pcp_fail("%zu points exceed the frame capacity %zu", n, f->cap);
    if (n && !xyz) return pcp_fail("xyz is null");
    PCP_CHECK(hipSetDevice(f->device));
    if (n) PCP_CHECK(hipMemcpyAsync(f->xyz, xyz, n * 24, hipMemcpyHostToDevice, f->stream));
    PCP_CHECK(hipStreamSynchronize(f->stream));
    f->n = n;
    return 0;
}

int pcp_frame_upload_rows(pcp_frame* f, const double* x, const double* y, const double* z, size_t n) {
    if (!f) return pcp_fail("null frame");
    if (n > f->cap) return pcp_fail("%zu points exceed the frame capacity %zu", n, f->cap);
    if (n && (!x || !y || !z)) return pcp_fail("a coordinate row is null");
    PCP_CHECK(hipSetDevice(f->device));
    if (n) {
        // staging in buffers this call owns anyway: x and z in the two halves of the raw records (16 B per
        // point), y in the float32 copy (rebuilt from the cloud when it is next asked for)
        double* sx = reinterpret_cast<double*>(f->raw);
        double* sz = sx + n;
        double* sy = reinterpret_cast<double*>(f->xyzw32);
        PCP_CHECK(hipMemcpyAsync(sx, x, n * 8, hipMemcpyHostToDevice, f->stream));
        PCP_CHECK(hipMemcpyAsync(sy, y, n * 8, hipMemcpyHostToDevice, f->stream));
        PCP_CHECK(hipMemcpyAsync(sz, z, n * 8, hipMemcpyHostToDevice, f->stream));
        hipLaunchKernelGGL(k_interleave_rows, dim3(blocks_for(n, 256)), dim3(256), 0, f->stream, sx, sy, sz, n, f->xyz);
        PCP_CHECK(hipGetLastError());
    }
    PCP_CHECK(hipStreamSynchronize(f->stream));
    f->n = n;
    return 0;
}

int pcp_frame_move(pcp_frame* f, const double* R9, const double* t3) {
    if (!f) return pcp_fail("null frame");
    if (!R9 || !t3) return pcp_fail("R or t is null");
    PCP_CHECK(hipSetDevice(f->device));
    if (f->n == 0) return 0;
    double rt[12];
    memcpy(rt, R9, 72);
    memcpy(rt + 9, t3, 24);
    PCP_CHECK(hipStreamSynchronize(f->stream));
    PCP_CHECK(hipMemcpyAsync(f->rt, rt, sizeof(rt), hipMemcpyHostToDevice, f->stream));
    hipLaunchKernelGGL(k_move, dim3(blocks_for(f->n, 256)), dim3(256), 0, f->stream, f->xyz, f->n, f->rt);
    PCP_CHECK(hipGetLastError());
    PCP_CHECK(hipStreamSynchronize(f->stream));
    return 0;
}

int pcp_frame_scan_lines(pcp_frame* f, const double* origin3, int max_lines, uint32_t* lines, int* n_lines) {
    if (!f) return pcp_fail("null frame");
    if (!origin3 || !lines || !n_lines) return pcp_fail("origin3, lines or n_lines is null");
    PCP_CHECK(hipSetDevice(f->device));
    *n_lines = 0;
    if (f->n < 2) return 0;
    const size_t m = f->n - 1;
    const int nblk = (int)((m + FR_TILE - 1) / FR_TILE);
    const Wrap pred{f->xyz, origin3[0], origin3[1], origin3[2]};
    const EmitIndex emit{f->ends};
    compact_ordered(f->stream, pred, emit, m, f->counts);
    PCP_CHECK(hipGetLastError());
    unsigned found = 0;
    PCP_CHECK(hipMemcpyAsync(&found, f->counts + nblk, 4, hipMemcpyDeviceToHost, f->stream));
    PCP_CHECK(hipStreamSynchronize(f->stream));
    if ((long long)found > (long long)max_lines)
        return pcp_fail("%u scan lines found, room for %d", found, max_lines);
    std::vector<unsigned> ends(found);
    if (found) PCP_CHECK(hipMemcpy(ends.data(), f->ends, (size_t)found * 4, hipMemcpyDeviceToHost));
    unsigned begin = 0;
    for (unsigned k = 0; k < found; ++k) {
        lines[2 * k + 0] = begin;
        lines[2 * k + 1] = ends[k];
        begin = ends[k] + 1;
    }
    *n_lines = (int)found;
    return 0;
}

int pcp_frame_read(pcp_frame* f, double* xyz, float* intensity) {
    if (!f) return pcp_fail("null frame");
    PCP_CHECK(hipSetDevice(f->device));
    if (f->n && xyz) PCP_CHECK(hipMemcpyAsync(xyz, f->xyz, f->n * 24, hipMemcpyDeviceToHost, f->stream));
    if (f->n && intensity) PCP_CHECK(hipMemcpyAsync(intensity, f->intensity, f->n * 4, hipMemcpyDeviceToHost, f->stream));
    PCP_CHECK(hipStreamSynchronize(f->stream));
    return 0;
}

int pcp_frame_points_dev(pcp_frame* f, const double** xyz_dev, size_t* n) {
    if (!f) return pcp_fail("null frame");
    if (xyz_dev) *xyz_dev = f->xyz;
    if (n) *n = f->n;
    return 0;
}

int pcp_frame_padded_dev(pcp_frame* f, const float** xyzw_dev, size_t* n) {
    if (!f) return pcp_fail("null frame");
    PCP_CHECK(hipSetDevice(f->device));
    if (f->n) {
        hipLaunchKernelGGL(k_pad32, dim3(blocks_for(f->n, 256)), dim3(256), 0, f->stream, f->xyz, f->n, f->xyzw32);
        PCP_CHECK(hipGetLastError());
        PCP_CHECK(hipStreamSynchronize(f->stream));
    }
    if (xyzw_dev) *xyzw_dev = reinterpret_cast<const float*>(f->xyzw32);
    if (n) *n = f->n;
    return 0;
}

}  // extern "C"
