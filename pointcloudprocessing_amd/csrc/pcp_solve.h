// Device-side pose algebra shared by the ring and the cell-grid registration
// kernels: the 6x6 solve, the rotation of a step, the upper levels of the tile
// tree and the per-iteration state update.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/pcp.h"
#include "pcp_ctx.h"
#include "pcp_device.h"

namespace pcp {

// 6x6 solve in float64: Gaussian elimination with partial pivoting, the
// arithmetic order of the host checker's solve6.  Every index is a compile-time
// constant after unrolling (the row exchange is a predicated select), so the
// system lives in registers.  Returns false when singular.
__device__ __forceinline__ bool solve6(const double (&H)[6][6], const double (&rhs)[6], double (&x)[6]) {
    double A[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c < 6; ++c) A[r][c] = H[r][c];
        x[r] = rhs[r];
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double piv = A[k][k];
#pragma unroll
        for (int r = k + 1; r < 6; ++r)
            if (fabs(A[r][k]) > fabs(piv)) { piv = A[r][k]; p = r; }
        if (piv == 0.0 || piv != piv) ok = false;
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const bool sw = (p == r);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double top = A[k][c], bot = A[r][c];
                A[k][c] = sw ? bot : top;
                A[r][c] = sw ? top : bot;
            }
            const double top = x[k], bot = x[r];
            x[k] = sw ? bot : top;
            x[r] = sw ? top : bot;
        }
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const double m = A[r][k] / A[k][k];
#pragma unroll
            for (int c = k + 1; c < 6; ++c) A[r][c] = A[r][c] - m * A[k][c];
            x[r] = x[r] - m * x[k];
        }
    }
#pragma unroll
    for (int k = 5; k >= 0; --k) {
        double sum = x[k];
#pragma unroll
        for (int c = k + 1; c < 6; ++c) sum = sum - A[k][c] * x[c];
        x[k] = sum / A[k][k];
    }
    return ok;
}

__device__ __forceinline__ void mm3(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j]) + A[i * 3 + 2] * B[2 * 3 + j];
}

// PointcloudAlignment.rotation (PointcloudAlignment.py:229-261): Rx @ Ry @ Rz with
// float32 trig values held in float64 matrices.
__device__ __forceinline__ void rotation_xyz(float xr, float yr, float zr, double* R) {
    const double cx = (double)(float)cos((double)xr), sx = (double)(float)sin((double)xr);
    const double cy = (double)(float)cos((double)yr), sy = (double)(float)sin((double)yr);
    const double cz = (double)(float)cos((double)zr), sz = (double)(float)sin((double)zr);
    const double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1.0};
    const double Ry[9] = {cy, 0, sy, 0, 1.0, 0, -sy, 0, cy};
    const double Rx[9] = {1.0, 0, 0, 0, cx, -sx, 0, sx, cx};
    double M[9];
    mm3(Rx, Ry, M);
    mm3(M, Rz, R);
}

// The host half of one iteration, dispatchPointToPoint / dispatchPointToPlane
// (PointcloudIcpContainer.py:24-29, :47-69), from the 27 summed values: solve in
// float64, round dx to float32, convergence test, rotation for the next step.
__device__ __forceinline__ void pose_step(const float (&v)[PCP_NVAL], IcpState* __restrict__ st,
                                          float* __restrict__ dx_out, const int it, const int mode,
                                          const int max_iter, const float tol, const int full_iter) {
    double H[6][6], rhs[6], x[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c < 6; ++c) H[r][c] = (double)v[r <= c ? tri(r, c) : tri(c, r)];
        rhs[r] = -(double)v[21 + r];
    }
    if (!solve6(H, rhs, x)) {
        st->singular = 1;
        st->done = 1;
        st->pending = 0;
        st->iter = it + 1;
        return;
    }
    float dx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { dx[k] = (float)x[k]; dx_out[(size_t)it * 6 + k] = dx[k]; }
    st->iter = it + 1;
    if (mode == PCP_MODE_PLANE && !full_iter) {
        float s = 0.0f;                       // np.mean(np.abs(delta_x)), float32
#pragma unroll
        for (int k = 0; k < 6; ++k) s = s + fabsf(dx[k]);
        if (s / 6.0f < tol) {
            st->converged = 1;
            st->done = 1;
            st->pending = 0;
            return;
        }
    }
    rotation_xyz(dx[3], dx[4], dx[5], st->R);
    st->t[0] = (double)dx[0]; st->t[1] = (double)dx[1]; st->t[2] = (double)dx[2];
    st->pending = 1;
    if (it + 1 >= max_iter) st->done = 1;
}

// Upper levels of the ParalellHBSum tree (ComputeShader.py:682-707) over the
// level-1 partials, then -- when solve != 0 -- the host half of one iteration:
// dispatchPointToPoint / dispatchPointToPlane (PointcloudIcpContainer.py:24-29,
// :47-69).  One block of 1024 threads.
struct FinishArgs {
    const float* partials; int n1; float* lvl0; float* lvl1;
    IcpState* st; float* hb_out; float* dx_out;
};

__device__ __forceinline__ void finish_body(const FinishArgs& F, const int solve, const int mode, const int max_iter,
                                            const float tol, const int full_iter) {
    const float* __restrict__ partials = F.partials;
    float* __restrict__ lvl0 = F.lvl0;
    float* __restrict__ lvl1 = F.lvl1;
    IcpState* __restrict__ st = F.st;
    float* __restrict__ hb_out = F.hb_out;
    float* __restrict__ dx_out = F.dx_out;
    const int n1 = F.n1;
    if (solve && st->done) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const float* in = partials;
    float* out = lvl0;
    float* other = lvl1;
    int n = n1;
    // Lane l keeps element (l & 31) of the records with parity (l >> 5): one load
    // instruction covers two adjacent 128-byte records, the tree steps s = 32..2
    // pair records of equal parity (register adds), the last step s = 1 pairs
    // record 0 with record 1 across the two half-waves.
    const int e = lane & 31, par = lane >> 5;
    while (n > 1) {
        const int groups = (n + 63) >> 6;
        for (int g = wave; g < groups; g += nwaves) {
            float v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int rec = g * 64 + 2 * j + par;
                v[j] = (rec < n) ? in[(size_t)rec * PCP_PSTRIDE + e] : 0.0f;
            }
#pragma unroll
            for (int s = 16; s > 0; s >>= 1) {
#pragma unroll
                for (int j = 0; j < s; ++j) v[j] = v[j] + v[j + s];
            }
            const float sum = v[0] + __shfl_down(v[0], 32, 64);
            if (par == 0) out[(size_t)g * PCP_PSTRIDE + e] = sum;
        }
        __threadfence_block();
        __syncthreads();
        in = out;
        float* tmp = out; out = other; other = tmp;
        n = groups;
    }
    if (threadIdx.x != 0) return;
    const int it = solve ? st->iter : 0;
    float v[PCP_NVAL];
#pragma unroll
    for (int k = 0; k < PCP_NVAL; ++k) v[k] = in[k];
    float* hb = hb_out + (size_t)it * 64;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float val = 0.0f;
            if (r < 6 && c < 6) val = v[r <= c ? tri(r, c) : tri(c, r)];
            else if (r < 6 && c == 6) val = v[21 + r];
            hb[r * 8 + c] = val;
        }
    if (!solve) return;
    pose_step(v, st, dx_out, it, mode, max_iter, tol, full_iter);
}

}  // namespace pcp
