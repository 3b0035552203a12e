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

// rotation_xyz with its six trigonometric values taken from the first six lanes of the calling wave (lane j < 3: cos of
// angle j, lane 3 + j: sin of angle j; float64 evaluation rounded to float32, as above): the evaluations, which are most
// of a step's serial time on one lane, run side by side.  Every lane of the wave must call it; all get the same matrix.
__device__ __forceinline__ void rotation_xyz_wave(const float xr, const float yr, const float zr, double* R) {
    const int lane = threadIdx.x & 63;
    const float ang = (lane % 3 == 0) ? xr : ((lane % 3 == 1) ? yr : zr);
    const double a = (double)ang;
    const float mine = (lane < 3) ? (float)cos(a) : (float)sin(a);
    const double cx = (double)__shfl(mine, 0, 64), cy = (double)__shfl(mine, 1, 64), cz = (double)__shfl(mine, 2, 64);
    const double sx = (double)__shfl(mine, 3, 64), sy = (double)__shfl(mine, 4, 64), sz = (double)__shfl(mine, 5, 64);
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
    // Called by every lane of one wave with the same values: the solve runs redundantly in all lanes (no lane waits
    // for another), the rotation's trigonometry is spread over six of them, lane 0 stores.
    const bool writer = (threadIdx.x & 63) == 0;
    double H[6][6], rhs[6], x[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c < 6; ++c) H[r][c] = (double)v[r <= c ? tri(r, c) : tri(c, r)];
        rhs[r] = -(double)v[21 + r];
    }
    if (!solve6(H, rhs, x)) {
        if (writer) {
            st->singular = 1;
            st->done = 1;
            st->pending = 0;
            st->iter = it + 1;
        }
        return;
    }
    float dx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        dx[k] = (float)x[k];
        if (writer) dx_out[(size_t)it * 6 + k] = dx[k];
    }
    if (writer) st->iter = it + 1;
    if (mode == PCP_MODE_PLANE && !full_iter) {
        float s = 0.0f;                       // np.mean(np.abs(delta_x)), float32
#pragma unroll
        for (int k = 0; k < 6; ++k) s = s + fabsf(dx[k]);
        if (s / 6.0f < tol) {
            if (writer) {
                st->converged = 1;
                st->done = 1;
                st->pending = 0;
            }
            return;
        }
    }
    double R[9];
    rotation_xyz_wave(dx[3], dx[4], dx[5], R);
    if (!writer) return;
#pragma unroll
    for (int k = 0; k < 9; ++k) st->R[k] = R[k];
    st->t[0] = (double)dx[0]; st->t[1] = (double)dx[1]; st->t[2] = (double)dx[2];
    st->pending = 1;
    if (it + 1 >= max_iter) st->done = 1;
}

// U V^T of the singular value decomposition W = U S V^T of a 3x3 matrix (row-major), what
// uniform_optimal_icp (PointcloudIcpContainer.py:342-344) forms from np.linalg.svd -- without a
// determinant fix, as there.  One-sided Jacobi: the columns of A = W V are made orthogonal by plane
// rotations collected in V; their lengths are the singular values and U = A / S.  For a W of full rank
// U V^T is its orthogonal polar factor and does not depend on how the decomposition orders or signs the
// singular triplets.  A column of zero length (rank-deficient W, where LAPACK's choice of U is arbitrary)
// is replaced by the cross product of the other two, and `deficient` is set.
__device__ inline void polar_uvt3(const double* W, double* R, int* deficient) {
    double A[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { A[i][j] = W[i * 3 + j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int k = 0; k < 3; ++k) { alpha += A[k][p] * A[k][p]; beta += A[k][q] * A[k][q]; gamma += A[k][p] * A[k][q]; }
                if (gamma == 0.0 || fabs(gamma) <= 2.3e-16 * sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double ap = A[k][p], aq = A[k][q];
                    A[k][p] = c * ap - s * aq; A[k][q] = s * ap + c * aq;
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sig[3], big = 0.0;
    for (int j = 0; j < 3; ++j) {
        sig[j] = sqrt((A[0][j] * A[0][j] + A[1][j] * A[1][j]) + A[2][j] * A[2][j]);
        big = sig[j] > big ? sig[j] : big;
    }
    int bad = 0;
    for (int j = 0; j < 3; ++j) {
        if (sig[j] > 1e-300 && sig[j] > 1e-14 * big) { for (int k = 0; k < 3; ++k) A[k][j] /= sig[j]; }
        else bad |= 1 << j;
    }
    if (bad) {
        for (int j = 0; j < 3; ++j) {
            if (!(bad & (1 << j))) continue;
            const int a = (j + 1) % 3, b = (j + 2) % 3;
            if ((bad & (1 << a)) || (bad & (1 << b))) {       // fewer than two directions left: any orthonormal completion
                for (int k = 0; k < 3; ++k) A[k][j] = (k == j) ? 1.0 : 0.0;
                continue;
            }
            A[0][j] = A[1][a] * A[2][b] - A[2][a] * A[1][b];
            A[1][j] = A[2][a] * A[0][b] - A[0][a] * A[2][b];
            A[2][j] = A[0][a] * A[1][b] - A[1][a] * A[0][b];
        }
    }
    if (deficient) *deficient = bad ? 1 : 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = (A[i][0] * V[j][0] + A[i][1] * V[j][1]) + A[i][2] * V[j][2];
}

// Coordinate k (0..9) of np.linspace(-1, 1, 10): arange * step + start with step = 2 / 9, the end point set to stop
// (PointcloudIcpContainer.getUniformGrid :429-438).
__device__ __forceinline__ double grid_axis10(const int k) {
    if (k == 9) return 1.0;
    const double step = 2.0 / 9.0;
    const double v = (double)k * step;
    return v + (-1.0);
}

// The pose a registration returns (full_pt2pt :84-100, full_pt2pl :118-140): the 10x10x10 tracking grid on
// [-1, 1]^3 is moved by every applied step, g <- R g + t (applyIcpStep :320-322, float64, left to right), and
// uniform_optimal_icp (:333-347) reads the pose off it: means, W = (g' - mean')^T (g - mean), R_opt = U V^T,
// t_opt = mean' - R_opt mean.  One block of PCP_POSE_THREADS threads; thread 0 leaves t_opt in out[0..3), R_opt (row-major)
// in out[3..12), out[12] = 1 when W was rank-deficient.  `applied` steps of 6 float32 (t, then the angles) are read from dx.
#define PCP_POSE_THREADS 256
#define PCP_POSE_CHUNK 32         // steps whose rotations are prepared together (6 threads per step)
__device__ inline void pose_from_steps(const float* __restrict__ dx, const int applied, double* __restrict__ out) {
    __shared__ double s_red[PCP_POSE_THREADS / 64][12];
    __shared__ double s_mean[6];
    __shared__ double s_Rt[PCP_POSE_CHUNK][12];
    __shared__ double s_trig[PCP_POSE_CHUNK][6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = (1000 + PCP_POSE_THREADS - 1) / PCP_POSE_THREADS;
    double g0[PER][3], g[PER][3];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + j * PCP_POSE_THREADS;
        const int ii = i < 1000 ? i : 0;
        g0[j][0] = grid_axis10(ii / 100); g0[j][1] = grid_axis10((ii / 10) % 10); g0[j][2] = grid_axis10(ii % 10);
        g[j][0] = g0[j][0]; g[j][1] = g0[j][1]; g[j][2] = g0[j][2];
    }
    // the steps' rotations, PCP_POSE_CHUNK at a time: thread 6 s + j takes trigonometric value j of step s (cos of the
    // three angles, then their sines), thread s then multiplies the three axis rotations out (rotation_xyz's products)
    for (int base = 0; base < applied; base += PCP_POSE_CHUNK) {
        const int chunk = min(applied - base, PCP_POSE_CHUNK);
        __syncthreads();
        if (tid < 6 * chunk) {
            const int s = tid / 6, j = tid % 6;
            const double a = (double)dx[(size_t)(base + s) * 6 + 3 + j % 3];
            s_trig[s][j] = (double)((j < 3) ? (float)cos(a) : (float)sin(a));
        }
        __syncthreads();
        if (tid < chunk) {
            const double cx = s_trig[tid][0], cy = s_trig[tid][1], cz = s_trig[tid][2];
            const double sx = s_trig[tid][3], sy = s_trig[tid][4], sz = s_trig[tid][5];
            const double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1.0};
            const double Ry[9] = {cy, 0, sy, 0, 1.0, 0, -sy, 0, cy};
            const double Rx[9] = {1.0, 0, 0, 0, cx, -sx, 0, sx, cx};
            double M[9], R[9];
            mm3(Rx, Ry, M);
            mm3(M, Rz, R);
            const float* d = dx + (size_t)(base + tid) * 6;
            for (int k = 0; k < 9; ++k) s_Rt[tid][k] = R[k];
            s_Rt[tid][9] = (double)d[0]; s_Rt[tid][10] = (double)d[1]; s_Rt[tid][11] = (double)d[2];
        }
        __syncthreads();
        for (int s = 0; s < chunk; ++s) {
            const double* Rt = s_Rt[s];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const double x = g[j][0], y = g[j][1], z = g[j][2];
                g[j][0] = ((Rt[0] * x + Rt[1] * y) + Rt[2] * z) + Rt[9];
                g[j][1] = ((Rt[3] * x + Rt[4] * y) + Rt[5] * z) + Rt[10];
                g[j][2] = ((Rt[6] * x + Rt[7] * y) + Rt[8] * z) + Rt[11];
            }
        }
    }
    // block sums in a fixed order: a thread's own points, the wave's butterfly, the waves one after the other
    auto block_sum = [&](double (&v)[12], const int n) {
        for (int k = 0; k < n; ++k)
            for (int sft = 32; sft > 0; sft >>= 1) v[k] += __shfl_down(v[k], sft, 64);
        __syncthreads();
        if (lane == 0) for (int k = 0; k < n; ++k) s_red[wave][k] = v[k];
        __syncthreads();
    };
    double v[12];
    for (int k = 0; k < 12; ++k) v[k] = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (tid + j * PCP_POSE_THREADS < 1000) {
            for (int k = 0; k < 3; ++k) { v[k] += g[j][k]; v[3 + k] += g0[j][k]; }
        }
    }
    block_sum(v, 6);
    if (tid == 0)
        for (int k = 0; k < 6; ++k) {
            double sum = 0.0;
            for (int w = 0; w < PCP_POSE_THREADS / 64; ++w) sum += s_red[w][k];
            s_mean[k] = sum / 1000.0;
        }
    __syncthreads();
    for (int k = 0; k < 12; ++k) v[k] = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (tid + j * PCP_POSE_THREADS < 1000) {
            const double a[3] = {g[j][0] - s_mean[0], g[j][1] - s_mean[1], g[j][2] - s_mean[2]};
            const double b[3] = {g0[j][0] - s_mean[3], g0[j][1] - s_mean[4], g0[j][2] - s_mean[5]};
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) v[r * 3 + c] += a[r] * b[c];
        }
    }
    block_sum(v, 9);
    if (tid == 0) {
        double W[9], R[9];
        for (int k = 0; k < 9; ++k) {
            double sum = 0.0;
            for (int w = 0; w < PCP_POSE_THREADS / 64; ++w) sum += s_red[w][k];
            W[k] = sum;
        }
        int deficient = 0;
        polar_uvt3(W, R, &deficient);
        for (int r = 0; r < 3; ++r)
            out[r] = s_mean[r] - ((R[r * 3 + 0] * s_mean[3] + R[r * 3 + 1] * s_mean[4]) + R[r * 3 + 2] * s_mean[5]);
        for (int k = 0; k < 9; ++k) out[3 + k] = R[k];
        out[12] = deficient ? 1.0 : 0.0;
    }
    __syncthreads();
}

// Upper levels of the ParalellHBSum tree (ComputeShader.py:682-707) over the
// level-1 partials, then -- when solve != 0 -- the host half of one iteration:
// dispatchPointToPoint / dispatchPointToPlane (PointcloudIcpContainer.py:24-29,
// :47-69).  One block of 1024 threads.
struct FinishArgs {
    const float* partials; int n1; float* lvl0; float* lvl1;
    IcpState* st; float* hb_out; float* dx_out;
    double* pose;        // 16 doubles: t_opt[3], R_opt[9], rank flag, steps applied, 2 unused (written by the pose kernel)
};

// The pose of a finished loop from its recorded steps (see pose_from_steps).  A converged point-to-plane loop did not
// apply its last step (PointcloudIcpContainer.py:121-122: break before apply); a singular system leaves NaN.
__device__ inline void pose_body(const FinishArgs& F) {
    const IcpState* st = F.st;
    const int iters = st->iter, singular = st->singular;
    int applied = iters - (st->converged ? 1 : 0);
    if (singular) applied = iters - 1;
    if (applied < 0) applied = 0;
    pose_from_steps(F.dx_out, applied, F.pose);
    if (threadIdx.x == 0) {
        F.pose[13] = (double)applied;
        if (singular) { const double nan = __longlong_as_double(0x7ff8000000000000ll); for (int k = 0; k < 12; ++k) F.pose[k] = nan; }
    }
}

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
    if (threadIdx.x >= 64) return;            // the first wave finishes: every lane holds the 27 sums
    const int it = solve ? st->iter : 0;
    float v[PCP_NVAL];
#pragma unroll
    for (int k = 0; k < PCP_NVAL; ++k) v[k] = in[k];
    {   // the summed tile as the reference reads it back (8 x 8, rows 6-7 and column 7 zero): element `lane`
        const int r = lane >> 3, c = lane & 7;
        float val = 0.0f;
#pragma unroll
        for (int rr = 0; rr < 6; ++rr)
#pragma unroll
            for (int cc = 0; cc < 7; ++cc)
                if (rr == r && cc == c) val = (cc < 6) ? v[rr <= cc ? tri(rr, cc) : tri(cc, rr)] : v[21 + rr];
        hb_out[(size_t)it * 64 + lane] = val;
    }
    if (!solve) return;
    pose_step(v, st, dx_out, it, mode, max_iter, tol, full_iter);
}

}  // namespace pcp
