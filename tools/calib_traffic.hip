// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access widths of the fused search kernel's
// streams, against known byte counts: (1) a float64 cloud read as three 8-byte loads per lane at a 24-byte
// stride and written back the same way (the working copy of the source points), (2) a float per lane read
// and written (the ring hints).  Tables of 768 MB / 512 MB, beyond the 256 MB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o calib_traffic calib_traffic.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_calib_move24(const double* __restrict__ in, double* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[3 * i + 0], y = in[3 * i + 1], z = in[3 * i + 2];
    out[3 * i + 0] = x + 1.0; out[3 * i + 1] = y + 1.0; out[3 * i + 2] = z + 1.0;
}

__global__ void k_calib_float_rw(float* __restrict__ h, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    h[i] = h[i] + 1.0f;
}

int main() {
    const size_t n = 32u << 20;          // 32 M points: 768 MB in, 768 MB out
    double *a = nullptr, *b = nullptr;
    float* h = nullptr;
    CHECK(hipMalloc(&a, n * 24));
    CHECK(hipMalloc(&b, n * 24));
    CHECK(hipMemset(a, 0, n * 24));
    CHECK(hipMemset(b, 0, n * 24));
    const size_t m = 128u << 20;         // 128 M floats: 512 MB read and written
    CHECK(hipMalloc(&h, m * 4));
    CHECK(hipMemset(h, 0, m * 4));
    CHECK(hipDeviceSynchronize());
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k_calib_move24, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, a, b, n);
        hipLaunchKernelGGL(k_calib_float_rw, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, 0, h, m);
    }
    CHECK(hipDeviceSynchronize());
    printf("k_calib_move24: %zu bytes read, %zu bytes written per launch\n", n * 24, n * 24);
    printf("k_calib_float_rw: %zu bytes read, %zu bytes written per launch\n", m * 4, m * 4);
    return 0;
}
