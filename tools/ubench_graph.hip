// What a kernel boundary costs on one stream: N dependent small kernels launched one by one, and the same N kernels
// captured into a HIP graph and launched as one.  Device time by HIP events around the run.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_graph tools/ubench_graph.hip && /tmp/ubench_graph
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_small(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

int main() {
    const int N = 24, reps = 200;
    float* p;
    CK(hipMalloc(&p, 1 << 20));
    CK(hipMemset(p, 0, 1 << 20));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {1, 64, 2048}) {
        // one by one
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r)
                for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(64), 0, s, p, blocks * 64);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
        }
        float ms_stream = 0;
        CK(hipEventElapsedTime(&ms_stream, e0, e1));
        // captured
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(64), 0, s, p, blocks * 64);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms_graph = 0;
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms_graph, e0, e1));
        printf("%5d blocks: stream launches %.2f us per kernel, graph of %d kernels %.2f us per kernel\n", blocks,
               ms_stream * 1e3 / (reps * N), N, ms_graph * 1e3 / (reps * N));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
