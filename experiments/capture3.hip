// Experiment: which stream-capture shapes with three streams survive hipStreamEndCapture / hipGraphInstantiate on this ROCm build?
// argv[1] = case number.  Each case prints "case N ok (nodes)" or dies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
__global__ void k(float* p) { p[threadIdx.x] += 1.f; }
int main(int argc, char** argv) {
    const int c = argc > 1 ? atoi(argv[1]) : 1;
    hipStream_t s0, s1, s2;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    float* d; CK(hipMalloc(&d, 4096));
    hipEvent_t e[16];
    for (auto& x : e) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    if (c == 3 || c == 5) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, d);          // origin has a kernel before the fork
    CK(hipEventRecord(e[0], s0));
    CK(hipStreamWaitEvent(s1, e[0], 0));
    CK(hipStreamWaitEvent(s2, e[0], 0));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, d + 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s2, d + 128);
    if (c == 2 || c == 5 || c == 6) {   // cross edges between the two forked streams, both directions
        CK(hipEventRecord(e[1], s1)); CK(hipStreamWaitEvent(s2, e[1], 0));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s2, d + 128);
        CK(hipEventRecord(e[2], s2)); CK(hipStreamWaitEvent(s1, e[2], 0));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, d + 64);
    }
    if (c == 4 || c == 6) {   // the same event object recorded twice on a forked stream (re-record inside one capture)
        CK(hipEventRecord(e[3], s1)); CK(hipStreamWaitEvent(s2, e[3], 0));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, d + 64);
        CK(hipEventRecord(e[3], s1)); CK(hipStreamWaitEvent(s2, e[3], 0));
    }
    if (c >= 8) {   // the engine's order: s1 = lane 0 (joins first), s2 = lane 1; the waited event is older than the waiting stream's own tail
        CK(hipEventRecord(e[6], s1)); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, d + 64);
        CK(hipStreamWaitEvent(s2, e[6], 0));
        for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s2, d + 128); if (i == 1) CK(hipEventRecord(e[7], s2)); }
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, d + 64);
        CK(hipStreamWaitEvent(s1, e[7], 0));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, d + 64);
        if (c >= 9) {   // and back again
            CK(hipEventRecord(e[8], s1)); CK(hipStreamWaitEvent(s2, e[8], 0));
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s2, d + 128);
        }
        if (c >= 10) {  // a second wait of lane 0 on a later lane-1 event, and lane 1 on lane 0
            CK(hipEventRecord(e[9], s2)); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s2, d + 128);
            CK(hipStreamWaitEvent(s1, e[9], 0)); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, d + 64);
        }
    }
    CK(hipEventRecord(e[4], s1)); CK(hipStreamWaitEvent(s0, e[4], 0));
    CK(hipEventRecord(e[5], s2)); CK(hipStreamWaitEvent(s0, e[5], 0));
    if (c == 7) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, d);   // origin kernel after the join only
    printf("case %d: end capture...\n", c); fflush(stdout);
    CK(hipStreamEndCapture(s0, &g));
    size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
    printf("case %d: captured %zu nodes, instantiate...\n", c, nn); fflush(stdout);
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
    printf("case %d ok\n", c);
    return 0;
}
