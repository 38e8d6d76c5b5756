// capture_legal.hip -- which HIP calls may a thread make on stream B while stream A is being captured (global capture mode, what
// torch.cuda.graph uses)?  Each call is tried in a fresh capture; prints its return code and whether the capture survived; then the same
// with the calling thread's capture mode exchanged to hipStreamCaptureModeRelaxed around the call (hipThreadExchangeStreamCaptureMode).
//   hipcc --offload-arch=gfx950 tools/probes/capture_legal.hip -o /tmp/capture_legal && /tmp/capture_legal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
__global__ void k(float *p) { p[threadIdx.x] += 1.f; }
int main() {
    hipStream_t a, b;
    hipStreamCreate(&a), hipStreamCreate(&b);
    float *d, *d2, *pinned;
    hipMalloc(&d, 4096), hipMalloc(&d2, 4096);
    hipHostMalloc(&pinned, 4096, hipHostMallocDefault);
    float pageable[16];
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    struct T { const char *name; std::function<hipError_t()> fn; };
    T tests[] = {
        {"kernel launch on B", [&] { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, b, d2); return hipGetLastError(); }},
        {"hipMemsetAsync on B", [&] { return hipMemsetAsync(d2, 0, 64, b); }},
        {"hipMemcpyAsync D2D on B", [&] { return hipMemcpyAsync(d2, d2 + 256, 64, hipMemcpyDeviceToDevice, b); }},
        {"hipMemcpyAsync D2H pinned on B", [&] { return hipMemcpyAsync(pinned, d2, 64, hipMemcpyDeviceToHost, b); }},
        {"hipMemcpyAsync D2H pageable on B", [&] { return hipMemcpyAsync(pageable, d2, 64, hipMemcpyDeviceToHost, b); }},
        {"hipStreamSynchronize(B)", [&] { return hipStreamSynchronize(b); }},
        {"hipEventRecord(e0, B)", [&] { return hipEventRecord(e0, b); }},
        {"hipEventSynchronize(e0)", [&] { hipEventRecord(e0, b); return hipEventSynchronize(e0); }},
        {"hipEventElapsedTime", [&] { hipEventRecord(e0, b); hipEventRecord(e1, b); hipEventSynchronize(e1); float ms; return hipEventElapsedTime(&ms, e0, e1); }},
        {"hipEventQuery(e0)", [&] { hipEventRecord(e0, b); hipError_t r = hipEventQuery(e0); return r == hipErrorNotReady ? hipSuccess : r; }},
        {"hipStreamQuery(B)", [&] { hipError_t r = hipStreamQuery(b); return r == hipErrorNotReady ? hipSuccess : r; }},
        {"hipStreamIsCapturing(B)", [&] { hipStreamCaptureStatus s; return hipStreamIsCapturing(b, &s); }},
        {"hipEventCreate/Destroy", [&] { hipEvent_t e; hipError_t r = hipEventCreate(&e); if (r == hipSuccess) r = hipEventDestroy(e); return r; }},
        {"hipMalloc/hipFree", [&] { void *p; hipError_t r = hipMalloc(&p, 4096); if (r == hipSuccess) r = hipFree(p); return r; }},
        {"hipHostMalloc/hipHostFree", [&] { void *p; hipError_t r = hipHostMalloc(&p, 4096, 0); if (r == hipSuccess) r = hipHostFree(p); return r; }},
        {"hipMemcpy (sync) D2H", [&] { return hipMemcpy(pageable, d2, 64, hipMemcpyDeviceToHost); }},
        {"hipDeviceSynchronize", [&] { return hipDeviceSynchronize(); }},
        {"hipGetDevice/hipGetDeviceProperties", [&] { int dv; hipGetDevice(&dv); hipDeviceProp_t p; return hipGetDeviceProperties(&p, dv); }},
        {"hipFuncGetAttributes", [&] { hipFuncAttributes at; return hipFuncGetAttributes(&at, (const void *)k); }},
    };
    for (int relaxed = 0; relaxed < 2; ++relaxed)
    for (auto &t : tests) {
        hipDeviceSynchronize();
        (void)hipGetLastError();
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(a, hipStreamCaptureModeGlobal) != hipSuccess) { printf("begin failed\n"); return 1; }
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, a, d);
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        if (relaxed)
            hipThreadExchangeStreamCaptureMode(&mode); // this thread: unsafe calls allowed; `mode` now holds the previous mode
        const hipError_t r = t.fn();
        if (relaxed)
            hipThreadExchangeStreamCaptureMode(&mode);
        (void)hipGetLastError();
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, a, d);
        const hipError_t e = hipStreamEndCapture(a, &g);
        (void)hipGetLastError();
        printf("%s%-40s rc=%-3d (%s)   capture %s (%d)\n", relaxed ? "[thread mode relaxed] " : "", t.name, (int)r, hipGetErrorName(r), e == hipSuccess ? "SURVIVED" : "INVALIDATED", (int)e);
        if (g) hipGraphDestroy(g);
    }
    return 0;
}
