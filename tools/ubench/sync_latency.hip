// how long does the host take to see the end of a short kernel: hipStreamSynchronize vs hipEventSynchronize vs spinning on a word the
// kernel writes into pinned host memory.   hipcc --offload-arch=gfx950 -O2 -o sync_latency sync_latency.hip && ./sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void work_kernel(volatile int* flag, int seq, int spin)
{
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = seq; }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* flag; hipHostMalloc((void**)&flag, 64, hipHostMallocDefault); *flag = 0;
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const int spin = 2000 * 20;      // ~20 us of kernel at ~2 GHz
    for (int mode = 0; mode < 3; mode++) {
        std::vector<double> t;
        for (int i = 1; i <= 300; i++) {
            const double t0 = now();
            hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, flag, mode * 1000 + i, spin);
            if (mode == 0) hipStreamSynchronize(s);
            else if (mode == 1) { hipEventRecord(ev, s); hipEventSynchronize(ev); }
            else { while (*(volatile int*)flag != mode * 1000 + i) {} }
            t.push_back(now() - t0);
            if (mode == 2) hipStreamSynchronize(s);
        }
        std::sort(t.begin(), t.end());
        printf("%s: median %.1f us, p10 %.1f, p90 %.1f (launch + ~20 us kernel + wait)\n", mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "event record + sync" : "spin on pinned word", t[150], t[30], t[270]);
    }
    return 0;
}
