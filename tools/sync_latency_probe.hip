// tools/sync_latency_probe.hip — how long after a kernel ends does the host notice?  hipStreamSynchronize (the runtime's wait) against spinning on
// hipEventQuery / hipStreamQuery.  A kernel parks the stream for `us` microseconds (s_memrealtime); the host clock brackets launch .. return.
// usage: tools/sync_latency_probe [us=2000]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_park(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const unsigned us = argc > 1 ? (unsigned)atoi(argv[1]) : 2000u;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int mode = 0; mode < 3; ++mode) {
        double best = 1e9, sum = 0; const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            const double t0 = now_us();
            hipLaunchKernelGGL(k_park, dim3(1), dim3(64), 0, st, (unsigned long long)us * 100ull);
            if (mode == 0) hipStreamSynchronize(st);
            else if (mode == 1) { hipEventRecord(ev, st); while (hipEventQuery(ev) == hipErrorNotReady) {} }
            else { while (hipStreamQuery(st) == hipErrorNotReady) {} }
            const double dt = now_us() - t0 - us;
            if (r >= 2) { sum += dt; if (dt < best) best = dt; }
        }
        printf("%-40s launch .. return minus the kernel's %u us: avg %.1f us, best %.1f us\n", mode == 0 ? "hipStreamSynchronize" : (mode == 1 ? "hipEventRecord + spin on hipEventQuery" : "spin on hipStreamQuery"), us, sum / reps, best);
    }
    return 0;
}
