// Latency of "kernel B on stream 2 starts when kernel A on stream 1 has ended", three ways, measured on the DEVICE clock
// (wall_clock64 at A's end and at B's start; no profiler attached):
//  0  same stream (the floor: back-to-back dispatch)
//  1  hipEventRecord(s1) + hipStreamWaitEvent(s2)            (what ccd() does between its two streams)
//  2  A's last wave sets a device word; a one-wave kernel on s2, enqueued beforehand, polls it (bounded) and B follows it on s2
//  3  same stream, with a hipEventRecord between A and B (what a stream pays for being waited on: the marker packet between its kernels)
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/xqueue_probe.hip -o /tmp/xqueue_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void a_k(unsigned long long* t_end, unsigned* done, unsigned* flag, unsigned seq, int spin)
{
    unsigned long long a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 6364136223846793005ull + 1442695040888963407ull;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (a == 12345ull) t_end[1] = a;
        const unsigned n = atomicAdd(done, 1u);
        if (n == gridDim.x - 1) { // the last block of the launch
            *done = 0;
            t_end[0] = wall_clock64();
            if (flag) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ void poll_k(const unsigned* flag, unsigned seq, unsigned* gave_up)
{
    if (threadIdx.x != 0) return;
    for (int i = 0; i < 2000000; i++) { // bounded: ~1 s at worst, never a hang
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == seq) return;
        __builtin_amdgcn_s_sleep(4);
    }
    *gave_up = 1;
}
__global__ void b_k(unsigned long long* t_start)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *t_start = wall_clock64();
}
int main()
{
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned long long *t_end, *t_start;
    unsigned *done, *flag, *gave_up;
    CK(hipMalloc(&t_end, 64)); CK(hipMalloc(&t_start, 64)); CK(hipMalloc(&done, 64)); CK(hipMalloc(&flag, 64)); CK(hipMalloc(&gave_up, 64));
    CK(hipMemset(done, 0, 64)); CK(hipMemset(flag, 0, 64)); CK(hipMemset(gave_up, 0, 64));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int N = 300;
    for (int mode = 0; mode < 4; mode++) {
        std::vector<double> gap;
        for (int i = 1; i <= N; i++) {
            if (mode == 2) hipLaunchKernelGGL(poll_k, dim3(1), dim3(64), 0, s2, flag, (unsigned)i, gave_up); // enqueued BEFORE A, as a step would
            hipLaunchKernelGGL(a_k, dim3(512), dim3(256), 0, s1, t_end, done, mode == 2 ? flag : nullptr, (unsigned)i, 20000);
            if (mode == 0) hipLaunchKernelGGL(b_k, dim3(512), dim3(256), 0, s1, t_start);
            else if (mode == 3) {
                CK(hipEventRecord(ev, s1));
                hipLaunchKernelGGL(b_k, dim3(512), dim3(256), 0, s1, t_start);
            }
            else {
                if (mode == 1) { CK(hipEventRecord(ev, s1)); CK(hipStreamWaitEvent(s2, ev, 0)); }
                hipLaunchKernelGGL(b_k, dim3(512), dim3(256), 0, s2, t_start);
            }
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            unsigned long long e, s;
            CK(hipMemcpy(&e, t_end, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&s, t_start, 8, hipMemcpyDeviceToHost));
            if (i > 20) gap.push_back(((double)s - (double)e) / 100.0); // wall_clock64: 100 MHz
        }
        std::sort(gap.begin(), gap.end());
        unsigned g = 0; CK(hipMemcpy(&g, gave_up, 4, hipMemcpyDeviceToHost));
        printf("mode %d: A's end -> B's start  p10 %.2f  p50 %.2f  p90 %.2f us   (poll gave up: %u)\n", mode, gap[gap.size() / 10], gap[gap.size() / 2], gap[gap.size() * 9 / 10], g);
    }
    return 0;
}
