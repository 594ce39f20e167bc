// Latency of a small device -> host read between two dependent kernels, three ways:
//  A  hipMemcpyAsync into pinned memory + polled event   (what ReadBack does)
//  B  a one-wave kernel that stores into host-mapped memory + a sequence word the host spins on
//  C  the producing kernel stores into host-mapped memory itself (last thread), host spins
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/readback_probe.hip -o /tmp/readback_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void work_k(unsigned long long* out, unsigned long long v, int spin)
{
    unsigned long long a = v;
    for (int i = 0; i < spin; i++) a = a * 6364136223846793005ull + 1442695040888963407ull;
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = a | 1ull;
}
__global__ void publish_k(const unsigned long long* src, volatile unsigned long long* host, unsigned long long seq)
{
    if (threadIdx.x == 0) {
        host[0] = *src;
        __threadfence_system();
        host[1] = seq;
    }
}
__global__ void work_publish_k(unsigned long long* out, unsigned long long v, int spin, volatile unsigned long long* host, unsigned long long seq)
{
    unsigned long long a = v;
    for (int i = 0; i < spin; i++) a = a * 6364136223846793005ull + 1442695040888963407ull;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *out = a | 1ull;
        host[0] = a | 1ull;
        __threadfence_system();
        host[1] = seq;
    }
}
int main()
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *d, *h_pinned, *h_mapped, *d_mapped;
    CK(hipMalloc(&d, 64));
    CK(hipHostMalloc(&h_pinned, 64, hipHostMallocDefault));
    CK(hipHostMalloc(&h_mapped, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_mapped, h_mapped, 0));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int N = 2000, spin = 2000;
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            h_mapped[1] = 0;
            auto t0 = std::chrono::steady_clock::now();
            unsigned long long v = 1;
            for (int i = 1; i <= N; i++) {
                if (mode == 0) {
                    hipLaunchKernelGGL(work_k, dim3(256), dim3(256), 0, s, d, v, spin);
                    CK(hipMemcpyAsync(h_pinned, d, 8, hipMemcpyDeviceToHost, s));
                    CK(hipEventRecord(ev, s));
                    for (;;) { hipError_t e = hipEventQuery(ev); if (e == hipSuccess) break; if (e != hipErrorNotReady) CK(e); }
                    v = h_pinned[0];
                } else if (mode == 1) {
                    hipLaunchKernelGGL(work_k, dim3(256), dim3(256), 0, s, d, v, spin);
                    hipLaunchKernelGGL(publish_k, dim3(1), dim3(64), 0, s, d, d_mapped, (unsigned long long)i);
                    while (((volatile unsigned long long*)h_mapped)[1] != (unsigned long long)i) { }
                    v = ((volatile unsigned long long*)h_mapped)[0];
                } else {
                    hipLaunchKernelGGL(work_publish_k, dim3(256), dim3(256), 0, s, d, v, spin, d_mapped, (unsigned long long)i);
                    while (((volatile unsigned long long*)h_mapped)[1] != (unsigned long long)i) { }
                    v = ((volatile unsigned long long*)h_mapped)[0];
                }
            }
            CK(hipStreamSynchronize(s));
            auto t1 = std::chrono::steady_clock::now();
            if (rep) printf("mode %c: %.2f us per (kernel + read-back) round trip  (v=%llx)\n", "ABC"[mode], std::chrono::duration<double, std::micro>(t1 - t0).count() / N, v);
        }
    }
    // kernel alone, back to back, for scale
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(work_k, dim3(256), dim3(256), 0, s, d, 1ull, spin);
    CK(hipStreamSynchronize(s));
    auto t1 = std::chrono::steady_clock::now();
    printf("kernel alone, back to back: %.2f us each\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    return 0;
}
