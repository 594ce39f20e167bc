// Host -> device upload of a caller's PAGEABLE buffer (the reference's ccd() takes host matrices, ccd.cu:103-106), four ways:
//  A  hipMemcpy from the pageable buffer (what the runtime does by itself)
//  B  hipHostRegister + hipMemcpyAsync + hipHostUnregister
//  C  T threads copy chunks into a ring of pinned slots, each slot sent with hipMemcpyAsync as soon as it is full
//  D  the device reads the (registered) host buffer directly through a kernel -- zero copy
// build: hipcc --offload-arch=gfx950 -O2 -pthread tools/probe/upload_probe.hip -o /tmp/upload_probe ; run: /tmp/upload_probe [MB]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void pull_k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 48) << 20;
    char* src = (char*)malloc(bytes);
    for (size_t i = 0; i < bytes; i += 4096) src[i] = (char)i; // touched
    memset(src, 3, bytes);
    char* dst;
    CK(hipMalloc(&dst, bytes));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 4; rep++) {
        double t0 = now_ms();
        CK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        double t1 = now_ms();
        if (rep) printf("A pageable hipMemcpy           %7.3f ms  %6.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e6);
    }
    for (int rep = 0; rep < 4; rep++) {
        double t0 = now_ms();
        CK(hipHostRegister(src, bytes, hipHostRegisterDefault));
        double t1 = now_ms();
        CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t2 = now_ms();
        CK(hipHostUnregister(src));
        double t3 = now_ms();
        if (rep) printf("B register %.3f + copy %.3f + unregister %.3f = %7.3f ms\n", t1 - t0, t2 - t1, t3 - t2, t3 - t0);
    }
    for (size_t slot_mb : { 1, 2, 4 })
        for (int T : { 1, 2, 4, 8 }) {
            const size_t slot = slot_mb << 20;
            const int n_slots = 8;
            char* pin;
            CK(hipHostMalloc(&pin, slot * n_slots, hipHostMallocDefault));
            hipEvent_t ev[8];
            for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            double best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                double t0 = now_ms();
                const size_t n_chunks = (bytes + slot - 1) / slot;
                // T copy threads take chunks in turn; the main thread sends them in order
                std::vector<std::atomic<int>> ready(n_chunks);
                for (auto& r : ready) r.store(0);
                std::atomic<size_t> sent { 0 };
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++)
                    th.emplace_back([&, t] {
                        for (size_t c = t; c < n_chunks; c += T) {
                            while (c >= sent.load(std::memory_order_acquire) + n_slots) { } // the slot is still in flight
                            const size_t off = c * slot, len = std::min(slot, bytes - off);
                            memcpy(pin + (c % n_slots) * slot, src + off, len);
                            ready[c].store(1, std::memory_order_release);
                        }
                    });
                size_t issued = 0, done = 0;
                auto advance_done = [&] { // a slot may be refilled once the copy out of it has completed
                    while (done < issued && hipEventQuery(ev[done % n_slots]) == hipSuccess) sent.store(++done, std::memory_order_release);
                };
                for (size_t c = 0; c < n_chunks; c++) {
                    while (!ready[c].load(std::memory_order_acquire)) advance_done();
                    const size_t off = c * slot, len = std::min(slot, bytes - off);
                    CK(hipMemcpyAsync(dst + off, pin + (c % n_slots) * slot, len, hipMemcpyHostToDevice, s));
                    CK(hipEventRecord(ev[c % n_slots], s));
                    issued = c + 1;
                    advance_done();
                }
                CK(hipStreamSynchronize(s));
                sent.store(n_chunks + n_slots);
                for (auto& t : th) t.join();
                double t1 = now_ms();
                if (rep) best = std::min(best, t1 - t0);
            }
            printf("C ring of %d x %zu MB pinned, %d copy threads  %7.3f ms  %6.1f GB/s\n", n_slots, slot_mb, T, best, bytes / best / 1e6);
            for (auto& e : ev) CK(hipEventDestroy(e));
            CK(hipHostFree(pin));
        }
    {
        CK(hipHostRegister(src, bytes, hipHostRegisterMapped));
        void* dsrc;
        CK(hipHostGetDevicePointer(&dsrc, src, 0));
        for (int rep = 0; rep < 4; rep++) {
            double t0 = now_ms();
            hipLaunchKernelGGL(pull_k, dim3(1024), dim3(256), 0, s, (const uint4*)dsrc, (uint4*)dst, bytes / 16);
            CK(hipStreamSynchronize(s));
            double t1 = now_ms();
            if (rep) printf("D kernel pulls from registered host memory  %7.3f ms  %6.1f GB/s (registration not counted)\n", t1 - t0, bytes / (t1 - t0) / 1e6);
        }
        CK(hipHostUnregister(src));
    }
    return 0;
}
