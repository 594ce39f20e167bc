// How fast does the vendor library sort 16M (u32 key, u32 value) pairs on this GPU?  (Measurement only:
// the product uses its own onesweep, csrc/sort.hip.)  hipcc --offload-arch=gfx950 -O3 rocprim_sort_probe.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>

int main()
{
    const size_t n = 16000000;
    std::vector<unsigned> h(n);
    std::mt19937 g(1);
    for (auto& x : h) x = g() & 0x7FFFFFFFu;
    unsigned *k0, *k1, *v0, *v1;
    (void)hipMalloc(&k0, n * 4); (void)hipMalloc(&k1, n * 4); (void)hipMalloc(&v0, n * 4); (void)hipMalloc(&v1, n * 4);
    (void)hipMemcpy(k0, h.data(), n * 4, hipMemcpyHostToDevice);
    size_t tmp_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, 32);
    void* tmp;
    (void)hipMalloc(&tmp, tmp_bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int bits : { 32, 24 }) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            (void)hipEventRecord(a);
            (void)rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            if (rep) best = ms < best ? ms : best;
        }
        printf("rocprim::radix_sort_pairs 16M pairs, %d key bits: %.3f ms\n", bits, best);
    }
    return 0;
}
