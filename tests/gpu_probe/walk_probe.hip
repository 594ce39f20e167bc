// Device-vs-host check of the stackless-walk helpers (ti_math.hpp: nq_descend / nq_backtrack / nq_donate) on random
// operation sequences reaching depths up to 93.  Build (from the repo root):
//   make tests/gpu_probe/walk_probe

#include <hip/hip_runtime.h>
#include "ti_math.hpp"
#define WP_QUAL __device__
#include "walk_probe_ops.h"
#include <cstdio>
#include <vector>
extern "C" void wp_host_generate(int n_seq, int n_ops, Op* ops, Out* href, unsigned* maxdepth);
__global__ void k(const Op* ops, int n_ops, int n_seq, Out* out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_seq) run_ops(ops + (size_t)s * n_ops, n_ops, NQDom{ 0, 0, 0, 0 }, out + (size_t)s * n_ops);
}
int main()
{
    const int n_seq = 4096, n_ops = 400;
    std::vector<Op> ops((size_t)n_seq * n_ops);
    std::vector<Out> href(ops.size()), hdev(ops.size());
    unsigned maxdepth = 0;
    wp_host_generate(n_seq, n_ops, ops.data(), href.data(), &maxdepth);
    Op* d_ops; Out* d_out;
    (void)hipMalloc(&d_ops, ops.size() * sizeof(Op));
    (void)hipMalloc(&d_out, ops.size() * sizeof(Out));
    (void)hipMemcpy(d_ops, ops.data(), ops.size() * sizeof(Op), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n_seq + 63) / 64), dim3(64), 0, 0, d_ops, n_ops, n_seq, d_out);
    (void)hipMemcpy(hdev.data(), d_out, ops.size() * sizeof(Out), hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < href.size(); i++) {
        const Out &a = href[i], &b = hdev[i];
        const bool same = a.dom.k0 == b.dom.k0 && a.dom.k1 == b.dom.k1 && a.dom.k2 == b.dom.k2 && a.dom.d == b.dom.d
            && a.w.pend.lo == b.w.pend.lo && a.w.pend.hi == b.w.pend.hi && a.w.s0.lo == b.w.s0.lo && a.w.s0.hi == b.w.s0.hi
            && a.w.s1.lo == b.w.s1.lo && a.w.s1.hi == b.w.s1.hi;
        if (!same) {
            if (bad < 5) printf("mismatch seq %zu op %zu kind %d: host dom %u %u %u %x pend %llx dev dom %u %u %u %x pend %llx\n", i / n_ops, i % n_ops, ops[i].kind,
                                a.dom.k0, a.dom.k1, a.dom.k2, a.dom.d, (unsigned long long)a.w.pend.lo, b.dom.k0, b.dom.k1, b.dom.k2, b.dom.d, (unsigned long long)b.w.pend.lo);
            bad++;
        }
    }
    printf("WALK PROBE: %ld mismatches over %zu states, max depth %u\n", bad, href.size(), maxdepth);
    return bad != 0;
}
