// Probe of global_load_lds_dwordx4 on gfx950: per-lane source, LDS destination = uniform base + lane*16,
// behaviour of inactive lanes.  hipcc --offload-arch=gfx950 -O2 glds_probe.hip -o glds_probe && ./glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const double* __restrict__ src, const int* __restrict__ perm, int n_active, double* __restrict__ out)
{
    __shared__ double2 buf[3 * 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 3 * 64; i += 64) buf[i] = make_double2(-1.0, -1.0);
    __syncthreads();
    if (lane < n_active) {
        const char* rec = reinterpret_cast<const char*>(src) + (size_t)perm[lane] * 48;
#pragma unroll
        for (int p = 0; p < 3; p++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + p * 16),
                                             (__attribute__((address_space(3))) void*)(buf + p * 64), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 3 * 64; i += 64) {
        out[2 * i] = buf[i].x;
        out[2 * i + 1] = buf[i].y;
    }
}

int main()
{
    const int nrec = 1000;
    std::vector<double> h(nrec * 6);
    for (int i = 0; i < nrec * 6; i++) h[i] = i;
    std::vector<int> perm(64);
    for (int i = 0; i < 64; i++) perm[i] = (i * 37 + 11) % nrec;
    double *d_src, *d_out;
    int* d_perm;
    (void)hipMalloc(&d_src, h.size() * 8);
    (void)hipMalloc(&d_out, 3 * 64 * 16);
    (void)hipMalloc(&d_perm, 64 * 4);
    (void)hipMemcpy(d_src, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_perm, perm.data(), 64 * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int n_active : { 64, 24, 1 }) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_src, d_perm, n_active, d_out);
        std::vector<double> o(3 * 64 * 2);
        (void)hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
        for (int p = 0; p < 3; p++)
            for (int l = 0; l < 64; l++) {
                const double want0 = l < n_active ? perm[l] * 6 + p * 2 : -1.0, want1 = l < n_active ? want0 + 1 : -1.0;
                if (o[2 * (p * 64 + l)] != want0 || o[2 * (p * 64 + l) + 1] != want1) {
                    if (bad < 10) printf("n_active %d part %d lane %d: got (%g,%g) want (%g,%g)\n", n_active, p, l, o[2 * (p * 64 + l)], o[2 * (p * 64 + l) + 1], want0, want1);
                    bad++;
                }
            }
    }
    printf(bad ? "GLDS PROBE: %d mismatches\n" : "GLDS PROBE: ok (dest = base + lane*16, inactive lanes untouched)\n", bad);
    return bad != 0;
}
