// fp64 FMA issue rate of one CU on gfx950 by waves per workgroup and independent chains per wave:
//   hipcc --offload-arch=gfx950 -O3 -o fp64_issue_probe scripts/probes/fp64_issue_probe.hip && ./fp64_issue_probe
// (what bounds k_lbp_lds: ~110 fp64 instructions per message, 2 waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ __launch_bounds__(1024) void k_fma(int iters, double a, double b, double *out, long long *cyc)
{
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = 1.0 + threadIdx.x * 1e-9 + i;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int l = 0; l < iters; ++l) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) x[i] = __builtin_fma(x[i], a, b);
        }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ILP> void run(int waves, double *out, long long *cyc)
{
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_fma<ILP>, dim3(blocks), dim3(waves * 64), 0, 0, iters, 0.999999, 1e-7, out, cyc);      // warm-up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_fma<ILP>, dim3(blocks), dim3(waves * 64), 0, 0, iters, 0.999999, 1e-7, out, cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> r(blocks);
    hipMemcpy(r.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : r) m += (double)v;
    m /= blocks;
    const double insts = (double)iters * 16 * ILP * waves;           // wave-instructions per CU
    printf("waves %2d x %d chains: %.1f wave-FMAs per 100 counter ticks per CU; kernel %.1f us = %.0f ticks -> %.0f ticks per us; "
           "%.2f wave-FMAs per ns per CU (peak at 64 lanes x 2.4 GHz: 2.4)\n", waves, ILP, insts / m * 100, ms * 1e3, m, m / (ms * 1e3), insts / (ms * 1e6));
}

int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
    for (int waves : {4, 8, 16}) { run<1>(waves, out, cyc); run<2>(waves, out, cyc); run<4>(waves, out, cyc); }
    return 0;
}
