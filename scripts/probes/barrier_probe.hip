// Cost of one s_barrier round on gfx950 as a function of the waves of the workgroup, alone and with the ingredients of a
// level of k_sweep_fused (scalar loads of level offsets, random byte gathers from LDS, one LDS write):
//   hipcc --offload-arch=gfx950 -O3 -o barrier_probe scripts/probes/barrier_probe.hip && ./barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void k_probe(int iters, int mode, const int *offs, long long *out, const unsigned short *addr)
{
    extern __shared__ unsigned char lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 40960; i += blockDim.x) lds[i] = (unsigned char)(i * 7);
    unsigned short a8[8];
    for (int q = 0; q < 8; ++q) a8[q] = addr[(tid * 8 + q) & 65535] & 16383;
    __syncthreads();
    int acc = 0;
    typedef const int __attribute__((address_space(4))) *cptr;
    const cptr op = (cptr)(unsigned long long)offs;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int l = 0; l < iters; ++l) {
        if (mode & 1) acc += op[l & 1023];                                   // scalar load of a level offset
        if (mode & 2) {                                                      // 8 random byte gathers + sum + 1 write
            int x = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) x += (signed char)lds[(a8[q] + l) & 16383];          // random byte gathers (bank conflicts as in the sweep)
            lds[20480 + ((tid * 13 + l) & 16383)] = (unsigned char)x;
            acc += x;
        }
        if (mode & 4) {                                                      // the same gathers, bank-conflict-free addresses
            int x = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) x += (signed char)lds[(((tid & 31) * 4 + q * 132 + ((tid >> 5) & 1) + l * 4) & 16383)];
            lds[20480 + (((tid & 63) * 4 + l * 4) & 16383)] = (unsigned char)x;
            acc += x;
        }
        if (mode & 8) {                                                      // 8 random DWORD gathers
            int x = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) x += reinterpret_cast<const int *>(lds)[((a8[q] + l) & 4095)];
            lds[20480 + ((tid * 13 + l) & 16383)] = (unsigned char)x;
            acc += x;
        }
        __syncthreads();
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (tid == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
    else if (acc == 0x7fffffff) out[1] = acc;
}

int main()
{
    const int iters = 2000, blocks = 256;
    int *offs; long long *out; unsigned short *addr;
    hipMalloc(&offs, 1024 * 4); hipMalloc(&out, blocks * 16); hipMalloc(&addr, 65536 * 2);
    std::vector<int> h(1024); for (int i = 0; i < 1024; ++i) h[i] = i * 3;
    std::vector<unsigned short> ha(65536); unsigned s = 12345; for (auto &v : ha) { s = s * 1664525u + 1013904223u; v = (unsigned short)(s >> 12); }
    hipMemcpy(offs, h.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(addr, ha.data(), 65536 * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    // Wall time per round from HIP events around the launch (2 x 10^4 rounds: launch overhead and the prologue are < 1 %).  The
    // value of __builtin_readcyclecounter is printed beside it for reference only: it is NOT a core-clock cycle count on this
    // chip (ticks per microsecond varied 4x with the number of resident waves when calibrated against events).
    const int iters_t = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode : {0, 2, 4, 8})
        for (int waves : {1, 2, 4, 8, 12, 16}) {
            hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(waves * 64), 140 * 1024, 0, iters, mode, offs, out, addr);      // warm-up
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(waves * 64), 140 * 1024, 0, iters_t, mode, offs, out, addr);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> r(blocks * 2);
            hipMemcpy(r.data(), out, blocks * 16, hipMemcpyDeviceToHost);
            double m = 0; for (int b = 0; b < blocks; ++b) m += (double)r[b * 2];
            printf("mode %d (%s%s) waves %2d: %7.1f ns per barrier round (events), %.2f counter ticks\n", mode,
                   (mode & 4) ? "conflict-free gathers " : (mode & 8) ? "random dword gathers " : "", (mode & 2) ? "random byte gathers+write" : "",
                   waves, (double)ms * 1e6 / iters_t, m / blocks / iters_t);
        }
    return 0;
}
