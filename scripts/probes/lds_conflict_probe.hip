// How gfx950's LDS serves a wave64 ds_read_u8 gather by the bank pattern of its 64 addresses (ns per round of 8 gathers + barrier, 16
// waves, HIP events):  hipcc --offload-arch=gfx950 -O3 -o lds_conflict_probe scripts/probes/lds_conflict_probe.hip && ./lds_conflict_probe
//   mode 0: lanes l and l+32 read bytes of the SAME dword, 32 distinct banks                (conflict-free)
//   mode 1: lane l bank l%32; lanes l and l+32 DIFFERENT dwords of the same bank            (2-way across the two halves of a wave)
//   mode 2: 2-way inside each half (16 banks used per half), halves on disjoint banks
//   mode 3: 4-way inside each half
//   mode 4: random addresses
//   mode 5: 64 distinct banks (bank = lane, if the LDS has 64 banks this is conflict-free and mode 1 is too)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(int rounds, int mode, int *sink)
{
    extern __shared__ unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536; i += blockDim.x) lds[i] = (unsigned char)(i * 7);
    unsigned a8[8];
    unsigned s = 12345u + 2654435761u * (unsigned)tid;
    for (int q = 0; q < 8; ++q) {
        s = s * 1664525u + 1013904223u;
        const unsigned row = (unsigned)q * 1024u;            // a different 1 KB row per gather slot
        unsigned a;
        switch (mode) {
        case 0: a = (lane & 31) * 4 + (lane >> 5); break;
        case 1: a = (lane & 31) * 4 + (lane >> 5) * 128; break;
        case 2: a = ((lane & 15) + (lane >> 5) * 16) * 4 + ((lane >> 4) & 1) * 128; break;
        case 3: a = ((lane & 7) + (lane >> 5) * 8) * 4 + ((lane >> 3) & 3) * 128; break;
        case 5: a = lane * 4; break;
        default: a = (s >> 12) & 1023u; break;
        }
        a8[q] = row + (a & 1023u);
    }
    __syncthreads();
    int acc = 0;
    for (int l = 0; l < rounds; ++l) {
        int x = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) x += (signed char)lds[(a8[q] + 8192u * (unsigned)(l & 3)) & 65535u];
        acc += x;
        __syncthreads();
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}
int main()
{
    int *sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 20000;
    for (int waves : {1, 16})
        for (int mode = 0; mode < 6; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 65536, 0, 256, mode, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 65536, 0, rounds, mode, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("waves %2d mode %d: %.1f ns per round\n", waves, mode, ms * 1e6 / rounds);
        }
    return 0;
}
