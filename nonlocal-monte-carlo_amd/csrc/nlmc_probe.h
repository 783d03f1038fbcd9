// nlmc_probe.h -- the floor of a level-synchronous step, measured on the device the library runs on.
//
// A level of the sweep kernels is: workgroup barrier -> gather the neighbour spins from LDS (8 byte reads per lane, all in
// flight together) -> sum -> decide -> one LDS write -> barrier.  No level can take less than that dependent chain at the
// workgroup's number of waves, whatever else the kernel does; k_level_round_probe runs exactly that chain and nothing else
// (no schedule loads, no random numbers, no energy), with the gather addresses either RANDOM (bank conflicts as in a sweep) or
// conflict-free (the best any placement of row entries could reach).  bench.py prices the sweep kernel against it:
// roofline.peak = levels per second at the conflict-free round.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void k_level_round_probe(int rounds, int conflict_free, int lds_bytes, int *sink)
{
    extern __shared__ __align__(16) unsigned char probe_lds[];
    const int tid = threadIdx.x;
    const unsigned span = (unsigned)lds_bytes / 2u, msk = span - 1u;   // spins in [0, span), written bytes in [span, 2 span); span = 2^k
    for (int i = tid; i < lds_bytes; i += blockDim.x) probe_lds[i] = (unsigned char)((i * 7) & 1 ? 1 : 0xFF);
    unsigned a8[8];
    unsigned s = 12345u + 2654435761u * (unsigned)tid;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        s = s * 1664525u + 1013904223u;
        a8[q] = conflict_free ? (unsigned)((tid & 31) * 4 + q * 132 + ((tid >> 5) & 1)) : (s >> 12);
    }
    const unsigned step = conflict_free ? 4u : 1u, wbase = conflict_free ? (unsigned)((tid & 63) * 4) : (unsigned)(tid * 13);
    __syncthreads();
    int acc = 0;
    for (int l = 0; l < rounds; ++l) {
        int x = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) x += (signed char)probe_lds[(a8[q] + step * (unsigned)l) & msk];
        probe_lds[span + ((wbase + step * (unsigned)l) & msk)] = (unsigned char)(x > 0 ? 1 : 0xFF);
        acc += x;
        __syncthreads();
    }
    if (acc == 0x7fffffff) sink[0] = acc;                        // keeps the loop alive
}
