// nlmc_host.h -- host-side graph logic exported through the same C-ABI (no device work).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

// Cluster growth of find_clusters (NMC/nmc.py:257-318) on CSR neighbour lists.
// Seeds = {|mag| >= thr_init} in ascending order; a seed not yet claimed opens a cluster and claims its unclaimed seed
// neighbours (ascending); then, for thr = thr_init - step, thr - step, ... while thr > thr_cut, every cluster in turn
// claims the unclaimed neighbours of its members with |mag| >= thr (ascending).  Neighbour = stored entry with J != 0.
static int host_find_clusters(int n, const int32_t *rowptr, const int32_t *col, const double *val, const double *mag,
                              double thr_init, double thr_cut, double thr_step, int32_t *out_members,
                              int64_t members_cap, int32_t *out_sizes, int32_t *out_n_clusters)
{
    std::vector<uint8_t> in_cluster((size_t)n, 0), seed((size_t)n, 0);
    std::vector<std::vector<int32_t>> clusters;
    for (int k = 0; k < n; ++k) seed[k] = std::fabs(mag[k]) >= thr_init;
    std::vector<int32_t> nb;
    auto neighbours = [&](const std::vector<int32_t> &nodes) {
        nb.clear();
        for (int32_t k : nodes)
            for (int e = rowptr[k]; e < rowptr[k + 1]; ++e)
                if (val[e] != 0.0) nb.push_back(col[e]);
        std::sort(nb.begin(), nb.end());
        nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
    };
    std::vector<int32_t> one(1);
    for (int s = 0; s < n; ++s) {
        if (!seed[s] || in_cluster[s]) continue;
        one[0] = s;
        neighbours(one);
        std::vector<int32_t> c;
        c.push_back(s);
        for (int32_t v : nb)
            if (!in_cluster[v] && seed[v]) c.push_back(v);
        for (int32_t v : c) in_cluster[v] = 1;
        clusters.push_back(std::move(c));
    }
    double thr = thr_init - thr_step;
    while (thr > thr_cut) {
        for (auto &c : clusters) {
            neighbours(c);
            const size_t before = c.size();
            for (int32_t v : nb)
                if (!in_cluster[v] && std::fabs(mag[v]) >= thr) c.push_back(v);
            for (size_t i = before; i < c.size(); ++i) in_cluster[c[i]] = 1;
        }
        thr -= thr_step;
    }
    int64_t at = 0;
    for (size_t i = 0; i < clusters.size(); ++i) {
        if (at + (int64_t)clusters[i].size() > members_cap) return -1;
        for (int32_t v : clusters[i]) out_members[at++] = v;
        out_sizes[i] = (int32_t)clusters[i].size();
    }
    *out_n_clusters = (int32_t)clusters.size();
    return 0;
}

// Read-out layout of a recorded trace: src [n_blocks][S][N] int8 (what the sweep kernels record, one block per chain)
// -> dst [n_dst_blocks][N][row_len], block b written to rows of block dst_block[b], columns dst_col[b] .. dst_col[b]+S-1
// (the reference's M block of a replica: rows = spins, columns = sweeps, NPT/npt.py:641; sub-replica j of APT_ICM in
// columns j S .., NPT/apt_ICM.py:188,207), as int8 or as float64.  Split over host threads by (block, spin range):
// first-touch page faults of a large float64 M are the dominant cost of a single-threaded fill.
#include <thread>
template <typename T>
static void host_trace_layout_part(const int8_t *src, int64_t S, int64_t N, const int32_t *dst_block, const int32_t *dst_col,
                                   int64_t row_len, T *dst, int64_t job0, int64_t job1, int64_t tiles_per_block, int64_t tile)
{
    for (int64_t j = job0; j < job1; ++j) {
        const int64_t b = j / tiles_per_block, k0 = (j % tiles_per_block) * tile, k1 = std::min(N, k0 + tile);
        const int8_t *sb = src + b * S * N;
        T *db = dst + (int64_t)(dst_block ? dst_block[b] : b) * N * row_len + (dst_col ? dst_col[b] : 0);
        for (int64_t k = k0; k < k1; ++k) {
            T *row = db + k * row_len;
            for (int64_t t = 0; t < S; ++t) row[t] = (T)sb[t * N + k];
        }
    }
}

template <typename T>
static void host_trace_layout(const int8_t *src, int64_t n_blocks, int64_t S, int64_t N, const int32_t *dst_block,
                              const int32_t *dst_col, int64_t row_len, T *dst, int n_threads)
{
    const int64_t tile = 1024, tiles_per_block = (N + tile - 1) / tile, jobs = n_blocks * tiles_per_block;
    int64_t nt = std::max<int64_t>(1, std::min<int64_t>(n_threads, jobs));
    if (n_blocks * S * N < (int64_t)1 << 18) nt = 1;
    if (nt == 1) {
        host_trace_layout_part<T>(src, S, N, dst_block, dst_col, row_len, dst, 0, jobs, tiles_per_block, tile);
        return;
    }
    std::vector<std::thread> th;
    for (int64_t i = 0; i < nt; ++i)
        th.emplace_back(host_trace_layout_part<T>, src, S, N, dst_block, dst_col, row_len, dst, jobs * i / nt,
                        jobs * (i + 1) / nt, tiles_per_block, tile);
    for (auto &t : th) t.join();
}
