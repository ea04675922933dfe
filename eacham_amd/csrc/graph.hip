// graph.hip — view-graph query on the CSR match graph.
//
// Replaces Graph::GetBestPairForValid (/root/reference/modules/sfm/data/Graph.h:59-106) on the wire
// format the matcher emits (pairs, counts, offsets, q, t): pair p with counts[p] > 0 is the factor
// f1 -> f2 with matches q -> t and the factor f2 -> f1 with t -> q (Graph::Connect both ways,
// apps/sfm/main.cpp:144-145). No unordered_map round trip: K1 counts, per directed factor, the matches
// whose keypoint already has a (not two-view) 3-D point — a gather over the edge list, one workgroup per
// pair; K2 picks the best (valid node, not-yet-valid neighbour). The reference replaces its best
// unless `bestScore > count`, walking nodes in ascending id, so among equal counts the last visited
// wins: with neighbours taken in ascending id (the reference's unordered_map has no order) that is
// the lexicographic maximum of (count, node, neighbour) — an order-free reduction.
#include "context.hpp"

namespace eacham {

namespace {

constexpr int GT = 256;

__global__ __launch_bounds__(GT) void graph_edge_counts_kernel(const int2* __restrict__ pairs, const int* __restrict__ counts,
                                                               const long long* __restrict__ offsets,
                                                               const unsigned* __restrict__ q, const unsigned* __restrict__ t,
                                                               const long long* __restrict__ kp_offsets,
                                                               const unsigned char* __restrict__ has3d,
                                                               unsigned* __restrict__ edge_counts) {
    __shared__ unsigned s12[GT / 64], s21[GT / 64];
    const int p = blockIdx.x;
    const int2 pr = pairs[p];
    const long long o = offsets[p], b1 = kp_offsets[pr.x], b2 = kp_offsets[pr.y];
    unsigned c12 = 0, c21 = 0;
    for (int k = threadIdx.x; k < counts[p]; k += GT) {
        c12 += has3d[b1 + q[o + k]];
        c21 += has3d[b2 + t[o + k]];
    }
    for (int off = 32; off > 0; off >>= 1) {
        c12 += __shfl_xor(c12, off);
        c21 += __shfl_xor(c21, off);
    }
    if ((threadIdx.x & 63) == 0) {
        s12[threadIdx.x >> 6] = c12;
        s21[threadIdx.x >> 6] = c21;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned a = 0, b = 0;
        for (int w = 0; w < GT / 64; ++w) {
            a += s12[w];
            b += s21[w];
        }
        edge_counts[2 * p] = a;
        edge_counts[2 * p + 1] = b;
    }
}

// key = (count, node, neighbour) packed for a lexicographic max; 0 = no candidate
__device__ __forceinline__ unsigned long long cand_key(unsigned count, unsigned node, unsigned other) {
    return ((unsigned long long)count << 42) | ((unsigned long long)node << 21) | (unsigned long long)other | (1ull << 63);
}

__global__ __launch_bounds__(1024) void graph_best_pair_kernel(const int2* __restrict__ pairs, int npairs,
                                                               const int* __restrict__ counts,
                                                               const unsigned* __restrict__ edge_counts,
                                                               const unsigned char* __restrict__ valid,
                                                               const unsigned char* __restrict__ excluded,
                                                               unsigned* __restrict__ best) {
    __shared__ unsigned long long sm[1024 / 64];
    unsigned long long k = 0;
    for (int p = threadIdx.x; p < npairs; p += 1024) {
        if (counts[p] <= 0) continue;
        const int2 pr = pairs[p];
        const bool v1 = valid[pr.x], v2 = valid[pr.y];
        if (v1 && !v2 && !(excluded && excluded[pr.y])) {
            const unsigned long long c = cand_key(edge_counts[2 * p], pr.x, pr.y);
            k = c > k ? c : k;
        }
        if (v2 && !v1 && !(excluded && excluded[pr.x])) {
            const unsigned long long c = cand_key(edge_counts[2 * p + 1], pr.y, pr.x);
            k = c > k ? c : k;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(k, off);
        k = o > k ? o : k;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 1024 / 64; ++w) k = sm[w] > k ? sm[w] : k;
        if (k) {
            best[0] = (unsigned)((k >> 21) & 0x1fffffu);
            best[1] = (unsigned)(k & 0x1fffffu);
            best[2] = (unsigned)((k >> 42) & 0x1fffffu);
        } else {
            best[0] = best[1] = 0xffffffffu;
            best[2] = 0;
        }
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace
}  // namespace eacham

using namespace eacham;

extern "C" int eacham_graph_best_pair(eacham_ctx* ctx, int n_frames, const int32_t* pairs, int npairs, const int32_t* counts,
                                      const int64_t* offsets, const uint32_t* q, const uint32_t* t, const uint8_t* valid,
                                      const uint8_t* excluded, const int64_t* kp_offsets, const uint8_t* kp_has3d,
                                      uint32_t* edge_counts, uint32_t* best) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_frames < 0 || npairs < 0 || !best || (n_frames > 0 && (!valid || !kp_offsets)) ||
        (npairs > 0 && (!pairs || !counts || !offsets)))
        return ctx->fail(EACHAM_ERR_INVALID, "graph_best_pair: null argument or negative size");
    best[0] = best[1] = 0xffffffffu;
    best[2] = 0;
    if (npairs == 0) return EACHAM_OK;
    if (n_frames >= (1 << 21)) return ctx->fail(EACHAM_ERR_CAPACITY, "graph_best_pair: at most 2^21 frames");
    long long n_edges = 0;
    for (int p = 0; p < npairs; ++p) {
        const int f1 = pairs[2 * p], f2 = pairs[2 * p + 1];
        if (f1 < 0 || f2 < 0 || f1 >= n_frames || f2 >= n_frames)
            return ctx->fail(EACHAM_ERR_INVALID, "graph_best_pair: pair %d names frame %d/%d of %d", p, f1, f2, n_frames);
        if (counts[p] < 0 || offsets[p] < 0) return ctx->fail(EACHAM_ERR_INVALID, "graph_best_pair: negative count/offset at pair %d", p);
        if (counts[p] >= (1 << 21)) return ctx->fail(EACHAM_ERR_CAPACITY, "graph_best_pair: more than 2^21 matches in pair %d", p);
        n_edges = std::max<long long>(n_edges, offsets[p] + counts[p]);
    }
    if (n_edges > 0 && (!q || !t || !kp_has3d)) return ctx->fail(EACHAM_ERR_INVALID, "graph_best_pair: null edge arrays");
    for (int f = 0; f < n_frames; ++f)
        if (kp_offsets[f + 1] < kp_offsets[f] || kp_offsets[f] < 0) return ctx->fail(EACHAM_ERR_INVALID, "graph_best_pair: kp_offsets not monotone");
    const long long n_kp = n_frames > 0 ? kp_offsets[n_frames] : 0;
    for (int p = 0; p < npairs; ++p) {  // keypoint indices must stay inside their frame: checked here, not in the kernel
        const long long n1 = kp_offsets[pairs[2 * p] + 1] - kp_offsets[pairs[2 * p]];
        const long long n2 = kp_offsets[pairs[2 * p + 1] + 1] - kp_offsets[pairs[2 * p + 1]];
        for (long long k = offsets[p]; k < offsets[p] + counts[p]; ++k)
            if (q[k] >= n1 || t[k] >= n2) return ctx->fail(EACHAM_ERR_INVALID, "graph_best_pair: match %lld of pair %d is out of range", k, p);
    }
    EACHAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    const size_t o_pairs = take(sizeof(int) * 2 * (size_t)npairs), o_cnt = take(sizeof(int) * (size_t)npairs);
    const size_t o_off = take(sizeof(long long) * (size_t)npairs), o_q = take(sizeof(unsigned) * (size_t)n_edges);
    const size_t o_t = take(sizeof(unsigned) * (size_t)n_edges), o_valid = take((size_t)n_frames), o_excl = take((size_t)n_frames);
    const size_t o_kpo = take(sizeof(long long) * ((size_t)n_frames + 1)), o_h3 = take((size_t)n_kp);
    const size_t o_ec = take(sizeof(unsigned) * 2 * (size_t)npairs), o_best = take(sizeof(unsigned) * 4);
    if (int rc = ensure_io(ctx, off)) return rc;
    char* base = (char*)ctx->io;
    hipStream_t st = ctx->stream;
    auto up = [&](size_t o, const void* src, size_t bytes) {
        return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess;
    };
    EACHAM_HIP_TRY(ctx, up(o_pairs, pairs, sizeof(int) * 2 * (size_t)npairs));
    EACHAM_HIP_TRY(ctx, up(o_cnt, counts, sizeof(int) * (size_t)npairs));
    EACHAM_HIP_TRY(ctx, up(o_off, offsets, sizeof(long long) * (size_t)npairs));
    EACHAM_HIP_TRY(ctx, up(o_q, q, sizeof(unsigned) * (size_t)n_edges));
    EACHAM_HIP_TRY(ctx, up(o_t, t, sizeof(unsigned) * (size_t)n_edges));
    EACHAM_HIP_TRY(ctx, up(o_valid, valid, (size_t)n_frames));
    if (excluded) EACHAM_HIP_TRY(ctx, up(o_excl, excluded, (size_t)n_frames));
    EACHAM_HIP_TRY(ctx, up(o_kpo, kp_offsets, sizeof(long long) * ((size_t)n_frames + 1)));
    EACHAM_HIP_TRY(ctx, up(o_h3, kp_has3d, (size_t)n_kp));
    graph_edge_counts_kernel<<<npairs, GT, 0, st>>>((const int2*)(base + o_pairs), (const int*)(base + o_cnt),
                                                     (const long long*)(base + o_off), (const unsigned*)(base + o_q),
                                                     (const unsigned*)(base + o_t), (const long long*)(base + o_kpo),
                                                     (const unsigned char*)(base + o_h3), (unsigned*)(base + o_ec));
    graph_best_pair_kernel<<<1, 1024, 0, st>>>((const int2*)(base + o_pairs), npairs, (const int*)(base + o_cnt),
                                               (const unsigned*)(base + o_ec), (const unsigned char*)(base + o_valid),
                                               excluded ? (const unsigned char*)(base + o_excl) : nullptr,
                                               (unsigned*)(base + o_best));
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(best, base + o_best, sizeof(unsigned) * 3, hipMemcpyDeviceToHost, st));
    if (edge_counts)
        EACHAM_HIP_TRY(ctx, hipMemcpyAsync(edge_counts, base + o_ec, sizeof(unsigned) * 2 * (size_t)npairs, hipMemcpyDeviceToHost, st));
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(st));
    return EACHAM_OK;
}
