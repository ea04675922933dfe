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

#include <algorithm>
#include <vector>

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

// ---- the resident form: the CSR match graph is validated and uploaded ONCE, the per-frame state (Node::IsValid, the keypoints that
// carry a not-two-view 3-D point) is updated frame by frame as the incremental loop changes it, a query is two small kernels.
// (The one-shot entry point below re-validates and re-uploads the whole graph per call: 3.9 ms per query on a 500-frame sequence,
// more than PnP + triangulation + bundle adjustment of the frame the query is for.)
struct eacham_graph {
    eacham_ctx* ctx = nullptr;
    int n_frames = 0, n_edges = 0;       // edges = pairs with matches
    long long n_matches = 0, n_kp = 0;
    char* dev = nullptr;
    int2* pairs = nullptr;
    int* counts = nullptr;
    long long* offsets = nullptr;
    unsigned *q = nullptr, *t = nullptr, *edge_counts = nullptr, *best = nullptr;
    unsigned char *valid = nullptr, *excluded = nullptr, *has3d = nullptr;
    long long* kp_offsets = nullptr;
    std::vector<long long> kp_offsets_h;
};

extern "C" int eacham_graph_create(eacham_ctx* ctx, int n_frames, const int32_t* pairs, int npairs, const int32_t* counts,
                                   const int64_t* offsets, const uint32_t* q, const uint32_t* t, const int64_t* kp_offsets,
                                   eacham_graph** out) {
    if (!ctx || !out) return EACHAM_ERR_INVALID;
    *out = nullptr;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_frames <= 0 || npairs < 0 || !kp_offsets || (npairs > 0 && (!pairs || !counts || !offsets)))
        return ctx->fail(EACHAM_ERR_INVALID, "graph_create: null argument or negative size");
    if (n_frames >= (1 << 21)) return ctx->fail(EACHAM_ERR_CAPACITY, "graph_create: at most 2^21 frames");
    for (int f = 0; f < n_frames; ++f)
        if (kp_offsets[f + 1] < kp_offsets[f] || kp_offsets[f] < 0) return ctx->fail(EACHAM_ERR_INVALID, "graph_create: kp_offsets not monotone");
    // the edges (pairs with matches), compacted, their match lists packed in pair order
    std::vector<int> e_pairs, e_counts;
    std::vector<long long> e_off;
    std::vector<unsigned> e_q, e_t;
    for (int p = 0; p < npairs; ++p) {
        const int f1 = pairs[2 * p], f2 = pairs[2 * p + 1];
        if (f1 < 0 || f2 < 0 || f1 >= n_frames || f2 >= n_frames)
            return ctx->fail(EACHAM_ERR_INVALID, "graph_create: pair %d names frame %d/%d of %d", p, f1, f2, n_frames);
        if (counts[p] < 0 || offsets[p] < 0) return ctx->fail(EACHAM_ERR_INVALID, "graph_create: negative count/offset at pair %d", p);
        if (counts[p] >= (1 << 21)) return ctx->fail(EACHAM_ERR_CAPACITY, "graph_create: more than 2^21 matches in pair %d", p);
        if (counts[p] == 0) continue;
        if (!q || !t) return ctx->fail(EACHAM_ERR_INVALID, "graph_create: null edge arrays");
        const long long n1 = kp_offsets[f1 + 1] - kp_offsets[f1], n2 = kp_offsets[f2 + 1] - kp_offsets[f2];
        e_pairs.push_back(f1);
        e_pairs.push_back(f2);
        e_counts.push_back(counts[p]);
        e_off.push_back((long long)e_q.size());
        for (long long k = offsets[p]; k < offsets[p] + counts[p]; ++k) {
            if (q[k] >= n1 || t[k] >= n2) return ctx->fail(EACHAM_ERR_INVALID, "graph_create: match %lld of pair %d is out of range", k, p);
            e_q.push_back(q[k]);
            e_t.push_back(t[k]);
        }
    }
    eacham_graph* g = new eacham_graph();
    g->ctx = ctx;
    g->n_frames = n_frames;
    g->n_edges = (int)e_counts.size();
    g->n_matches = (long long)e_q.size();
    g->n_kp = kp_offsets[n_frames];
    g->kp_offsets_h.assign(kp_offsets, kp_offsets + n_frames + 1);
    (void)hipSetDevice(ctx->device);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + std::max<size_t>(bytes, 8)); return o; };
    const size_t o_pairs = take(sizeof(int) * 2 * (size_t)g->n_edges), o_cnt = take(sizeof(int) * (size_t)g->n_edges);
    const size_t o_off = take(sizeof(long long) * (size_t)g->n_edges), o_q = take(sizeof(unsigned) * (size_t)g->n_matches);
    const size_t o_t = take(sizeof(unsigned) * (size_t)g->n_matches), o_valid = take((size_t)n_frames), o_excl = take((size_t)n_frames);
    const size_t o_kpo = take(sizeof(long long) * ((size_t)n_frames + 1)), o_h3 = take((size_t)g->n_kp);
    const size_t o_ec = take(sizeof(unsigned) * 2 * (size_t)g->n_edges), o_best = take(sizeof(unsigned) * 4);
    if (hipMalloc((void**)&g->dev, off) != hipSuccess) {
        delete g;
        return ctx->fail(EACHAM_ERR_HIP, "graph_create: allocating %zu bytes failed", off);
    }
    char* base = g->dev;
    g->pairs = (int2*)(base + o_pairs); g->counts = (int*)(base + o_cnt); g->offsets = (long long*)(base + o_off);
    g->q = (unsigned*)(base + o_q); g->t = (unsigned*)(base + o_t); g->valid = (unsigned char*)(base + o_valid);
    g->excluded = (unsigned char*)(base + o_excl); g->kp_offsets = (long long*)(base + o_kpo); g->has3d = (unsigned char*)(base + o_h3);
    g->edge_counts = (unsigned*)(base + o_ec); g->best = (unsigned*)(base + o_best);
    hipStream_t st = ctx->stream;
    bool ok = hipMemsetAsync(base, 0, off, st) == hipSuccess;
    auto up = [&](void* dst, const void* src, size_t bytes) { return !bytes || hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) == hipSuccess; };
    ok = ok && up(g->pairs, e_pairs.data(), sizeof(int) * e_pairs.size()) && up(g->counts, e_counts.data(), sizeof(int) * e_counts.size()) &&
         up(g->offsets, e_off.data(), sizeof(long long) * e_off.size()) && up(g->q, e_q.data(), sizeof(unsigned) * e_q.size()) &&
         up(g->t, e_t.data(), sizeof(unsigned) * e_t.size()) && up(g->kp_offsets, kp_offsets, sizeof(long long) * ((size_t)n_frames + 1));
    ok = ok && hipStreamSynchronize(st) == hipSuccess;   // (the host vectors die here)
    if (!ok) {
        (void)hipFree(g->dev);
        delete g;
        return ctx->fail(EACHAM_ERR_HIP, "graph_create: upload failed");
    }
    *out = g;
    return EACHAM_OK;
}

extern "C" void eacham_graph_destroy(eacham_graph* g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lock(g->ctx->mu);
        (void)hipSetDevice(g->ctx->device);
        (void)hipStreamSynchronize(g->ctx->stream);
        if (g->dev) (void)hipFree(g->dev);
    }
    delete g;
}

// Node::IsValid() of a frame and, per keypoint, HasPoint3d(k) && !IsPoint3dTwoView(k) (has3d == NULL: the flags stay as they are)
extern "C" int eacham_graph_set_frame(eacham_graph* g, int frame, int valid, const uint8_t* has3d, int n_keypoints) {
    if (!g) return EACHAM_ERR_INVALID;
    eacham_ctx* ctx = g->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (frame < 0 || frame >= g->n_frames) return ctx->fail(EACHAM_ERR_INVALID, "graph_set_frame: frame %d of %d", frame, g->n_frames);
    const long long nk = g->kp_offsets_h[frame + 1] - g->kp_offsets_h[frame];
    if (has3d && n_keypoints != nk) return ctx->fail(EACHAM_ERR_INVALID, "graph_set_frame: frame %d has %lld keypoints, got %d flags", frame, nk, n_keypoints);
    (void)hipSetDevice(ctx->device);
    const unsigned char v = valid ? 1 : 0;
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(g->valid + frame, &v, 1, hipMemcpyHostToDevice, ctx->stream));
    if (has3d && nk > 0) EACHAM_HIP_TRY(ctx, hipMemcpyAsync(g->has3d + g->kp_offsets_h[frame], has3d, (size_t)nk, hipMemcpyHostToDevice, ctx->stream));
    // the sources are a stack byte and the caller's buffer: both may die when this call returns, and whether the runtime has
    // staged a pageable copy by then is its business — wait (eacham_graph_set_frames, what the loop calls, goes through the pinned mirror)
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return EACHAM_OK;
}

// the flags of several frames, staged one behind the other, to their places (block = frame of the batch)
__global__ __launch_bounds__(256) void graph_scatter_frames_kernel(int n, const int* __restrict__ frames, const long long* __restrict__ src_off,
                                                                     const unsigned char* __restrict__ valid_in, const unsigned char* __restrict__ flags,
                                                                     const long long* __restrict__ kp_offsets, unsigned char* __restrict__ valid,
                                                                     unsigned char* __restrict__ has3d) {
    const int i = blockIdx.x;
    if (i >= n) return;
    const int f = frames[i];
    const long long s0 = src_off[i], cnt = src_off[i + 1] - s0, d0 = kp_offsets[f];
    for (long long k = threadIdx.x; k < cnt; k += 256) has3d[d0 + k] = flags[s0 + k];
    if (threadIdx.x == 0) valid[f] = valid_in[i];
}

extern "C" int eacham_graph_set_frames(eacham_graph* g, int n, const int32_t* frames, const uint8_t* valid, const uint8_t* has3d,
                                       const int64_t* has3d_offsets) {
    if (!g) return EACHAM_ERR_INVALID;
    eacham_ctx* ctx = g->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n < 0 || (n > 0 && (!frames || !valid || !has3d_offsets))) return ctx->fail(EACHAM_ERR_INVALID, "graph_set_frames: null argument or negative size");
    if (n == 0) return EACHAM_OK;
    if (has3d_offsets[0] != 0) return ctx->fail(EACHAM_ERR_INVALID, "graph_set_frames: has3d_offsets[0] must be 0");
    for (int i = 0; i < n; ++i) {
        if (frames[i] < 0 || frames[i] >= g->n_frames) return ctx->fail(EACHAM_ERR_INVALID, "graph_set_frames: frame %d of %d", frames[i], g->n_frames);
        const long long nk = g->kp_offsets_h[frames[i] + 1] - g->kp_offsets_h[frames[i]];
        if (has3d_offsets[i + 1] - has3d_offsets[i] != nk)
            return ctx->fail(EACHAM_ERR_INVALID, "graph_set_frames: frame %d has %lld keypoints, got %lld flags", frames[i], nk,
                             (long long)(has3d_offsets[i + 1] - has3d_offsets[i]));
    }
    const long long total = has3d_offsets[n];
    if (total > 0 && !has3d) return ctx->fail(EACHAM_ERR_INVALID, "graph_set_frames: null flags");
    EACHAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_fr = 0, o_off = align(sizeof(int) * (size_t)n), o_v = o_off + align(sizeof(long long) * ((size_t)n + 1));
    const size_t o_fl = o_v + align((size_t)n), bytes = o_fl + align((size_t)std::max<long long>(total, 1));
    if (int rc = ensure_io(ctx, bytes)) return rc;
    if (int rc = ensure_io_host(ctx, bytes)) return rc;
    IoPack io(ctx, ctx->stream);
    if (int rc = io.in(o_fr, frames, sizeof(int) * (size_t)n)) return rc;
    if (int rc = io.in(o_off, has3d_offsets, sizeof(long long) * ((size_t)n + 1))) return rc;
    if (int rc = io.in(o_v, valid, (size_t)n)) return rc;
    if (total > 0)
        if (int rc = io.in(o_fl, has3d, (size_t)total)) return rc;
    if (int rc = io.flush_in()) return rc;
    char* base = (char*)ctx->io;
    graph_scatter_frames_kernel<<<n, 256, 0, ctx->stream>>>(n, (const int*)(base + o_fr), (const long long*)(base + o_off), (const unsigned char*)(base + o_v),
                                                            (const unsigned char*)(base + o_fl), g->kp_offsets, g->valid, g->has3d);
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    // (the staging buffer is the context's: the next call that uses it is ordered behind this kernel on the same stream, and the
    // pinned mirror is rewritten only after a call that synchronised — every user of IoPack ends with finish() or is this one)
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->io_busy = false;  // (synchronised: nothing reads the mirror any more)
    return EACHAM_OK;
}

// Graph::GetBestPairForValid(excluded) on the resident state
extern "C" int eacham_graph_query(eacham_graph* g, const int32_t* excluded_frames, int n_excluded, uint32_t* best) {
    if (!g || !best || n_excluded < 0 || (n_excluded > 0 && !excluded_frames)) return EACHAM_ERR_INVALID;
    eacham_ctx* ctx = g->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    best[0] = best[1] = 0xffffffffu;
    best[2] = 0;
    if (g->n_edges == 0) return EACHAM_OK;
    std::vector<unsigned char> ex((size_t)g->n_frames, 0);
    for (int k = 0; k < n_excluded; ++k) {
        if (excluded_frames[k] < 0 || excluded_frames[k] >= g->n_frames) return ctx->fail(EACHAM_ERR_INVALID, "graph_query: excluded frame %d", excluded_frames[k]);
        ex[excluded_frames[k]] = 1;
    }
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(g->excluded, ex.data(), ex.size(), hipMemcpyHostToDevice, st));
    graph_edge_counts_kernel<<<g->n_edges, GT, 0, st>>>(g->pairs, g->counts, g->offsets, g->q, g->t, g->kp_offsets, g->has3d, g->edge_counts);
    graph_best_pair_kernel<<<1, 1024, 0, st>>>(g->pairs, g->n_edges, g->counts, g->edge_counts, g->valid, g->excluded, g->best);
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(best, g->best, sizeof(unsigned) * 3, hipMemcpyDeviceToHost, st));
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(st));
    return EACHAM_OK;
}

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
