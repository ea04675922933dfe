// comm.hip — single-process multi-GPU sharding of the pair loop behind the C-ABI.
//
// Replaces the std::for_each(par_unseq) + std::async over pairs of the reference's ONE C++ process
// (apps/sfm/main.cpp:84-147, :98-109) across the GPUs of a node: one context + one host thread per device, every device
// holds all descriptors (replicated), the train-frame-ordered pair list is cut into contiguous shards, every device
// matches its shard (eacham_match_all_pairs_dev), and the match graph is assembled on EVERY device by RCCL all-gathers
// over xGMI — ncclAllGather of the per-pair counts and of the edge lists padded to the largest shard — enqueued on each
// context's own stream behind its matching, so no host synchronisation sits between the kernels and the collective.
// RCCL is loaded at run time (dlopen of the librccl that sits next to the HIP runtime this library is linked against: no
// link-time dependency, and never the copy a PyTorch wheel bundles for its own runtime); without it eacham_comm_init fails with EACHAM_ERR_UNSUPPORTED — there is no host-staged fallback.
#include "context.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <thread>

struct eacham_comm {
    int ndev = 0;
    std::vector<eacham_ctx*> ctx;
    std::vector<ncclComm_t> comms;
    std::string err;
    std::mutex mu;
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // per device: pairs | counts | offsets | total | edges, and the gathered counts / edges
    struct Buf { void* dev = nullptr; size_t bytes = 0; };
    std::vector<Buf> work, gathered;

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

namespace {

int grow(eacham_comm* c, int r, eacham_comm::Buf& b, size_t bytes) {
    if (bytes <= b.bytes) return EACHAM_OK;
    if (b.dev) {
        (void)hipStreamSynchronize(c->ctx[r]->stream);
        (void)hipFree(b.dev);
        b.dev = nullptr;
        b.bytes = 0;
    }
    if (hipMalloc(&b.dev, bytes) != hipSuccess) return EACHAM_ERR_HIP;
    b.bytes = bytes;
    return EACHAM_OK;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

// Host-side assembly of gathered shards (no device needed): g_counts = world x shard_cap per-pair counts (zero padded),
// g_edges = world x edge_cap x {q, t}; the shards are the contiguous ranges of eacham_shard_bounds over the SORTED pair
// list, sorted_index[k] = position of sorted pair k in the caller's list. Writes the CSR over the caller's pair order.
int eacham_assemble_match_graph(const int32_t* g_counts, const uint32_t* g_edges, int npairs, int world, int shard_cap,
                                int64_t edge_cap, const int32_t* sorted_index, int32_t* counts, int64_t* offsets,
                                uint32_t* out_q, uint32_t* out_t, int64_t cap, int64_t* out_total) {
    if (npairs < 0 || world <= 0 || shard_cap < 0 || edge_cap < 0 || !out_total || (npairs > 0 && (!g_counts || !counts || !offsets)))
        return EACHAM_ERR_INVALID;
    std::vector<int64_t> src_off((size_t)npairs, 0);  // offset of sorted pair k inside its shard's edge list
    std::vector<int> src_rank((size_t)npairs, 0);
    for (int r = 0; r < world; ++r) {
        int b = 0, e = 0;
        if (eacham_shard_bounds(npairs, world, r, &b, &e) != EACHAM_OK || e - b > shard_cap) return EACHAM_ERR_INVALID;
        int64_t run = 0;
        for (int k = b; k < e; ++k) {
            const int32_t cnt = g_counts[(size_t)r * shard_cap + (k - b)];
            if (cnt < 0 || run + cnt > edge_cap) return EACHAM_ERR_INVALID;
            const int dst = sorted_index ? sorted_index[k] : k;
            if (dst < 0 || dst >= npairs) return EACHAM_ERR_INVALID;
            counts[dst] = cnt;
            src_off[dst] = run;
            src_rank[dst] = r;
            run += cnt;
        }
    }
    int64_t total = 0;
    for (int p = 0; p < npairs; ++p) {
        offsets[p] = total;
        total += counts[p];
    }
    if (offsets) offsets[npairs] = total;
    *out_total = total;
    if (total > cap) return EACHAM_ERR_CAPACITY;
    for (int p = 0; p < npairs; ++p) {
        const uint32_t* e = g_edges + 2 * ((size_t)src_rank[p] * (size_t)edge_cap + (size_t)src_off[p]);
        for (int32_t k = 0; k < counts[p]; ++k) {
            out_q[offsets[p] + k] = e[2 * k];
            out_t[offsets[p] + k] = e[2 * k + 1];
        }
    }
    return EACHAM_OK;
}

int eacham_comm_init(int ndev, const int* devices, eacham_comm** out) {
    if (!out || ndev <= 0) return EACHAM_ERR_INVALID;
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return EACHAM_ERR_NO_DEVICE;
    eacham_comm* c = new (std::nothrow) eacham_comm();
    if (!c) return EACHAM_ERR_INVALID;
    {
        // The RCCL that belongs to the HIP runtime THIS library is linked against: a process may carry a second ROCm
        // (the PyTorch wheel bundles its own runtime and its own librccl.so), and a communicator built on the other
        // runtime could not take this library's streams. So: the librccl next to our libamdhip64 first, by path and
        // RTLD_LOCAL (its symbols are only ever reached through this handle), the loader's default search last.
        std::vector<std::string> names;
        Dl_info info;
        if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) names.push_back(dir.substr(0, slash) + "/librccl.so.1");
        }
        names.push_back("/opt/rocm/lib/librccl.so.1");
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        for (const std::string& name : names) {
            c->lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (c->lib) break;
        }
    }
    if (c->lib) {
        c->CommInitAll = (decltype(c->CommInitAll))dlsym(c->lib, "ncclCommInitAll");
        c->CommDestroy = (decltype(c->CommDestroy))dlsym(c->lib, "ncclCommDestroy");
        c->AllGather = (decltype(c->AllGather))dlsym(c->lib, "ncclAllGather");
        c->GetErrorString = (decltype(c->GetErrorString))dlsym(c->lib, "ncclGetErrorString");
    }
    if (!c->CommInitAll || !c->CommDestroy || !c->AllGather) {
        delete c;
        return EACHAM_ERR_UNSUPPORTED;  // RCCL not loadable: no host-staged fallback
    }
    std::vector<int> devs(ndev);
    for (int r = 0; r < ndev; ++r) {
        devs[r] = devices ? devices[r] : r;
        if (devs[r] < 0 || devs[r] >= have) {
            delete c;
            return EACHAM_ERR_INVALID;
        }
    }
    c->ndev = ndev;
    c->ctx.assign(ndev, nullptr);
    c->work.resize(ndev);
    c->gathered.resize(ndev);
    int rc = EACHAM_OK;
    for (int r = 0; r < ndev && rc == EACHAM_OK; ++r) rc = eacham_ctx_create(devs[r], &c->ctx[r]);
    if (rc == EACHAM_OK) {
        c->comms.assign(ndev, nullptr);
        const ncclResult_t nr = c->CommInitAll(c->comms.data(), ndev, devs.data());
        if (nr != ncclSuccess) {
            c->comms.clear();
            rc = EACHAM_ERR_HIP;
        }
    }
    if (rc != EACHAM_OK) {
        for (eacham_ctx* x : c->ctx)
            if (x) eacham_ctx_destroy(x);
        delete c;
        return rc;
    }
    *out = c;
    return EACHAM_OK;
}

void eacham_comm_destroy(eacham_comm* c) {
    if (!c) return;
    for (int r = 0; r < c->ndev; ++r) {
        (void)hipSetDevice(c->ctx[r]->device);
        (void)hipStreamSynchronize(c->ctx[r]->stream);
        if (c->work[r].dev) (void)hipFree(c->work[r].dev);
        if (c->gathered[r].dev) (void)hipFree(c->gathered[r].dev);
    }
    for (ncclComm_t m : c->comms)
        if (m) (void)c->CommDestroy(m);
    for (eacham_ctx* x : c->ctx)
        if (x) eacham_ctx_destroy(x);
    delete c;
}

const char* eacham_comm_last_error(const eacham_comm* c) { return c ? c->err.c_str() : "null communicator"; }
int eacham_comm_size(const eacham_comm* c) { return c ? c->ndev : 0; }
eacham_ctx* eacham_comm_ctx(eacham_comm* c, int rank) { return (c && rank >= 0 && rank < c->ndev) ? c->ctx[rank] : nullptr; }

// Replicates one frame's descriptors on every device (the S200 set is 102 MB as int8: trivial against 288 GB).
int eacham_comm_upload_descriptors(eacham_comm* c, int frame_id, const float* rowmajor, int n, int dim) {
    if (!c) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(c->mu);
    for (int r = 0; r < c->ndev; ++r) {
        const int rc = eacham_upload_descriptors(c->ctx[r], frame_id, rowmajor, n, dim);
        if (rc) return c->fail(rc, "device %d: %s", r, eacham_last_error(c->ctx[r]));
    }
    return EACHAM_OK;
}

int eacham_match_all_pairs_sharded(eacham_comm* c, const int32_t* pairs, int npairs, double ratio, int min_dir, int min_mutual,
                                   int32_t* counts, int64_t* offsets, uint32_t* out_q, uint32_t* out_t, int64_t cap,
                                   int64_t* out_total) {
    if (!c || npairs < 0 || !out_total || (npairs > 0 && (!pairs || !counts || !offsets)) || cap < 0 || (cap > 0 && (!out_q || !out_t)))
        return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(c->mu);
    *out_total = 0;
    if (npairs == 0) {
        if (offsets) offsets[0] = 0;
        return EACHAM_OK;
    }
    const int world = c->ndev;
    // every device sees the same order: by train frame, then query frame (eacham_order_pairs), remembered for the way back
    std::vector<int32_t> order((size_t)npairs);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return pairs[2 * a + 1] != pairs[2 * b + 1] ? pairs[2 * a + 1] < pairs[2 * b + 1] : pairs[2 * a] < pairs[2 * b];
    });
    std::vector<int32_t> sorted(2 * (size_t)npairs);
    for (int k = 0; k < npairs; ++k) sorted[2 * k] = pairs[2 * order[k]], sorted[2 * k + 1] = pairs[2 * order[k] + 1];
    const int shard_cap = (npairs + world - 1) / world;
    std::vector<int> lo(world), hi(world);
    std::vector<long long> bound(world, 1);  // edges a shard can produce at most: the rows of its query frames
    for (int r = 0; r < world; ++r) {
        (void)eacham_shard_bounds(npairs, world, r, &lo[r], &hi[r]);
        for (int k = lo[r]; k < hi[r]; ++k) {
            const int rows = eacham_frame_rows(c->ctx[r], sorted[2 * k]);
            if (rows < 0 || eacham_frame_rows(c->ctx[r], sorted[2 * k + 1]) < 0)
                return c->fail(EACHAM_ERR_INVALID, "pair %d names a frame (%d, %d) that is not resident", order[k], sorted[2 * k], sorted[2 * k + 1]);
            bound[r] += rows;
        }
    }
    // ---- phase 1: every device matches its shard; the exact totals size the padded edge lists of the gather ----
    std::vector<int> rcs(world, EACHAM_OK);
    std::vector<long long> totals(world, 0);
    std::vector<size_t> o_counts(world), o_offsets(world), o_total(world), o_edges(world);
    auto phase1 = [&](int r) {
        eacham_ctx* x = c->ctx[r];
        (void)hipSetDevice(x->device);
        const int n = hi[r] - lo[r];
        o_counts[r] = align256((size_t)shard_cap * 2 * sizeof(int32_t));
        o_offsets[r] = align256(o_counts[r] + (size_t)shard_cap * sizeof(int32_t));
        o_total[r] = align256(o_offsets[r] + (size_t)(shard_cap + 1) * sizeof(int64_t));
        o_edges[r] = align256(o_total[r] + sizeof(int64_t));
        if (grow(c, r, c->work[r], o_edges[r] + (size_t)bound[r] * 2 * sizeof(uint32_t))) { rcs[r] = EACHAM_ERR_HIP; return; }
        char* w = (char*)c->work[r].dev;
        if (hipMemsetAsync(w + o_counts[r], 0, (size_t)shard_cap * sizeof(int32_t), x->stream) != hipSuccess ||
            hipMemsetAsync(w + o_total[r], 0, sizeof(int64_t), x->stream) != hipSuccess ||
            (n > 0 && hipMemcpyAsync(w, sorted.data() + 2 * (size_t)lo[r], (size_t)n * 2 * sizeof(int32_t), hipMemcpyHostToDevice, x->stream) != hipSuccess)) {
            rcs[r] = EACHAM_ERR_HIP;
            return;
        }
        if (n > 0)
            rcs[r] = eacham_match_all_pairs_dev(x, (const int32_t*)w, n, ratio, min_dir, min_mutual, (int32_t*)(w + o_counts[r]),
                                                (int64_t*)(w + o_offsets[r]), (uint32_t*)(w + o_edges[r]), bound[r],
                                                (int64_t*)(w + o_total[r]), nullptr);
        if (rcs[r] == EACHAM_OK && (hipMemcpyAsync(&totals[r], w + o_total[r], sizeof(long long), hipMemcpyDeviceToHost, x->stream) != hipSuccess ||
                                    hipStreamSynchronize(x->stream) != hipSuccess))
            rcs[r] = EACHAM_ERR_HIP;
    };
    {
        std::vector<std::thread> th;
        for (int r = 1; r < world; ++r) th.emplace_back(phase1, r);
        phase1(0);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < world; ++r)
        if (rcs[r]) return c->fail(rcs[r], "device %d: %s", r, rcs[r] == EACHAM_ERR_HIP ? "HIP call failed while matching a shard" : eacham_last_error(c->ctx[r]));
    const long long edge_cap = std::max<long long>(1, *std::max_element(totals.begin(), totals.end()));
    // ---- phase 2: the RCCL all-gathers, on every context's own stream (one host thread per device) ----
    std::vector<int32_t> g_counts((size_t)world * shard_cap);
    std::vector<uint32_t> g_edges((size_t)world * edge_cap * 2);
    auto phase2 = [&](int r) {
        eacham_ctx* x = c->ctx[r];
        (void)hipSetDevice(x->device);
        const size_t gc = align256((size_t)world * shard_cap * sizeof(int32_t));
        if (grow(c, r, c->gathered[r], gc + (size_t)world * edge_cap * 2 * sizeof(uint32_t))) { rcs[r] = EACHAM_ERR_HIP; return; }
        char* w = (char*)c->work[r].dev;
        char* g = (char*)c->gathered[r].dev;
        ncclResult_t nr = c->AllGather(w + o_counts[r], g, (size_t)shard_cap, ncclInt32, c->comms[r], x->stream);
        if (nr == ncclSuccess) nr = c->AllGather(w + o_edges[r], g + gc, (size_t)edge_cap * 2, ncclUint32, c->comms[r], x->stream);
        if (nr != ncclSuccess) { rcs[r] = EACHAM_ERR_HIP; return; }
        if (r == 0) {  // every device holds the graph now; the host reads device 0's copy
            if (hipMemcpyAsync(g_counts.data(), g, g_counts.size() * sizeof(int32_t), hipMemcpyDeviceToHost, x->stream) != hipSuccess ||
                hipMemcpyAsync(g_edges.data(), g + gc, g_edges.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, x->stream) != hipSuccess)
                rcs[r] = EACHAM_ERR_HIP;
        }
        if (hipStreamSynchronize(x->stream) != hipSuccess) rcs[r] = EACHAM_ERR_HIP;
    };
    {
        std::vector<std::thread> th;
        for (int r = 1; r < world; ++r) th.emplace_back(phase2, r);
        phase2(0);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < world; ++r)
        if (rcs[r]) return c->fail(rcs[r], "device %d: the all-gather of the match graph failed", r);
    const int rc = eacham_assemble_match_graph(g_counts.data(), g_edges.data(), npairs, world, shard_cap, edge_cap, order.data(), counts,
                                               offsets, out_q, out_t, cap, out_total);
    if (rc) return c->fail(rc, rc == EACHAM_ERR_CAPACITY ? "output capacity %lld too small" : "gathered match graph is inconsistent", (long long)cap);
    return EACHAM_OK;
}

}  // extern "C"
