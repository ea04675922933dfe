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
    // what the last eacham_comm_match_run left resident (eacham_comm_match_fetch reads it)
    int run_npairs = -1, run_shard_cap = 0;
    long long run_edge_cap = 0, run_total = 0;
    size_t run_gc = 0;
    std::vector<int32_t> run_order, run_bounds;

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

// Work-balanced contiguous cut of the ordered pair list: bounds[r] = first pair of shard r, bounds[world] = npairs.
// With weights w_k >= 0 (the matcher's cost of pair k is its distance matrix, rows(f1) x rows(f2)) and prefix sums
// P[k] = w_0 + .. + w_{k-1}, shard r starts at the smallest k with P[k] * world >= r * P[npairs]: every shard's work is
// within one pair of the mean, whatever the mix of frame sizes. weights == NULL (or all zero) = equal pair COUNTS, the cut
// of eacham_shard_bounds.
int eacham_shard_bounds_weighted(int npairs, int world, const int64_t* weights, int32_t* bounds) {
    if (npairs < 0 || world <= 0 || !bounds) return EACHAM_ERR_INVALID;
    long long W = 0;
    if (weights)
        for (int k = 0; k < npairs; ++k) {
            if (weights[k] < 0 || weights[k] > ((long long)1 << 40)) return EACHAM_ERR_INVALID;
            W += weights[k];
        }
    if (!weights || W == 0) {
        for (int r = 0; r < world; ++r) {
            int b = 0, e = 0;
            (void)eacham_shard_bounds(npairs, world, r, &b, &e);
            bounds[r] = b;
        }
        bounds[world] = npairs;
        return EACHAM_OK;
    }
    if (W > (long long)0x7fffffffffffffffLL / world) return EACHAM_ERR_UNSUPPORTED;
    long long P = 0;
    int k = 0;
    for (int r = 0; r < world; ++r) {
        while (k < npairs && P * world < (long long)r * W) P += weights[k++];
        bounds[r] = k;
    }
    bounds[world] = npairs;
    return EACHAM_OK;
}

// Host-side assembly of gathered shards (no device needed): g_counts = world x shard_cap per-pair counts (zero padded),
// g_edges = world x edge_cap x {q, t}; shard r = pairs [bounds[r], bounds[r + 1]) of the SORTED pair list (bounds == NULL:
// the equal-count cut of eacham_shard_bounds), sorted_index[k] = position of sorted pair k in the caller's list. Writes
// the CSR over the caller's pair order.
int eacham_assemble_match_graph_bounds(const int32_t* g_counts, const uint32_t* g_edges, int npairs, int world, int shard_cap,
                                       int64_t edge_cap, const int32_t* bounds, const int32_t* sorted_index, int32_t* counts,
                                       int64_t* offsets, uint32_t* out_q, uint32_t* out_t, int64_t cap, int64_t* out_total) {
    if (npairs < 0 || world <= 0 || shard_cap < 0 || edge_cap < 0 || !out_total || (npairs > 0 && (!g_counts || !counts || !offsets)))
        return EACHAM_ERR_INVALID;
    if (bounds && (bounds[0] != 0 || bounds[world] != npairs)) return EACHAM_ERR_INVALID;
    std::vector<int64_t> src_off((size_t)npairs, 0);  // offset of sorted pair k inside its shard's edge list
    std::vector<int> src_rank((size_t)npairs, 0);
    for (int r = 0; r < world; ++r) {
        int b = 0, e = 0;
        if (bounds) {
            b = bounds[r];
            e = bounds[r + 1];
            if (b < 0 || e < b || e > npairs) return EACHAM_ERR_INVALID;
        } else if (eacham_shard_bounds(npairs, world, r, &b, &e) != EACHAM_OK) {
            return EACHAM_ERR_INVALID;
        }
        if (e - b > shard_cap) return EACHAM_ERR_INVALID;
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

int eacham_assemble_match_graph(const int32_t* g_counts, const uint32_t* g_edges, int npairs, int world, int shard_cap,
                                int64_t edge_cap, const int32_t* sorted_index, int32_t* counts, int64_t* offsets,
                                uint32_t* out_q, uint32_t* out_t, int64_t cap, int64_t* out_total) {
    return eacham_assemble_match_graph_bounds(g_counts, g_edges, npairs, world, shard_cap, edge_cap, nullptr, sorted_index, counts,
                                              offsets, out_q, out_t, cap, out_total);
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

// Sizes of the edge region of every rank's send buffer (host logic, exported for the CPU tests): a shard can produce at
// most the rows of its query frames, and because ncclAllGather sends the SAME number of elements from every rank — the
// largest exact total — every rank's region must hold the largest bound, not just its own.
int eacham_comm_edge_region(int world, const int64_t* shard_bound, int64_t* region) {
    if (world <= 0 || !shard_bound || !region) return EACHAM_ERR_INVALID;
    int64_t m = 1;
    for (int r = 0; r < world; ++r) {
        if (shard_bound[r] < 0) return EACHAM_ERR_INVALID;
        m = std::max<int64_t>(m, shard_bound[r]);
    }
    *region = m;
    return EACHAM_OK;
}

// The device part: ordering, work-balanced shards, matching on every device, the two all-gathers. The gathered graph
// stays resident on EVERY device (c->gathered[r]); eacham_comm_match_fetch reads device 0's copy.
int eacham_comm_match_run(eacham_comm* c, const int32_t* pairs, int npairs, double ratio, int min_dir, int min_mutual,
                          int balance, int64_t* out_total) {
    if (!c || npairs < 0 || !out_total || (npairs > 0 && !pairs)) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(c->mu);
    *out_total = 0;
    c->run_npairs = -1;
    if (npairs == 0) {
        c->run_npairs = 0;
        return EACHAM_OK;
    }
    const int world = c->ndev;
    // every device sees the same order: by train frame, then query frame (eacham_order_pairs), remembered for the way back
    std::vector<int32_t>& order = c->run_order;
    order.resize((size_t)npairs);
    int max_frame = 0;
    for (int i = 0; i < 2 * npairs; ++i) {
        if (pairs[i] < 0 || pairs[i] >= (1 << 20)) return c->fail(EACHAM_ERR_INVALID, "pair %d names frame %d", i / 2, pairs[i]);
        max_frame = std::max(max_frame, (int)pairs[i]);
    }
    {   // two stable counting passes (query frame, then train frame) = the order of eacham_order_pairs, O(npairs + frames)
        std::vector<int32_t> tmp((size_t)npairs), head((size_t)max_frame + 2);
        for (int col = 0; col < 2; ++col) {
            std::fill(head.begin(), head.end(), 0);
            for (int k = 0; k < npairs; ++k) head[pairs[2 * (col ? tmp[k] : k) + col] + 1]++;
            for (int f = 0; f <= max_frame; ++f) head[f + 1] += head[f];
            for (int k = 0; k < npairs; ++k) {
                const int32_t src = col ? tmp[k] : k;
                (col ? order : tmp)[head[pairs[2 * src + col]]++] = src;
            }
        }
    }
    std::vector<int> rows((size_t)max_frame + 1, -2);  // one locked query per frame, not per pair
    auto rows_of = [&](int f) {
        if (rows[f] == -2) rows[f] = eacham_frame_rows(c->ctx[0], f);
        return rows[f];
    };
    std::vector<int32_t> sorted(2 * (size_t)npairs);
    std::vector<int64_t> weight((size_t)npairs), rows_q((size_t)npairs);
    for (int k = 0; k < npairs; ++k) {
        sorted[2 * k] = pairs[2 * order[k]], sorted[2 * k + 1] = pairs[2 * order[k] + 1];
        const int r1 = rows_of(sorted[2 * k]), r2 = rows_of(sorted[2 * k + 1]);
        if (r1 < 0 || r2 < 0)
            return c->fail(EACHAM_ERR_INVALID, "pair %d names a frame (%d, %d) that is not resident", order[k], sorted[2 * k], sorted[2 * k + 1]);
        rows_q[k] = r1;
        weight[k] = (int64_t)r1 * r2;  // the pair's distance matrix: what the tile kernel's time is proportional to
    }
    // every device holds every frame at the same size (a partial upload would otherwise lose matches silently: a pair naming a
    // frame that is missing on ITS device is redirected to the empty stand-in by the kernels' own sanitising)
    for (int r = 1; r < world; ++r)
        for (int f = 0; f <= max_frame; ++f)
            if (rows[f] >= 0 && eacham_frame_rows(c->ctx[r], f) != rows[f])
                return c->fail(EACHAM_ERR_INVALID, "frame %d has %d rows on device 0 and %d on device %d: upload it through the communicator", f, rows[f],
                               eacham_frame_rows(c->ctx[r], f), r);
    std::vector<int32_t>& bnd = c->run_bounds;
    bnd.assign((size_t)world + 1, 0);
    int rc = eacham_shard_bounds_weighted(npairs, world, balance ? weight.data() : nullptr, bnd.data());
    if (rc) return c->fail(rc, "cutting %d pairs into %d shards failed", npairs, world);
    int shard_cap = 1;
    std::vector<int64_t> bound(world, 1);  // edges a shard can produce at most: the rows of its query frames
    for (int r = 0; r < world; ++r) {
        shard_cap = std::max(shard_cap, bnd[r + 1] - bnd[r]);
        for (int k = bnd[r]; k < bnd[r + 1]; ++k) bound[r] += rows_q[k];
    }
    int64_t region = 1;
    (void)eacham_comm_edge_region(world, bound.data(), &region);
    // ---- phase 0: every allocation of every rank, before anything is enqueued (a rank that fails here fails the call
    // while no other rank has entered a collective) ----
    const size_t o_counts = align256((size_t)shard_cap * 2 * sizeof(int32_t));
    const size_t o_offsets = align256(o_counts + (size_t)shard_cap * sizeof(int32_t));
    const size_t o_total = align256(o_offsets + (size_t)(shard_cap + 1) * sizeof(int64_t));
    const size_t o_edges = align256(o_total + sizeof(int64_t));
    for (int r = 0; r < world; ++r) {
        (void)hipSetDevice(c->ctx[r]->device);
        if (grow(c, r, c->work[r], o_edges + (size_t)region * 2 * sizeof(uint32_t)))
            return c->fail(EACHAM_ERR_HIP, "device %d: allocating the shard workspace failed", r);
    }
    // ---- phase 1: every device matches its shard; the exact totals size the padded edge lists of the gather ----
    std::vector<int> rcs(world, EACHAM_OK);
    std::vector<long long> totals(world, 0);
    auto phase1 = [&](int r) {
        eacham_ctx* x = c->ctx[r];
        (void)hipSetDevice(x->device);
        const int n = bnd[r + 1] - bnd[r];
        char* w = (char*)c->work[r].dev;
        if (hipMemsetAsync(w + o_counts, 0, (size_t)shard_cap * sizeof(int32_t), x->stream) != hipSuccess ||
            hipMemsetAsync(w + o_total, 0, sizeof(int64_t), x->stream) != hipSuccess ||
            (n > 0 && hipMemcpyAsync(w, sorted.data() + 2 * (size_t)bnd[r], (size_t)n * 2 * sizeof(int32_t), hipMemcpyHostToDevice, x->stream) != hipSuccess)) {
            rcs[r] = EACHAM_ERR_HIP;
            return;
        }
        if (n > 0)
            rcs[r] = eacham_match_all_pairs_dev(x, (const int32_t*)w, n, ratio, min_dir, min_mutual, (int32_t*)(w + o_counts),
                                                (int64_t*)(w + o_offsets), (uint32_t*)(w + o_edges), bound[r],
                                                (int64_t*)(w + o_total), nullptr);
        if (rcs[r] == EACHAM_OK && (hipMemcpyAsync(&totals[r], w + o_total, sizeof(long long), hipMemcpyDeviceToHost, x->stream) != hipSuccess ||
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
    if (edge_cap > region) return c->fail(EACHAM_ERR_HIP, "a shard reports %lld matches, more than its query rows (%lld)", edge_cap, (long long)region);
    // ---- phase 2: the receive buffers of every rank first, then the RCCL all-gathers on every context's own stream ----
    const size_t gc = align256((size_t)world * shard_cap * sizeof(int32_t));
    for (int r = 0; r < world; ++r) {
        (void)hipSetDevice(c->ctx[r]->device);
        if (grow(c, r, c->gathered[r], gc + (size_t)world * edge_cap * 2 * sizeof(uint32_t)))
            return c->fail(EACHAM_ERR_HIP, "device %d: allocating the gathered match graph failed", r);
    }
    auto phase2 = [&](int r) {
        eacham_ctx* x = c->ctx[r];
        (void)hipSetDevice(x->device);
        char* w = (char*)c->work[r].dev;
        char* g = (char*)c->gathered[r].dev;
        // the slots between this rank's total and the common edge_cap travel too: cleared, never stale
        if (totals[r] < edge_cap &&
            hipMemsetAsync(w + o_edges + (size_t)totals[r] * 2 * sizeof(uint32_t), 0, (size_t)(edge_cap - totals[r]) * 2 * sizeof(uint32_t), x->stream) != hipSuccess) {
            rcs[r] = EACHAM_ERR_HIP;  // (still joins the collectives below: nobody may be left waiting)
        }
        ncclResult_t nr = c->AllGather(w + o_counts, g, (size_t)shard_cap, ncclInt32, c->comms[r], x->stream);
        if (nr == ncclSuccess) nr = c->AllGather(w + o_edges, g + gc, (size_t)edge_cap * 2, ncclUint32, c->comms[r], x->stream);
        if (nr != ncclSuccess) rcs[r] = EACHAM_ERR_HIP;
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
    long long total = 0;
    for (int r = 0; r < world; ++r) total += totals[r];
    c->run_npairs = npairs;
    c->run_shard_cap = shard_cap;
    c->run_edge_cap = edge_cap;
    c->run_gc = gc;
    c->run_total = total;
    *out_total = total;
    return EACHAM_OK;
}

// Downloads the graph gathered by the last eacham_comm_match_run from device 0 and assembles the CSR in the caller's
// pair order.
int eacham_comm_match_fetch(eacham_comm* c, int32_t* counts, int64_t* offsets, uint32_t* out_q, uint32_t* out_t, int64_t cap,
                            int64_t* out_total) {
    if (!c || !out_total || cap < 0 || (cap > 0 && (!out_q || !out_t))) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->run_npairs < 0) return c->fail(EACHAM_ERR_INVALID, "no gathered match graph: eacham_comm_match_run has not succeeded");
    *out_total = 0;
    const int npairs = c->run_npairs, world = c->ndev;
    if (npairs == 0) {
        if (offsets) offsets[0] = 0;
        return EACHAM_OK;
    }
    if (!counts || !offsets) return EACHAM_ERR_INVALID;
    std::vector<int32_t> g_counts((size_t)world * c->run_shard_cap);
    std::vector<uint32_t> g_edges((size_t)world * c->run_edge_cap * 2);
    eacham_ctx* x = c->ctx[0];
    (void)hipSetDevice(x->device);
    const char* g = (const char*)c->gathered[0].dev;
    if (hipMemcpyAsync(g_counts.data(), g, g_counts.size() * sizeof(int32_t), hipMemcpyDeviceToHost, x->stream) != hipSuccess ||
        hipMemcpyAsync(g_edges.data(), g + c->run_gc, g_edges.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, x->stream) != hipSuccess ||
        hipStreamSynchronize(x->stream) != hipSuccess)
        return c->fail(EACHAM_ERR_HIP, "reading the gathered match graph back failed");
    const int rc = eacham_assemble_match_graph_bounds(g_counts.data(), g_edges.data(), npairs, world, c->run_shard_cap, c->run_edge_cap,
                                                      c->run_bounds.data(), c->run_order.data(), counts, offsets, out_q, out_t, cap, out_total);
    if (rc) return c->fail(rc, rc == EACHAM_ERR_CAPACITY ? "output capacity %lld too small" : "gathered match graph is inconsistent", (long long)cap);
    return EACHAM_OK;
}

int eacham_match_all_pairs_sharded(eacham_comm* c, const int32_t* pairs, int npairs, double ratio, int min_dir, int min_mutual,
                                   int32_t* counts, int64_t* offsets, uint32_t* out_q, uint32_t* out_t, int64_t cap,
                                   int64_t* out_total) {
    if (!c || npairs < 0 || !out_total || (npairs > 0 && (!pairs || !counts || !offsets)) || cap < 0 || (cap > 0 && (!out_q || !out_t)))
        return EACHAM_ERR_INVALID;
    int64_t total = 0;
    int rc = eacham_comm_match_run(c, pairs, npairs, ratio, min_dir, min_mutual, /*balance=*/1, &total);
    if (rc) return rc;
    return eacham_comm_match_fetch(c, counts, offsets, out_q, out_t, cap, out_total);
}

}  // extern "C"
