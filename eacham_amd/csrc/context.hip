// context.hip — context lifetime, workspace, profiling: the non-kernel part of the C-ABI
// declared in include/eacham_hip.h.
#include "context.hpp"
#include <algorithm>

#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <new>

namespace eacham {

int ensure_workspace(eacham_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return EACHAM_OK;
    if (ctx->ws) {
        EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        EACHAM_HIP_TRY(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    EACHAM_HIP_TRY(ctx, hipMalloc(&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return EACHAM_OK;
}

int ensure_io(eacham_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->io_bytes) return EACHAM_OK;
    if (ctx->io) {
        EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        EACHAM_HIP_TRY(ctx, hipFree(ctx->io));
        ctx->io = nullptr;
        ctx->io_bytes = 0;
    }
    EACHAM_HIP_TRY(ctx, hipMalloc(&ctx->io, bytes));
    ctx->io_bytes = bytes;
    return EACHAM_OK;
}

// pinned mirror of the first `bytes` of the staging buffer (grown, never shrunk; capped: large arrays are copied directly)
int ensure_io_host(eacham_ctx* ctx, size_t bytes) {
    bytes = std::min<size_t>(bytes, 4u << 20);
    if (bytes <= ctx->io_host_bytes) return EACHAM_OK;
    if (ctx->io_host) {
        EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        EACHAM_HIP_TRY(ctx, hipHostFree(ctx->io_host));
        ctx->io_host = nullptr;
        ctx->io_host_bytes = 0;
    }
    const size_t want = std::max<size_t>(bytes, 256 * 1024);
    EACHAM_HIP_TRY(ctx, hipHostMalloc(&ctx->io_host, want, hipHostMallocDefault));
    ctx->io_host_bytes = want;
    return EACHAM_OK;
}

// the two kernels of the second-stream probe of eacham_ctx_create
__global__ void stream_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void stream_probe_kernel() {}

__global__ void sanitize_pairs_kernel(const int2* __restrict__ in, int npairs, const FrameDev* __restrict__ frames, int n_frames,
                                      int2* __restrict__ out, int* __restrict__ flag) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    int2 pr = in[p];
    const bool ok = pr.x >= 0 && pr.y >= 0 && pr.x < n_frames && pr.y < n_frames && frames[pr.x].resident && frames[pr.y].resident;
    if (!ok) {
        pr = make_int2(n_frames, n_frames);  // the empty stand-in: the pair yields no match
        atomicOr(flag, 1);
    }
    out[p] = pr;
}

int sanitize_pairs(eacham_ctx* ctx, const int2* pairs_dev, int npairs, const int2** out) {
    if (npairs > ctx->pairs_safe_cap) {
        if (ctx->pairs_safe) {
            EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            EACHAM_HIP_TRY(ctx, hipFree(ctx->pairs_safe));
            ctx->pairs_safe = nullptr;
            ctx->pairs_safe_cap = 0;
        }
        const int cap = npairs + npairs / 2 + 64;
        EACHAM_HIP_TRY(ctx, hipMalloc((void**)&ctx->pairs_safe, sizeof(int2) * (size_t)cap));
        ctx->pairs_safe_cap = cap;
    }
    sanitize_pairs_kernel<<<(npairs + 255) / 256, 256, 0, ctx->stream>>>(pairs_dev, npairs, ctx->frame_table_dev, (int)ctx->frames.size(),
                                                                          ctx->pairs_safe, ctx->flag_dev + 1);
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    *out = ctx->pairs_safe;
    return EACHAM_OK;
}

__global__ void gather_tiles_kernel(const FrameDev* __restrict__ frames, int n, int* __restrict__ used) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) used[i] = frames[i].meta ? frames[i].meta[1] : frames[i].ntiles;
}

int sync_frame_table(eacham_ctx* ctx) {
    if (!ctx->frame_table_dirty) return EACHAM_OK;
    int need = (int)ctx->frames.size();
    if (need + 1 > ctx->frame_table_cap) {
        if (ctx->frame_table_dev) {
            EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            EACHAM_HIP_TRY(ctx, hipFree(ctx->frame_table_dev));
            ctx->frame_table_dev = nullptr;
        }
        int cap = need < 64 ? 64 : need * 2;
        EACHAM_HIP_TRY(ctx, hipMalloc((void**)&ctx->frame_table_dev, sizeof(FrameDev) * cap));
        ctx->frame_table_cap = cap;
    }
    std::vector<FrameDev> tab(need + 1);
    {   // entry [need]: an empty frame that invalid device-side pairs are redirected to
        FrameDev& e = tab[need];
        e.frag = nullptr; e.norm = nullptr; e.normb = nullptr;
        e.orig = ctx->flag_dev + 16; e.pos = ctx->flag_dev + 16; e.meta = ctx->flag_dev + 8;
        e.n = 0; e.ntiles = 0; e.resident = 1;
    }
    for (int i = 0; i < need; ++i) {
        const FrameHost& f = ctx->frames[i];
        tab[i].frag = f.frag;
        tab[i].norm = f.norm;
        tab[i].normb = f.normb;
        tab[i].orig = f.orig;
        tab[i].pos = f.pos;
        tab[i].meta = f.meta;
        tab[i].n = f.n < 0 ? 0 : f.n;
        tab[i].ntiles = f.n < 0 ? 0 : f.ntiles;
        tab[i].resident = f.n < 0 ? 0 : 1;
    }
    {
        // pageable source: hipMemcpyAsync stages it before returning, so `tab` may die here
        EACHAM_HIP_TRY(ctx, hipMemcpyAsync(ctx->frame_table_dev, tab.data(), sizeof(FrameDev) * (need + 1),
                                           hipMemcpyHostToDevice, ctx->stream));
        // tiles in use per frame (the parity split of an int8 frame is decided on the device): one
        // small read-back per table rebuild lets the host size strides and grids exactly
        int* used_dev = nullptr;
        if (need == 0) {
            ctx->frame_table_dirty = false;
            return EACHAM_OK;
        }
        EACHAM_HIP_TRY(ctx, hipMalloc((void**)&used_dev, sizeof(int) * need));
        gather_tiles_kernel<<<(need + 255) / 256, 256, 0, ctx->stream>>>(ctx->frame_table_dev, need, used_dev);
        std::vector<int> used(need);
        hipError_t e = hipMemcpyAsync(used.data(), used_dev, sizeof(int) * need, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(used_dev);
        EACHAM_HIP_TRY(ctx, e);
        for (int i = 0; i < need; ++i) ctx->frames[i].tiles_used = used[i];
    }
    ctx->frame_table_dirty = false;
    return EACHAM_OK;
}

// Optional ROCTx ranges around the stages ("roctx ranges"): EACHAM_ROCTX=1 in the environment loads
// the marker library at run time (no link-time dependency) and every ProfileScope pushes / pops a named range, visible
// to `rocprofv3 --marker-trace` (never together with --pmc).
namespace {
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)(void);
roctx_push_t g_roctx_push = nullptr;
roctx_pop_t g_roctx_pop = nullptr;
std::once_flag g_roctx_once;
const char* const kStageNames[EACHAM_KERNEL_COUNT] = {"eacham.match.tile", "eacham.match.finalize", "eacham.ba.linearize", "eacham.ba.schur",
                                                      "eacham.ba.solve", "eacham.ba.error", "eacham.triangulate", "eacham.score"};
void roctx_init() {
    const char* e = getenv("EACHAM_ROCTX");
    if (!e || !*e || *e == '0') return;
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    g_roctx_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
    g_roctx_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
    if (!g_roctx_push || !g_roctx_pop) g_roctx_push = nullptr, g_roctx_pop = nullptr;
}
}  // namespace

ProfileScope::ProfileScope(eacham_ctx* c, int kernel_id, hipStream_t on) : ctx(c), id(kernel_id), stream(on ? on : c->stream) {
    std::call_once(g_roctx_once, roctx_init);
    if (g_roctx_push) {
        (void)g_roctx_push(kStageNames[id]);
        range = true;
    }
    if (!ctx->profile) return;
    ProfileSlot& s = ctx->prof[id];
    if (s.used == s.events.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        s.events.emplace_back(a, b);
    }
    auto& ev = s.events[s.used++];
    (void)hipEventRecord(ev.first, stream);
    stop = ev.second;
}

ProfileScope::~ProfileScope() {
    if (stop) (void)hipEventRecord(stop, stream);
    if (range && g_roctx_pop) (void)g_roctx_pop();
}

static void profile_drain(eacham_ctx* ctx) {
    for (int k = 0; k < EACHAM_KERNEL_COUNT; ++k) {
        ProfileSlot& s = ctx->prof[k];
        for (size_t i = 0; i < s.used; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s.events[i].first, s.events[i].second) == hipSuccess) {
                s.total_ms += ms;
                s.launches += 1;
            }
        }
        s.used = 0;
    }
}

}  // namespace eacham

using namespace eacham;

extern "C" {

const char* eacham_version(void) { return "eacham_hip 0.1 gfx950"; }

int eacham_order_pairs(int32_t* pairs, int npairs) {
    if (npairs < 0 || (npairs > 0 && !pairs)) return EACHAM_ERR_INVALID;
    struct P2 { int32_t q, t; };
    P2* p = reinterpret_cast<P2*>(pairs);
    std::stable_sort(p, p + npairs, [](const P2& a, const P2& b) { return a.t != b.t ? a.t < b.t : a.q < b.q; });
    return EACHAM_OK;
}

int eacham_shard_bounds(int npairs, int world, int rank, int* begin, int* end) {
    if (npairs < 0 || world <= 0 || rank < 0 || rank >= world || !begin || !end) return EACHAM_ERR_INVALID;
    const int base = npairs / world, rem = npairs % world;
    *begin = rank * base + std::min(rank, rem);
    *end = *begin + base + (rank < rem ? 1 : 0);
    return EACHAM_OK;
}

int eacham_ctx_create(int device_id, eacham_ctx** out_ctx) {
    if (!out_ctx) return EACHAM_ERR_INVALID;
    *out_ctx = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return EACHAM_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= ndev) return EACHAM_ERR_INVALID;
    eacham_ctx* ctx = new (std::nothrow) eacham_ctx();
    if (!ctx) return EACHAM_ERR_INVALID;
    ctx->device = device_id;
    ctx->match_no_overlap = getenv("EACHAM_NO_OVERLAP") != nullptr;
    ctx->exp_no_coltop2 = getenv("EACHAM_EXP_NO_COLTOP2") != nullptr;
    ctx->match_full_columns = getenv("EACHAM_MATCH_FULL_COLUMNS") != nullptr;
    ctx->exp_all_candidates = getenv("EACHAM_EXP_ALL_CANDIDATES") != nullptr;
    ctx->match_tile_sweep = getenv("EACHAM_MATCH_TILE_SWEEP") != nullptr;
    if (const char* v = getenv("EACHAM_MATCH_SWEEP_FORM")) ctx->match_sweep_form = !strcmp(v, "exact") ? 1 : !strcmp(v, "bound") ? 2 : 0;
    if (const char* v = getenv("EACHAM_EXP_SWEEP_PRIO")) ctx->exp_sweep_prio = std::max(0, std::min(3, atoi(v)));
    if (const char* v = getenv("EACHAM_EXP_STREAM2_CUS")) ctx->exp_stream2_cus = std::max(0, std::min(256, atoi(v)));
    if (const char* b = getenv("EACHAM_MATCH_BUDGET_MB")) {
        const int v = atoi(b);
        if (v >= 16 && v <= 65536) ctx->match_budget_mb = v;
    }
    if (const char* l = getenv("EACHAM_BA_LPL_STEP")) {
        const int v = atoi(l);
        if (v == 1 || v == 2 || v == 4 || v == 8) ctx->ba_lpl_step = v;
    }
    if (const char* l = getenv("EACHAM_BA_LPL_LIN")) {
        const int v = atoi(l);
        if (v == 1 || v == 2 || v == 4 || v == 8) ctx->ba_lpl_lin = v;
    }
    if (const char* m = getenv("EACHAM_BA_PREPARE")) ctx->ba_prepare_mode = !strcmp(m, "host") ? 1 : !strcmp(m, "device") ? 2 : 0;
    if (const char* m = getenv("EACHAM_BA_SCHUR")) ctx->ba_schur_mode = !strcmp(m, "pairs") ? 2 : !strcmp(m, "groups") ? 1 : !strcmp(m, "dense") ? 3 : 0;
    if (const char* r = getenv("EACHAM_BA_WINDOW_ROWS")) {
        const int v = atoi(r);
        if (v >= 64 && v <= 256 && v % 4 == 0) ctx->ba_window_rows = v;
    }
    if (const char* r = getenv("EACHAM_BA_GROUP_ROWS")) {
        const int v = atoi(r);
        if (v >= 16 && v <= 512 && v % 4 == 0) ctx->ba_group_rows = v;
    }
    if (const char* o = getenv("EACHAM_BA_ORDERING"))
        ctx->ba_ordering = !strcmp(o, "natural") ? EACHAM_BA_ORDER_NATURAL : !strcmp(o, "rcm") ? EACHAM_BA_ORDER_RCM : !strcmp(o, "nd") ? EACHAM_BA_ORDER_ND : EACHAM_BA_ORDER_AUTO;
    // The second stream carries the work behind a batch's distance sweep, which is meant to run BESIDE the next batch's sweep. The
    // runtime deals its few hardware queues out to all streams of the process in creation order, so two streams may share one and
    // then run strictly one behind the other (the kernel traces of rounds 3 and 4 show exactly that for the first context of
    // bench.py: every kernel of the second stream behind the sweep). Priorities are no way out: a high-priority stream that lands
    // on a queue in use is slower than no second stream (measured, tools/experiments/c5_probe2.py). So the context TRIES: a 40 us
    // spin on the first stream, an empty kernel on the candidate; a candidate whose kernel does not finish well before the spin
    // shares the queue and is set aside (destroyed after the search, so that the next candidate gets the next queue).
    int stream2_mode = 0;   // 0 = probe for a stream on a queue of its own (default); 1 / -1 = high / low priority, no probe (A/B: EACHAM_STREAM2_PRIORITY)
    if (const char* sp = getenv("EACHAM_STREAM2_PRIORITY")) stream2_mode = !strcmp(sp, "high") ? 1 : !strcmp(sp, "low") ? -1 : !strcmp(sp, "noprobe") ? 2 : 0;
    auto bail = [&]() {   // whatever exists by now goes with the context
        if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        for (hipEvent_t e : {ctx->ev_tile[0], ctx->ev_tile[1], ctx->ev_fin[0], ctx->ev_fin[1], ctx->ev_join})
            if (e) (void)hipEventDestroy(e);
        if (ctx->flag_dev) (void)hipFree(ctx->flag_dev);
        delete ctx;
        return EACHAM_ERR_HIP;
    };
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail();
    if (stream2_mode == 1 || stream2_mode == -1) {
        int prio_least = 0, prio_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, stream2_mode > 0 ? prio_greatest : prio_least) != hipSuccess)
            return bail();
        ctx->stream2_attempt = -1;
    } else {
        hipStream_t rejected[4];
        int n_rejected = 0;
        hipEvent_t e_spin = nullptr, e_probe = nullptr;
        bool ok = hipEventCreate(&e_spin) == hipSuccess && hipEventCreate(&e_probe) == hipSuccess;
        for (int attempt = 0; ok && attempt < 5; ++attempt) {
            hipStream_t cand = nullptr;
            if (ctx->exp_stream2_cus > 0) {  // (A/B: n CUs spread evenly over the 256 — every (256 / n)-th bit of the mask)
                uint32_t cu_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const int stride = std::max(1, 256 / ctx->exp_stream2_cus);
                for (int cu = 0; cu < 256; cu += stride) cu_mask[cu >> 5] |= 1u << (cu & 31);
                if (hipExtStreamCreateWithCUMask(&cand, 8, cu_mask) != hipSuccess) { ok = false; break; }
            } else if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) { ok = false; break; }
            bool beside = stream2_mode == 2 || attempt == 4;   // (the last candidate is kept whatever it does)
            float ms = -1.f;   // from the probe's end to the spin's end
            if (!beside) {
                stream_spin_kernel<<<1, 64, 0, ctx->stream>>>(4000);   // 40 us of the 100 MHz wall clock
                ok = ok && hipEventRecord(e_spin, ctx->stream) == hipSuccess;
                stream_probe_kernel<<<1, 64, 0, cand>>>();
                ok = ok && hipEventRecord(e_probe, cand) == hipSuccess;
                ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess && hipStreamSynchronize(cand) == hipSuccess;
                ok = ok && hipEventElapsedTime(&ms, e_probe, e_spin) == hipSuccess;
                beside = ok && ms > 0.010f;
            }
            if (beside) {   // what the search decided stays with the context (eacham_ctx_stream2_info: the bench line reports it)
                ctx->stream2 = cand;
                ctx->stream2_attempt = attempt;
                ctx->stream2_lead_ms = ms;
                break;
            }
            rejected[n_rejected++] = cand;
        }
        for (int k = 0; k < n_rejected; ++k) (void)hipStreamDestroy(rejected[k]);
        if (e_spin) (void)hipEventDestroy(e_spin);
        if (e_probe) (void)hipEventDestroy(e_probe);
        if (!ok || !ctx->stream2) return bail();
    }
    if (
        hipEventCreateWithFlags(&ctx->ev_tile[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_tile[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fin[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fin[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void**)&ctx->flag_dev, 64 * sizeof(int)) != hipSuccess ||
        hipMemsetAsync(ctx->flag_dev, 0, 64 * sizeof(int), ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return bail();
    *out_ctx = ctx;
    return EACHAM_OK;
}

int eacham_ctx_stream2_info(const eacham_ctx* ctx, int* attempt, float* lead_ms) {
    if (!ctx) return EACHAM_ERR_INVALID;
    if (attempt) *attempt = ctx->stream2_attempt;
    if (lead_ms) *lead_ms = ctx->stream2_lead_ms;
    return EACHAM_OK;
}

int eacham_clear_descriptors(eacham_ctx* ctx) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& f : ctx->frames) {
        if (f.frag) (void)hipFree(f.frag);
        if (f.norm) (void)hipFree(f.norm);
        f = FrameHost();
    }
    ctx->frames.clear();
    ctx->ks_common = 0;
    ctx->kind_common = 0;
    ctx->frame_table_dirty = true;
    return EACHAM_OK;
}

void eacham_ctx_destroy(eacham_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& f : ctx->frames) {
        if (f.frag) (void)hipFree(f.frag);
        if (f.norm) (void)hipFree(f.norm);
    }
    if (ctx->frame_table_dev) (void)hipFree(ctx->frame_table_dev);
    if (ctx->flag_dev) (void)hipFree(ctx->flag_dev);
    if (ctx->pairs_safe) (void)hipFree(ctx->pairs_safe);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->io) (void)hipFree(ctx->io);
    if (ctx->io_host) (void)hipHostFree(ctx->io_host);
    for (auto& b : ctx->ba_pool) {
        if (b.dev) (void)hipFree(b.dev);
        if (b.pinned) (void)hipHostFree(b.pinned);
    }
    for (auto& sc : ctx->ba_scratch)
        if (sc.dev) (void)hipFree(sc.dev);
    for (auto& s : ctx->prof)
        for (auto& ev : s.events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    for (hipEvent_t e : {ctx->ev_tile[0], ctx->ev_tile[1], ctx->ev_fin[0], ctx->ev_fin[1], ctx->ev_join})
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* eacham_last_error(const eacham_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int eacham_ctx_sync(eacham_ctx* ctx) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int bad = 0;
    EACHAM_HIP_TRY(ctx, hipMemcpy(&bad, ctx->flag_dev + 1, sizeof(int), hipMemcpyDeviceToHost));
    if (bad) {
        EACHAM_HIP_TRY(ctx, hipMemset(ctx->flag_dev + 1, 0, sizeof(int)));
        return ctx->fail(EACHAM_ERR_INVALID, "a device-side pair list named a frame that is not resident (those pairs produced no match)");
    }
    return EACHAM_OK;
}

void* eacham_ctx_stream(eacham_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int eacham_profile_enable(eacham_ctx* ctx, int on) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->profile = on != 0;
    return EACHAM_OK;
}

int eacham_profile_reset(eacham_ctx* ctx) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& s : ctx->prof) {
        s.used = 0;
        s.launches = 0;
        s.total_ms = 0.0;
    }
    return EACHAM_OK;
}

int eacham_profile_get(eacham_ctx* ctx, int kernel_id, int64_t* launches, double* total_ms) {
    if (!ctx || kernel_id < 0 || kernel_id >= EACHAM_KERNEL_COUNT) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    profile_drain(ctx);
    if (launches) *launches = ctx->prof[kernel_id].launches;
    if (total_ms) *total_ms = ctx->prof[kernel_id].total_ms;
    return EACHAM_OK;
}

}  // extern "C"
