// context.hpp — per-device context shared by the matcher and the bundle adjuster.
// One context = one HIP device + one stream; the C-ABI in include/eacham_hip.h serialises calls on
// a context with `mu` so the reference's concurrent Match() pattern (apps/sfm/main.cpp:98-109)
// stays legal.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/eacham_hip.h"

namespace eacham {

// Descriptor of one resident frame as the kernels see it (device-side table entry).
// int8 frames are stored PARITY-SORTED: rows whose centred squared norm is even come first (stable),
// padded to whole 32-row tiles, then the odd ones; `orig`/`pos` translate between stored position
// and the caller's row index (see matcher.hip). fp32 frames keep the caller's order (orig = pos = null).
struct FrameDev {
    const int4* frag;  // [ntiles][KS][64] 16-byte MFMA operand fragments (int8, centred by -128)
    const int* norm;   // int8: floor(|c|^2 / 2) per stored position (query role, MFMA C-init); fp32: |x|^2 as float bits
    const int* normb;  // int8: floor(|c|^2 / 2) + sum(c) per stored position (train role)
    const int* orig;   // int8: stored position -> caller's row (-1 = padding)
    const int* pos;    // int8: caller's row -> stored position
    const int* meta;   // int8: {tiles of the even class, tiles in use}; written by the upload kernels
    int n;             // real rows
    int ntiles;        // allocated 32-row tiles (upper bound of meta[1] for int8 frames)
    int resident;      // 0: no descriptors uploaded under this id (device-side pair lists naming it are neutralised)
};

struct FrameHost {
    int4* frag = nullptr;
    int* norm = nullptr;   // one allocation: norm | normb | orig | pos | scratch (int8 path)
    int* normb = nullptr;
    int* orig = nullptr;
    int* pos = nullptr;
    int* meta = nullptr;
    int n = -1;  // -1 = not resident
    int dim = 0;
    int ks = 0;
    int ntiles = 0;      // allocated tiles
    int tiles_used = 0;  // tiles in use (<= ntiles), known after sync_frame_table
};

// One arena of the bundle adjuster: every device array of a prepared problem is carved out of ONE allocation, and the
// arena (with its block of pinned host scalars) goes back to the context's pool when the problem is released. The
// reference calls RefineBA for a new local window after every frame (apps/sfm/main.cpp:207): ~60 hipMalloc + hipFree +
// a pinned allocation per call were 0.8 ms of a 1.9 ms window.
struct BaBlock {
    void* dev = nullptr;
    size_t bytes = 0;
    double* pinned = nullptr;
    bool busy = false;
};

// Scratch of the device-side problem construction (temporaries of its sorts and scans): kept with the context, grown on demand.
struct BaScratch {
    void* dev = nullptr;
    size_t bytes = 0;
};

struct ProfileSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t used = 0;
    int64_t launches = 0;
    double total_ms = 0.0;
};

}  // namespace eacham

struct eacham_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // finalize/compaction of batch i runs here beside the tile kernel of batch i+1
    hipEvent_t ev_tile[2] = {nullptr, nullptr}, ev_fin[2] = {nullptr, nullptr}, ev_join = nullptr;
    std::mutex mu;
    std::string err;

    // descriptor store
    std::vector<eacham::FrameHost> frames;
    eacham::FrameDev* frame_table_dev = nullptr;
    int frame_table_cap = 0;
    bool frame_table_dirty = true;
    int* flag_dev = nullptr;  // [0] = non-integer descriptor seen, [1] = a device-side pair list named a frame that is not
                              // resident, [8..9] = meta of the empty stand-in frame (zeros), [16..] scratch
    int ks_common = 0;        // k-step class shared by all resident frames (0 = none yet)
    int kind_common = 0;      // 0 = int8 fragments (matcher.hip), 1 = fp32 fragments (matcher_f32.hip)
    void* last_matches = nullptr;  // per-pair match lists of the last run (directed API reads them back)

    // matcher workspace (grown on demand, never inside a timed launch sequence after warm-up)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    int2* pairs_safe = nullptr;  // sanitised copy of the caller's device-side pair list
    int pairs_safe_cap = 0;
    void* io = nullptr;  // staging for the host-pointer entry points
    size_t io_bytes = 0;
    void* io_host = nullptr;  // pinned mirror of the head of `io`: the small arrays of a call travel as ONE copy each way (IoPack)
    size_t io_host_bytes = 0;

    std::vector<eacham::BaBlock> ba_pool;  // arenas of released BA problems, reused by the next eacham_ba_prepare
    eacham::BaScratch ba_scratch[4];
    int ba_prepare_mode = 0;  // EACHAM_BA_PREPARE=host|device (diagnostic / tests: force one form of the structure construction;
                              // default: device for >= 65536 observations), read at create
    int stream2_attempt = -1;        // which candidate of the second-stream search was kept (0..4; 4 = the last, kept unprobed; -1 = no search)
    float stream2_lead_ms = -1.f;    // how long before the spin's end the probe on it finished (> 0.010: a hardware queue of its own)
    bool io_busy = false;            // an IoPack call has not reached its finish(): copies out of the pinned mirror may be in flight
    bool ba_groups_lds_set = false;  // ba_schur_groups has been granted its dynamic LDS size on this context's device
    bool ba_dense_lds_set = false;   // the same for ba_schur_dense
    int ba_window_rows = 0;   // EACHAM_BA_WINDOW_ROWS=<n> (diagnostic): rows per group of the dense form instead of ba_window.hpp's choice
    int ba_group_rows = 0;    // EACHAM_BA_GROUP_ROWS=<n> (diagnostic): rows per landmark group instead of ba_groups.hpp's choice
    int ba_schur_mode = 0;    // EACHAM_BA_SCHUR=groups|pairs|dense (diagnostic / tests). 0 = by problem size: the landmark groups of ba_groups.hpp for the
                              // problems eacham_ba_prepare builds on the device, the pair lists of rounds 1-4 (ba_schur_pairs) for the
                              // small ones it builds with host loops (a local window: building the group structure on the host costs
                              // more than the 13 us per LM iteration it saves there); 1 = groups whenever they apply, 2 = always pairs,
                              // 3 = the dense form of ba_window.hpp wherever it applies (by default: eacham_ba_solve with the direct LM solve)
    int ba_ordering = 0;  // EACHAM_BA_ORDERING=natural|rcm|nd read ONCE at eacham_ctx_create (diagnostic override of
                          // eacham_ba_problem.ordering == AUTO); nothing on the solve path reads the environment
    int ba_lpl_lin = 0;             // EACHAM_BA_LPL_LIN=1|2|4|8 (diagnostic: lanes per landmark of the linearisation), read at create
    int ba_lpl_step = 0;            // EACHAM_BA_LPL_STEP=1|2|4|8 (diagnostic: lanes per landmark of the step's tail kernels), read at create
    bool exp_no_coltop2 = false;    // EACHAM_EXP_NO_COLTOP2 (diagnostic, WRONG RESULTS: the tile sweep without its column direction — timing only)
    bool exp_all_candidates = false;  // EACHAM_EXP_ALL_CANDIDATES (diagnostic, timing only: every row is a candidate of the column pass)
    int exp_sweep_prio = 0;           // EACHAM_EXP_SWEEP_PRIO=1..3 (diagnostic A/B): s_setprio of the sweep's waves (the candidate pass beside it stays at 0)
    int exp_stream2_cus = 0;          // EACHAM_EXP_STREAM2_CUS=<n> (diagnostic A/B): the second stream may use n of the 256 CUs only (hipExtStreamCreateWithCUMask)
    int match_sweep_form = 0;         // EACHAM_MATCH_SWEEP_FORM=exact|bound (diagnostic A/B, tests): the lean form's row sweep keeps every row's exact top-2 (1), or
                                      // runs its bound form + the exact pass over the rows left open (2); 0 = by descriptor dimension (bound up to 128-D)
    bool match_tile_sweep = false;    // EACHAM_MATCH_TILE_SWEEP (diagnostic A/B: the lean form's sweep by match_tile_kernel, the first round-4 form)
    bool match_full_columns = false;  // EACHAM_MATCH_FULL_COLUMNS (diagnostic A/B: every column's top-2 from the sweep, the round-1..3 form)
    int match_budget_mb = 1024;     // EACHAM_MATCH_BUDGET_MB (diagnostic: workspace budget of one batch of pairs), read at create
    bool match_no_overlap = false;  // EACHAM_NO_OVERLAP (diagnostic: finalize on the tile kernel's stream), read at create

    // profiling
    bool profile = false;
    eacham::ProfileSlot prof[EACHAM_KERNEL_COUNT];

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

#define EACHAM_HIP_TRY(ctx, expr)                                                              \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return (ctx)->fail(EACHAM_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                 \
                               hipGetErrorString(e_), __FILE__, __LINE__);                     \
    } while (0)

namespace eacham {

int ensure_workspace(eacham_ctx* ctx, size_t bytes);
int ensure_io(eacham_ctx* ctx, size_t bytes);
int ensure_io_host(eacham_ctx* ctx, size_t bytes);

// The host-pointer entry points of the estimators, the triangulation and the graph query are called once or more per frame of
// the incremental loop with a handful of small arrays each way (points, a few models, counts): as separate copies from pageable
// memory every one of them is a submission and a wait of ~10 us (25 copies per solvePnPRansac of the loop). IoPack lays the
// arrays that are small into a pinned mirror of the device staging buffer at the SAME offsets and moves each direction in one
// copy (the bytes between two packed arrays travel along); arrays above PACK_MAX keep their own direct copy.
struct IoPack {
    static constexpr size_t PACK_MAX = 256 * 1024;
    eacham_ctx* ctx;
    char* dev;
    hipStream_t st;
    size_t in_lo = ~(size_t)0, in_hi = 0, out_lo = ~(size_t)0, out_hi = 0;
    size_t direct_lo = ~(size_t)0, direct_hi = 0;  // what has been copied directly so far: a packed span must not cover it
    struct Out { void* dst; size_t off, bytes; };
    struct In { size_t off, bytes; };
    Out outs[8];
    In ins[16];
    int n_outs = 0, n_ins = 0;
    IoPack(eacham_ctx* c, hipStream_t s) : ctx(c), dev((char*)c->io), st(s) {
        // a call that returned on an error may have left a copy out of the mirror in flight: wait before writing into it again
        if (ctx->io_busy) (void)hipStreamSynchronize(st);
        ctx->io_busy = true;
    }
    // host -> device: packed (memcpy now, one copy at flush_in) or direct
    int in(size_t off, const void* src, size_t bytes) {
        if (bytes == 0) return EACHAM_OK;
        if (bytes <= PACK_MAX && off + bytes <= ctx->io_host_bytes && n_ins < 16) {
            memcpy((char*)ctx->io_host + off, src, bytes);
            in_lo = std::min(in_lo, off), in_hi = std::max(in_hi, off + bytes);
            ins[n_ins++] = In{off, bytes};
            return EACHAM_OK;
        }
        // (what has been packed so far goes first; the direct copy is remembered: a later packed span that would cover it — the
        // gaps of the mirror hold stale bytes — is sent piece by piece instead)
        if (int rc = flush_in()) return rc;
        EACHAM_HIP_TRY(ctx, hipMemcpyAsync(dev + off, src, bytes, hipMemcpyHostToDevice, st));
        direct_lo = std::min(direct_lo, off), direct_hi = std::max(direct_hi, off + bytes);
        return EACHAM_OK;
    }
    int flush_in() {
        if (in_hi > in_lo) {
            if (in_lo < direct_hi && direct_lo < in_hi) {
                for (int k = 0; k < n_ins; ++k)
                    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(dev + ins[k].off, (char*)ctx->io_host + ins[k].off, ins[k].bytes, hipMemcpyHostToDevice, st));
            } else {
                EACHAM_HIP_TRY(ctx, hipMemcpyAsync(dev + in_lo, (char*)ctx->io_host + in_lo, in_hi - in_lo, hipMemcpyHostToDevice, st));
            }
        }
        in_lo = ~(size_t)0, in_hi = 0, n_ins = 0;
        return EACHAM_OK;
    }
    // device -> host: registered now, moved by finish()
    int out(void* dst, size_t off, size_t bytes) {
        if (bytes == 0 || !dst) return EACHAM_OK;
        if (bytes <= PACK_MAX && off + bytes <= ctx->io_host_bytes && n_outs < 8) {
            outs[n_outs++] = Out{dst, off, bytes};
            out_lo = std::min(out_lo, off), out_hi = std::max(out_hi, off + bytes);
            return EACHAM_OK;
        }
        EACHAM_HIP_TRY(ctx, hipMemcpyAsync(dst, dev + off, bytes, hipMemcpyDeviceToHost, st));
        return EACHAM_OK;
    }
    int finish() {   // one copy back, the stream's synchronisation, the hand-over to the caller's arrays
        if (out_hi > out_lo) EACHAM_HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->io_host + out_lo, dev + out_lo, out_hi - out_lo, hipMemcpyDeviceToHost, st));
        EACHAM_HIP_TRY(ctx, hipStreamSynchronize(st));
        for (int k = 0; k < n_outs; ++k) memcpy(outs[k].dst, (char*)ctx->io_host + outs[k].off, outs[k].bytes);
        ctx->io_busy = false;  // (an early error return leaves it set: the next IoPack waits for the stream first)
        return EACHAM_OK;
    }
};
int sync_frame_table(eacham_ctx* ctx);
// copies `pairs` into the workspace tail with every pair that names a missing frame redirected to the empty
// stand-in entry frames[n_frames] (and flags it); returns the sanitised device pointer in *out
int sanitize_pairs(eacham_ctx* ctx, const int2* pairs_dev, int npairs, const int2** out);

// matcher_f32.hip
int upload_frame_f32(eacham_ctx* ctx, int frame_id, const float* src_dev, int n, int dim);
int run_match_f32(eacham_ctx* ctx, const int2* pairs_dev, int npairs, double ratio, int min_dir, int min_mutual, int mode,
                  int* counts_dev, long long* offsets_dev, uint2* edges_dev, long long edge_cap, long long* total_dev,
                  int4* stats_dev);
// matcher.hip
void launch_scan_counts(eacham_ctx* ctx, const int* counts, int n, long long* offsets, long long* total, int first, int is_last);
void launch_compact_edges(eacham_ctx* ctx, int nb, const uint2* matches, const int* counts, const long long* offsets,
                          int row_stride, uint2* edges, long long edge_cap);

// RAII: records a start/stop HIP event pair around a launch sequence when profiling is on.
struct ProfileScope {
    eacham_ctx* ctx;
    int id;
    hipEvent_t stop = nullptr;
    hipStream_t stream;
    bool range = false;  // a ROCTx range is open (EACHAM_ROCTX=1)
    ProfileScope(eacham_ctx* c, int kernel_id, hipStream_t on = nullptr);
    ~ProfileScope();
};

}  // namespace eacham
