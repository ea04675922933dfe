// matcher_f32.hip — float-descriptor path of the matcher (SuperPoint / LightGlue-style 256-D floats,
// modules/onnx/lightglue/feature/Types.h:11-14): all-pairs "L2 via dot product" on the f32 MFMA.
//
// Same contract as the int8 path in matcher.hip (FeatureMatcherFlann::Match semantics + the pair loop
// of apps/sfm/main.cpp:84-147); only the arithmetic differs:
//     d2(q,t) = max(fma(-2, a_q.b_t, |a_q|^2 + |b_t|^2), 0)      all fp32
// with a.b accumulated by v_mfma_f32_32x32x2_f32, which is bit-for-bit a k-ordered fmaf chain, and the
// norms accumulated by the same chain. The CPU restatement used by the tests (its dot-product mode) evaluates exactly this,
// so indices are bit-exact against it; agreement with the sum-of-squared-differences form is
// reported by tests/test_match_gpu.py. Ties resolve to the lower index like the reference's scan.
//
// Layout in HBM: fragf[tile = row/32][k2 = k/2][lane = 32*(k%2) + row%32] fp32 — one wave-wide
// 4-byte access is 256 B contiguous and is the A/B operand image of the 32x32x2 MFMA; norms fp32,
// padding rows carry PAD_F so they never win.
// Kernel: workgroup = 4 waves x 32 query rows (A fragments stay in VGPRs: D/2 registers), train tiles
// of 32 rows (D x 128 B) stream through a 2-slot LDS ring filled by LDS-DMA; the MFMA dominates
// (D/2 instructions of 64 cycles per 32x32 tile), the (value, index) top-2 epilogue is minor.
#include "context.hpp"

#include <algorithm>

namespace eacham {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef const float __attribute__((address_space(1)))* gfloat_t;

constexpr float PAD_F = 1.0e30f;      // norm of padding rows
constexpr float INVALID_F = 1.0e29f;  // d2 at or above this marks "no such neighbour"
constexpr float INIT_F = 3.0e38f;
constexpr int F_THREADS = 256, F_WAVES = 4;

// ---- upload ---------------------------------------------------------------------------------------
__global__ void pack_f32_kernel(const float* __restrict__ src, int n, int dim, int D2, int npad, float* __restrict__ frag) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)npad * D2 * 2) return;
    const int lane = (int)(idx % 64);
    const long long rest = idx / 64;
    const int k2 = (int)(rest % D2), tile = (int)(rest / D2);
    const int row = tile * 32 + (lane & 31), k = 2 * k2 + (lane >> 5);
    frag[idx] = (row < n && k < dim) ? src[(size_t)row * dim + k] : 0.0f;
}
// |x|^2 as a k-ordered fmaf chain (one thread per row)
__global__ void norm_f32_kernel(const float* __restrict__ src, int n, int dim, int npad, float* __restrict__ norm) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= npad) return;
    float s = PAD_F;
    if (row < n) {
        s = 0.0f;
        for (int k = 0; k < dim; ++k) s = __fmaf_rn(src[(size_t)row * dim + k], src[(size_t)row * dim + k], s);
    }
    norm[row] = s;
}

// ---- K1f: distance tiles + fused row/column top-2 ---------------------------------------------------
// rowres[p][q]       = {bits(v1), col1, bits(v2), 0}        final over all columns
// colpart[p][wb][c]  = {bits(v1), row1, bits(v2), 0}        over the 32 rows of tile wb of frame A
template <int D2>
__global__ __launch_bounds__(F_THREADS, 2) void match_tile_f32_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs, int wgs_per_pair,
    int4* __restrict__ rowres, int4* __restrict__ colpart, int wb_stride, int row_stride) {
    constexpr int TILE_F = 64 * D2;  // floats per 32-row tile
    constexpr int SLABS_F = F_WAVES * 3 * 32 * 33;
    constexpr int LDS_F = 2 * TILE_F > SLABS_F ? 2 * TILE_F : SLABS_F;
    __shared__ float sMem[LDS_F];    // 2-slot tile ring; after the sweep reused as the row slabs
    float (*sB)[TILE_F] = reinterpret_cast<float (*)[TILE_F]>(sMem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 31, h = lane >> 5;
    const int p = blockIdx.x / wgs_per_pair, rb = blockIdx.x % wgs_per_pair;
    const int2 pr = pairs[p];
    const FrameDev A = frames[pr.x], B = frames[pr.y];
    if (rb * F_WAVES >= A.ntiles) return;  // workgroup-uniform
    const int wb = rb * F_WAVES + wave;    // 32-row tile of frame A owned by this wave
    const bool active = wb < A.ntiles;
    const int wbc = active ? wb : 0;
    const int T = B.ntiles;
    // the frame table is read through vector memory: the pointers of the train frame, live for the whole sweep, are moved
    // to scalar registers (they are workgroup-uniform)
    auto uniform_ptr = [](const void* q) {
        const unsigned long long u = (unsigned long long)q;
        return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32 |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)u);
    };
    const gfloat_t Af = (gfloat_t)A.frag, Bf = (gfloat_t)uniform_ptr(B.frag);
    const gfloat_t An = (gfloat_t)A.norm, Bn = (gfloat_t)uniform_ptr(B.norm);

    float a[D2];
#pragma unroll
    for (int k2 = 0; k2 < D2; ++k2) a[k2] = Af[((size_t)wbc * D2 + k2) * 64 + lane];
    float nar[16], rv1[16], rv2[16];
    int rt1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        nar[r] = An[32 * wbc + (r & 3) + 8 * (r >> 2) + 4 * h];
        rv1[r] = INIT_F;
        rv2[r] = INIT_F;
        rt1[r] = 0;
    }
    constexpr int PIECES = TILE_F / 256;  // 1 KiB LDS-DMA pieces per tile
    auto stage_tile = [&](int tile, int slot) {
#pragma unroll
        for (int i = 0; i < (PIECES + F_WAVES - 1) / F_WAVES; ++i) {
            const int piece = wave + i * F_WAVES;
            if (piece < PIECES)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(Bf + (size_t)tile * TILE_F + piece * 256 + lane * 4),
                    (__attribute__((address_space(3))) void*)(&sB[slot][piece * 256]), 16, 0, 0);
        }
    };
    if (T > 0) stage_tile(0, 0);
    float nb_cur = T > 0 ? Bn[cl] : 0.0f;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    int4* cp;  // wave-uniform: kept as a scalar base (the kernel sits at the 256-register limit of two waves per SIMD)
    {
        const unsigned long long u = (unsigned long long)(colpart + ((size_t)p * wb_stride + wb) * row_stride);
        cp = (int4*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)u));
    }
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        const float nbc = nb_cur;
        const int t1 = min(t + 1, T - 1);
        nb_cur = Bn[32 * t1 + cl];
        stage_tile(t1, cur ^ 1);  // that slot was last read before the previous barrier
        if (active) {
            v16f acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k2 = 0; k2 < D2; ++k2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2], sB[cur][k2 * 64 + lane], acc, 0, 0, 0);
            float cv1 = INIT_F, cv2 = INIT_F;
            int cr1 = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d2 = fmaxf(__fmaf_rn(-2.0f, acc[r], nar[r] + nbc), 0.0f);
                const bool lr = d2 < rv1[r];  // ascending t: strict '<' keeps the lower column
                rv2[r] = lr ? rv1[r] : fminf(rv2[r], d2);
                rt1[r] = lr ? t : rt1[r];
                rv1[r] = fminf(rv1[r], d2);
                const bool lc = d2 < cv1;     // ascending rows within the lane
                cv2 = lc ? cv1 : fminf(cv2, d2);
                cr1 = lc ? (r & 3) + 8 * (r >> 2) + 4 * h : cr1;
                cv1 = fminf(cv1, d2);
            }
            // the two lane halves hold interleaved rows of the same column: (value, row) lexicographic merge
            const float ov1 = __shfl_xor(cv1, 32), ov2 = __shfl_xor(cv2, 32);
            const int or1 = __shfl_xor(cr1, 32);
            const bool take = ov1 < cv1 || (ov1 == cv1 && or1 < cr1);
            const float n1 = take ? ov1 : cv1, n2 = take ? fminf(cv1, ov2) : fminf(cv2, ov1);
            const int nr = take ? or1 : cr1;
            if (h == 0) cp[(unsigned)(32 * t + cl)] = make_int4(__float_as_int(n1), 32 * wb + nr, __float_as_int(n2), 0);
        }
        __syncthreads();
    }
    if (!active) return;
    // every wave is past the last barrier: the tile ring is dead, reuse it for the row transposition
    float* sv1 = sMem + wave * (3 * 32 * 33);
    float* sv2 = sv1 + 32 * 33;
    int* sc1 = (int*)(sv2 + 32 * 33);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        sv1[row * 33 + cl] = rv1[r];
        sv2[row * 33 + cl] = rv2[r];
        sc1[row * 33 + cl] = 32 * rt1[r] + cl;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float b1 = INIT_F, b2 = INIT_F;
    int bc = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int idx = cl * 33 + 16 * h + j;
        const float v1 = sv1[idx], v2 = sv2[idx];
        const int c = sc1[idx];
        const bool lt = v1 < b1 || (v1 == b1 && c < bc);
        b2 = lt ? fminf(b1, v2) : fminf(b2, v1);
        bc = lt ? c : bc;
        b1 = lt ? v1 : b1;
    }
    const float o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
    const int oc = __shfl_xor(bc, 32);
    if (h == 0) {
        const bool lt = o1 < b1 || (o1 == b1 && oc < bc);
        const float f1 = lt ? o1 : b1, f2 = lt ? fminf(b1, o2) : fminf(b2, o1);
        rowres[(size_t)p * row_stride + 32 * wb + cl] = make_int4(__float_as_int(f1), lt ? oc : bc, __float_as_int(f2), 0);
    }
}

// ---- K2f: merge, ratio test, mutual check, thresholds, ordered compaction -----------------------------
__device__ __forceinline__ bool ratio_pass_f32(float d2_best, float d2_second, double ratio) {
    const float q = __fdiv_rn(__fsqrt_rn(d2_best), __fsqrt_rn(d2_second));  // FeatureMatcherFlann.cpp:23
    return (double)q < ratio;
}

constexpr int FIN_T = 256;
__global__ __launch_bounds__(FIN_T) void match_finalize_f32_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs, const int4* __restrict__ rowres,
    const int4* __restrict__ colpart, int wb_stride, int row_stride, double ratio, int min_dir, int min_mutual,
    int mode, uint2* __restrict__ out_matches, int* __restrict__ counts, int4* __restrict__ stats) {
    extern __shared__ int smem[];
    const int tid = threadIdx.x, p = blockIdx.x;
    const int2 pr = pairs[p];
    const FrameDev A = frames[pr.x], B = frames[pr.y];
    int* fwd = smem;
    int* bwd = smem + row_stride;
    __shared__ int s_cnt[2];
    __shared__ int s_scan[FIN_T];
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    int c12 = 0, c21 = 0;
    for (int q = tid; q < A.n; q += FIN_T) {
        const int4 r = rowres[(size_t)p * row_stride + q];
        const float v1 = __int_as_float(r.x), v2 = __int_as_float(r.z);
        const bool ok = B.ntiles > 0 && v2 < INVALID_F && ratio_pass_f32(v1, v2, ratio);
        fwd[q] = ok ? r.y : -1;
        c12 += ok;
    }
    for (int c = tid; c < B.n; c += FIN_T) {
        float v1 = INIT_F, v2 = INIT_F;
        int r1 = -1;
        const int4* cp = colpart + (size_t)p * wb_stride * row_stride + c;
        for (int wb = 0; wb < A.ntiles; ++wb) {  // ascending rows; strict '<' keeps the lower row on ties
            const int4 e = cp[(size_t)wb * row_stride];
            const float a1 = __int_as_float(e.x), a2 = __int_as_float(e.z);
            if (a1 < v1) {
                v2 = fminf(v1, a2);
                v1 = a1;
                r1 = e.y;
            } else {
                v2 = fminf(v2, a1);
            }
        }
        const bool ok = v2 < INVALID_F && ratio_pass_f32(v1, v2, ratio);
        bwd[c] = ok ? r1 : -1;
        c21 += ok;
    }
    atomicAdd(&s_cnt[0], c12);
    atomicAdd(&s_cnt[1], c21);
    __syncthreads();
    uint2* out = out_matches + (size_t)p * row_stride;
    int base = 0;
    for (int q0 = 0; q0 < A.n; q0 += FIN_T) {
        const int q = q0 + tid;
        const int t = q < A.n ? fwd[q] : -1;
        const bool keep = t >= 0 && (mode == 1 || bwd[t] == q);
        s_scan[tid] = keep;
        __syncthreads();
        for (int off = 1; off < FIN_T; off <<= 1) {
            const int v = tid >= off ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        if (keep) out[base + s_scan[tid] - 1] = make_uint2((unsigned)q, (unsigned)t);
        base += s_scan[FIN_T - 1];
        __syncthreads();
    }
    if (tid == 0) {
        const int n12 = s_cnt[0], n21 = s_cnt[1];
        const bool edge = n12 >= min_dir && n21 >= min_dir && base > min_mutual;
        counts[p] = mode == 1 ? base : (edge ? base : 0);
        if (stats) stats[p] = make_int4(n12, n21, base, edge ? 1 : 0);
    }
}

// ---- host ------------------------------------------------------------------------------------------
static int d2_for_dim(int dim) {
    if (dim <= 0 || dim > 256) return 0;
    return dim <= 64 ? 32 : (dim <= 128 ? 64 : 128);
}

int upload_frame_f32(eacham_ctx* ctx, int frame_id, const float* src_dev, int n, int dim) {
    if (frame_id < 0 || frame_id >= (1 << 20)) return ctx->fail(EACHAM_ERR_INVALID, "frame_id %d out of range", frame_id);
    if (n < 0) return ctx->fail(EACHAM_ERR_INVALID, "negative row count");
    const int D2 = d2_for_dim(dim);
    if (!D2) return ctx->fail(EACHAM_ERR_UNSUPPORTED, "descriptor dim %d: need 1..256", dim);
    if (ctx->ks_common && (ctx->kind_common != 1 || ctx->ks_common != D2))
        return ctx->fail(EACHAM_ERR_UNSUPPORTED, "all resident frames must share one descriptor kind (int8 / f32) and dim class");
    int ntiles = (n + 31) / 32;
    ntiles = (ntiles + 3) / 4 * 4;
    if (ntiles > 512) return ctx->fail(EACHAM_ERR_UNSUPPORTED, "frame has %d rows; this build supports <= 16384", n);
    if ((size_t)frame_id >= ctx->frames.size()) ctx->frames.resize(frame_id + 1);
    FrameHost& f = ctx->frames[frame_id];
    if (f.frag || f.norm) {
        EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (f.frag) (void)hipFree(f.frag);
        if (f.norm) (void)hipFree(f.norm);
        f = FrameHost();
    }
    const int npad = ntiles * 32;
    if (npad > 0) {
        EACHAM_HIP_TRY(ctx, hipMalloc((void**)&f.frag, (size_t)npad * D2 * 2 * sizeof(float)));
        EACHAM_HIP_TRY(ctx, hipMalloc((void**)&f.norm, (size_t)npad * sizeof(float)));
        f.normb = f.norm;
        const long long work = (long long)npad * D2 * 2;
        pack_f32_kernel<<<(unsigned)((work + 255) / 256), 256, 0, ctx->stream>>>(src_dev, n, dim, D2, npad, (float*)f.frag);
        norm_f32_kernel<<<(npad + 255) / 256, 256, 0, ctx->stream>>>(src_dev, n, dim, npad, (float*)f.norm);
        EACHAM_HIP_TRY(ctx, hipGetLastError());
    }
    f.n = n;
    f.dim = dim;
    f.ks = D2;
    f.ntiles = ntiles;
    ctx->ks_common = D2;
    ctx->kind_common = 1;
    ctx->frame_table_dirty = true;
    return EACHAM_OK;
}

int run_match_f32(eacham_ctx* ctx, const int2* pairs_dev, int npairs, double ratio, int min_dir, int min_mutual, int mode,
                  int* counts_dev, long long* offsets_dev, uint2* edges_dev, long long edge_cap, long long* total_dev,
                  int4* stats_dev) {
    int max_tiles = 4;
    for (const auto& f : ctx->frames)
        if (f.n >= 0) max_tiles = std::max(max_tiles, f.ntiles);
    const int row_stride = max_tiles * 32, wb_stride = max_tiles;
    const int wgs_per_pair = (max_tiles + F_WAVES - 1) / F_WAVES;
    const size_t per_pair = (size_t)row_stride * sizeof(int4) + (size_t)wb_stride * row_stride * sizeof(int4) +
                            (size_t)row_stride * sizeof(uint2) + sizeof(int);
    const int batch = (int)std::max<size_t>(1, std::min<size_t>(((size_t)1 << 30) / per_pair, (size_t)npairs));
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t off_rowres = 0;
    const size_t off_colpart = align((size_t)batch * row_stride * sizeof(int4));
    const size_t off_matches = align(off_colpart + (size_t)batch * wb_stride * row_stride * sizeof(int4));
    const size_t total = align(off_matches + (size_t)batch * row_stride * sizeof(uint2));
    int rc = ensure_workspace(ctx, total);
    if (rc) return rc;
    char* ws = (char*)ctx->ws;
    ctx->last_matches = ws + off_matches;
    const size_t fin_smem = (size_t)2 * row_stride * sizeof(int);
    if (fin_smem > 48 * 1024)
        EACHAM_HIP_TRY(ctx, hipFuncSetAttribute((const void*)match_finalize_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_smem));
    for (int first = 0; first < npairs; first += batch) {
        const int nb = std::min(batch, npairs - first);
        const int2* pb = pairs_dev + first;
        {
            ProfileScope ps(ctx, EACHAM_KERNEL_MATCH_TILE);
            int4* rr = (int4*)(ws + off_rowres);
            int4* cp = (int4*)(ws + off_colpart);
            switch (ctx->ks_common) {
                case 32: match_tile_f32_kernel<32><<<nb * wgs_per_pair, F_THREADS, 0, ctx->stream>>>(ctx->frame_table_dev, pb, wgs_per_pair, rr, cp, wb_stride, row_stride); break;
                case 64: match_tile_f32_kernel<64><<<nb * wgs_per_pair, F_THREADS, 0, ctx->stream>>>(ctx->frame_table_dev, pb, wgs_per_pair, rr, cp, wb_stride, row_stride); break;
                default: match_tile_f32_kernel<128><<<nb * wgs_per_pair, F_THREADS, 0, ctx->stream>>>(ctx->frame_table_dev, pb, wgs_per_pair, rr, cp, wb_stride, row_stride); break;
            }
        }
        const bool csr = offsets_dev != nullptr;  // mode 1 without offsets: the single directed pair of eacham_match_pair
        int* cnt = csr ? counts_dev + first : counts_dev;
        {
            ProfileScope ps(ctx, EACHAM_KERNEL_MATCH_FINALIZE);
            match_finalize_f32_kernel<<<nb, FIN_T, fin_smem, ctx->stream>>>(
                ctx->frame_table_dev, pb, (const int4*)(ws + off_rowres), (const int4*)(ws + off_colpart), wb_stride, row_stride,
                ratio, min_dir, min_mutual, mode, (uint2*)(ws + off_matches), cnt, stats_dev ? stats_dev + first : nullptr);
            if (csr) {
                launch_scan_counts(ctx, cnt, nb, offsets_dev, total_dev, first, first + nb == npairs);
                launch_compact_edges(ctx, nb, (const uint2*)(ws + off_matches), cnt, offsets_dev + first, row_stride, edges_dev, edge_cap);
            }
        }
        EACHAM_HIP_TRY(ctx, hipGetLastError());
    }
    return EACHAM_OK;
}

}  // namespace eacham

using namespace eacham;

extern "C" int eacham_upload_descriptors_f32(eacham_ctx* ctx, int frame_id, const float* rowmajor, int n, int dim) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (n > 0 && !rowmajor) return ctx->fail(EACHAM_ERR_INVALID, "null descriptor pointer");
    if (n < 0 || dim <= 0) return ctx->fail(EACHAM_ERR_INVALID, "bad descriptor shape %d x %d", n, dim);
    const size_t bytes = (size_t)n * dim * sizeof(float);
    int rc = ensure_io(ctx, std::max<size_t>(bytes, 256));
    if (rc) return rc;
    if (bytes) EACHAM_HIP_TRY(ctx, hipMemcpyAsync(ctx->io, rowmajor, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = upload_frame_f32(ctx, frame_id, (const float*)ctx->io, n, dim);
    if (rc) return rc;
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the staging buffer is reused by the next call
    return EACHAM_OK;
}
