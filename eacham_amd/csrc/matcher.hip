// matcher.hip — exact all-pairs descriptor matching for gfx950 (MI355X).
//
// Replaces   FeatureMatcherFlann::Match   modules/base/features/FeatureMatcherFlann.cpp:14-30
//            pair loop + mutual check     apps/sfm/main.cpp:84-147
//
// Data layout in HBM ("fragment-major int8"): descriptors are integers in [0,255] (OpenCV SIFT),
// stored centred (x-128) as int8 in the exact register image of v_mfma_i32_32x32x32_i8 operands:
//     frag[tile = row/32][ks = k/32][lane = 32*((k%32)/16) + row%32] = 16 bytes  (k%16 ascending)
// so one wave-wide 16-byte load is 1 KiB contiguous and needs no LDS swizzle. The same image
// serves as the A operand (queries, rows of the distance tile) and the B operand (train, columns).
//
// Arithmetic (all integer, hence independent of summation order => bit-exact vs the CPU oracle):
//     d2(q,t) = |a_q|^2 + |b_t|^2 - 2 a_q.b_t = 2 H + pa + pb
//     H = floor(|a_q|^2 / 2) + floor(|b_t|^2 / 2) - a_q.b_t   (int8 MFMA, int32 accumulate, C-init;
//         stored with a +1 bias so that it is never negative)
//     pa, pb = parities of the two squared norms
// One int32 key per distance serves BOTH directions:
//     key = (H << 8) | (pb << 7) | column-tile index
// Row direction (q -> best t): pa is constant, keys order by (2H + pb, tile) — exact, ties to the
// lower column. Column direction (t -> best q): pb and the tile are constant, keys order by H, which
// is the order of d2 as long as pa is constant among the rows compared — so frames are stored
// PARITY-SORTED (even squared norms first, each class padded to whole 32-row tiles): a sub-tile has
// one parity, and the one workgroup that may straddle the boundary keeps two column partials.
// The column direction only needs VALUES: a row q is column t's unique best
// iff d2(q,t) equals the column minimum and the column passes the ratio test (a tie gives quotient
// 1, which no ratio <= 1 accepts), so no row index is carried through the column reduction.
// A running top-2 costs two VALU ops:  m2 = med3(m1, m2, key); m1 = min(m1, key).
#include "context.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <type_traits>

namespace eacham {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// Pointers read out of the device-side frame table are generic to the compiler; casting them to the
// global address space turns flat_load (which ties vmcnt to lgkmcnt) into global_load.
typedef const v4i __attribute__((address_space(1)))* gfrag_t;
typedef const int __attribute__((address_space(1)))* gint_t;

// H constant of padding rows/columns: above any real H (<= 256*255^2/2 = 8,323,200) and small
// enough that pad x pad (2 * PADH) still fits the 24-bit H field of a key
constexpr int PADH = 8350000;
constexpr int KEY_SHIFT = 7;               // tile field; key >> 7 = 2H + pb
constexpr int KEY_MASK = (1 << KEY_SHIFT) - 1;
constexpr int CHUNK_TILES = 1 << KEY_SHIFT;  // column tiles one sweep can tag in a key: 128 * 32 = 4096 train rows
constexpr int MAX_ROWS = 16384;              // rows per frame (K2 keeps two int per stored row in LDS)
constexpr int WG_THREADS = 256;            // 4 waves (1 per SIMD); 2 workgroups per CU drift out of phase so MFMA and VALU overlap
constexpr int WAVES = WG_THREADS / 64;
#ifndef EACHAM_MATCH_NSUB
#define EACHAM_MATCH_NSUB 2
#endif
constexpr int MATCH_NSUB = EACHAM_MATCH_NSUB;        // 32-row MFMA sub-tiles per wave (A fragments live in VGPRs)
constexpr int ROWS_PER_WAVE = 32 * MATCH_NSUB;
constexpr int ROWS_PER_WG = WAVES * ROWS_PER_WAVE;
constexpr int GROUP_TILES = MATCH_NSUB;              // tiles in use are padded to whole wave-blocks

// ------------------------------------------------------------------------------------------------
// upload: fp32 row-major -> fragment-major int8 + squared norms
// ------------------------------------------------------------------------------------------------

// U1: per-row sum of squares / sum of the centred values + integrality flag (thread per 16 values)
__global__ void rowsum_kernel(const float* __restrict__ src, int n, int dim, int* __restrict__ s2,
                              int* __restrict__ s1, int* __restrict__ bad_flag) {
    const int chunks = (dim + 15) / 16;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * chunks) return;
    const int row = (int)(idx / chunks), ch = (int)(idx % chunks);
    int sq = 0, sum = 0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = ch * 16 + j;
        if (k < dim) {
            const float v = src[(size_t)row * dim + k];
            if (!(v >= 0.0f && v <= 255.0f) || v != floorf(v)) bad = true;
            const int c = (int)v - 128;
            sq += c * c;
            sum += c;
        }
    }
    if (sq) atomicAdd(&s2[row], sq);
    if (sum) atomicAdd(&s1[row], sum);
    if (bad) atomicOr(bad_flag, 1);
}

// U2 (one workgroup): stable partition of the rows by the parity of |c|^2; fills the per-position
// constants and the two index maps. meta = {tiles of the even class, tiles in use}.
__global__ __launch_bounds__(1024) void partition_kernel(const int* __restrict__ s2, const int* __restrict__ s1, int n,
                                                         int npad, int group_rows, int* __restrict__ ca,
                                                         int* __restrict__ hb, int* __restrict__ orig,
                                                         int* __restrict__ pos, int* __restrict__ meta) {
    __shared__ int sc[1024];
    __shared__ int s_n0, s_carry;
    const int tid = threadIdx.x;
    for (int j = tid; j < npad; j += 1024) {
        ca[j] = PADH;
        hb[j] = PADH;
        orig[j] = -1;
    }
    if (tid == 0) s_n0 = 0, s_carry = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < n; i += 1024) mine += !(s2[i] & 1);
    atomicAdd(&s_n0, mine);
    __syncthreads();
    const int n0 = s_n0;
    const int even_pad = (n0 + 31) / 32 * 32;   // each parity class fills whole 32-row tiles
    const int odd_pad = (n - n0 + 31) / 32 * 32;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const int even = i < n ? !(s2[i] & 1) : 0;
        sc[tid] = even;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            const int v = tid >= off ? sc[tid - off] : 0;
            __syncthreads();
            sc[tid] += v;
            __syncthreads();
        }
        if (i < n) {
            const int rank_even = s_carry + sc[tid] - even;          // evens before row i
            const int j = even ? rank_even : even_pad + (i - rank_even);
            const int half = s2[i] >> 1;
            ca[j] = half + 1;  // +1 keeps H >= 0 when d2 = 0 between two odd-norm rows (d2 = 2 (H - 1) + pa + pb)
            hb[j] = half + s1[i];
            orig[j] = i;
            pos[i] = j;
        }
        __syncthreads();
        if (tid == 0) s_carry += sc[1023];
        __syncthreads();
    }
    if (tid == 0) {
        meta[0] = even_pad / 32;
        meta[1] = (even_pad + odd_pad + group_rows - 1) / group_rows * (group_rows / 32);  // whole wave-blocks
    }
}

// U3: fp32 row-major -> fragment-major int8 at the stored (parity-sorted) position
__global__ void quantize_kernel(const float* __restrict__ src, int dim, int KS, int npad,
                                const int* __restrict__ orig, v4i* __restrict__ frag) {
    const int chunks = KS * 2;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)npad * chunks) return;
    const int j = (int)(idx / chunks), ch = (int)(idx % chunks);
    const int ks = ch >> 1, h = ch & 1, tile = j >> 5, r = j & 31;
    const int row = orig[j];
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (row >= 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = ks * 32 + h * 16 + e;
            const int c = k < dim ? (int)src[(size_t)row * dim + k] - 128 : 0;  // padded dimensions are centred zeros
            w[e >> 2] |= (unsigned)(c & 0xff) << (8 * (e & 3));
        }
    }
    const v4i out = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    frag[((size_t)tile * KS + ks) * 64 + h * 32 + r] = out;
}

// ------------------------------------------------------------------------------------------------
// K1: distance tiles + fused row/column top-2
// ------------------------------------------------------------------------------------------------
//
// grid  = npairs * wgs_per_pair workgroups of 256 threads; workgroup (p, rb) owns rows
//         [256*rb, 256*rb+256) of frame A = pairs[p].x against ALL rows of frame B = pairs[p].y.
// wave  = 64 rows (two 32-row MFMA tiles): A fragments stay in VGPRs for the whole sweep
//         (A-stationary), B tiles (32 train rows = KS KiB) stream through LDS once per workgroup.
//
// The kernel is VALU-bound (each wave64 integer op costs 4 cycles per SIMD, the int8 MFMAs of a
// tile only 16 x 32), so the epilogue is cut to ~4.7 ops per distance:
//   * the query fragments are complemented once (a' = ~a = -a-1 per byte) and the accumulator starts
//     at floor(|a|^2/2) (C-init), so the MFMA returns acc = floor(|a|^2/2) - a.b - sum(b) and the
//     key is ONE v_lshl_add_u32:  key = (acc << 8) + ((hb_c << 8) | pb << 7 | t),
//     hb = floor(|b|^2/2) + sum(b) precomputed at upload;
//   * the row top-2 update is v_med3_u32 + v_min_u32 (2 ops per key);
//   * the column top-2 runs on the SAME keys, three at a time (5 ops per 3 keys).
// Measured on MI355X (tools/valu_ubench.hip, tools/mfma_ubench.hip): a wave64 integer VALU op costs
// ~4 cycles of its SIMD whatever the number of resident waves, the int8 32x32x32 MFMA 32 cycles, so
// the MFMA chains are software-pipelined UNDER the epilogue (see the main loop).
// out   rowres[p][cc][j]   = {v1, col1, v2, 0}   top-2 of stored row j over the columns of chunk cc,
//                                                v = 2H + pb = d2 - pa_j, col = stored column position
//       colpart[p][k][c]   = {key1, key2}        top-2 keys of stored column c over the rows of one
//                                                parity of a workgroup (d2 = 2 (key >> 8) + pa_k + pb_c - 2)
__device__ __forceinline__ unsigned vmed3(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }


__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
// 16-byte LDS read the compiler does not track (the caller places the s_waitcnt).
template <class V>
__device__ __forceinline__ void lds_read_frag(V& dst, unsigned addr, int offset_bytes) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(offset_bytes));
}

template <int KS, int NSUB, bool COL>
__global__ __launch_bounds__(WG_THREADS, (NSUB <= 2 ? 2 : 1)) void match_tile_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs, int wgs_per_pair, int col_chunks,
    uint4* __restrict__ rowres, uint2* __restrict__ colpart, int wb_stride, int row_stride) {
    constexpr int TILE_V4 = KS * 64;          // int4 per B tile
    constexpr int ROWS_WAVE = 32 * NSUB;      // query rows a wave keeps in registers
    constexpr int ROWS_WG = WAVES * ROWS_WAVE;
    static_assert(ROWS_WAVE <= (1 << KEY_SHIFT), "the local row must fit the key's code field");
    __shared__ v4i sB[3][TILE_V4];
    // One LDS region, two lives: during the sweep it buffers the column partials of the 4 waves for
    // bursts of BURST tiles (double-buffered), after the sweep it is the row-transposition slab.
    constexpr int BURST = 8;
    struct ColBuf { uint2 e[2][BURST][2][WAVES][32]; };  // [buffer][tile % BURST][row-parity group][wave][column]
    struct RowSlab { uint2 e[WAVES][32 * 33]; };
    __shared__ union { ColBuf c; RowSlab r; } sU;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 31, h = lane >> 5;
    // workgroup (pair p, row block rb, column chunk cc): frames with more than 4096 rows are swept in
    // chunks of CHUNK_TILES column tiles because a key has 7 bits for the tile index; K2 merges the
    // per-chunk row results (ascending chunks keep the lower column on ties).
    const int cc = blockIdx.x % col_chunks;
    const int rb = (blockIdx.x / col_chunks) % wgs_per_pair;
    const int p = blockIdx.x / (col_chunks * wgs_per_pair);
    const int2 pr = pairs[p];
    const FrameDev A = frames[pr.x], B = frames[pr.y];
    const int tbeg = cc * CHUNK_TILES;
    const int A_even = ((gint_t)A.meta)[0], A_tiles = ((gint_t)A.meta)[1];  // tiles of the even class / in use
    const int B_even = ((gint_t)B.meta)[0], B_tiles = ((gint_t)B.meta)[1];
    if (rb * (ROWS_WG / 32) >= A_tiles || tbeg >= B_tiles) return;  // workgroup-uniform
    const int wb = rb * WAVES + wave;                  // wave-block (ROWS_WAVE rows) of frame A
    const bool active = NSUB * wb < A_tiles;           // wave-uniform (tiles in use are a multiple of NSUB)
    const int T = min(B_tiles - tbeg, CHUNK_TILES);    // tiles of this chunk, t below is chunk-local
    const gfrag_t Afrag = (gfrag_t)A.frag, Bfrag = (gfrag_t)B.frag + (size_t)tbeg * TILE_V4;
    const gint_t Aca = (gint_t)A.norm, Bhb = (gint_t)B.normb + 32 * tbeg;

    v4i a[NSUB][KS];
    v16i cinit[NSUB];                 // floor(|a|^2/2) of this lane's 16 rows per sub-tile: the MFMA C-init
    unsigned rm1[NSUB][16], rm2[NSUB][16];
    const int wbc = active ? wb : 0;  // inactive waves load a valid block and never use it
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[s][ks] = ~Afrag[((size_t)(NSUB * wbc + s) * KS + ks) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int lrow = 32 * s + (r & 3) + 8 * (r >> 2) + 4 * h;  // C/D layout of the 32x32 MFMA
            cinit[s][r] = Aca[ROWS_WAVE * wbc + lrow];
            rm1[s][r] = 0xffffffffu;
            rm2[s][r] = 0xffffffffu;
        }
    }

    // Software pipeline (no extra registers): the NSUB accumulators of a wave are staggered by one
    // sub-tile. The MFMA chains are issued in the order (t,0) (t,1) .. (t,NSUB-1) (t+1,0) ..., and
    // while chain c+1 is being issued the VALU runs the epilogue of chain c — an MFMA chain
    // (8 x 32 cycles) always hides under ~100 VALU ops (4 cycles each). Waves issue in order, hence
    // the source interleaves one MFMA with two epilogue elements and pins that order with
    // sched_barrier. B fragments are read from LDS just in time through a 3-deep register ring (a tile
    // is read once per accumulator) instead of holding a whole tile in VGPRs.
    // LDS ring of 3 tiles: sB[t % 3] holds tile t. Tile t+2 is DMA-ed into sB[(t+2) % 3] from the
    // top of iteration t (that slot held tile t-1, last read in iteration t-1, i.e. before the
    // previous barrier); the barrier at the end of the iteration (vmcnt(0) + s_barrier) publishes it.
    // Staging is LDS-DMA (global_load_lds_dwordx4): a wave-instruction moves 1 KiB = one k-step of
    // the fragment-major tile straight into LDS (destination = wave-uniform base + lane*16, which
    // is exactly the linear tile image), no VGPR round trip and no ds_write pass.
    static_assert(TILE_V4 % 64 == 0, "a tile is a whole number of 1 KiB pieces");
    constexpr int PIECES = TILE_V4 / 64;                 // 1 KiB pieces per tile (= KS)
    constexpr int PPW = (PIECES + WAVES - 1) / WAVES;    // pieces per wave
    auto stage_tile = [&](int tile, int slot) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + i * WAVES;
            if (piece < PIECES)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(Bfrag + (size_t)tile * TILE_V4 + piece * 64 + lane),
                    (__attribute__((address_space(3))) void*)(&sB[slot][piece * 64]), 16, 0, 0);
        }
    };
    if (T > 0) {
        stage_tile(0, 0);
        stage_tile(min(1, T - 1), 1);
    }
    int hb_cur = T > 0 ? Bhb[cl] : 0;
    __builtin_amdgcn_s_waitcnt(0);  // every prologue load has landed (keeps vmcnt(0) out of the loop)
    __syncthreads();

    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16i acc[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) acc[s] = zero16;
    v4i bq[3];  // fragment ring: step i of the (phase, ks) sequence lives in bq[i % 3]
    if (T > 0 && active) {
        v4i b0[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b0[ks] = sB[0][ks * 64 + lane];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][ks], b0[ks], ks ? acc[0] : cinit[0], 0, 0, 0);
        bq[0] = sB[0][lane];       // steps 0 and 1 of iteration 0 (chain (0,1) re-reads tile 0)
        bq[1] = sB[0][64 + lane];
    }

    // Column partials leave the workgroup merged over its 4 wave-blocks (a quarter of the traffic of
    // per-wave partials). Rows of different parity must not be merged, so a workgroup owns the slots
    // rb (its even rows) and rb + 1 (its odd rows); only the one workgroup of a frame that straddles
    // the even/odd boundary fills both, every other workgroup fills the one of its parity. K2 knows
    // from A_even which slots are in use: slot k holds even rows iff 8k < A_even.
    const int wg_tile0 = rb * (ROWS_WG / 32);
    const bool split = wg_tile0 < A_even && wg_tile0 + ROWS_WG / 32 > A_even;  // workgroup-uniform
    const int slot0 = rb + ((!split && wg_tile0 >= A_even) ? 1 : 0);
    // The workgroup's part of the partial array as a SCALAR base (two s-registers) + a 32-bit per-lane index: hoisted as a
    // per-lane 64-bit address it cost two VGPRs across the whole sweep on a kernel that sits at the 256-register limit of two
    // waves per SIMD (and was the value the compiler spilled).
    uint2* colpart_wg;
    {
        const unsigned long long u = (unsigned long long)(colpart + ((size_t)p * wb_stride + slot0) * row_stride + 32 * (size_t)tbeg);
        colpart_wg = (uint2*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                              (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)u));
    }
    // Burst merge of the published tiles [t0, t0 + BURST) by the whole workgroup: thread -> (tile, column).
    auto merge_burst = [&](int t0) {
        const int tt = t0 + 2 * wave + h;  // tid >> 5
        if (tt < T) {
            for (int g = 0; g < (split ? 2 : 1); ++g) {
                const uint2(*src)[32] = sU.c.e[(t0 / BURST) & 1][tt % BURST][g];
                uint2 m = src[0][cl];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) {
                    const uint2 e = src[w][cl];
                    m.y = umin(umin(umax(m.x, e.x), m.y), e.y);
                    m.x = umin(m.x, e.x);
                }
                colpart_wg[(unsigned)(g * row_stride + 32 * tt + cl)] = m;
            }
        }
    };
    constexpr int EPK = 16 / KS;         // epilogue elements interleaved per MFMA (KS = 8 -> 2)
    constexpr int NSTEP = NSUB * KS;     // (phase, ks) steps per tile
    // One tile of the sweep. The LDS slot ring (tile t lives in slot t % 3) and the fragment register ring (step i of the
    // (phase, ks) sequence lives in bq[(i + PH) % 3]; a tile advances it by NSTEP % 3) both have period 3 in t: the body
    // is instantiated for the three residues and the loop below runs three tiles per trip, so every ring index is a
    // compile-time constant and no register is moved to rotate a ring (8 v_mov per wave-tile before, 4 % of its VALU
    // instructions, on a kernel bound by its issue port).
    auto tile = [&](auto PHc, auto SLc, const int t) {
        constexpr int PH = decltype(PHc)::value, SL = decltype(SLc)::value;
        constexpr int slot_cur = SL, slot_nxt = (SL + 1) % 3, slot_new = (SL + 2) % 3;
        // per-column part of this tile's keys: (hb << 8) | pb << 7 | t  (pb is wave-uniform per tile)
        const unsigned lowc = ((unsigned)hb_cur << (KEY_SHIFT + 1)) | (unsigned)((tbeg + t >= B_even ? (1 << KEY_SHIFT) : 0) | t);
        const int t1 = min(t + 1, T - 1), t2 = min(t + 2, T - 1);
        hb_cur = Bhb[32 * t1 + cl];
        if (COL && t > 0 && t % BURST == 0) merge_burst(t - BURST);  // tiles t-8 .. t-1 are published; their buffer is rewritten from tile t+8 on
        stage_tile(t2, slot_new);  // lands during this iteration; the barrier below publishes it
        unsigned cm1[NSUB], cm2[NSUB], pend[3] = {0, 0, 0};  // column top-2 per sub-tile (a sub-tile has one parity)
        if (active) {
            // LDS byte addresses of this lane's 16 bytes in the two live slots. The fragment reads
            // are inline asm with hand-placed s_waitcnt lgkmcnt(1): hipcc waits with lgkmcnt(0)
            // before every second MFMA, i.e. also for the read it issued one step earlier
            // (~50 cycles ago, well short of the LDS latency). Nothing else in the loop uses lgkmcnt.
            const unsigned curB = lds_addr(&sB[slot_cur][lane]);
            const unsigned nxtB = lds_addr(&sB[slot_nxt][lane]);
            // The ring phase advances by NSTEP % 3 per iteration, so the loop body is written for a
            // fixed phase and the ring is rotated at the end (register moves) to keep every index a
            // compile-time constant.
#pragma unroll
            for (int i = 0; i < NSTEP; ++i) {
                const int ph = i / KS, ks = i % KS;  // phase ph: epilogue of acc[ph], MFMAs of the next chain
                // prefetch the fragment of step i+2. Phase q < NSUB-1 issues chain (t, q+1): tile t;
                // phase NSUB-1 issues chain (t+1, 0) and phase 0 of the next iteration chain (t+1, 1):
                // tile t+1.
                const int j = i + 2;
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bq[(i + PH) % 3]));  // step i landed (i+1 may be in flight)
                lds_read_frag(bq[(j + PH) % 3], (j / KS < NSUB - 1) ? curB : nxtB, (j % KS) * 1024);
                const int q = (ph + 1) % NSUB;       // accumulator of the chain being issued
                // (the last iteration recomputes tile T-1 into acc[0]; it is never read)
                acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q][ks], bq[(i + PH) % 3], ks ? acc[q] : cinit[q], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < EPK; ++e) {
                    const int r = ks * EPK + e;
                    const unsigned key = ((unsigned)acc[ph][r] << (KEY_SHIFT + 1)) + lowc;  // one v_lshl_add_u32
                    rm2[ph][r] = vmed3(rm1[ph][r], rm2[ph][r], key);
                    rm1[ph][r] = umin(rm1[ph][r], key);
                    // Column direction: the 16 keys of a sub-tile belong to this lane's column, so they
                    // are taken three at a time — {min3, med3} of a triple (2 ops) then one sorted-pair
                    // insert (3 ops) = 5 ops per 3 keys instead of 6 (and the first triple needs no insert).
                    const int eg = ks * EPK + e;  // 0..15 within the sub-tile, compile-time after unrolling
                    if constexpr (COL) {
                        pend[eg % 3] = key;
                        if (eg % 3 == 2) {
                            const unsigned s1 = umin(umin(pend[0], pend[1]), pend[2]);
                            const unsigned s2 = vmed3(pend[0], pend[1], pend[2]);
                            if (eg == 2) {
                                cm1[ph] = s1;
                                cm2[ph] = s2;
                            } else {
                                cm2[ph] = umin(umin(umax(cm1[ph], s1), cm2[ph]), s2);
                                cm1[ph] = umin(cm1[ph], s1);
                            }
                        } else if (eg == 15) {  // 16 = 5 triples + 1
                            cm2[ph] = vmed3(cm1[ph], cm2[ph], key);
                            cm1[ph] = umin(cm1[ph], key);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // steps NSTEP and NSTEP + 1 (= steps 0, 1 of the next tile, whose ring phase is PH + NSTEP) are in flight:
            // they have landed before the barrier below lets anybody overwrite their slot
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]));
            // Merge the sub-tiles that share a parity: all of them, except in the workgroup that
            // straddles the even/odd boundary of frame A, which keeps the two parities apart.
            unsigned g1[2] = {0xffffffffu, 0xffffffffu}, g2[2] = {0xffffffffu, 0xffffffffu};
#pragma unroll
            for (int s = 0; COL && s < NSUB; ++s) {
                const int g = (split && NSUB * wb + s >= A_even) ? 1 : 0;  // wave-uniform
                if (g == 0) {
                    g2[0] = umin(umin(umax(g1[0], cm1[s]), g2[0]), cm2[s]);
                    g1[0] = umin(g1[0], cm1[s]);
                } else {
                    g2[1] = umin(umin(umax(g1[1], cm1[s]), g2[1]), cm2[s]);
                    g1[1] = umin(g1[1], cm1[s]);
                }
            }
            // lanes l and l+32 hold the same column, rows 4h.. of each 8-row group: merge halves
            // (v_permlane32_swap: lanes 0-31 of the 2nd operand <-> lanes 32-63 of the 1st, pure VALU)
            if constexpr (COL) {
                auto w1 = __builtin_amdgcn_permlane32_swap(g1[0], g1[0], false, false);
                auto w2 = __builtin_amdgcn_permlane32_swap(g2[0], g2[0], false, false);
                const unsigned n1 = umin(w1[0], w1[1]);
                const unsigned n2 = umin(umax(w1[0], w1[1]), umin(w2[0], w2[1]));
                if (h == 0) sU.c.e[(t / BURST) & 1][t % BURST][0][wave][cl] = make_uint2(n1, n2);
            }
            if (COL && split) {
                auto w1 = __builtin_amdgcn_permlane32_swap(g1[1], g1[1], false, false);
                auto w2 = __builtin_amdgcn_permlane32_swap(g2[1], g2[1], false, false);
                const unsigned n1 = umin(w1[0], w1[1]);
                const unsigned n2 = umin(umax(w1[0], w1[1]), umin(w2[0], w2[1]));
                if (h == 0) sU.c.e[(t / BURST) & 1][t % BURST][1][wave][cl] = make_uint2(n1, n2);
            }
        } else if constexpr (COL) {  // a wave beyond the frame's rows contributes nothing
            if (h == 0) sU.c.e[(t / BURST) & 1][t % BURST][0][wave][cl] = make_uint2(0xffffffffu, 0xffffffffu);
            if (split && h == 0) sU.c.e[(t / BURST) & 1][t % BURST][1][wave][cl] = make_uint2(0xffffffffu, 0xffffffffu);
        }
        __syncthreads();
    };
    {
        constexpr int ADV = NSTEP % 3;  // ring advance per tile
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, ADV % 3>;
        using P2 = std::integral_constant<int, (2 * ADV) % 3>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        int t = 0;
        for (; t + 3 <= T; t += 3) {
            tile(P0{}, P0{}, t);
            tile(P1{}, S1{}, t + 1);
            tile(P2{}, S2{}, t + 2);
        }
        if (t < T) {
            tile(P0{}, P0{}, t);
            if (t + 1 < T) tile(P1{}, S1{}, t + 1);
        }
    }
    if (COL && T > 0) merge_burst((T - 1) / BURST * BURST);  // the last 1..8 tiles, published by the loop's last barrier
    __syncthreads();                                  // the region becomes the row slab
    if (!active) return;

    // Row direction: every lane holds, per row, its top-2 over the columns {32t + cl}. Transpose
    // through this wave's private LDS slab (rows padded to 33 entries: conflict-free) so that lane
    // (row, half) scans 16 of the 32 lane-partials of its row in ASCENDING lane-column order with a
    // strict '<': equal keys (same rank, same tile) then keep the lower column, which makes plain
    // 32-bit key compares exact — (key, lane-column) lexicographic == (rank, column) lexicographic.
    uint2* slab = sU.r.e[wave];
    uint4* rr = rowres + ((size_t)p * col_chunks + cc) * row_stride + ROWS_WAVE * wb;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            slab[row * 33 + cl] = make_uint2(rm1[s][r], rm2[s][r]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned b1 = 0xffffffffu, b2 = 0xffffffffu;
        int c1 = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int cidx = 16 * h + j;
            const uint2 e = slab[cl * 33 + cidx];
            const bool lt = e.x < b1;
            b2 = lt ? b1 : umin(b2, e.x);
            c1 = lt ? cidx : c1;
            b1 = umin(b1, e.x);
            b2 = umin(b2, e.y);  // e.y >= e.x: it can only be a runner-up
        }
        const unsigned o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
        const int oc = __shfl_xor(c1, 32);
        if (h == 0) {  // the other half holds the higher columns: it wins only with a strictly smaller key
            const bool lt = o1 < b1;
            const unsigned key1 = lt ? o1 : b1, key2 = lt ? umin(b1, o2) : umin(b2, o1);
            const unsigned col1 = 32 * (tbeg + (key1 & KEY_MASK)) + (lt ? oc : c1);
            rr[32 * s + cl] = make_uint4(key1 >> KEY_SHIFT, col1, key2 >> KEY_SHIFT, 0);  // key >> 7 = 2H + pb
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ------------------------------------------------------------------------------------------------
// K1r: the ROW sweep with the MFMA operands swapped (round 4) — the sweep of the candidate-only pipeline.
//
// match_tile_kernel keeps its query rows as the A operand: a lane then holds one train COLUMN of the tile, its 16 accumulators are 16
// different query rows, and every distance needs a key of its own — one VALU op to add the column's constant and tag the tile, two
// for the running top-2 of its row: 3 ops per distance on a kernel bound by the VALU issue port. Swapped — the train tile as the A
// operand, the wave's 64 query rows as the B operand — a lane IS a query row, its 16 accumulators are 16 train rows against it, and
// the per-train-row constant hb is the accumulator's C-init (read from LDS next to the tile: no VALU op), the per-query constant
// is added once after the sweep. The row's top-2 VALUES then run on the raw accumulators three at a time — {min3, med3} of a
// triple, a sorted-pair merge: 27 ops per 16 distances — and what a key used to carry is recovered elsewhere:
//   * the parity of the train row's norm: frames are stored parity-sorted, so the sweep passes all even tiles, then all odd ones;
//     the running state is set aside ONCE at the boundary and the two classes are joined when 2H + pb is formed;
//   * the column of the minimum: only the TILE of the first strict improvement is tracked (a compare + select per 16 distances);
//     the row inside it is found afterwards for the rows that pass the ratio test only (match_argmin_kernel: one 32 x 32 MFMA
//     group per (train tile, <= 32 passing rows)), lowest train index on ties as the reference's scan.
// 31 VALU ops per 16 distances instead of 48, and no 7-bit tile field: frames of any size in one sweep (no column chunks).
// Pipeline, LDS-DMA ring, just-in-time fragment ring and the three-tiles-per-trip loop are match_tile_kernel's.
// out   rowres[p][j] = {v1, tile1, v2, 0}: v = 2H + pb = d2 - pa_j + 2 of stored query row j, tile1 = first tile holding the minimum
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imed3(int a, int b, int c) {
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// sorted pair (s1 <= s2) merged into the running top-2 (m1 <= m2): 3 ops
__device__ __forceinline__ void top2_merge(int& m1, int& m2, int s1, int s2) {
    m2 = imin(imin(imax(m1, s1), m2), s2);
    m1 = imin(m1, s1);
}

//
// BOUND = true (round 5, the default of the lean form): the sweep keeps NO second-smallest value and no tile. Per sub-tile a lane
// runs FOUR minima, one per four of its sixteen accumulators (two v_min3 per group and tile: 8 VALU operations per 16 distances
// instead of 31), over the train rows of four disjoint subsets; with the partner lane and the two parity classes a query row ends
// with 16 partial minima over 16 DISJOINT subsets of the train frame. Their smallest is the row's minimum v1, and their second
// smallest u is an UPPER BOUND of the row's true second-smallest value (the runner-up is either the minimum of another subset, or
// hides behind v1 in v1's own subset and is then even smaller). The ratio test is monotone in the second value, so a row that
// fails it against u fails it against the truth: such rows are finished. The others — the rows that pass (a few per cent) and the
// few whose runner-up shared the minimum's subset — are candidates: match_rowpick_kernel lists them and match_colverify_kernel<KS, true>
// computes their exact {v1, tile, v2} against the whole train frame (the candidate-only pass of the columns with the frames' roles
// swapped), overwriting their rowres entries before match_rows2_kernel reads them. out: rowres[p][j] = {v1, 0, u, 0}.
template <int KS, bool BOUND = false>
__global__ __launch_bounds__(WG_THREADS, KS >= 4 ? 3 : 4) void match_sweep_kernel(const FrameDev* __restrict__ frames, const int2* __restrict__ pairs,
                                                                   int wgs_per_pair, uint4* __restrict__ rowres, int row_stride, int prio) {
    if (prio == 1) __builtin_amdgcn_s_setprio(1);  // (A/B switch EACHAM_EXP_SWEEP_PRIO: workgroup-uniform, off by default)
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 3) __builtin_amdgcn_s_setprio(3);
    constexpr int NSUB = 2;
    constexpr int TILE_V4 = KS * 64;
    constexpr int ROWS_WAVE = 32 * NSUB, ROWS_WG = WAVES * ROWS_WAVE;
    constexpr int BIG = 0x7fffffff;
    __shared__ v4i sB[3][TILE_V4];
    __shared__ __attribute__((aligned(16))) int sHb[3][32];  // hb of the tile in the same ring slot: the C-init of its MFMA chains
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 31, h = lane >> 5;
    const int rb = blockIdx.x % wgs_per_pair;
    const int p = blockIdx.x / wgs_per_pair;
    const int2 pr = pairs[p];
    const FrameDev A = frames[pr.x], B = frames[pr.y];
    const int A_tiles = ((gint_t)A.meta)[1];
    const int B_even = ((gint_t)B.meta)[0], T = ((gint_t)B.meta)[1];
    if (rb * (ROWS_WG / 32) >= A_tiles) return;  // workgroup-uniform
    const int wb = rb * WAVES + wave;
    const bool active = NSUB * wb < A_tiles;
    const gfrag_t Afrag = (gfrag_t)A.frag, Bfrag = (gfrag_t)B.frag;
    const gint_t Aca = (gint_t)A.norm, Bhb = (gint_t)B.normb;
    const int wbc = active ? wb : 0;

    v4i a[NSUB][KS];   // this wave's 64 query rows, complemented: the B operand (a lane = a query row of the sub-tile)
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[s][ks] = ~Afrag[((size_t)(NSUB * wbc + s) * KS + ks) * 64 + lane];
    // running top-2 values and the tile of the minimum, per sub-tile; the even class is set aside at the parity boundary
    // (set aside in LDS, each lane its own words: six registers less through the sweep)
    constexpr int NG = 4;                    // BOUND: running minima per sub-tile (eight: 14 registers spilled at 256-D)
    constexpr int NSTATE = BOUND ? NG : 3;   // words of running state per sub-tile
    __shared__ int sEven[NSTATE * NSUB][WG_THREADS];
    int x1[NSUB], x2[NSUB], xt[NSUB];
    int xm[NSUB][NG];  // BOUND: the minimum over accumulators 4 g .. 4 g + 3 of every tile of the class
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        x1[s] = x2[s] = BIG, xt[s] = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) xm[s][g] = BIG;
    }

    static_assert(TILE_V4 % 64 == 0, "a tile is a whole number of 1 KiB pieces");
    constexpr int PIECES = TILE_V4 / 64;
    constexpr int PPW = (PIECES + WAVES - 1) / WAVES;
    // (a wave stages PPW consecutive 1 KiB pieces: a scalar base, ONE 32-bit lane offset and immediate offsets — no 64-bit
    // per-lane addresses held through the sweep)
    const unsigned stage_off = (unsigned)(wave * PPW * 64 + lane) * 16u;
    auto stage_tile = [&](int tile, int slot) {
        const __attribute__((address_space(1))) char* base = (const __attribute__((address_space(1))) char*)(Bfrag + (size_t)tile * TILE_V4);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave * PPW + i;
            if (piece < PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + stage_off + i * 1024),
                                                 (__attribute__((address_space(3))) void*)(&sB[slot][piece * 64]), 16, 0, 0);
        }
    };
    if (T > 0) {
        stage_tile(0, 0);
        stage_tile(min(1, T - 1), 1);
        stage_tile(min(2, T - 1), 2);
        if (tid < 96) sHb[tid >> 5][tid & 31] = Bhb[32 * min(tid >> 5, T - 1) + (tid & 31)];
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    // C-init of a chain on the tile in ring slot `slot`: accumulator r is train row (r & 3) + 8 (r >> 2) + 4 h
    auto cinit_of = [&](int slot) {
        v16i c;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4i q = *reinterpret_cast<const v4i*>(&sHb[slot][8 * g + 4 * h]);
#pragma unroll
            for (int k = 0; k < 4; ++k) c[4 * g + k] = q[k];
        }
        return c;
    };
    // Software pipeline: the MFMAs of tile t + 1 (both sub-tiles: every fragment is read from LDS ONCE and feeds two independent
    // chains) are issued interleaved with the epilogue of tile t, so the accumulators are double-buffered (tile parity). With one
    // LDS read per fragment and accumulator — the first form of this kernel, and match_tile_kernel's — the CU's LDS port is as busy as
    // its matrix pipes (16 KiB per wave-tile at 128 B / cycle = the 1024 cycles the MFMAs of two resident waves take) and the 256-D
    // sweep stayed where the VALU-bound kernel was; read once, the port is at half of that.
    // The chains' C-init (hb of the tile) is read from LDS straight into the accumulators that will take the chain: the ones whose
    // epilogue has just finished (a separate register set for it cost 16 VGPRs and, by the allocator's choice, a copy of a live chain).
    static_assert(KS >= 2, "the epilogue of a tile is spread over the first KS - 1 steps of the next tile's MFMAs");
    v16i accA[NSUB], accB[NSUB];
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NSUB; ++s) accA[s] = zero16, accB[s] = zero16;
    v4i bq[3];  // fragment ring: step i of the ks sequence lives in bq[i % 3]
    constexpr int DIST = KS >= 2 ? 2 : 1;   // fragments are read DIST steps ahead of their MFMAs (a 32-D tile is a single step)
    if (T > 0 && active) {
        const v16i c0 = cinit_of(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const v4i b0 = sB[0][ks * 64 + lane];
#pragma unroll
            for (int s = 0; s < NSUB; ++s) accA[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b0, a[s][ks], ks ? accA[s] : c0, 0, 0, 0);
        }
        bq[0] = sB[1][lane];   // the first fragments of tile 1 (the ring holds tile T - 1 again past the end: recomputed, never read)
        if constexpr (DIST == 2) bq[1] = sB[1][64 + lane];
#pragma unroll
        for (int s = 0; s < NSUB; ++s) accB[s] = cinit_of(1);
    }
    __syncthreads();   // slot 0 is staged into again by the first call below
    // keys of tile t handled by step ks of tile t + 1's MFMAs: all 32 within the first KS - 1 steps, so that the accumulators are
    // free for tile t + 2's constants a step before the call's drain
    auto key_lo = [](int ks) constexpr { return ks >= KS - 1 ? 32 : (32 * ks + KS - 2) / (KS - 1); };
    // One call: epilogue of tile t (held in `cur`), MFMAs of tile t + 1 into `nxt`, tile t + 3 staged into tile t's ring slot (its
    // fragments were consumed a call ago), the first fragments and the constants of tile t + 2 fetched at the end (published by the
    // previous call's barrier) and drained before this call's: nothing is in flight across a call. The LDS slot ring (period 3), the
    // fragment register ring (advance KS % 3 per call) and the accumulator parity (period 2) are compile-time: six instantiations.
    // key idx (sub-tile idx / 16, accumulator idx % 16) of tile t into the running state: triples through {min3, med3} + a sorted-pair
    // merge, the sixteenth key alone; the tile is recorded when the sub-tile has improved the class minimum strictly
    auto consume = [&](const v16i(&cur)[NSUB], const int idx, const int t, int(&pend)[3], int& before) {
        const int ph = idx / 16, eg = idx % 16;
        if constexpr (BOUND) {
            if (eg & 1) {   // accumulators 4 g .. 4 g + 3 -> xm[g], two at a time
                int d;
                asm("v_min3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(xm[ph][eg >> 2]), "v"(cur[ph][eg - 1]), "v"(cur[ph][eg]));
                xm[ph][eg >> 2] = d;
            }
            return;
        }
        const int key = cur[ph][eg];
        if (eg == 0) before = x1[ph];  // the class minimum this sub-tile meets
        pend[eg % 3] = key;
        if (eg % 3 == 2) {
            const int s1 = imin(imin(pend[0], pend[1]), pend[2]);
            const int s2 = imed3(pend[0], pend[1], pend[2]);
            top2_merge(x1[ph], x2[ph], s1, s2);
        } else if (eg == 15) {
            x2[ph] = imed3(x1[ph], x2[ph], key);
            x1[ph] = imin(x1[ph], key);
            xt[ph] = x1[ph] < before ? t : xt[ph];  // a strict improvement somewhere in this tile: it is the minimum's tile now
        }
    };
    auto set_aside_even = [&]() {
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            if constexpr (BOUND) {
#pragma unroll
                for (int g = 0; g < NG; ++g) sEven[NG * s + g][tid] = xm[s][g], xm[s][g] = BIG;
            } else {
                sEven[3 * s][tid] = x1[s], sEven[3 * s + 1][tid] = x2[s], sEven[3 * s + 2][tid] = xt[s];
                x1[s] = x2[s] = BIG;
            }
        }
    };
    auto tile = [&](auto PHc, auto SLc, auto PARc, const int t) {
        constexpr int PH = decltype(PHc)::value, SL = decltype(SLc)::value, PAR = decltype(PARc)::value;
        constexpr int slot_nxt = (SL + 1) % 3, slot_nn = (SL + 2) % 3;
        v16i(&cur)[NSUB] = PAR ? accB : accA;
        v16i(&nxt)[NSUB] = PAR ? accA : accB;
        const int t3 = min(t + 3, T - 1);
        const int hb_new = (tid < 32) ? *reinterpret_cast<const __attribute__((address_space(1))) int*>(reinterpret_cast<const __attribute__((address_space(1))) char*>(Bhb + 32 * t3) + (unsigned)tid * 4u) : 0;   // lands during this call; stored to the ring before the barrier
        stage_tile(t3, SL);
        if (t == B_even) set_aside_even();  // (workgroup-uniform, once per sweep) the even tiles are done: start the odd class
        if (active) {
            const unsigned nxtB = lds_addr(&sB[slot_nxt][lane]);
            const unsigned nnB = lds_addr(&sB[slot_nn][lane]);
            int pend[3] = {0, 0, 0}, before = 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (DIST == 2) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bq[(ks + PH) % 3]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[(ks + PH) % 3]));
                const int j = ks + DIST;
                lds_read_frag(bq[(j + PH) % 3], j < KS ? nxtB : nnB, (j % KS) * 1024);
#ifndef EXP_SWEEP_NO_MFMA   // (knock-out, timing only: the epilogue, the LDS traffic and the staging without the matrix pipes)
#pragma unroll
                for (int s = 0; s < NSUB; ++s)
                    nxt[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[(ks + PH) % 3], a[s][ks], nxt[s], 0, 0, 0);
#else
#pragma unroll
                for (int s = 0; s < NSUB; ++s) asm volatile("" : "+v"(nxt[s]) : "v"(bq[(ks + PH) % 3]), "v"(a[s][ks]));
#endif
#pragma unroll
                for (int idx = key_lo(ks); idx < key_lo(ks + 1); ++idx) {
#ifdef EXP_SWEEP_NO_EPI    // (knock-out, timing only: the MFMA chains and the data movement without the top-2 epilogue)
                    if (idx % 16 == 15) {
                        asm volatile("" : "+v"(cur[idx / 16]));
                        x1[idx / 16] = imin(x1[idx / 16], cur[idx / 16][0]);
                    }
#else
                    consume(cur, idx, t, pend, before);
#endif
                    if (idx % 16 == 15) cur[idx / 16] = cinit_of(slot_nn);   // done with these accumulators: tile t + 2's constants, for the next call's chains
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]));
        }
        if (tid < 32) sHb[SL][tid] = hb_new;
        __syncthreads();
    };
    // the last tile has nothing to issue and nothing to stage: its epilogue alone (which accumulators hold it is its parity)
    auto last = [&](auto PARc, const int t) {
        constexpr int PAR = decltype(PARc)::value;
        v16i(&cur)[NSUB] = PAR ? accB : accA;
        if (t == B_even) set_aside_even();
        if (active) {
            int pend[3] = {0, 0, 0}, before = 0;
#pragma unroll
            for (int idx = 0; idx < 32; ++idx) consume(cur, idx, t, pend, before);
        }
    };
    if (T > 0) {
        // tile(t) consumes tile t's accumulators and issues tile t + 1's, for t < T - 1
        constexpr int ADV = KS % 3;
        const int TM = T - 1;
        int t = 0;
        auto run = [&](auto K6c, int tt) {
            constexpr int K6 = decltype(K6c)::value;
            tile(std::integral_constant<int, (K6 * ADV) % 3>{}, std::integral_constant<int, K6 % 3>{}, std::integral_constant<int, K6 % 2>{}, tt);
        };
        for (; t + 6 <= TM; t += 6) {
            run(std::integral_constant<int, 0>{}, t);
            run(std::integral_constant<int, 1>{}, t + 1);
            run(std::integral_constant<int, 2>{}, t + 2);
            run(std::integral_constant<int, 3>{}, t + 3);
            run(std::integral_constant<int, 4>{}, t + 4);
            run(std::integral_constant<int, 5>{}, t + 5);
        }
        if (t < TM) run(std::integral_constant<int, 0>{}, t);
        if (t + 1 < TM) run(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < TM) run(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < TM) run(std::integral_constant<int, 3>{}, t + 3);
        if (t + 4 < TM) run(std::integral_constant<int, 4>{}, t + 4);
        if (TM & 1) last(std::integral_constant<int, 1>{}, TM);
        else last(std::integral_constant<int, 0>{}, TM);
    }
    if (!active) return;
    if constexpr (BOUND) {
        uint4* rr = rowres + (size_t)p * row_stride + ROWS_WAVE * wb;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const int ca = Aca[ROWS_WAVE * wb + 32 * s + cl];
            unsigned v1 = 0xffffffffu, v2 = 0xffffffffu;   // the two smallest of the 16 partial minima (4 groups x 2 lanes x 2 classes)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    // a frame without odd rows never met the boundary: what ran is the even class
                    const int mine = c == 0 ? (B_even >= T ? xm[s][g] : sEven[NG * s + g][tid]) : (B_even >= T ? BIG : xm[s][g]);
                    const int other = __shfl_xor(mine, 32);
                    const unsigned w0 = mine == BIG ? 0xffffffffu : (unsigned)(2 * (mine + ca) + c);
                    const unsigned w1 = other == BIG ? 0xffffffffu : (unsigned)(2 * (other + ca) + c);
                    v2 = umin(v2, umax(v1, w0)), v1 = umin(v1, w0);
                    v2 = umin(v2, umax(v1, w1)), v1 = umin(v1, w1);
                }
            if (h == 0) rr[32 * s + cl] = make_uint4(v1, 0u, v2, 0u);
        }
        return;
    }
    int e1[NSUB], e2[NSUB], et[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        if (B_even >= T) {  // a frame without odd rows never met the boundary: what ran is the even class
            e1[s] = x1[s], e2[s] = x2[s], et[s] = xt[s], x1[s] = x2[s] = BIG;
        } else {
            e1[s] = sEven[3 * s][tid], e2[s] = sEven[3 * s + 1][tid], et[s] = sEven[3 * s + 2][tid];
        }
    }
    uint4* rr = rowres + (size_t)p * row_stride + ROWS_WAVE * wb;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        // lanes l and l + 32 hold the same query row against complementary train rows of every tile: join them (equal minima:
        // the earlier tile), then the two parity classes as v = 2 (m + ca) + pb (never equal across classes)
        const int ca = Aca[ROWS_WAVE * wb + 32 * s + cl];
        int cls1[2] = {e1[s], x1[s]}, cls2[2] = {e2[s], x2[s]}, clst[2] = {et[s], xt[s]};
        unsigned v1 = 0xffffffffu, v2 = 0xffffffffu, t1 = 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int o1 = __shfl_xor(cls1[c], 32), o2 = __shfl_xor(cls2[c], 32), ot = __shfl_xor(clst[c], 32);
            const int tt = o1 < cls1[c] ? ot : (o1 == cls1[c] ? imin(ot, clst[c]) : clst[c]);
            int m1 = cls1[c], m2 = cls2[c];
            top2_merge(m1, m2, o1, o2);
            const unsigned w1 = m1 == BIG ? 0xffffffffu : (unsigned)(2 * (m1 + ca) + c);
            const unsigned w2 = m2 == BIG ? 0xffffffffu : (unsigned)(2 * (m2 + ca) + c);
            // (w1 <= w2) into (v1 <= v2)
            t1 = w1 < v1 ? (unsigned)tt : t1;
            v2 = umin(umin(umax(v1, w1), v2), w2);
            v1 = umin(v1, w1);
        }
        if (h == 0) rr[32 * s + cl] = make_uint4(v1, t1, v2, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// K2: merge column partials, ratio test, mutual check, thresholds, ordered compaction
// ------------------------------------------------------------------------------------------------

// FeatureMatcherFlann.cpp:23 — `m[0].distance / m[1].distance < 0.8`: distances are
// sqrtf(squared L2) in fp32, fp32 quotient, compared as double. d2 values are exact integers.
__device__ __forceinline__ bool ratio_pass(int d2_best, int d2_second, double ratio) {
    float q = __fdiv_rn(__fsqrt_rn((float)d2_best), __fsqrt_rn((float)d2_second));
    return (double)q < ratio;  // 0/0 = NaN -> false
}

constexpr int FIN_THREADS = 256;

// mode 0: mutual matches + thresholds (apps/sfm/main.cpp:111-146); mode 1: directed list m12.
// out_matches[p][k] = {q, t} sorted by q; counts[p]; stats[p] = {|m12|, |m21|, |mutual|, edge}.
// Rows/columns are addressed by stored position inside; q and t leave in the caller's numbering.
__global__ __launch_bounds__(FIN_THREADS) void match_finalize_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs,
    const uint4* __restrict__ rowres, const uint2* __restrict__ colpart, int col_chunks, int wb_stride,
    int row_stride, double ratio, int min_dir, int min_mutual, int mode,
    uint2* __restrict__ out_matches, int* __restrict__ counts, int4* __restrict__ stats) {
    extern __shared__ int smem[];
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const int2 pr = pairs[p];
    const FrameDev A = frames[pr.x], B = frames[pr.y];
    const int A_even = A.meta[0], A_tiles = A.meta[1], B_even = B.meta[0], B_tiles = B.meta[1];
    int* keepcol = smem;           // [row_stride] by stored row: stored column of the kept match or -1
    int* bwd = smem + row_stride;  // [row_stride] by stored column: d2 of the column's best if it passes the ratio test, else -1
    __shared__ int s_cnt[3];
    __shared__ int s_scan[FIN_THREADS];
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();
    constexpr unsigned PAD_V = 2u * PADH;  // v = 2H + parity of anything involving a padding row/column

    // columns first: top-2 VALUES of every stored column over the wave-blocks of A
    int c12 = 0, c21 = 0;
    constexpr int WG_TILES = ROWS_PER_WG / 32;
    const int nslots = (A_tiles + WG_TILES - 1) / WG_TILES + 1;  // workgroup rb fills slot rb (even rows) and/or rb + 1 (odd rows)
    const int first_odd = A_even / WG_TILES + 1;
    for (int c = tid; c < 32 * B_tiles; c += FIN_THREADS) {
        int d = -1;
        if (B.orig[c] >= 0) {
            unsigned v1 = 0xffffffffu, v2 = 0xffffffffu;
            const uint2* cp = colpart + (size_t)p * wb_stride * row_stride + c;
            for (int k = 0; k < nslots; ++k) {
                const bool even = WG_TILES * k < A_even;
                if (!even && k < first_odd) continue;                     // the slot between the classes when A_even % 8 == 0
                const uint2 e = cp[(size_t)k * row_stride];
                const unsigned pa = even ? 0u : 1u;                        // one parity per slot
                const unsigned va = ((e.x >> (KEY_SHIFT + 1)) << 1) | pa, vb = ((e.y >> (KEY_SHIFT + 1)) << 1) | pa;
                v2 = umin(umin(umax(v1, va), v2), vb);  // (va <= vb) merged into (v1 <= v2)
                v1 = umin(v1, va);
            }
            const unsigned pb = (c >> 5) >= B_even ? 1u : 0u;
            const bool ok = v2 < PAD_V && ratio_pass((int)(v1 + pb) - 2, (int)(v2 + pb) - 2, ratio);  // pad second => < 2 query rows
            d = ok ? (int)(v1 + pb) - 2 : -1;
            c21 += ok;
        }
        bwd[c] = d;
    }
    __syncthreads();
    const int nchunks = (B_tiles + CHUNK_TILES - 1) / CHUNK_TILES;
    for (int j = tid; j < 32 * A_tiles; j += FIN_THREADS) {
        int keep = -1;
        if (A.orig[j] >= 0) {
            unsigned v1 = 0xffffffffu, v2 = 0xffffffffu, col = 0;
            for (int ch = 0; ch < nchunks; ++ch) {  // ascending columns; strict '<' keeps the lower column on ties
                const uint4 e = rowres[((size_t)p * col_chunks + ch) * row_stride + j];
                if (e.x < v1) {
                    v2 = umin(v1, e.z);
                    v1 = e.x;
                    col = e.y;
                } else {
                    v2 = umin(v2, e.x);
                }
            }
            const unsigned pa = (j >> 5) >= A_even ? 1u : 0u;
            const bool ok = v2 < PAD_V && ratio_pass((int)(v1 + pa) - 2, (int)(v2 + pa) - 2, ratio);  // pad second => < 2 train rows
            c12 += ok;
            // main.cpp:133-140: q is kept iff t's own best match is q, i.e. iff t passes the ratio
            // test (unique minimum for any ratio <= 1) and its minimum is d2(q, t)
            if (ok && (mode == 1 || bwd[col] == (int)(v1 + pa) - 2)) keep = (int)col;
        }
        keepcol[j] = keep;
    }
    atomicAdd(&s_cnt[0], c12);
    atomicAdd(&s_cnt[1], c21);
    __syncthreads();

    // ordered compaction over the caller's q in chunks of FIN_THREADS
    uint2* out = out_matches + (size_t)p * row_stride;
    int base = 0;
    for (int q0 = 0; q0 < A.n; q0 += FIN_THREADS) {
        const int q = q0 + tid;
        const int t = q < A.n ? keepcol[A.pos[q]] : -1;
        const bool keep = t >= 0;
        s_scan[tid] = keep;
        __syncthreads();
        for (int off = 1; off < FIN_THREADS; off <<= 1) {  // Hillis-Steele inclusive scan
            int v = tid >= off ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        if (keep) out[base + s_scan[tid] - 1] = make_uint2((unsigned)q, (unsigned)B.orig[t]);
        base += s_scan[FIN_THREADS - 1];
        __syncthreads();
    }
    if (tid == 0) {
        int n12 = s_cnt[0], n21 = s_cnt[1];
        bool edge = n12 >= min_dir && n21 >= min_dir && base > min_mutual;  // main.cpp:111,142
        counts[p] = mode == 1 ? base : (edge ? base : 0);
        if (stats) stats[p] = make_int4(n12, n21, base, edge ? 1 : 0);
    }
}

// ------------------------------------------------------------------------------------------------
// The candidate-only column pass (round 4). The tile sweep is bound by the VALU issue port and its column direction is
// 1.67 of the 4.67 operations it spends per distance (profiles/r04_coltop_knockout.txt: the sweep alone is 26 % / 39 %
// faster at 256-D / 128-D). But the pair loop of apps/sfm/main.cpp:111,133-142 only ever looks at column t = m12[q] of a
// row q that passed the ratio test: with |mutual| > min_mutual >= min_dir - 1 the two direction thresholds are implied
// (every mutual match is in m12 and in m21), so a pair with no more than min_mutual passing rows is dead, and for the
// others the column top-2 is needed for the passing rows' best columns only. Three small kernels replace K2:
//   R  match_rows_kernel       per pair: merge the row results, ratio test, ordered list of the passing rows (candidates),
//                              one work item per 64 candidates of a live pair
//   V  match_colverify_kernel  per item: d2 of 64 candidate columns against ALL rows of the query frame on the int8 MFMA
//                              with the operands swapped (candidates = B operand, one per lane), top-2 VALUES per column
//   F  match_finalize2_kernel  per pair: keep candidate (q, t) iff column t passes the ratio test with minimum d2(q, t);
//                              ordered compaction in the caller's row order, threshold
// ------------------------------------------------------------------------------------------------

// exclusive rank of this thread's flag among the workgroup's flags (thread order) + the workgroup's total
__device__ __forceinline__ int block_rank(bool flag, int tid, int* s_wave /* [FIN_THREADS / 64] */, int& total) {
    const unsigned long long bal = __ballot(flag);
    const int lane = tid & 63, wave = tid >> 6;
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < FIN_THREADS / 64; ++w) {
        const int c = s_wave[w];
        before += w < wave ? c : 0;
        all += c;
    }
    __syncthreads();
    total = all;
    return before + in_wave;
}

constexpr int VER_GROUPS = 2;                 // 32-candidate groups per wave: one fetched A fragment feeds two MFMA chains
constexpr int VER_CANDS = 32 * VER_GROUPS;    // candidates per work item

// rowcand[p][j] = {stored column of row j's best, d2} if stored row j passes the ratio test, else {~0, 0};
// candlist[p][i] = stored row of the i-th passing row (ascending j); state[p] = {passing rows, live, 0, 0}
__global__ __launch_bounds__(FIN_THREADS) void match_rows_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs, const uint4* __restrict__ rowres, int col_chunks,
    int row_stride, double ratio, int min_dir, int min_mutual, int mode, uint2* __restrict__ rowcand,
    int* __restrict__ candlist, int4* __restrict__ state, int2* __restrict__ items, int* __restrict__ n_items, int exp_all) {
    __shared__ int s_wave[FIN_THREADS / 64];
    __shared__ int s_item0;
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const int2 pr = pairs[p];
    const FrameDev A = frames[pr.x], B = frames[pr.y];
    const int A_even = A.meta[0], A_tiles = A.meta[1], B_tiles = B.meta[1];
    constexpr unsigned PAD_V = 2u * PADH;
    const int nchunks = (B_tiles + CHUNK_TILES - 1) / CHUNK_TILES;
    uint2* rc = rowcand + (size_t)p * row_stride;
    int* cl = candlist + (size_t)p * row_stride;
    int base = 0;
    for (int j0 = 0; j0 < 32 * A_tiles; j0 += FIN_THREADS) {
        const int j = j0 + tid;
        bool ok = false;
        unsigned col = 0;
        int d2 = 0;
        if (j < 32 * A_tiles && A.orig[j] >= 0) {
            unsigned v1 = 0xffffffffu, v2 = 0xffffffffu;
            for (int ch = 0; ch < nchunks; ++ch) {  // ascending columns; strict '<' keeps the lower column on ties
                const uint4 e = rowres[((size_t)p * col_chunks + ch) * row_stride + j];
                if (e.x < v1) {
                    v2 = umin(v1, e.z);
                    v1 = e.x;
                    col = e.y;
                } else {
                    v2 = umin(v2, e.x);
                }
            }
            const unsigned pa = (j >> 5) >= A_even ? 1u : 0u;
            d2 = (int)(v1 + pa) - 2;
            ok = v2 < PAD_V && ratio_pass(d2, (int)(v2 + pa) - 2, ratio);  // pad second => < 2 train rows
            if (exp_all) ok = true;  // (diagnostic: every row a candidate — the candidate pass then sweeps the whole pair)
        }
        if (j < 32 * A_tiles) rc[j] = ok ? make_uint2(col, (unsigned)d2) : make_uint2(0xffffffffu, 0u);
        int total;
        const int rank = block_rank(ok, tid, s_wave, total);
        if (ok) cl[base + rank] = j;
        base += total;
    }
    // main.cpp:111,142: an edge needs |m12| >= min_dir and |mutual| > min_mutual, and mutual is a subset of m12
    const bool live = (mode == 0 && base >= min_dir && base > min_mutual) || exp_all;
    const int groups = (base + VER_CANDS - 1) / VER_CANDS;
    if (tid == 0) {
        state[p] = make_int4(base, live ? 1 : 0, 0, 0);
        if (live) s_item0 = atomicAdd(n_items, groups);
    }
    if (live) {  // workgroup-uniform
        __syncthreads();
        for (int g = tid; g < groups; g += FIN_THREADS) items[s_item0 + g] = make_int2(p, g);
    }
}

// Behind the BOUND form of the sweep (match_sweep_kernel<KS, true>): rowres[p][j] = {v1, 0, u, 0} with u an upper bound of the row's
// second-smallest value. A row that fails the ratio test against u fails it against the truth and stays as it is (match_rows2_kernel
// will fail it on the same two numbers); every other row with a real minimum is a candidate of the exact pass — also a row whose bound
// is a padding value (its runner-up, if it has one, shares the minimum's subset: a train frame of two adjacent rows). Lists the
// candidates of the pair (candlist, state[p].x = their number) and appends one item per 64 of them for match_colverify_kernel<KS, true>.
__global__ __launch_bounds__(FIN_THREADS) void match_rowpick_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs, const uint4* __restrict__ rowres, int row_stride, double ratio,
    int* __restrict__ candlist, int4* __restrict__ state, int2* __restrict__ items, int* __restrict__ n_items) {
    __shared__ int s_wave[FIN_THREADS / 64];
    __shared__ int s_item0;
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const int2 pr = pairs[p];
    const FrameDev A = frames[pr.x];
    const int A_even = A.meta[0], A_tiles = A.meta[1];
    constexpr unsigned PAD_V = 2u * PADH;
    int* cl = candlist + (size_t)p * row_stride;
    int base = 0;
    for (int j0 = 0; j0 < 32 * A_tiles; j0 += FIN_THREADS) {
        const int j = j0 + tid;
        bool ok = false;
        if (j < 32 * A_tiles && A.orig[j] >= 0) {
            const uint4 e = rowres[(size_t)p * row_stride + j];
            const unsigned pa = (j >> 5) >= A_even ? 1u : 0u;
            ok = e.x < PAD_V && (e.z >= PAD_V || ratio_pass((int)(e.x + pa) - 2, (int)(e.z + pa) - 2, ratio));
        }
        int total;
        const int rank = block_rank(ok, tid, s_wave, total);
        if (ok) cl[base + rank] = j;
        base += total;
    }
    const int groups = (base + VER_CANDS - 1) / VER_CANDS;
    if (tid == 0) {
        state[p] = make_int4(base, 0, 0, 0);
        s_item0 = groups ? atomicAdd(n_items, groups) : 0;
    }
    __syncthreads();
    for (int g = tid; g < groups; g += FIN_THREADS) items[s_item0 + g] = make_int2(p, g);
}

// The rows kernel of the operand-swapped sweep (match_sweep_kernel): rowres carries the TILE of a row's minimum, not its column.
// Besides what match_rows_kernel does, the passing rows of a pair that goes on are grouped by that tile (bytile[p][..], a counting
// sort in LDS) and one argmin item is appended per (tile, <= 32 rows) — match_argmin_kernel turns the tile into the column.
constexpr int MAX_TILES = MAX_ROWS / 32;
__global__ __launch_bounds__(FIN_THREADS) void match_rows2_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs, const uint4* __restrict__ rowres, int row_stride, double ratio,
    int min_dir, int min_mutual, int mode, uint2* __restrict__ rowcand, int* __restrict__ candlist, int* __restrict__ bytile,
    int4* __restrict__ state, int2* __restrict__ vitems, int* __restrict__ n_vitems, int4* __restrict__ aitems, int* __restrict__ n_aitems,
    int exp_all) {
    __shared__ int s_wave[FIN_THREADS / 64];
    __shared__ int s_cnt[MAX_TILES], s_pos[MAX_TILES], s_grp[MAX_TILES];
    __shared__ int s_scan[FIN_THREADS], s_scan2[FIN_THREADS];
    __shared__ int s_item0, s_aitem0;
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const int2 pr = pairs[p];
    const FrameDev A = frames[pr.x], B = frames[pr.y];
    const int A_even = A.meta[0], A_tiles = A.meta[1], B_tiles = B.meta[1];
    constexpr unsigned PAD_V = 2u * PADH;
    uint2* rc = rowcand + (size_t)p * row_stride;
    int* cl = candlist + (size_t)p * row_stride;
    for (int t = tid; t < MAX_TILES; t += FIN_THREADS) s_cnt[t] = 0;
    __syncthreads();
    int base = 0;
    for (int j0 = 0; j0 < 32 * A_tiles; j0 += FIN_THREADS) {
        const int j = j0 + tid;
        bool ok = false;
        unsigned tile = 0;
        int d2 = 0;
        if (j < 32 * A_tiles && A.orig[j] >= 0) {
            const uint4 e = rowres[(size_t)p * row_stride + j];
            const unsigned pa = (j >> 5) >= A_even ? 1u : 0u;
            tile = e.y;
            d2 = (int)(e.x + pa) - 2;
            ok = e.z < PAD_V && ratio_pass(d2, (int)(e.z + pa) - 2, ratio);  // pad second => < 2 train rows
            if (exp_all) ok = e.x < PAD_V;
        }
        if (j < 32 * A_tiles) rc[j] = ok ? make_uint2(tile, (unsigned)d2) : make_uint2(0xffffffffu, 0u);
        if (ok) atomicAdd(&s_cnt[tile], 1);
        int total;
        const int rank = block_rank(ok, tid, s_wave, total);
        if (ok) cl[base + rank] = j;
        base += total;
    }
    // main.cpp:111,142: an edge needs |m12| >= min_dir and |mutual| > min_mutual, and mutual is a subset of m12
    const bool live = (mode == 0 && base >= min_dir && base > min_mutual) || (exp_all && base > 0);
    const bool want_col = live || (mode == 1 && base > 0);   // directed lists need the column of every passing row
    const int groups = (base + VER_CANDS - 1) / VER_CANDS;
    __syncthreads();
    if (want_col) {  // workgroup-uniform
        // exclusive prefix of the per-tile counts (positions in bytile) and of the per-tile item counts, two tiles per thread
        static_assert(MAX_TILES == 2 * FIN_THREADS, "two tiles per thread");
        const int c0 = 2 * tid < B_tiles ? s_cnt[2 * tid] : 0, c1 = 2 * tid + 1 < B_tiles ? s_cnt[2 * tid + 1] : 0;
        const int g0 = (c0 + 31) / 32, g1 = (c1 + 31) / 32;
        s_scan[tid] = c0 + c1;
        s_scan2[tid] = g0 + g1;
        __syncthreads();
        for (int off = 1; off < FIN_THREADS; off <<= 1) {
            const int u = tid >= off ? s_scan[tid - off] : 0, v = tid >= off ? s_scan2[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += u;
            s_scan2[tid] += v;
            __syncthreads();
        }
        const int pos0 = s_scan[tid] - (c0 + c1), grp0 = s_scan2[tid] - (g0 + g1), n_groups = s_scan2[FIN_THREADS - 1];
        s_pos[2 * tid] = pos0;
        s_pos[2 * tid + 1] = pos0 + c0;
        s_grp[2 * tid] = grp0;
        s_grp[2 * tid + 1] = grp0 + g0;
        if (tid == 0) s_aitem0 = atomicAdd(n_aitems, n_groups);
        __syncthreads();
        for (int t = tid; t < B_tiles; t += FIN_THREADS) {  // the items of tile t: rows [pos, pos + cnt) of bytile in groups of 32
            const int cnt = s_cnt[t], pos = s_pos[t];
            for (int g = 0; g < (cnt + 31) / 32; ++g) aitems[s_aitem0 + s_grp[t] + g] = make_int4(p, t, pos + 32 * g, min(32, cnt - 32 * g));
        }
        __syncthreads();
        int* bt = bytile + (size_t)p * row_stride;
        for (int i = tid; i < base; i += FIN_THREADS) {
            const int j = cl[i];
            bt[atomicAdd(&s_pos[rc[j].x], 1)] = j;   // (any order inside a tile: every row is resolved on its own)
        }
    }
    if (tid == 0) {
        state[p] = make_int4(base, live ? 1 : 0, 0, 0);
        if (live) s_item0 = atomicAdd(n_vitems, groups);
    }
    if (live) {  // workgroup-uniform
        __syncthreads();
        for (int g = tid; g < groups; g += FIN_THREADS) vitems[s_item0 + g] = make_int2(p, g);
    }
}

// rowcand[p][j].x: the tile of row j's minimum -> its column. One wave per item (train tile T of the pair, <= 32 passing rows whose
// minimum lies in T): the 32 x 32 distance block on the MFMA with the operands of match_sweep_kernel (the tile as A, the gathered
// query rows as B, hb as C-init), then every lane scans its 16 train rows in ascending order with a strict '<', the two halves
// of a query row are joined lower-index-first: the lowest train row of the minimum, the reference's tie rule.
template <int KS>
__global__ __launch_bounds__(64) void match_argmin_kernel(const FrameDev* __restrict__ frames, const int2* __restrict__ pairs,
                                                          uint2* __restrict__ rowcand, const int* __restrict__ bytile,
                                                          const int4* __restrict__ aitems, const int* __restrict__ n_aitems, int row_stride) {
    const int lane = threadIdx.x & 63, cl = lane & 31, h = lane >> 5;
    const int n = *n_aitems;
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
        const int4 it = aitems[w];
        const int p = it.x, tile = it.y;
        const int2 pr = pairs[p];
        const FrameDev A = frames[pr.x], B = frames[pr.y];
        const gfrag_t Afrag = (gfrag_t)A.frag, Bfrag = (gfrag_t)B.frag;
        const gint_t Bhb = (gint_t)B.normb;
        const int j = bytile[(size_t)p * row_stride + it.z + (cl < it.w ? cl : 0)];
        v16i acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4i q = *(const v4i __attribute__((address_space(1)))*)(Bhb + 32 * tile + 8 * g + 4 * h);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[4 * g + k] = q[k];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const v4i t = Bfrag[((size_t)tile * KS + ks) * 64 + lane];
            const v4i q = ~Afrag[((size_t)(j >> 5) * KS + ks) * 64 + 32 * h + (j & 31)];
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(t, q, acc, 0, 0, 0);
        }
        int best = 0x7fffffff, bi = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;   // ascending in r
            if (acc[r] < best) best = acc[r], bi = row;
        }
        const int o = __shfl_xor(best, 32), oi = __shfl_xor(bi, 32);
        if (o < best || (o == best && oi < bi)) bi = oi;
        if (h == 0 && cl < it.w) rowcand[(size_t)p * row_stride + j].x = (unsigned)(32 * tile + bi);
    }
}

// colres[p][j] = {v1, v2}: the two smallest 2H + pa over ALL stored rows of frame A against column rowcand[p][j].x of
// frame B (d2 = v + pb - 2), for the candidates of live pairs. Persistent workgroups over the item list.
//
// The MFMA of the sweep with its operands swapped: the 64 candidate columns of an item are the B operand (gathered once,
// two 32-column groups held in registers: a candidate is a LANE), the query frame's tiles stream as the A operand straight
// from L2, each fetched fragment feeding both groups' chains. Because a lane's 16 accumulators are 16 rows against ITS
// column, the column's constant hb is added once at the very end and the top-2 runs on the raw accumulators, three at a
// time: 27 VALU operations per 16 distances (the sweep needs 48) under 8 MFMAs — the kernel is bound by the matrix pipe,
// which the sweep (bound by the VALU port) leaves idle more than half of the time. Rows of the two parity classes are kept
// apart (a tile has one parity) and joined when 2H + pa is formed.
// ROWS = true: the same pass with the frames' roles swapped — the candidates are stored ROWS of the query frame (candlist holds them
// directly), the train frame's tiles stream — for the rows the bound form of the sweep (match_sweep_kernel<KS, true>) could not
// finish: their exact {v1 = 2H + pb minimum, first tile holding it, v2} go to rowres[p][j], what match_rows2_kernel reads. A wave
// meets its tiles in ascending order and takes a tile only on a STRICT improvement; equal minima of two waves / the two lanes of
// a candidate keep the lower tile: the first tile of the minimum, as the exact sweep records it.
template <int KS, bool ROWS = false>
__global__ __launch_bounds__(WG_THREADS) void match_colverify_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs, const uint2* __restrict__ rowcand,
    const int* __restrict__ candlist, const int4* __restrict__ state, const int2* __restrict__ items,
    const int* __restrict__ n_items, int row_stride, uint2* __restrict__ colres, uint4* __restrict__ rowres_out = nullptr) {
    __shared__ int4 sM[WAVES][VER_GROUPS][32];  // per wave, group, column: {m1 even, m2 even, m1 odd, m2 odd}
    __shared__ int2 sT[WAVES][VER_GROUPS][32];  // ROWS: {tile of m1 even, tile of m1 odd}
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 31, h = lane >> 5;
    const int n = *n_items;
    constexpr int BIG = 0x7fffffff;
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
        const int2 it = items[w];
        const int p = it.x;
        const int2 pr = pairs[p];
        const FrameDev A = frames[ROWS ? pr.y : pr.x], B = frames[ROWS ? pr.x : pr.y];   // A: the frame whose tiles stream, B: the candidates'
        const int A_even = ((gint_t)A.meta)[0], A_tiles = ((gint_t)A.meta)[1];
        const int ncand = state[p].x;
        const gfrag_t Afrag = (gfrag_t)A.frag, Bfrag = (gfrag_t)B.frag;
        const gint_t Aca = (gint_t)A.norm;
        const int* cand = candlist + (size_t)p * row_stride;
        const uint2* rc = rowcand + (size_t)p * row_stride;
        v4i b[VER_GROUPS][KS];
        int jg[VER_GROUPS];  // stored row of this lane's candidate per group (-1: past the list)
#pragma unroll
        for (int g = 0; g < VER_GROUPS; ++g) {
            const int i = VER_CANDS * it.y + 32 * g + cl;
            const int j = cand[i < ncand ? i : VER_CANDS * it.y];  // (an item has at least one candidate)
            jg[g] = i < ncand ? j : -1;
            const unsigned col = ROWS ? (unsigned)j : rc[j].x;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[g][ks] = Bfrag[((size_t)(col >> 5) * KS + ks) * 64 + 32 * h + (col & 31)];
        }
        int m1[VER_GROUPS][2], m2[VER_GROUPS][2];  // [group][row parity]
        int mt[VER_GROUPS][2];                      // ROWS: the tile of m1
#pragma unroll
        for (int g = 0; g < VER_GROUPS; ++g) m1[g][0] = m1[g][1] = m2[g][0] = m2[g][1] = BIG, mt[g][0] = mt[g][1] = 0;
        // this wave's tiles: wave, wave + 4, ...; the next tile's fragments and C-init are in flight under the current chains
        v4i a_nxt[KS];
        v4i c_nxt[4];
        auto fetch = [&](int t) {
            const int tc = min(t, A_tiles - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a_nxt[ks] = Afrag[((size_t)tc * KS + ks) * 64 + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q) c_nxt[q] = *(const v4i __attribute__((address_space(1)))*)(Aca + 32 * tc + 8 * q + 4 * h);  // rows 8q + 4h .. + 3: registers 4q .. 4q + 3
        };
        fetch(wave);
        for (int t = wave; t < A_tiles; t += WAVES) {
            v4i a[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a[ks] = ~a_nxt[ks];
            v16i c0;
#pragma unroll
            for (int r = 0; r < 16; ++r) c0[r] = c_nxt[r >> 2][r & 3];
            fetch(t + WAVES);
            v16i acc[VER_GROUPS];
#pragma unroll
            for (int g = 0; g < VER_GROUPS; ++g) {
                acc[g] = c0;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], b[g][ks], acc[g], 0, 0, 0);
            }
            const bool odd = t >= A_even;  // wave-uniform
#pragma unroll
            for (int g = 0; g < VER_GROUPS; ++g) {
                int x1 = odd ? m1[g][1] : m1[g][0], x2 = odd ? m2[g][1] : m2[g][0];
#pragma unroll
                for (int k = 0; k < 15; k += 3) {  // five triples: {min3, med3} + a sorted-pair merge
                    const int s1 = imin(imin(acc[g][k], acc[g][k + 1]), acc[g][k + 2]);
                    const int s2 = imed3(acc[g][k], acc[g][k + 1], acc[g][k + 2]);
                    top2_merge(x1, x2, s1, s2);
                }
                x2 = imed3(x1, x2, acc[g][15]);
                x1 = imin(x1, acc[g][15]);
                if constexpr (ROWS) {
                    const int before = odd ? m1[g][1] : m1[g][0];
                    if (odd) mt[g][1] = x1 < before ? t : mt[g][1];
                    else mt[g][0] = x1 < before ? t : mt[g][0];
                }
                if (odd) m1[g][1] = x1, m2[g][1] = x2;
                else m1[g][0] = x1, m2[g][0] = x2;
            }
        }
#pragma unroll
        for (int g = 0; g < VER_GROUPS; ++g) {  // lanes l and l + 32 hold the same column, different rows
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int o1 = __shfl_xor(m1[g][e], 32), o2 = __shfl_xor(m2[g][e], 32);
                if constexpr (ROWS) {
                    const int ot = __shfl_xor(mt[g][e], 32);
                    mt[g][e] = o1 < m1[g][e] ? ot : (o1 == m1[g][e] ? imin(ot, mt[g][e]) : mt[g][e]);
                }
                top2_merge(m1[g][e], m2[g][e], o1, o2);
            }
            if (h == 0) sM[wave][g][cl] = make_int4(m1[g][0], m2[g][0], m1[g][1], m2[g][1]);
            if (ROWS && h == 0) sT[wave][g][cl] = make_int2(mt[g][0], mt[g][1]);
        }
        __syncthreads();
        if (tid < 32 * VER_GROUPS) {
            const int g = tid >> 5, c = tid & 31;
            int4 m = sM[0][g][c];
            int2 mtile = ROWS ? sT[0][g][c] : make_int2(0, 0);
#pragma unroll
            for (int k = 1; k < WAVES; ++k) {
                const int4 e = sM[k][g][c];
                if constexpr (ROWS) {
                    const int2 et = sT[k][g][c];
                    mtile.x = e.x < m.x ? et.x : (e.x == m.x ? imin(et.x, mtile.x) : mtile.x);
                    mtile.y = e.z < m.z ? et.y : (e.z == m.z ? imin(et.y, mtile.y) : mtile.y);
                }
                top2_merge(m.x, m.y, e.x, e.y);
                top2_merge(m.z, m.w, e.z, e.w);
            }
            // H = acc + hb; v = 2 H + pa. A class without rows stays at BIG (never the minimum of a frame with rows)
            const int i = VER_CANDS * it.y + tid;
            if (i < ncand) {
                const int j = cand[i];
                const long long hb2 = 2ll * ((gint_t)B.normb)[ROWS ? (unsigned)j : rc[j].x];
                auto val = [&](int m, int pa) { return m == BIG ? 0xffffffffull : (unsigned long long)(2ll * m + hb2 + pa); };
                unsigned long long e1 = val(m.x, 0), e2 = val(m.y, 0), o1 = val(m.z, 1), o2 = val(m.w, 1);
                const unsigned long long v1 = e1 < o1 ? e1 : o1;
                const unsigned long long hi = e1 < o1 ? o1 : e1, lo2 = e2 < o2 ? e2 : o2;
                const unsigned long long v2 = hi < lo2 ? hi : lo2;
                if constexpr (ROWS) rowres_out[(size_t)p * row_stride + j] = make_uint4((unsigned)v1, (unsigned)(e1 < o1 ? mtile.x : mtile.y), (unsigned)v2, 0u);
                else colres[(size_t)p * row_stride + j] = make_uint2((unsigned)v1, (unsigned)v2);
            }
        }
        __syncthreads();
        (void)jg;
    }
}

// out_matches[p][k] = {q, t} sorted by q; counts[p] (mode 0: |mutual| if it exceeds min_mutual, else 0; mode 1: |m12|)
__global__ __launch_bounds__(FIN_THREADS) void match_finalize2_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs, const uint2* __restrict__ rowcand,
    const uint2* __restrict__ colres, const int4* __restrict__ state, int row_stride, double ratio, int min_mutual, int mode,
    uint2* __restrict__ out_matches, int* __restrict__ counts) {
    extern __shared__ int smem[];
    __shared__ int s_wave[FIN_THREADS / 64];
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const int4 st = state[p];
    if (st.x == 0 || (mode == 0 && !st.y)) {  // workgroup-uniform: no passing row, or a pair that cannot reach the thresholds
        if (tid == 0) counts[p] = 0;
        return;
    }
    const int2 pr = pairs[p];
    const FrameDev A = frames[pr.x], B = frames[pr.y];
    const int A_tiles = A.meta[1], B_even = B.meta[0];
    constexpr unsigned PAD_V = 2u * PADH;
    int* keepcol = smem;  // [row_stride] by stored row: stored column of the kept match or -1
    const uint2* rc = rowcand + (size_t)p * row_stride;
    const uint2* cr = colres + (size_t)p * row_stride;
    for (int j = tid; j < 32 * A_tiles; j += FIN_THREADS) {
        const uint2 c = rc[j];
        int keep = -1;
        if (c.x != 0xffffffffu) {
            if (mode == 1) {
                keep = (int)c.x;
            } else {
                // main.cpp:133-140: q is kept iff t's own best match is q, i.e. iff column t passes the ratio test (a unique
                // minimum for any ratio <= 1) and its minimum is d2(q, t)
                const uint2 v = cr[j];
                const unsigned pb = (c.x >> 5) >= (unsigned)B_even ? 1u : 0u;
                const int d1 = (int)(v.x + pb) - 2;
                if (d1 == (int)c.y && v.y < PAD_V && ratio_pass(d1, (int)(v.y + pb) - 2, ratio)) keep = (int)c.x;
            }
        }
        keepcol[j] = keep;
    }
    __syncthreads();
    uint2* out = out_matches + (size_t)p * row_stride;
    int base = 0;
    for (int q0 = 0; q0 < A.n; q0 += FIN_THREADS) {  // ordered compaction over the caller's q
        const int q = q0 + tid;
        const int t = q < A.n ? keepcol[A.pos[q]] : -1;
        int total;
        const int rank = block_rank(t >= 0, tid, s_wave, total);
        if (t >= 0) out[base + rank] = make_uint2((unsigned)q, (unsigned)B.orig[t]);
        base += total;
    }
    if (tid == 0) counts[p] = mode == 1 ? base : (base > min_mutual ? base : 0);  // main.cpp:142
}

// offsets[first + i] = running total; single workgroup, sequential over chunks (npairs is small)
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int* __restrict__ counts, int n,
                                                           long long* __restrict__ offsets,
                                                           long long* __restrict__ total,
                                                           int first, int is_last) {
    __shared__ long long s[1024];
    __shared__ long long carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = first == 0 ? 0 : *total;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        int i = i0 + tid;
        long long v = i < n ? counts[i] : 0;
        s[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            long long u = tid >= off ? s[tid - off] : 0;
            __syncthreads();
            s[tid] += u;
            __syncthreads();
        }
        if (i < n) offsets[first + i] = carry + s[tid] - v;
        __syncthreads();
        if (tid == 0) carry += s[1023];
        __syncthreads();
    }
    if (tid == 0) {
        *total = carry;
        if (is_last) offsets[first + n] = carry;
    }
}

__global__ void compact_edges_kernel(const uint2* __restrict__ matches, const int* __restrict__ counts,
                                     const long long* __restrict__ offsets, int row_stride,
                                     uint2* __restrict__ edges, long long edge_cap) {
    const int p = blockIdx.x;
    const int n = counts[p];
    const long long off = offsets[p];
    for (int k = threadIdx.x; k < n; k += blockDim.x)
        if (off + k < edge_cap) edges[off + k] = matches[(size_t)p * row_stride + k];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

static int ks_for_dim(int dim) {
    if (dim <= 0 || dim > 256 || dim % 16) return 0;
    if (dim <= 64) return 2;
    if (dim <= 128) return 4;
    return 8;
}

static int upload_frame(eacham_ctx* ctx, int frame_id, const float* src_dev, int n, int dim) {
    if (frame_id < 0 || frame_id >= (1 << 20)) return ctx->fail(EACHAM_ERR_INVALID, "frame_id %d out of range", frame_id);
    if (n < 0) return ctx->fail(EACHAM_ERR_INVALID, "negative row count");
    int ks = ks_for_dim(dim);
    if (!ks) return ctx->fail(EACHAM_ERR_UNSUPPORTED, "descriptor dim %d: need a multiple of 16, <= 256", dim);
    if (ctx->ks_common && (ctx->ks_common != ks || ctx->kind_common != 0))
        return ctx->fail(EACHAM_ERR_UNSUPPORTED, "all resident frames must share one descriptor kind (int8 / f32) and dim class");
    // each parity class is padded to whole tiles (at most one extra tile), the total to whole wave-blocks
    const int group_rows = 32 * GROUP_TILES;
    const int ntiles = n > 0 ? ((n + 31) / 32 + 1 + GROUP_TILES - 1) / GROUP_TILES * GROUP_TILES : 0;
    if (n > MAX_ROWS)
        return ctx->fail(EACHAM_ERR_UNSUPPORTED, "frame has %d rows; this build supports <= %d", n, MAX_ROWS);
    if ((size_t)frame_id >= ctx->frames.size()) ctx->frames.resize(frame_id + 1);
    FrameHost& f = ctx->frames[frame_id];
    if (f.frag || f.norm) {
        EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (f.frag) (void)hipFree(f.frag);
        if (f.norm) (void)hipFree(f.norm);  // the other per-row arrays share the allocation
        f = FrameHost();
    }
    const int npad = ntiles * 32;
    {
        // norm (ca) | normb (hb) | orig | pos | s2 | s1 | meta[2]
        const size_t ints = (size_t)6 * npad + 2;
        EACHAM_HIP_TRY(ctx, hipMalloc((void**)&f.norm, ints * sizeof(int)));
        EACHAM_HIP_TRY(ctx, hipMemsetAsync(f.norm, 0, ints * sizeof(int), ctx->stream));
        f.normb = f.norm + npad;
        f.orig = f.norm + 2 * (size_t)npad;
        f.pos = f.norm + 3 * (size_t)npad;
        int* s2 = f.norm + 4 * (size_t)npad;
        int* s1 = f.norm + 5 * (size_t)npad;
        f.meta = f.norm + 6 * (size_t)npad;
        if (npad > 0) {
            EACHAM_HIP_TRY(ctx, hipMalloc((void**)&f.frag, (size_t)ntiles * ks * 64 * sizeof(int4)));
            const long long sums = (long long)n * ((dim + 15) / 16);
            rowsum_kernel<<<(unsigned)((sums + 255) / 256), 256, 0, ctx->stream>>>(src_dev, n, dim, s2, s1, ctx->flag_dev);
            partition_kernel<<<1, 1024, 0, ctx->stream>>>(s2, s1, n, npad, group_rows, f.norm, f.normb, f.orig, f.pos, f.meta);
            const long long work = (long long)npad * ks * 2;
            quantize_kernel<<<(unsigned)((work + 255) / 256), 256, 0, ctx->stream>>>(src_dev, dim, ks, npad, f.orig, (v4i*)f.frag);
            EACHAM_HIP_TRY(ctx, hipGetLastError());
        }
    }
    f.n = n;
    f.dim = dim;
    f.ks = ks;
    f.ntiles = ntiles;
    ctx->ks_common = ks;
    ctx->kind_common = 0;
    ctx->frame_table_dirty = true;
    return EACHAM_OK;
}

static int check_integer_flag(eacham_ctx* ctx) {
    if (ctx->kind_common == 1) return EACHAM_OK;  // fp32 frames take any value
    int flag = 0;
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(&flag, ctx->flag_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (flag) {
        EACHAM_HIP_TRY(ctx, hipMemsetAsync(ctx->flag_dev, 0, sizeof(int), ctx->stream));
        return ctx->fail(EACHAM_ERR_NOT_INTEGER,
                         "descriptors must be integer-valued in [0,255] for the exact int8 path");
    }
    return EACHAM_OK;
}

struct MatchPlan {
    int batch;       // pairs per launch
    int round_pairs; // pairs of one round of 512 workgroups (launches are sized in whole rounds)
    int wb_stride;   // wave-blocks per frame (max over resident frames)
    int row_stride;  // padded rows per frame (max)
    int wgs_per_pair;
    int col_chunks;  // sweeps of <= 4096 train rows per pair
    int slots;       // workspace copies: batch i+1's tile kernel overlaps batch i's finalize
    size_t off_rowres, off_colpart, off_matches, slot_bytes, total;
    // the candidate-only column pass: rowcand | candlist | colres | state | items | n_items
    size_t off_rowcand, off_candlist, off_colres, off_state, off_items, off_nitems, off_bytile, off_aitems;
};

static MatchPlan make_plan(const eacham_ctx* ctx, int npairs, bool full_cols) {
    int max_tiles = std::max(4, GROUP_TILES);
    for (const auto& f : ctx->frames)
        if (f.n >= 0) max_tiles = std::max(max_tiles, f.tiles_used);
    MatchPlan pl;
    pl.row_stride = max_tiles * 32;
    pl.wb_stride = (max_tiles + (ROWS_PER_WG / 32) - 1) / (ROWS_PER_WG / 32) + 1;  // column-partial slots per pair: one per workgroup + 1
    pl.wgs_per_pair = (max_tiles + (ROWS_PER_WG / 32) - 1) / (ROWS_PER_WG / 32);
    pl.col_chunks = (max_tiles + CHUNK_TILES - 1) / CHUNK_TILES;
    // per pair: row results + match list, and EITHER the column partials of the full sweep OR the arrays of the candidate pass
    const size_t colpart_pp = full_cols ? (size_t)pl.wb_stride * pl.row_stride * sizeof(int2) : 0;
    const size_t cand_pp = full_cols ? 0 : (size_t)pl.row_stride * (sizeof(uint2) + 2 * sizeof(int) + sizeof(uint2)) + sizeof(int4) +
                                               (size_t)max_tiles * (sizeof(int2) + 2 * sizeof(int4));
    size_t per_pair = (size_t)pl.col_chunks * pl.row_stride * sizeof(int4) + (size_t)pl.row_stride * sizeof(uint2) + colpart_pp + cand_pp;
    // bound a slot near 1 GiB so the column partials of one batch stay cache-friendly
    size_t budget = (size_t)ctx->match_budget_mb << 20;
    int batch = (int)std::min<size_t>(std::max<size_t>(budget / per_pair, 1), (size_t)npairs);
    // whole rounds of workgroups (2 per CU x 256 CUs) keep the tail of a launch short
    const int wgs = pl.wgs_per_pair * pl.col_chunks;
    int round_pairs = 512;
    for (int g = 512; g >= 1; g >>= 1)
        if (wgs % g == 0) { round_pairs = 512 / g; break; }
    if (batch < npairs && batch > round_pairs) batch -= batch % round_pairs;
    // a job that fits one launch (a shard of a multi-GPU run) is still cut in two, so that the finalize
    // of its first half runs beside the tile kernel of the second
    if (batch >= npairs && npairs >= 8 * round_pairs) batch = ((npairs + 1) / 2 + round_pairs - 1) / round_pairs * round_pairs;
    pl.batch = std::max(batch, 1);
    pl.round_pairs = round_pairs;
    pl.slots = (pl.batch < npairs && !ctx->match_no_overlap) ? 2 : 1;  // (diagnostic switch, read once at eacham_ctx_create)
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    pl.off_rowres = 0;
    pl.off_colpart = align(pl.off_rowres + (size_t)pl.batch * pl.col_chunks * pl.row_stride * sizeof(int4));
    pl.off_matches = align(pl.off_colpart + (full_cols ? (size_t)pl.batch * pl.wb_stride * pl.row_stride * sizeof(int2) : 0));
    pl.off_rowcand = align(pl.off_matches + (size_t)pl.batch * pl.row_stride * sizeof(uint2));
    const size_t cb = full_cols ? 0 : (size_t)pl.batch;  // the candidate arrays exist in the other form only
    pl.off_candlist = align(pl.off_rowcand + cb * pl.row_stride * sizeof(uint2));
    pl.off_colres = align(pl.off_candlist + cb * pl.row_stride * sizeof(int));
    pl.off_state = align(pl.off_colres + cb * pl.row_stride * sizeof(uint2));
    pl.off_items = align(pl.off_state + cb * sizeof(int4));
    pl.off_nitems = align(pl.off_items + cb * max_tiles * sizeof(int2));
    pl.off_bytile = align(pl.off_nitems + 256);
    pl.off_aitems = align(pl.off_bytile + cb * pl.row_stride * sizeof(int));
    pl.slot_bytes = align(pl.off_aitems + cb * 2 * max_tiles * sizeof(int4));
    pl.total = pl.slot_bytes * pl.slots;
    return pl;
}

// Pairs of the launch that starts at `first`: full batches, the job's last one cut short (an eighth of a full one) when two
// workspace slots are in use. One definition for run_match and for eacham_match_debug_batches.
static int next_batch(const MatchPlan& pl, int first, int npairs) {
    const int tail_pairs = std::max(pl.round_pairs, pl.batch / 8 / pl.round_pairs * pl.round_pairs);
    int nb = std::min(pl.batch, npairs - first);
    if (pl.slots == 2 && first + nb == npairs && nb > 2 * tail_pairs)
        nb = (nb - tail_pairs + pl.round_pairs - 1) / pl.round_pairs * pl.round_pairs;  // leave the tail for one more, short launch
    return nb;
}

template <int KS>
static void launch_tile(eacham_ctx* ctx, const MatchPlan& pl, const int2* pairs_dev, int nb, char* ws, bool col) {
    if (!col)  // the sweep without its column direction: the candidate-only pass follows (or nothing, for directed lists)
        match_tile_kernel<KS, MATCH_NSUB, false><<<nb * pl.wgs_per_pair * pl.col_chunks, WG_THREADS, 0, ctx->stream>>>(
            ctx->frame_table_dev, pairs_dev, pl.wgs_per_pair, pl.col_chunks, (uint4*)(ws + pl.off_rowres),
            (uint2*)(ws + pl.off_colpart), pl.wb_stride, pl.row_stride);
    else
        match_tile_kernel<KS, MATCH_NSUB, true><<<nb * pl.wgs_per_pair * pl.col_chunks, WG_THREADS, 0, ctx->stream>>>(
            ctx->frame_table_dev, pairs_dev, pl.wgs_per_pair, pl.col_chunks, (uint4*)(ws + pl.off_rowres),
            (uint2*)(ws + pl.off_colpart), pl.wb_stride, pl.row_stride);
}

// Core driver. mode 0 = mutual (CSR out), mode 1 = directed single pair (fixed-stride out in ws).
static int run_match(eacham_ctx* ctx, const int2* pairs_dev, int npairs, double ratio, int min_dir,
                     int min_mutual, int mode, int* counts_dev, long long* offsets_dev,
                     uint2* edges_dev, long long edge_cap, long long* total_dev, int4* stats_dev) {
    int rc = sync_frame_table(ctx);
    if (rc) return rc;
    if (npairs <= 0) return EACHAM_OK;
    if (mode == 0 && !(ratio <= 1.0))  // the mutual check relies on a passing column having a unique minimum
        return ctx->fail(EACHAM_ERR_INVALID, "ratio %g: mutual matching supports 0 < ratio <= 1 (the reference uses 0.8)", ratio);
    rc = sanitize_pairs(ctx, pairs_dev, npairs, &pairs_dev);  // a bad frame id in a device-side list must not reach the kernels
    if (rc) return rc;
    if (ctx->kind_common == 1)
        return run_match_f32(ctx, pairs_dev, npairs, ratio, min_dir, min_mutual, mode, counts_dev, offsets_dev, edges_dev,
                             edge_cap, total_dev, stats_dev);
    // Which form of the column direction: the reference's thresholds (30 / 30, main.cpp:111,142) make the direction counts
    // redundant once |mutual| > min_mutual >= min_dir - 1, so only the columns that are some passing row's best are ever
    // looked at (candidate-only pass); directed lists need no column at all. Callers that ask for the per-pair statistics
    // (|m12|, |m21|) or use other thresholds get every column's top-2 from the sweep itself, as before.
    const bool full_cols = ctx->match_full_columns || (mode == 0 && (stats_dev != nullptr || (long long)min_mutual < (long long)min_dir - 1));
    MatchPlan pl = make_plan(ctx, npairs, full_cols);
    rc = ensure_workspace(ctx, pl.total);
    if (rc) return rc;
    ctx->last_matches = (char*)ctx->ws + pl.off_matches;
    const size_t fin_smem = (size_t)(full_cols ? 2 : 1) * pl.row_stride * sizeof(int);
    if (fin_smem > 48 * 1024) {
        if (full_cols) EACHAM_HIP_TRY(ctx, hipFuncSetAttribute((const void*)match_finalize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_smem));
        else EACHAM_HIP_TRY(ctx, hipFuncSetAttribute((const void*)match_finalize2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_smem));
    }
    // Two streams: the tile kernels run back to back on the context stream; finalize, scan and
    // compaction of a batch run on stream2 beside the next batch's tile kernel (they are bound by
    // HBM/L2, the tile kernel by the vector ALU), each batch in its own workspace slot.
    hipStream_t st1 = ctx->stream, st2 = pl.slots == 2 || npairs <= pl.batch ? ctx->stream2 : ctx->stream;
    EACHAM_HIP_TRY(ctx, hipEventRecord(ctx->ev_join, st1));        // inputs queued on the context stream
    EACHAM_HIP_TRY(ctx, hipStreamWaitEvent(st2, ctx->ev_join, 0));
    // The work behind a batch's sweep (rows / candidate columns / finalize / compaction) runs on the second stream beside the NEXT
    // batch's sweep — except the last batch's, which nothing hides: the job's last batch is cut short (an eighth of a full one),
    // so the exposed tail is that of ~1 500 pairs instead of ~10 000 (0.8 ms of a 21 ms S200 step).
    int b = 0;
    for (int first = 0, nb = 0; first < npairs; first += nb, ++b) {
        nb = next_batch(pl, first, npairs);
        const int slot = b % pl.slots;
        char* ws = (char*)ctx->ws + (size_t)slot * pl.slot_bytes;
        const int2* pb = pairs_dev + first;
        if (b >= pl.slots) EACHAM_HIP_TRY(ctx, hipStreamWaitEvent(st1, ctx->ev_fin[slot], 0));  // slot free again
        const bool row_sweep = !full_cols && !ctx->match_tile_sweep;   // the operand-swapped row sweep (match_sweep_kernel): the default of the lean form
        // its BOUND form + the exact pass over the rows it could not finish: up to 128-D, where the sweep is bound by its epilogue's
        // VALU work (S200 at 128-D +5 %, the 1000-frame KITTI stand-in +12 %); at 256-D the sweep is bound by the matrix pipe and gains
        // 4 % while the exact pass (one more stream of the train frame per pair beside the next sweep) costs the step 4-8 %
        // (profiles/r05_match_bound_sweep_ab.txt)
        const bool bound_sweep = row_sweep && (ctx->match_sweep_form == 2 || (ctx->match_sweep_form == 0 && ctx->ks_common <= 4));
        {
            ProfileScope ps(ctx, EACHAM_KERNEL_MATCH_TILE, st1);
            const bool col = full_cols && !ctx->exp_no_coltop2;
            if (row_sweep) {
                switch (ctx->ks_common) {
#define EACHAM_SWEEP(KS_, BOUND_) match_sweep_kernel<KS_, BOUND_><<<nb * pl.wgs_per_pair, WG_THREADS, 0, st1>>>(ctx->frame_table_dev, pb, pl.wgs_per_pair, (uint4*)(ws + pl.off_rowres), pl.row_stride, ctx->exp_sweep_prio)
                    case 2: if (bound_sweep) EACHAM_SWEEP(2, true); else EACHAM_SWEEP(2, false); break;
                    case 4: if (bound_sweep) EACHAM_SWEEP(4, true); else EACHAM_SWEEP(4, false); break;
                    default: if (bound_sweep) EACHAM_SWEEP(8, true); else EACHAM_SWEEP(8, false); break;
#undef EACHAM_SWEEP
                }
            } else {
                switch (ctx->ks_common) {
                    case 2: launch_tile<2>(ctx, pl, pb, nb, ws, col); break;
                    case 4: launch_tile<4>(ctx, pl, pb, nb, ws, col); break;
                    default: launch_tile<8>(ctx, pl, pb, nb, ws, col); break;
                }
            }
        }
        EACHAM_HIP_TRY(ctx, hipEventRecord(ctx->ev_tile[slot], st1));
        EACHAM_HIP_TRY(ctx, hipStreamWaitEvent(st2, ctx->ev_tile[slot], 0));
        const bool csr = offsets_dev != nullptr;  // mode 1 without offsets: the single directed pair of eacham_match_pair
        int* cnt = csr ? counts_dev + first : counts_dev;
        {
            ProfileScope ps(ctx, EACHAM_KERNEL_MATCH_FINALIZE, st2);
            if (full_cols) {
                match_finalize_kernel<<<nb, FIN_THREADS, fin_smem, st2>>>(
                    ctx->frame_table_dev, pb, (const uint4*)(ws + pl.off_rowres),
                    (const uint2*)(ws + pl.off_colpart), pl.col_chunks, pl.wb_stride, pl.row_stride, ratio, min_dir,
                    min_mutual, mode, (uint2*)(ws + pl.off_matches), cnt,
                    stats_dev ? stats_dev + first : nullptr);
            } else {
                uint2* rowcand = (uint2*)(ws + pl.off_rowcand);
                int* candlist = (int*)(ws + pl.off_candlist);
                uint2* colres = (uint2*)(ws + pl.off_colres);
                int4* state = (int4*)(ws + pl.off_state);
                int2* items = (int2*)(ws + pl.off_items);
                int* n_items = (int*)(ws + pl.off_nitems);
                int* n_aitems = n_items + 16;
                EACHAM_HIP_TRY(ctx, hipMemsetAsync(n_items, 0, 32 * sizeof(int), st2));
                if (bound_sweep) {   // the rows the bound form left open: exact {v1, tile, v2} into rowres, before the rows kernel reads it
                    int* n_pre = n_items + 8;
                    match_rowpick_kernel<<<nb, FIN_THREADS, 0, st2>>>(ctx->frame_table_dev, pb, (const uint4*)(ws + pl.off_rowres), pl.row_stride, ratio,
                                                                      candlist, state, items, n_pre);
                    const int vgrid = std::min(std::max(nb * pl.wgs_per_pair, 1), 512);
                    switch (ctx->ks_common) {
                        case 2: match_colverify_kernel<2, true><<<vgrid, WG_THREADS, 0, st2>>>(ctx->frame_table_dev, pb, nullptr, candlist, state, items, n_pre, pl.row_stride, nullptr, (uint4*)(ws + pl.off_rowres)); break;
                        case 4: match_colverify_kernel<4, true><<<vgrid, WG_THREADS, 0, st2>>>(ctx->frame_table_dev, pb, nullptr, candlist, state, items, n_pre, pl.row_stride, nullptr, (uint4*)(ws + pl.off_rowres)); break;
                        default: match_colverify_kernel<8, true><<<vgrid, WG_THREADS, 0, st2>>>(ctx->frame_table_dev, pb, nullptr, candlist, state, items, n_pre, pl.row_stride, nullptr, (uint4*)(ws + pl.off_rowres)); break;
                    }
                }
                if (row_sweep) {
                    int* bytile = (int*)(ws + pl.off_bytile);
                    int4* aitems = (int4*)(ws + pl.off_aitems);
                    match_rows2_kernel<<<nb, FIN_THREADS, 0, st2>>>(ctx->frame_table_dev, pb, (const uint4*)(ws + pl.off_rowres), pl.row_stride, ratio,
                                                                    min_dir, min_mutual, mode, rowcand, candlist, bytile, state, items, n_items, aitems,
                                                                    n_aitems, ctx->exp_all_candidates ? 1 : 0);
                    const int agrid = std::min(std::max(nb * pl.wgs_per_pair, 1), 4096);   // one wave per item, persistent over the list
                    switch (ctx->ks_common) {
                        case 2: match_argmin_kernel<2><<<agrid, 64, 0, st2>>>(ctx->frame_table_dev, pb, rowcand, bytile, aitems, n_aitems, pl.row_stride); break;
                        case 4: match_argmin_kernel<4><<<agrid, 64, 0, st2>>>(ctx->frame_table_dev, pb, rowcand, bytile, aitems, n_aitems, pl.row_stride); break;
                        default: match_argmin_kernel<8><<<agrid, 64, 0, st2>>>(ctx->frame_table_dev, pb, rowcand, bytile, aitems, n_aitems, pl.row_stride); break;
                    }
                } else {
                    match_rows_kernel<<<nb, FIN_THREADS, 0, st2>>>(ctx->frame_table_dev, pb, (const uint4*)(ws + pl.off_rowres), pl.col_chunks,
                                                                   pl.row_stride, ratio, min_dir, min_mutual, mode, rowcand, candlist, state,
                                                                   items, n_items, ctx->exp_all_candidates ? 1 : 0);
                }
                if (mode == 0) {
                    // persistent workgroups over the item list (its length is only known on the device): one round of the chip
                    const int vgrid = std::min(std::max(nb * pl.wgs_per_pair, 1), 512);
                    switch (ctx->ks_common) {
                        case 2: match_colverify_kernel<2><<<vgrid, WG_THREADS, 0, st2>>>(ctx->frame_table_dev, pb, rowcand, candlist, state, items, n_items, pl.row_stride, colres); break;
                        case 4: match_colverify_kernel<4><<<vgrid, WG_THREADS, 0, st2>>>(ctx->frame_table_dev, pb, rowcand, candlist, state, items, n_items, pl.row_stride, colres); break;
                        default: match_colverify_kernel<8><<<vgrid, WG_THREADS, 0, st2>>>(ctx->frame_table_dev, pb, rowcand, candlist, state, items, n_items, pl.row_stride, colres); break;
                    }
                }
                match_finalize2_kernel<<<nb, FIN_THREADS, fin_smem, st2>>>(ctx->frame_table_dev, pb, rowcand, colres, state, pl.row_stride, ratio,
                                                                          min_mutual, mode, (uint2*)(ws + pl.off_matches), cnt);
            }
            if (csr) {
                scan_counts_kernel<<<1, 1024, 0, st2>>>(cnt, nb, offsets_dev, total_dev, first, first + nb == npairs);
                compact_edges_kernel<<<nb, 256, 0, st2>>>((const uint2*)(ws + pl.off_matches), cnt, offsets_dev + first,
                                                          pl.row_stride, edges_dev, edge_cap);
            }
        }
        EACHAM_HIP_TRY(ctx, hipEventRecord(ctx->ev_fin[slot], st2));
        EACHAM_HIP_TRY(ctx, hipGetLastError());
    }
    EACHAM_HIP_TRY(ctx, hipEventRecord(ctx->ev_join, st2));         // later work on the context stream sees the results
    EACHAM_HIP_TRY(ctx, hipStreamWaitEvent(st1, ctx->ev_join, 0));
    return EACHAM_OK;
}

void launch_scan_counts(eacham_ctx* ctx, const int* counts, int n, long long* offsets, long long* total, int first, int is_last) {
    scan_counts_kernel<<<1, 1024, 0, ctx->stream>>>(counts, n, offsets, total, first, is_last);
}
void launch_compact_edges(eacham_ctx* ctx, int nb, const uint2* matches, const int* counts, const long long* offsets,
                          int row_stride, uint2* edges, long long edge_cap) {
    compact_edges_kernel<<<nb, 256, 0, ctx->stream>>>(matches, counts, offsets, row_stride, edges, edge_cap);
}

static int check_pairs_host(eacham_ctx* ctx, const int32_t* pairs, int npairs) {
    for (int i = 0; i < 2 * npairs; ++i) {
        int f = pairs[i];
        if (f < 0 || (size_t)f >= ctx->frames.size() || ctx->frames[f].n < 0)
            return ctx->fail(EACHAM_ERR_INVALID, "pair %d references frame %d which is not resident", i / 2, f);
    }
    return EACHAM_OK;
}

}  // namespace eacham

using namespace eacham;

extern "C" {


int eacham_upload_descriptors_dev(eacham_ctx* ctx, int frame_id, const float* rowmajor_dev, int n, int dim) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (n > 0 && !rowmajor_dev) return ctx->fail(EACHAM_ERR_INVALID, "null descriptor pointer");
    return upload_frame(ctx, frame_id, rowmajor_dev, n, dim);
}

int eacham_upload_descriptors(eacham_ctx* ctx, int frame_id, const float* rowmajor, int n, int dim) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (n > 0 && !rowmajor) return ctx->fail(EACHAM_ERR_INVALID, "null descriptor pointer");
    if (n < 0 || dim <= 0) return ctx->fail(EACHAM_ERR_INVALID, "bad descriptor shape %d x %d", n, dim);
    size_t bytes = (size_t)n * dim * sizeof(float);
    int rc = ensure_io(ctx, std::max<size_t>(bytes, 256));
    if (rc) return rc;
    if (bytes) {
        EACHAM_HIP_TRY(ctx, hipMemcpyAsync(ctx->io, rowmajor, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = upload_frame(ctx, frame_id, (const float*)ctx->io, n, dim);
    if (rc) return rc;
    return check_integer_flag(ctx);  // also orders reuse of the staging buffer
}

int eacham_frame_rows(eacham_ctx* ctx, int frame_id) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (frame_id < 0 || (size_t)frame_id >= ctx->frames.size() || ctx->frames[frame_id].n < 0)
        return ctx->fail(EACHAM_ERR_INVALID, "frame %d is not resident", frame_id);
    return ctx->frames[frame_id].n;
}

int eacham_match_pair(eacham_ctx* ctx, int f1, int f2, double ratio, uint32_t* out_q, uint32_t* out_t,
                      int cap, int* out_count) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (!out_count || cap < 0 || (cap > 0 && (!out_q || !out_t))) return ctx->fail(EACHAM_ERR_INVALID, "null output");
    int32_t pr[2] = {f1, f2};
    int rc = check_pairs_host(ctx, pr, 1);
    if (rc) return rc;
    rc = check_integer_flag(ctx);
    if (rc) return rc;
    rc = ensure_io(ctx, 256);
    if (rc) return rc;
    int2* pairs_dev = (int2*)ctx->io;
    int* count_dev = (int*)((char*)ctx->io + 64);
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(pairs_dev, pr, sizeof(pr), hipMemcpyHostToDevice, ctx->stream));
    rc = run_match(ctx, pairs_dev, 1, ratio, 0, 0, /*mode=*/1, count_dev, nullptr, nullptr, 0, nullptr, nullptr);
    if (rc) return rc;
    int count = 0;
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(&count, count_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out_count = count;
    if (count > cap) return ctx->fail(EACHAM_ERR_CAPACITY, "%d matches but capacity %d", count, cap);
    if (count > 0) {
        std::vector<uint2> tmp(count);
        EACHAM_HIP_TRY(ctx, hipMemcpy(tmp.data(), ctx->last_matches, sizeof(uint2) * count, hipMemcpyDeviceToHost));
        for (int k = 0; k < count; ++k) {
            out_q[k] = tmp[k].x;
            out_t[k] = tmp[k].y;
        }
    }
    return EACHAM_OK;
}

int eacham_match_all_pairs_dev(eacham_ctx* ctx, const int32_t* pairs_dev, int npairs, double ratio,
                               int min_dir, int min_mutual, int32_t* counts_dev, int64_t* offsets_dev,
                               uint32_t* edges_dev, int64_t edge_cap, int64_t* total_dev,
                               int32_t* stats_dev) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (npairs < 0 || (npairs > 0 && (!pairs_dev || !counts_dev || !offsets_dev || !total_dev)) ||
        edge_cap < 0 || (edge_cap > 0 && !edges_dev))
        return ctx->fail(EACHAM_ERR_INVALID, "bad arguments to match_all_pairs_dev");
    return run_match(ctx, (const int2*)pairs_dev, npairs, ratio, min_dir, min_mutual, 0, counts_dev,
                     (long long*)offsets_dev, (uint2*)edges_dev, edge_cap, (long long*)total_dev,
                     (int4*)stats_dev);
}

// host pointers in, CSR over the pairs out: mode 0 = mutual + thresholds, mode 1 = directed lists
static int match_pairs_host(eacham_ctx* ctx, const int32_t* pairs, int npairs, double ratio, int min_dir,
                            int min_mutual, int mode, int32_t* counts, int64_t* offsets, uint32_t* out_q,
                            uint32_t* out_t, int64_t cap, int64_t* out_total, int32_t* stats) {
    if (npairs < 0 || (npairs > 0 && (!pairs || !counts || !offsets)) || !out_total || cap < 0 || (cap > 0 && (!out_q || !out_t)))
        return ctx->fail(EACHAM_ERR_INVALID, "bad arguments to match_all_pairs");
    int rc = check_pairs_host(ctx, pairs, npairs);
    if (rc) return rc;
    rc = check_integer_flag(ctx);
    if (rc) return rc;
    *out_total = 0;
    if (npairs == 0) {
        if (offsets) offsets[0] = 0;
        return EACHAM_OK;
    }
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o_pairs = 0;
    size_t o_counts = align(o_pairs + (size_t)npairs * 2 * sizeof(int32_t));
    size_t o_offsets = align(o_counts + (size_t)npairs * sizeof(int32_t));
    size_t o_total = align(o_offsets + (size_t)(npairs + 1) * sizeof(int64_t));
    size_t o_stats = align(o_total + sizeof(int64_t));
    size_t o_edges = align(o_stats + (size_t)npairs * 4 * sizeof(int32_t));
    size_t bytes = o_edges + (size_t)cap * sizeof(uint2);
    rc = ensure_io(ctx, bytes);
    if (rc) return rc;
    char* io = (char*)ctx->io;
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(io + o_pairs, pairs, (size_t)npairs * 2 * sizeof(int32_t),
                                       hipMemcpyHostToDevice, ctx->stream));
    rc = run_match(ctx, (const int2*)(io + o_pairs), npairs, ratio, min_dir, min_mutual, mode,
                   (int*)(io + o_counts), (long long*)(io + o_offsets), (uint2*)(io + o_edges), cap,
                   (long long*)(io + o_total), stats ? (int4*)(io + o_stats) : nullptr);
    if (rc) return rc;
    long long total = 0;
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(&total, io + o_total, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(counts, io + o_counts, (size_t)npairs * sizeof(int32_t),
                                       hipMemcpyDeviceToHost, ctx->stream));
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(offsets, io + o_offsets, (size_t)(npairs + 1) * sizeof(int64_t),
                                       hipMemcpyDeviceToHost, ctx->stream));
    if (stats)
        EACHAM_HIP_TRY(ctx, hipMemcpyAsync(stats, io + o_stats, (size_t)npairs * 4 * sizeof(int32_t),
                                           hipMemcpyDeviceToHost, ctx->stream));
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out_total = total;
    if (total > cap) return ctx->fail(EACHAM_ERR_CAPACITY, "%lld matches but capacity %lld", total, (long long)cap);
    if (total > 0) {
        std::vector<uint2> tmp((size_t)total);
        EACHAM_HIP_TRY(ctx, hipMemcpy(tmp.data(), io + o_edges, sizeof(uint2) * (size_t)total, hipMemcpyDeviceToHost));
        for (long long k = 0; k < total; ++k) {
            out_q[k] = tmp[k].x;
            out_t[k] = tmp[k].y;
        }
    }
    return EACHAM_OK;
}

int eacham_match_all_pairs(eacham_ctx* ctx, const int32_t* pairs, int npairs, double ratio, int min_dir,
                           int min_mutual, int32_t* counts, int64_t* offsets, uint32_t* out_q,
                           uint32_t* out_t, int64_t cap, int64_t* out_total, int32_t* stats) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return match_pairs_host(ctx, pairs, npairs, ratio, min_dir, min_mutual, 0, counts, offsets, out_q, out_t, cap, out_total, stats);
}

int eacham_match_debug_batches(eacham_ctx* ctx, int npairs, int with_stats, int32_t* starts, int cap, int* n_batches, int* n_slots) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (npairs < 0 || !n_batches || cap < 0 || (cap > 0 && !starts)) return ctx->fail(EACHAM_ERR_INVALID, "bad arguments to match_debug_batches");
    if (ctx->kind_common == 1) return ctx->fail(EACHAM_ERR_UNSUPPORTED, "the fp32 path plans its batches on its own");
    const bool full_cols = ctx->match_full_columns || with_stats != 0;
    MatchPlan pl = make_plan(ctx, std::max(npairs, 1), full_cols);
    int b = 0;
    for (int first = 0, nb = 0; first < npairs; first += nb, ++b) {
        nb = next_batch(pl, first, npairs);
        if (b < cap) starts[b] = first;
    }
    *n_batches = b;
    if (n_slots) *n_slots = pl.slots;
    return EACHAM_OK;
}

int eacham_match_pairs_directed(eacham_ctx* ctx, const int32_t* pairs, int npairs, double ratio, int32_t* counts,
                                int64_t* offsets, uint32_t* out_q, uint32_t* out_t, int64_t cap, int64_t* out_total) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return match_pairs_host(ctx, pairs, npairs, ratio, 0, 0, 1, counts, offsets, out_q, out_t, cap, out_total, nullptr);
}

}  // extern "C"
