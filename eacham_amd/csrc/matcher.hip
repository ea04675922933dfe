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
//     d2(q,t) = |a_q|^2 + |b_t|^2 - 2 a_q.b_t         a.b from the int8 MFMA, int32 accumulate
// Row direction (q -> best t) ranks by  |b_t|^2 - 2 a.b  (|a_q|^2 is constant per row),
// column direction (t -> best q) by     |a_q|^2 - 2 a.b.
// Both rank values fit 25 signed bits for D <= 256, so a candidate is one int32 key
//     key = (rank << 7) | code        code = column-tile index (row dir) / local row (col dir)
// and a running top-2 costs two VALU ops:  m2 = med3(m1, m2, key); m1 = min(m1, key).
// Ties resolve to the lower index because `code` is monotone in the index within a lane.
#include "context.hpp"

#include <algorithm>
#include <climits>

namespace eacham {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// Pointers read out of the device-side frame table are generic to the compiler; casting them to the
// global address space turns flat_load (which ties vmcnt to lgkmcnt) into global_load.
typedef const v4i __attribute__((address_space(1)))* gfrag_t;
typedef const int __attribute__((address_space(1)))* gint_t;

// norm of padding rows/columns: above any real d2 (<= 256*255^2 = 16,646,400) and low enough that
// PAD_VALUE + 2^16 still fits the 25-bit signed rank field of a key
constexpr int PAD_VALUE = 16700000;
constexpr int KEY_SHIFT = 7;
constexpr int KEY_MASK = (1 << KEY_SHIFT) - 1;
constexpr int CHUNK_TILES = 1 << KEY_SHIFT;  // column tiles one sweep can tag in a key: 128 * 32 = 4096 train rows
constexpr int MAX_TILES = 512;               // 16384 rows per frame (K2 keeps two int per row in LDS)
constexpr int WG_THREADS = 256;            // 4 waves (1 per SIMD); 2 workgroups per CU drift out of phase so MFMA and VALU overlap
constexpr int WAVES = WG_THREADS / 64;
#ifndef EACHAM_MATCH_NSUB
#define EACHAM_MATCH_NSUB 2
#endif
constexpr int MATCH_NSUB = EACHAM_MATCH_NSUB;        // 32-row MFMA sub-tiles per wave (A fragments live in VGPRs)
constexpr int ROWS_PER_WAVE = 32 * MATCH_NSUB;
constexpr int ROWS_PER_WG = WAVES * ROWS_PER_WAVE;
constexpr int TILE_ALIGN = 4;                        // frames are padded to a multiple of 4 tiles (128 rows)

__device__ __forceinline__ int med3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// ------------------------------------------------------------------------------------------------
// upload: fp32 row-major -> fragment-major int8 + squared norms
// ------------------------------------------------------------------------------------------------

__global__ void init_norm_kernel(int* __restrict__ norm, int* __restrict__ normb, int n, int npad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) {
        norm[i] = i < n ? 0 : PAD_VALUE;
        normb[i] = i < n ? 0 : PAD_VALUE;
    }
}

// norm[row]  = sum c^2             (c = x - 128, the stored int8)            -> query role
// normb[row] = sum c^2 + 2 sum c   (absorbs the -1 of the ~a trick, see K1)  -> train role
__global__ void quantize_kernel(const float* __restrict__ src, int n, int dim, int KS, int npad,
                                v4i* __restrict__ frag, int* __restrict__ norm,
                                int* __restrict__ normb, int* __restrict__ bad_flag) {
    const int chunks = KS * 2;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)npad * chunks) return;
    int row = (int)(idx / chunks), ch = (int)(idx % chunks);
    int ks = ch >> 1, h = ch & 1, tile = row >> 5, r = row & 31;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    int sq = 0, sum = 0;
    bool bad = false;
    if (row < n) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int k = ks * 32 + h * 16 + j;
            int c = 0;  // centred value of a padded dimension
            if (k < dim) {
                float v = src[(size_t)row * dim + k];
                if (!(v >= 0.0f && v <= 255.0f) || v != floorf(v)) bad = true;
                c = (int)v - 128;
            }
            sq += c * c;
            sum += c;
            w[j >> 2] |= (unsigned)(c & 0xff) << (8 * (j & 3));
        }
    }
    v4i out = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    frag[((size_t)tile * KS + ks) * 64 + h * 32 + r] = out;
    if (sq) atomicAdd(&norm[row], sq);
    if (sq + 2 * sum) atomicAdd(&normb[row], sq + 2 * sum);
    if (bad) atomicOr(bad_flag, 1);
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
// tile only 16 x 32), so the epilogue is cut to 6 ops per distance:
//   * the query fragments are complemented once (a' = ~a = -a-1 per byte), so the MFMA returns
//     acc = -a.b - sum(b) and each key is ONE v_lshl_add_u32:
//         row key  = (acc << 8) + ((normb_c << 7) | t)        rank = |b|^2 - 2 a.b
//         col key  = (acc << 8) + ((norm_r  << 7) | lrow)     rank = |a|^2 - 2 a.b - 2 sum(b)
//     (normb = |b|^2 + 2 sum(b) is precomputed at upload; the -2 sum(b) offset of the column
//     rank is constant per column and is added back by K2);
//   * each running top-2 update is v_med3_i32 + v_min_i32.
// Measured on MI355X (tools/valu_ubench.hip, tools/mfma_ubench.hip): a wave64 integer VALU op costs
// ~4 cycles of its SIMD whatever the number of resident waves, the int8 32x32x32 MFMA 32 cycles.
// With 6 ops per distance the VALU floor is 192 ops = 768 cycles per 64x32 wave-tile against 512
// MFMA cycles, so the MFMA chains are software-pipelined UNDER the epilogue (see the main loop).
// out   rowres[p][q]      = {rank1, col1, rank2, 0}   final over all columns (d2 = rank + norm_q)
//       colpart[p][wb][c] = {key1, key2}              top-2 over the 64 rows of wave-block wb
//                                                     (d2 = (key >> 7) + normb_c, row = key & 127)
__device__ __forceinline__ int vmed3(int a, int b, int c) {
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Two keys from one accumulator, each ONE v_lshl_add_u32. The empty asm hides the second use of
// `acc` from CSE (hipcc would otherwise share the shift: 3 ops); it emits no instruction, so the
// MFMA->VALU wait states stay under the compiler's control (an asm that READS an MFMA result
// directly gets no hazard padding and returned stale data).
__device__ __forceinline__ void vkeys(int acc, int base_row_dir, int base_col_dir, int& kr, int& kc) {
    kr = (acc << (KEY_SHIFT + 1)) + base_row_dir;
    int again = acc;
    asm("" : "+v"(again));
    kc = (again << (KEY_SHIFT + 1)) + base_col_dir;
}

#ifdef EXP_STAMPS
__device__ unsigned long long g_dbg[16];
#endif

template <int KS, int NSUB>
__global__ __launch_bounds__(WG_THREADS, (NSUB <= 2 ? 2 : 1)) void match_tile_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs, int wgs_per_pair, int col_chunks,
    int4* __restrict__ rowres, int2* __restrict__ colpart, int wb_stride, int row_stride) {
    constexpr int TILE_V4 = KS * 64;          // int4 per B tile
    constexpr int ROWS_WAVE = 32 * NSUB;      // query rows a wave keeps in registers
    constexpr int ROWS_WG = WAVES * ROWS_WAVE;
    static_assert(ROWS_WAVE <= (1 << KEY_SHIFT), "the local row must fit the key's code field");
    __shared__ v4i sB[3][TILE_V4];
    __shared__ int2 sR[WAVES][32 * 33];

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
    if (rb * (ROWS_WG / 32) >= A.ntiles || tbeg >= B.ntiles) return;  // workgroup-uniform
    const int wb = rb * WAVES + wave;                  // wave-block (ROWS_WAVE rows) of frame A
    const bool active = NSUB * wb < A.ntiles;          // wave-uniform (ntiles is a multiple of NSUB)
    const int T = min(B.ntiles - tbeg, CHUNK_TILES);   // tiles of this chunk, t below is chunk-local
    const gfrag_t Afrag = (gfrag_t)A.frag, Bfrag = (gfrag_t)B.frag + (size_t)tbeg * TILE_V4;
    const gint_t Anorm = (gint_t)A.norm, Bnormb = (gint_t)B.normb + 32 * tbeg;

    v4i a[NSUB][KS];
    int base_r[NSUB][16], rm1[NSUB][16], rm2[NSUB][16];
    const int wbc = active ? wb : 0;  // inactive waves load a valid block and never use it
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[s][ks] = ~Afrag[((size_t)(NSUB * wbc + s) * KS + ks) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int lrow = 32 * s + (r & 3) + 8 * (r >> 2) + 4 * h;  // C/D layout of the 32x32 MFMA
            base_r[s][r] = (Anorm[ROWS_WAVE * wbc + lrow] << KEY_SHIFT) | lrow;
            rm1[s][r] = INT_MAX;
            rm2[s][r] = INT_MAX;
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
    int nb_cur = T > 0 ? Bnormb[cl] : 0;
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
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][ks], b0[ks], ks ? acc[0] : zero16, 0, 0, 0);
        bq[0] = sB[0][lane];       // steps 0 and 1 of iteration 0 (chain (0,1) re-reads tile 0)
        bq[1] = sB[0][64 + lane];
    }

    int2* cp = colpart + ((size_t)p * wb_stride + wb) * row_stride + 32 * tbeg;
#ifdef EXP_STAMPS
    // diagnostic build: where does a wave's time go? (sums of s_memtime deltas per segment)
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_prev; st_prev = now_; __builtin_amdgcn_sched_barrier(0);} while (0)
#else
#define STAMP(i) do {} while (0)
#endif
    constexpr int EPK = 16 / KS;         // epilogue elements interleaved per MFMA (KS = 8 -> 2)
    constexpr int NSTEP = NSUB * KS;     // (phase, ks) steps per tile
    int slot_cur = 0, slot_nxt = 1, slot_new = 2;  // t % 3, (t+1) % 3, (t+2) % 3
    for (int t = 0; t < T; ++t) {
        const int base_c = (nb_cur << KEY_SHIFT) | t;
        const int t1 = min(t + 1, T - 1), t2 = min(t + 2, T - 1);
        nb_cur = Bnormb[32 * t1 + cl];
        stage_tile(t2, slot_new);  // lands during this iteration; the barrier below publishes it
        int cm1 = INT_MAX, cm2 = INT_MAX, pend[3] = {0, 0, 0};
        STAMP(0);
        if (active) {
            const v4i* curB = sB[slot_cur];
            const v4i* nxtB = sB[slot_nxt];
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
                const v4i* src = (j / KS < NSUB - 1) ? curB : nxtB;
                bq[j % 3] = src[(j % KS) * 64 + lane];
                const int q = (ph + 1) % NSUB;       // accumulator of the chain being issued
                // (the last iteration recomputes tile T-1 into acc[0]; it is never read)
                acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q][ks], bq[i % 3], ks ? acc[q] : zero16, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < EPK; ++e) {
                    const int r = ks * EPK + e;
                    int kr, kc;
                    vkeys(acc[ph][r], base_c, base_r[ph][r], kr, kc);
                    rm2[ph][r] = vmed3(rm1[ph][r], rm2[ph][r], kr);
                    rm1[ph][r] = min(rm1[ph][r], kr);
                    // Column direction: all NCOL keys of the tile belong to this lane's column, so they
                    // are taken three at a time — {min3, med3} of a triple (2 ops) then one sorted-pair
                    // insert (3 ops) = 5 ops per 3 keys instead of 6 (and the first triple needs no insert).
                    constexpr int NCOL = NSUB * 16;
                    const int eg = i * EPK + e;  // compile-time after unrolling
                    pend[eg % 3] = kc;
                    if (eg % 3 == 2) {
                        const int s1 = min(min(pend[0], pend[1]), pend[2]);
                        const int s2 = vmed3(pend[0], pend[1], pend[2]);
                        if (eg == 2) {
                            cm1 = s1;
                            cm2 = s2;
                        } else {
                            cm2 = min(min(max(cm1, s1), cm2), s2);
                            cm1 = min(cm1, s1);
                        }
                    } else if (eg >= NCOL - NCOL % 3) {  // leftover keys of an incomplete last triple
                        cm2 = vmed3(cm1, cm2, kc);
                        cm1 = min(cm1, kc);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i == KS - 1) STAMP(1);
            }
            STAMP(2);
            // steps NSTEP and NSTEP+1 (= steps 0, 1 of the next iteration) sit in bq[NSTEP % 3], ...
            {
                v4i s0 = bq[NSTEP % 3], s1 = bq[(NSTEP + 1) % 3];
                bq[0] = s0;
                bq[1] = s1;
            }
            // lanes l and l+32 hold the same column, rows 4h.. of each 8-row group: merge halves
            // (v_permlane32_swap: lanes 0-31 of the 2nd operand <-> lanes 32-63 of the 1st, pure VALU)
            auto w1 = __builtin_amdgcn_permlane32_swap(cm1, cm1, false, false);
            auto w2 = __builtin_amdgcn_permlane32_swap(cm2, cm2, false, false);
            int n1 = min((int)w1[0], (int)w1[1]);
            int n2 = min(max((int)w1[0], (int)w1[1]), min((int)w2[0], (int)w2[1]));
            if (h == 0) cp[32 * t + cl] = make_int2(n1, n2);
        }
        STAMP(3);
        STAMP(4);
        const int tmp = slot_cur;
        slot_cur = slot_nxt;
        slot_nxt = slot_new;
        slot_new = tmp;
        __syncthreads();
        STAMP(5);
    }
#ifdef EXP_STAMPS
    if (lane == 0 && blockIdx.x % 97 == 0) {
        unsigned long long* dbg = g_dbg;
        for (int i = 0; i < 6; ++i) atomicAdd(&dbg[i], st_acc[i]);
        atomicAdd(&dbg[6], 1ull);
        atomicAdd(&dbg[7], (unsigned long long)T);
    }
#endif
    if (!active) return;

    // Row direction: every lane holds, per row, its top-2 over the columns {32t + cl}. Transpose
    // through this wave's private LDS slab (rows padded to 33 entries: conflict-free) so that lane
    // (row, half) scans 16 of the 32 lane-partials of its row in ASCENDING lane-column order with a
    // strict '<': equal keys (same rank, same tile) then keep the lower column, which makes plain
    // 32-bit key compares exact — (key, lane-column) lexicographic == (rank, column) lexicographic.
    int2* slab = sR[wave];
    int4* rr = rowres + ((size_t)p * col_chunks + cc) * row_stride + ROWS_WAVE * wb;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            slab[row * 33 + cl] = make_int2(rm1[s][r], rm2[s][r]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int b1 = INT_MAX, b2 = INT_MAX, c1 = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int cidx = 16 * h + j;
            const int2 e = slab[cl * 33 + cidx];
            const bool lt = e.x < b1;
            b2 = lt ? b1 : min(b2, e.x);
            c1 = lt ? cidx : c1;
            b1 = min(b1, e.x);
            b2 = min(b2, e.y);  // e.y >= e.x: it can only be a runner-up
        }
        const int o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32), oc = __shfl_xor(c1, 32);
        if (h == 0) {  // the other half holds the higher columns: it wins only with a strictly smaller key
            const bool lt = o1 < b1;
            const int key1 = lt ? o1 : b1, key2 = lt ? min(b1, o2) : min(b2, o1);
            const int col1 = 32 * (tbeg + (key1 & KEY_MASK)) + (lt ? oc : c1);
            rr[32 * s + cl] = make_int4(key1 >> KEY_SHIFT, col1, key2 >> KEY_SHIFT, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
__global__ __launch_bounds__(FIN_THREADS) void match_finalize_kernel(
    const FrameDev* __restrict__ frames, const int2* __restrict__ pairs,
    const int4* __restrict__ rowres, const int2* __restrict__ colpart, int col_chunks, int wb_stride,
    int row_stride, double ratio, int min_dir, int min_mutual, int mode,
    uint2* __restrict__ out_matches, int* __restrict__ counts, int4* __restrict__ stats) {
    extern __shared__ int smem[];
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const int2 pr = pairs[p];
    const FrameDev A = frames[pr.x], B = frames[pr.y];
    int* fwd = smem;               // [row_stride]
    int* bwd = smem + row_stride;  // [row_stride]
    __shared__ int s_cnt[3];
    __shared__ int s_scan[FIN_THREADS];
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();

    int c12 = 0, c21 = 0;
    const int nchunks = (B.ntiles + CHUNK_TILES - 1) / CHUNK_TILES;
    for (int q = tid; q < A.n; q += FIN_THREADS) {
        int4 r = make_int4(INT_MAX >> KEY_SHIFT, -1, INT_MAX >> KEY_SHIFT, 0);
        for (int ch = 0; ch < nchunks; ++ch) {  // ascending columns; strict '<' keeps the lower column on ties
            const int4 e = rowres[((size_t)p * col_chunks + ch) * row_stride + q];
            if (e.x < r.x) {
                r.z = min(r.x, e.z);
                r.x = e.x;
                r.y = e.y;
            } else {
                r.z = min(r.z, e.x);
            }
        }
        int na = A.norm[q];
        bool ok = r.z + na < PAD_VALUE && ratio_pass(r.x + na, r.z + na, ratio);  // pad second => < 2 train rows
        fwd[q] = ok ? r.y : -1;
        c12 += ok;
    }
    const int nwb = A.ntiles / MATCH_NSUB;
    for (int c = tid; c < B.n; c += FIN_THREADS) {
        int v1 = INT_MAX >> KEY_SHIFT, v2 = INT_MAX >> KEY_SHIFT, r1 = -1;
        const int2* cp = colpart + (size_t)p * wb_stride * row_stride + c;
        for (int wb = 0; wb < nwb; ++wb) {  // ascending rows; strict '<' keeps the lower row on ties
            int2 e = cp[(size_t)wb * row_stride];
            int va = e.x >> KEY_SHIFT, vb = e.y >> KEY_SHIFT;
            if (va < v1) {
                v2 = v1;
                v1 = va;
                r1 = ROWS_PER_WAVE * wb + (e.x & KEY_MASK);
            } else if (va < v2) {
                v2 = va;
            }
            if (vb < v2) v2 = vb;  // e.y >= e.x, it can only become the runner-up
        }
        int nb = B.normb[c];  // column ranks are d2 - normb_c
        bool ok = v2 + nb < PAD_VALUE && ratio_pass(v1 + nb, v2 + nb, ratio);
        bwd[c] = ok ? r1 : -1;
        c21 += ok;
    }
    atomicAdd(&s_cnt[0], c12);
    atomicAdd(&s_cnt[1], c21);
    __syncthreads();

    // ordered compaction over q in chunks of FIN_THREADS
    uint2* out = out_matches + (size_t)p * row_stride;
    int base = 0;
    for (int q0 = 0; q0 < A.n; q0 += FIN_THREADS) {
        int q = q0 + tid;
        int t = q < A.n ? fwd[q] : -1;
        bool keep = t >= 0 && (mode == 1 || bwd[t] == q);  // main.cpp:133-140
        s_scan[tid] = keep;
        __syncthreads();
        for (int off = 1; off < FIN_THREADS; off <<= 1) {  // Hillis-Steele inclusive scan
            int v = tid >= off ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        if (keep) out[base + s_scan[tid] - 1] = make_uint2((unsigned)q, (unsigned)t);
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
    int ntiles = (n + 31) / 32;
    ntiles = (ntiles + TILE_ALIGN - 1) / TILE_ALIGN * TILE_ALIGN;
    if (ntiles > MAX_TILES)
        return ctx->fail(EACHAM_ERR_UNSUPPORTED, "frame has %d rows; this build supports <= %d", n, MAX_TILES * 32);
    if ((size_t)frame_id >= ctx->frames.size()) ctx->frames.resize(frame_id + 1);
    FrameHost& f = ctx->frames[frame_id];
    if (f.frag || f.norm) {
        EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (f.frag) (void)hipFree(f.frag);
        if (f.norm) (void)hipFree(f.norm);  // normb shares the allocation
        f = FrameHost();
    }
    const int npad = ntiles * 32;
    if (npad > 0) {
        EACHAM_HIP_TRY(ctx, hipMalloc((void**)&f.frag, (size_t)ntiles * ks * 64 * sizeof(int4)));
        EACHAM_HIP_TRY(ctx, hipMalloc((void**)&f.norm, (size_t)2 * npad * sizeof(int)));
        f.normb = f.norm + npad;
        init_norm_kernel<<<(npad + 255) / 256, 256, 0, ctx->stream>>>(f.norm, f.normb, n, npad);
        long long work = (long long)npad * ks * 2;
        quantize_kernel<<<(unsigned)((work + 255) / 256), 256, 0, ctx->stream>>>(
            src_dev, n, dim, ks, npad, (v4i*)f.frag, f.norm, f.normb, ctx->flag_dev);
        EACHAM_HIP_TRY(ctx, hipGetLastError());
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
    int wb_stride;   // wave-blocks per frame (max over resident frames)
    int row_stride;  // padded rows per frame (max)
    int wgs_per_pair;
    int col_chunks;  // sweeps of <= 4096 train rows per pair
    size_t off_rowres, off_colpart, off_matches, off_counts, total;
};

static MatchPlan make_plan(const eacham_ctx* ctx, int npairs) {
    int max_tiles = TILE_ALIGN;
    for (const auto& f : ctx->frames)
        if (f.n >= 0) max_tiles = std::max(max_tiles, f.ntiles);
    MatchPlan pl;
    pl.row_stride = max_tiles * 32;
    pl.wb_stride = max_tiles / MATCH_NSUB;
    pl.wgs_per_pair = (max_tiles + (ROWS_PER_WG / 32) - 1) / (ROWS_PER_WG / 32);
    pl.col_chunks = (max_tiles + CHUNK_TILES - 1) / CHUNK_TILES;
    size_t per_pair = (size_t)pl.col_chunks * pl.row_stride * sizeof(int4) + (size_t)pl.wb_stride * pl.row_stride * sizeof(int2) +
                      (size_t)pl.row_stride * sizeof(uint2) + sizeof(int);
    // bound the workspace near 1 GiB so the column partials of one batch stay cache-friendly
    size_t budget = (size_t)1 << 30;
    int batch = (int)std::min<size_t>(std::max<size_t>(budget / per_pair, 1), (size_t)npairs);
    pl.batch = std::max(batch, 1);
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    pl.off_rowres = 0;
    pl.off_colpart = align(pl.off_rowres + (size_t)pl.batch * pl.col_chunks * pl.row_stride * sizeof(int4));
    pl.off_matches = align(pl.off_colpart + (size_t)pl.batch * pl.wb_stride * pl.row_stride * sizeof(int2));
    pl.off_counts = align(pl.off_matches + (size_t)pl.batch * pl.row_stride * sizeof(uint2));
    pl.total = align(pl.off_counts + (size_t)pl.batch * sizeof(int));
    return pl;
}

template <int KS>
static void launch_tile(eacham_ctx* ctx, const MatchPlan& pl, const int2* pairs_dev, int nb, char* ws) {
    match_tile_kernel<KS, MATCH_NSUB><<<nb * pl.wgs_per_pair * pl.col_chunks, WG_THREADS, 0, ctx->stream>>>(
        ctx->frame_table_dev, pairs_dev, pl.wgs_per_pair, pl.col_chunks, (int4*)(ws + pl.off_rowres),
        (int2*)(ws + pl.off_colpart), pl.wb_stride, pl.row_stride);
}

// Core driver. mode 0 = mutual (CSR out), mode 1 = directed single pair (fixed-stride out in ws).
static int run_match(eacham_ctx* ctx, const int2* pairs_dev, int npairs, double ratio, int min_dir,
                     int min_mutual, int mode, int* counts_dev, long long* offsets_dev,
                     uint2* edges_dev, long long edge_cap, long long* total_dev, int4* stats_dev) {
    int rc = sync_frame_table(ctx);
    if (rc) return rc;
    if (npairs <= 0) return EACHAM_OK;
    if (ctx->kind_common == 1)
        return run_match_f32(ctx, pairs_dev, npairs, ratio, min_dir, min_mutual, mode, counts_dev, offsets_dev, edges_dev,
                             edge_cap, total_dev, stats_dev);
    MatchPlan pl = make_plan(ctx, npairs);
    rc = ensure_workspace(ctx, pl.total);
    if (rc) return rc;
    char* ws = (char*)ctx->ws;
    ctx->last_matches = ws + pl.off_matches;
    const size_t fin_smem = (size_t)2 * pl.row_stride * sizeof(int);
    if (fin_smem > 48 * 1024)
        EACHAM_HIP_TRY(ctx, hipFuncSetAttribute((const void*)match_finalize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_smem));
    for (int first = 0; first < npairs; first += pl.batch) {
        int nb = std::min(pl.batch, npairs - first);
        const int2* pb = pairs_dev + first;
        {
            ProfileScope ps(ctx, EACHAM_KERNEL_MATCH_TILE);
            switch (ctx->ks_common) {
                case 2: launch_tile<2>(ctx, pl, pb, nb, ws); break;
                case 4: launch_tile<4>(ctx, pl, pb, nb, ws); break;
                default: launch_tile<8>(ctx, pl, pb, nb, ws); break;
            }
        }
        int* cnt = mode == 1 ? counts_dev : counts_dev + first;
        {
            ProfileScope ps(ctx, EACHAM_KERNEL_MATCH_FINALIZE);
            match_finalize_kernel<<<nb, FIN_THREADS, fin_smem, ctx->stream>>>(
                ctx->frame_table_dev, pb, (const int4*)(ws + pl.off_rowres),
                (const int2*)(ws + pl.off_colpart), pl.col_chunks, pl.wb_stride, pl.row_stride, ratio, min_dir,
                min_mutual, mode, (uint2*)(ws + pl.off_matches), cnt,
                stats_dev ? stats_dev + first : nullptr);
            if (mode == 0) {
                scan_counts_kernel<<<1, 1024, 0, ctx->stream>>>(cnt, nb, offsets_dev, total_dev, first,
                                                                first + nb == npairs);
                compact_edges_kernel<<<nb, 256, 0, ctx->stream>>>((const uint2*)(ws + pl.off_matches), cnt,
                                                                  offsets_dev + first, pl.row_stride,
                                                                  edges_dev, edge_cap);
            }
        }
        EACHAM_HIP_TRY(ctx, hipGetLastError());
    }
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

#ifdef EXP_STAMPS
int eacham_debug_read(unsigned long long* out, int n, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * n) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z));
    }
    return 0;
}
#endif

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

int eacham_match_all_pairs(eacham_ctx* ctx, const int32_t* pairs, int npairs, double ratio, int min_dir,
                           int min_mutual, int32_t* counts, int64_t* offsets, uint32_t* out_q,
                           uint32_t* out_t, int64_t cap, int64_t* out_total, int32_t* stats) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (npairs < 0 || (npairs > 0 && (!pairs || !counts || !offsets)) || !out_total || cap < 0)
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
    rc = run_match(ctx, (const int2*)(io + o_pairs), npairs, ratio, min_dir, min_mutual, 0,
                   (int*)(io + o_counts), (long long*)(io + o_offsets), (uint2*)(io + o_edges), cap,
                   (long long*)(io + o_total), (int4*)(io + o_stats));
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

}  // extern "C"
