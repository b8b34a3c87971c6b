// s3d_wino.hip — 3x3 TriplaneConv as a fused Winograd F(2x2, 3x3) convolution on the fp32 matrix cores.
//
// Y = A^T [ (G g G^T) .* (B^T d B) ] A per 4x4 input patch d -> 2x2 outputs: 16 "frequency" GEMMs with 4 multiplies
// per output instead of 9 (2.25x fewer MFMA flops than the direct kernel in s3d_conv.hip).  fp32 error measured
// against an fp64 direct convolution: 4.5e-7 relative (direct fp32: 2.8e-7), far inside the 1e-3 gate.
//
// Everything happens inside one kernel — no transformed tensors ever touch HBM:
//   * block = 16x16 output pixels (8x8 Winograd tiles) x 64 output channels, 4 waves = 2 (tile halves) x 2 (32 couts);
//   * the 18x18x32-channel input halo of a chunk sits in LDS; a lane owns ONE Winograd tile (MFMA row) and reads its
//     4x4 patch as 16 ds_read_b128, applies B^T d B in registers (64 packed adds) and thereby holds the A operands
//     of all 16 frequencies for 4 channels — already in the "lane half 0 takes k0..3, half 1 k4..7" order;
//   * the transformed weights U = G g G^T are pre-packed on the host in MFMA *fragment order*, so a B operand is one
//     fully coalesced 1 KB global load per (frequency, 8 channels) — no LDS, no barrier for weights;
//   * a wave keeps 16 accumulators (one per frequency, 256 AGPRs) for its 32 tiles x 32 channels, so the inverse
//     transform A^T M A needs only values the lane already holds; the epilogue then matches the direct kernel
//     (bias, rank-1 rollout terms, residual, GroupNorm partial sums);
//   * software pipeline inside the wave (one wave per SIMD): while the 64 MFMAs of a k-step run, the next step's
//     patch is read and transformed, the weight fragments stream through an 8-deep register ring and a slice of the
//     next chunk's halo is fetched; one barrier per 32-channel chunk.
#include "s3d_common.h"

namespace s3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* wgf4;
__device__ __forceinline__ wgf4 wg4(const float* p) { return (wgf4)(uintptr_t)p; }

#ifndef W_ABL
#define W_ABL 0                        // tools/wino_ubench.hip only: 1 no epilogue stores, 2 no epilogue, 4 no k-loop, 8 weights from one hot 16 KB, 16 no halo staging
#endif
constexpr int W_T = 16;                 // output tile side
constexpr int W_H = W_T + 2;            // halo side
constexpr int W_KC = 32;                // channels per chunk
constexpr int W_LD = W_KC + 4;          // padded LDS pixel row (floats)
constexpr int W_AELEMS = W_H * W_H * W_LD;
constexpr int W_ITEMS = W_H * W_H * (W_KC / 4);           // float4 items per chunk
constexpr int W_NIT = (W_ITEMS + 255) / 256;              // per thread (11)

__device__ __forceinline__ int w_edge_variant(int idx, int n) { return n == 1 ? 3 : (idx == 0 ? 1 : (idx == n - 1 ? 2 : 0)); }

__device__ __forceinline__ void wino_row_pass(f32x4* r) {      // one patch row, along b
    const f32x4 d0 = r[0], d1 = r[1], d2 = r[2], d3 = r[3];
    r[0] = d0 - d2; r[1] = d1 + d2; r[2] = d2 - d1; r[3] = d1 - d3;
}
__device__ __forceinline__ void wino_col_pass(f32x4* c) {      // one patch column (stride 4), along a
    const f32x4 d0 = c[0], d1 = c[4], d2 = c[8], d3 = c[12];
    c[0] = d0 - d2; c[4] = d1 + d2; c[8] = d2 - d1; c[12] = d1 - d3;
}
// B^T d B for the 4x4 patch p[a*4+b] (each a float4 of channels), in place -> V[u*4+v]
__device__ __forceinline__ void wino_input_transform(f32x4* p) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {       // along b: s0 = d0-d2, s1 = d1+d2, s2 = d2-d1, s3 = d1-d3
        const f32x4 d0 = p[a * 4 + 0], d1 = p[a * 4 + 1], d2 = p[a * 4 + 2], d3 = p[a * 4 + 3];
        p[a * 4 + 0] = d0 - d2; p[a * 4 + 1] = d1 + d2; p[a * 4 + 2] = d2 - d1; p[a * 4 + 3] = d1 - d3;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {       // along a
        const f32x4 d0 = p[0 * 4 + b], d1 = p[1 * 4 + b], d2 = p[2 * 4 + b], d3 = p[3 * 4 + b];
        p[0 * 4 + b] = d0 - d2; p[1 * 4 + b] = d1 + d2; p[2 * 4 + b] = d2 - d1; p[3 * 4 + b] = d1 - d3;
    }
}

__global__ __launch_bounds__(256, 1) void k_conv_wino(ConvArgs args) {
    __shared__ __attribute__((aligned(16))) float smem[2 * W_AELEMS];
    const int bid = blockIdx.x;
    int j = 0;
    while (j + 1 < args.njobs && bid >= args.job[j + 1].block_begin) ++j;
    const ConvJob& J = args.job[j];
    int local = bid - J.block_begin;
    const int ntile = local % J.n_tiles_n; local /= J.n_tiles_n;
    const int b = local / J.tiles_per_img; local %= J.tiles_per_img;
    const int tile_idx = local;
    const int ty0 = (local / J.tiles_x) * W_T, tx0 = (local % J.tiles_x) * W_T;
    const int h = J.h, w = J.w, cin = args.cin, cout = args.cout;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int i = lane & 31, half = lane >> 5;
    const int tr = wm * 4 + (i >> 3), tc = i & 7;               // this lane's Winograd tile inside the 8x8 grid
    const int patch0 = ((2 * tr) * W_H + 2 * tc) * W_LD + half * 4;   // LDS float offset of its patch origin

    // fragment-ordered weights: [n32 tile][k8 step][16 freq][64 lanes][4]
    const int n32_total = (cout + 31) / 32;
    int n32 = ntile * 2 + wn;
    const bool n_live = n32 < n32_total;
    if (!n_live) n32 = n32_total - 1;                            // clamp (outputs masked by co < cout below)
    const int k8_total = cin / 8;
    const float* ub = J.wgt + (size_t(n32) * k8_total) * (16 * 256) + lane * 4;

    const float* inb = J.in + size_t(b) * h * w * cin;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // halo staging of one float4 item of chunk `ch` into LDS buffer `buf`
    auto item_load = [&](int it, int ch) -> f32x4 {
        const int idx = it * 256 + tid;
        const int pix = idx >> 3, q = idx & 7;
        const int hy = pix / W_H, hx = pix - hy * W_H;
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        const bool ok = idx < W_ITEMS && gy >= 0 && gy < h && gx >= 0 && gx < w;
        f32x4 v = wg4(inb + (size_t(ok ? gy : 0) * w + (ok ? gx : 0)) * cin + ch * W_KC + q * 4)[0];
        return ok ? v : zero4;
    };
    auto item_store = [&](int it, int buf, f32x4 v) {
        const int idx = it * 256 + tid;
        if (idx < W_ITEMS) *reinterpret_cast<f32x4*>(smem + buf * W_AELEMS + (idx >> 3) * W_LD + (idx & 7) * 4) = v;
    };

    // (A per-block rotation of the K-chunk order was tried against L2 hot-spotting on the shared weight lines: no gain,
    // and it made a sample's rounding depend on its position in the batch, so the order is the natural one.)
    const int nchunks = cin / W_KC;
    const int rot = 0;
    auto rot_chunk = [&](int c) { const int g = c + rot; return g >= nchunks ? g - nchunks : g; };

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // ---- prologue: chunk 0 -> LDS buffer 0; first patch transformed; first 8 weight fragments in flight
#pragma unroll
    for (int it = 0; it < W_NIT; ++it) item_store(it, 0, item_load(it, rot_chunk(0)));
    __syncthreads();
    f32x4 VA[16], VB[16], ring[8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) VA[a * 4 + bb] = *reinterpret_cast<const f32x4*>(smem + patch0 + (a * W_H + bb) * W_LD);
    wino_input_transform(VA);
    if (W_ABL & 32) {
#pragma unroll
        for (int k = 0; k < 16; ++k) VB[k] = VA[k] + 1.f;
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) ring[f] = wg4(ub + (size_t(rot_chunk(0)) * 4 * 16 + f) * 256)[0];

    // one k-step: 64 MFMAs with Vc; meanwhile read+transform the next patch into Vn, refill the ring, stage halo items
#define WINO_STEP(Vc, Vn, K8)                                                                                         \
    {                                                                                                                 \
        const int step = gch * 4 + (K8);                                     /* global k8 index (rotated chunk) */    \
        const int nstep = (K8) < 3 ? step + 1 : gnext * 4;                                                            \
        const int nbuf = ((K8) == 3 ? (chunk + 1) : chunk) & 1;              /* buffer holding the next step's patch */ \
        const float* nsrc = smem + nbuf * W_AELEMS + patch0 + (((K8) + 1) & 3) * 8;                                    \
        f32x4 pf[4];                                                                                                  \
        constexpr int it0 = (K8) * 4, itn = (K8) == 2 ? 3 : ((K8) == 3 ? 0 : 4);                                       \
        _Pragma("unroll") for (int t = 0; t < itn; ++t) pf[t] = (W_ABL & 16) ? zero4 : item_load(it0 + t, gnext);     \
        _Pragma("unroll") for (int f = 0; f < 16; ++f) {                                                              \
            if (!(W_ABL & 32) && f < 8 && (f & 1) == 0) {                    /* next patch: row f/2 */               \
                _Pragma("unroll") for (int bb = 0; bb < 4; ++bb)                                                      \
                    Vn[(f >> 1) * 4 + bb] = *reinterpret_cast<const f32x4*>(nsrc + ((f >> 1) * W_H + bb) * W_LD);     \
            }                                                                                                         \
            const f32x4 bq = ring[f & 7];                                                                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                             \
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][e], bq[e], acc[f], 0, 0, 0);                      \
            {                                                                /* refill the slot 8 fragments ahead */ \
                const int fs = f + 8 < 16 ? step : nstep, ff = (f + 8) & 15;                                          \
                ring[f & 7] = wg4(ub + (size_t((W_ABL & 8) ? 0 : fs) * 16 + ff) * 256)[0];                                              \
            }                                                                                                         \
            if (!(W_ABL & 32) && f >= 2 && f <= 8 && (f & 1) == 0) wino_row_pass(Vn + ((f >> 1) - 1) * 4);  /* row read two groups ago */ \
            if (!(W_ABL & 32) && f >= 9 && f <= 12) wino_col_pass(Vn + (f - 9));                                      \
            if ((W_ABL & 64) && !(W_ABL & 128)) continue;                                                               \
            __builtin_amdgcn_sched_barrier(0);                               /* keep loads this far ahead of use */   \
        }                                                                                                             \
        _Pragma("unroll") for (int t = 0; t < itn; ++t) item_store(it0 + t, (chunk + 1) & 1, pf[t]);                   \
        if ((K8) == 2) __syncthreads();                                      /* next chunk's halo complete */         \
    }

    for (int chunk = 0; chunk < ((W_ABL & 4) ? 0 : nchunks); ++chunk) {
        const int gch = rot_chunk(chunk);
        const int gnext = chunk + 1 < nchunks ? rot_chunk(chunk + 1) : gch;
        WINO_STEP(VA, VB, 0)
        WINO_STEP(VB, VA, 1)
        WINO_STEP(VA, VB, 2)
        WINO_STEP(VB, VA, 3)
    }
#undef WINO_STEP

    // ---- epilogue: inverse transform A^T M A per tile, then the same fused tail as the direct kernel
    const float* __restrict__ p_bias = J.bias;
    const float* __restrict__ p_bbias = J.bbias;
    const float* __restrict__ p_rcol = J.rcol;
    const float* __restrict__ p_rrow = J.rrow;
    const float* __restrict__ p_res = J.res;
    float* __restrict__ p_out = J.out;
    double* p_gn = J.gn_part;
    const int co = (ntile * 2 + wn) * 32 + i;
    const bool co_ok = n_live && co < cout;
    const int coc = co_ok ? co : 0;
    float base = p_bias ? p_bias[coc] : 0.f;
    if (p_bbias) base += p_bbias[size_t(b) * J.bbias_stride + coc];
    float gs = 0.f, gss = 0.f;
    if (W_ABL & 2) { if (co_ok && acc[0][0] == 12345.f) p_out[0] = acc[3][1] + VA[0][0] + ring[0][0]; return; }
    // 4 MFMA rows (= 4 tiles = 16 output pixels) per round: inverse transform, then every rank-1 / residual load of
    // the round is issued before any is consumed (predicated, clamped addresses, no per-element branches)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        float val[16]; bool ok[16]; size_t oidx[16];
        int yy_[16], xx_[16];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = rg * 4 + rr;
            const int ti = (r & 3) + 8 * (r >> 2) + 4 * half;                  // MFMA row -> tile of this wave
            const int ytile = ty0 + 2 * (wm * 4 + (ti >> 3)), xtile = tx0 + 2 * (ti & 7);
            float P[2][4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float m0 = acc[0 * 4 + v][r], m1 = acc[1 * 4 + v][r], m2 = acc[2 * 4 + v][r], m3 = acc[3 * 4 + v][r];
                P[0][v] = m0 + m1 + m2; P[1][v] = m1 - m2 - m3;
            }
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                const float Y0 = P[yy][0] + P[yy][1] + P[yy][2], Y1 = P[yy][1] - P[yy][2] - P[yy][3];
#pragma unroll
                for (int xx = 0; xx < 2; ++xx) {
                    const int k = rr * 4 + yy * 2 + xx;
                    const int y = ytile + yy, x = xtile + xx;
                    yy_[k] = y; xx_[k] = x;
                    ok[k] = y < h && x < w && co_ok;
                    oidx[k] = ok[k] ? ((size_t(b) * h + y) * w + x) * cout + co : 0;
                    val[k] = (xx == 0 ? Y0 : Y1) + base;
                }
            }
        }
        // issue every load of the round first (three uniform branches, no use in between), then consume
        float tc[16], tr_[16], ts[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { tc[k] = 0.f; tr_[k] = 0.f; ts[k] = 0.f; }
        if (p_rcol) {
#pragma unroll
            for (int k = 0; k < 16; ++k) tc[k] = p_rcol[ok[k] ? ((size_t(b) * w + xx_[k]) * 4 + w_edge_variant(yy_[k], h)) * cout + co : 0];
        }
        if (p_rrow) {
#pragma unroll
            for (int k = 0; k < 16; ++k) tr_[k] = p_rrow[ok[k] ? ((size_t(b) * h + yy_[k]) * 4 + w_edge_variant(xx_[k], w)) * cout + co : 0];
        }
        if (p_res) {
#pragma unroll
            for (int k = 0; k < 16; ++k) ts[k] = p_res[oidx[k]];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) val[k] += (tc[k] + tr_[k]) + ts[k];    // !ok lanes hold clamped garbage, never stored
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (ok[k] && !((W_ABL & 1) && val[k] != 12345.f)) { p_out[oidx[k]] = val[k]; gs += val[k]; gss = fmaf(val[k], val[k], gss); }
    }
    if (p_gn) {
        gs += __shfl_xor(gs, 32, 64); gss += __shfl_xor(gss, 32, 64);
        for (int off = 1; off < args.gn_sg; off <<= 1) { gs += __shfl_xor(gs, off, 64); gss += __shfl_xor(gss, off, 64); }
        if (lane < 32 && co_ok && (co % args.gn_sg) == 0) {
            const int part = tile_idx * 2 + wm;
            double* dst = p_gn + ((size_t(b) * 3 * args.gn_maxparts + part) * args.gn_nsub + co / args.gn_sg) * 2;
            dst[0] = double(gs); dst[1] = double(gss);
        }
    }
}

// ------------------------------------------------------------------ two waves per SIMD: the frequencies split in halves
// Same data flow as k_conv_wino with an 8x16-pixel tile (one 32-row MFMA tile); a (tile, 32-channel) unit is shared by TWO waves that each own 8 of
// the 16 frequencies (rows u = 2*fh, 2*fh+1 of the 4x4 frequency grid): 128 accumulator registers per wave instead of
// 256, so two waves fit on every SIMD.  That (a) lets one wave's barrier / prologue / epilogue time be covered by its
// neighbour's MFMAs and (b) halves the scheduling quantum, which removes most of the last-round imbalance at batch 1
// (1536 units on 1024 SIMDs became 3072 half-units that the dispatcher packs 3 per SIMD).  Costs: each wave redoes the
// cheap column pass of the input transform for its rows only (12 of the 16 patch reads), and the inverse transform
// needs one 4-float exchange per output tile between the two waves through LDS.
constexpr int W2_TH = 8, W2_TW = 16;                      // output tile: 4 x 8 Winograd tiles = one 32-row MFMA tile
constexpr int W2_HH = W2_TH + 2, W2_HW = W2_TW + 2;
constexpr int W2_ITEMS = W2_HH * W2_HW * (W_KC / 4);
constexpr int W2_ITEMS_PT = (W2_ITEMS + 255) / 256;       // halo float4 items per thread and chunk (6)
constexpr int W2_ABUF = W2_ITEMS_PT * 32 * W_LD;          // LDS buffer stride: 6 rounds x 32 pixels, so no store of a round needs a predicate
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef W_TIMING
__device__ unsigned long long* g_wtime;       // tools/wino_ubench.hip: per block {entry, k-loop begin, k-loop end, exit} (wall clock) + hw id
#define W_STAMP(k) if (threadIdx.x == 0) g_wtime[size_t(blockIdx.x) * 8 + (k)] = wall_clock64();
#else
#define W_STAMP(k)
#endif
__global__ __launch_bounds__(256, 2) void k_conv_wino2(ConvArgs args) {
    // k-loop: two halo buffers + the halo offsets; epilogue: two [128 pixels][64 channels] share images (64 KB, two blocks per CU)
    __shared__ __attribute__((aligned(16))) float smem[2 * W2_TH * W2_TW * 64];
    static_assert(2 * W2_ABUF + W2_ITEMS_PT * 256 <= 2 * W2_TH * W2_TW * 64, "LDS plan");
    W_STAMP(0)
    // The prologue and epilogue are short and latency-bound; a partner wave on the same SIMD that is in its k-loop is older
    // and would win every VALU issue slot (priority, then age): run them at raised priority, the k-loop at 0.
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    // workgroup i runs on XCD i % 8 (each XCD has its own L2): give every XCD a contiguous range of logical blocks, so
    // that the two 64-channel column blocks of a pixel tile and the neighbouring tiles (shared halo) hit the same L2
    int bid = blockIdx.x;
    if (args.xcd_swizzle & 1) {
        const int chunk = int(gridDim.x) >> 3;
        if (bid < (chunk << 3)) bid = (bid & 7) * chunk + (bid >> 3);
    }
    int j = 0;
    while (j + 1 < args.njobs && bid >= args.job[j + 1].block_begin) ++j;
    const ConvJob& J = args.job[j];
    int local = bid - J.block_begin;
    const int ntile = local % J.n_tiles_n; local /= J.n_tiles_n;
    const int b = local / J.tiles_per_img; local %= J.tiles_per_img;
    const int tile_idx = local;
    const int ty0 = (local / J.tiles_x) * W2_TH, tx0 = (local % J.tiles_x) * W2_TW;
    const int h = J.h, w = J.w, cin = args.cin, cout = args.cout;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fh = wid & 1, wn = wid >> 1;
    const int i = lane & 31, half = lane >> 5;
    const int tr = i >> 3, tc = i & 7;
    // The wave's two frequency rows need three rows of the 4x4 patch: u0 = d0 - d2, u1 = d1 + d2 (fh = 0) or
    // u2 = d2 - d1, u3 = d1 - d3 (fh = 1).  Both are t0 = x - y, t1 = y + sgn * z with the rows picked per wave
    // (x, y, z) = (d0, d2, d1) / (d2, d1, d3): the choice lives in three LDS addresses, the loop has no branch.
    const int prow = (2 * tr * W2_HW + 2 * tc) * W_LD + half * 4;
    const int px0 = prow + (fh ? 2 : 0) * W2_HW * W_LD, py0 = prow + (fh ? 1 : 2) * W2_HW * W_LD, pz0 = prow + (fh ? 3 : 1) * W2_HW * W_LD;
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(fh ? 0xBF800000 : 0x3F800000));

    const int n32_total = (cout + 31) / 32;
    int n32 = ntile * 2 + wn;
    const bool n_live = n32 < n32_total;
    if (!n_live) n32 = n32_total - 1;
    const int k8_total = cin / 8;
    // weights: this wave's fragments of step k8, frequency f sit (k8 * 16 + f) KB into its slab; lane * 16 bytes is the only VGPR
    const float* ub = J.wgt + ((size_t(n32) * k8_total) * 16 + fh * 8) * 256;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, k8_total * 16 * 1024, 0x00020000);
    const int wlane = lane * 16;
    auto wfrag = [&](int step, int f) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (step * 16 + f) * 1024, 0));
    };
    const float* inb = J.in + size_t(b) * h * w * cin;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Halo staging.  Round `it` gives a thread the float4 (pixel it*32 + tid/8, channel quad tid%8) of the 10x18-pixel
    // halo.  The loads are raw buffer loads: one byte offset per round in a VGPR, the chunk offset in an SGPR, and an
    // offset beyond the descriptor's range for everything outside the image (or past the halo), which the hardware
    // answers with zeros - the conv's padding costs no select and no predicate, and nothing depends on the loaded
    // data until the ds_write at the end of the k-step.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inb), 0, h * w * cin * 4, 0x00020000);
    // The six byte offsets of a thread live in LDS behind the two halo buffers (a k-step fetches its pair with one
    // ds_read_b64): six more live VGPRs made the allocator spill, and a scratch reload waits vmcnt(0) inside the k-loop.
    unsigned* gtab = reinterpret_cast<unsigned*>(smem + 2 * W2_ABUF);
    auto item_offset = [&](int it) -> unsigned {
        const int pix = it * 32 + (tid >> 3);
        const int hy = pix / W2_HW, hx = pix - hy * W2_HW;
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        const bool ok = pix < W2_HH * W2_HW && gy >= 0 && gy < h && gx >= 0 && gx < w;
        return ok ? unsigned((gy * w + gx) * cin + (tid & 7) * 4) * 4u : 0x80000000u;
    };
    const int lds_w = (tid >> 3) * W_LD + (tid & 7) * 4;
    auto item_load = [&](unsigned off, int ch) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, ch * (W_KC * 4), 0));
    };
    auto item_store = [&](int it, int buf, f32x4 v) {
        *reinterpret_cast<f32x4*>(smem + buf * W2_ABUF + lds_w + it * (32 * W_LD)) = v;
    };
    auto col_pair = [&](const f32x4& x, const f32x4& y, const f32x4& z, f32x4& t0, f32x4& t1) {
        t0 = x - y;
#pragma unroll
        for (int e = 0; e < 4; ++e) t1[e] = fmaf(sgn, z[e], y[e]);
        asm volatile("" : "+v"(t0), "+v"(t1));            // computed here: keeps the three raw rows from staying live until the row pass
    };

    f32x16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    const int nchunks = cin / W_KC;
    f32x4 VA[8], VB[8], ring[8];
    // first weight fragments before anything else (their latency hides behind the halo staging); in order, so that the
    // loop's first wait is for fragment 0 only
#pragma unroll
    for (int f = 0; f < 8; ++f) { ring[f] = wfrag(0, f); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int it = 0; it < W2_ITEMS_PT; ++it) {
        const unsigned off = item_offset(it);
        gtab[it * 256 + tid] = off;
        item_store(it, 0, item_load(off, 0));
    }
    W_STAMP(7)
    __syncthreads();
    W_STAMP(4)
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(smem + px0 + bb * W_LD);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(smem + py0 + bb * W_LD);
        const f32x4 r2 = *reinterpret_cast<const f32x4*>(smem + pz0 + bb * W_LD);
        col_pair(r0, r1, r2, VA[bb], VA[4 + bb]);
    }
    wino_row_pass(VA); wino_row_pass(VA + 4);

    // One k-step = 32 slots of {one MFMA + a small piece of the other work}, pinned with sched_barrier: the wave issues
    // in order, so work placed between two MFMAs runs in the shadow of the first (64 cycles) and the pipe never waits
    // for it.  Per step: 12 patch reads + column pass for the NEXT step's operands (slots of f = 0..3), the row passes
    // (f = 4, 5), 8 weight fragments (one per f), 2 halo loads whose data is only touched by the ds_write at the end.
#define W2_LDS4(off) (*static_cast<const f32x4*>(__builtin_assume_aligned(reinterpret_cast<const char*>(smem) + (off), 16)))
#define W2_PIN(v) asm volatile("" : "+v"(v))
#define WINO2_STEP(Vc, Vn, K8)                                                                                        \
    {                                                                                                                 \
        const int step = chunk * 4 + (K8);                                                                            \
        const int nstep = (K8) < 3 ? step + 1 : gnext * 4;                                                            \
        if ((K8) == 3) { ax += tog; ay += tog; az += tog; tog = -tog; }      /* the next patch is in the other buffer */ \
        constexpr int koff = (((K8) + 1) & 3) * 32;                                                                   \
        constexpr int it0 = (K8) * 2, itn = (K8) == 3 ? 0 : 2;                                                        \
        f32x4 pf[2], cx, cy, cz;                                                                                      \
        unsigned g0 = 0, g1 = 0;                                                                                      \
        _Pragma("unroll") for (int f = 0; f < 8; ++f) {                                                               \
            const f32x4 bq = ring[f];                                                                                 \
            /* slot 0 */                                                                                              \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][0], bq[0], acc[f], 0, 0, 0);                          \
            if (f < 4) { cx = W2_LDS4(ax + koff + f * (W_LD * 4)); cy = W2_LDS4(ay + koff + f * (W_LD * 4)); cz = W2_LDS4(az + koff + f * (W_LD * 4)); } \
            if (f == 0 && itn) { g0 = gtab[it0 * 256 + tid]; g1 = gtab[(it0 + 1) * 256 + tid]; }                      \
            if (f == 4) { const f32x4 d0 = Vn[0], d2 = Vn[2]; Vn[0] = d0 - d2; W2_PIN(Vn[0]); rp = Vn[1] + d2; W2_PIN(rp); rq = d2 - Vn[1]; W2_PIN(rq); } \
            if (f == 5) { const f32x4 d0 = Vn[4], d2 = Vn[6]; Vn[4] = d0 - d2; W2_PIN(Vn[4]); rp = Vn[5] + d2; W2_PIN(rp); rq = d2 - Vn[5]; W2_PIN(rq); } \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            /* slot 1 */                                                                                              \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][1], bq[1], acc[f], 0, 0, 0);                          \
            ring[f] = wfrag((W_ABL & 8) ? 0 : nstep, f);                     /* same frequency, next step */          \
            if (f == 1 && itn) { pf[0] = (W_ABL & 16) ? zero4 : item_load(g0, gnext); pf[1] = (W_ABL & 16) ? zero4 : item_load(g1, gnext); } \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            /* slot 2 */                                                                                              \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][2], bq[2], acc[f], 0, 0, 0);                          \
            if (f < 4) { Vn[f] = cx - cy; W2_PIN(Vn[f]); }                                                            \
            if (f == 4) { Vn[3] = Vn[1] - Vn[3]; W2_PIN(Vn[3]); Vn[1] = rp; Vn[2] = rq; }                             \
            if (f == 5) { Vn[7] = Vn[5] - Vn[7]; W2_PIN(Vn[7]); Vn[5] = rp; Vn[6] = rq; }                             \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            /* slot 3 */                                                                                              \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][3], bq[3], acc[f], 0, 0, 0);                          \
            if (f < 4) { _Pragma("unroll") for (int e = 0; e < 4; ++e) Vn[4 + f][e] = fmaf(sgn, cz[e], cy[e]); W2_PIN(Vn[4 + f]); } \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
        _Pragma("unroll") for (int t = 0; t < itn; ++t) item_store(it0 + t, (chunk + 1) & 1, pf[t]);                   \
        if ((K8) == 2) __syncthreads();                                                                               \
    }

    int ax = px0 * 4, ay = py0 * 4, az = pz0 * 4, tog = W2_ABUF * 4;          // byte offsets of the patch rows in the buffer being read
    f32x4 rp, rq;
    W_STAMP(1)
    __builtin_amdgcn_s_setprio(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int gnext = chunk + 1 < nchunks ? chunk + 1 : chunk;
        WINO2_STEP(VA, VB, 0)
        WINO2_STEP(VB, VA, 1)
        WINO2_STEP(VA, VB, 2)
        WINO2_STEP(VB, VA, 3)
    }
#undef WINO2_STEP
#undef W2_LDS4
#undef W2_PIN
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    W_STAMP(2)

    // ---- epilogue.  M[u][v] = acc[ui*4+v] with u = 2*fh + ui.  Y = A^T M A with A^T = [[1,1,1,0],[0,1,-1,-1]]:
    //   row pass  P[0][v] = M0 + M1 + M2,  P[1][v] = M1 - M2 - M3;   column pass  Y[y][0] = P0 + P1 + P2,  Y[y][1] = P1 - P2 - P3.
    // Both passes are linear, so each wave pushes its two frequency rows through them alone: wave fh = 0 holds
    // (M0 + M1, M1), wave fh = 1 holds (M2, -M2 - M3) as its shares of (P[0], P[1]); an output is the sum of the two
    // waves' shares.  The shares go to LDS as two [128 pixels][64 channels] images (ONE barrier), and the tile is
    // finished by threads that own 4 consecutive channels of a pixel column: the rank-1 tables, the residual and the
    // output move as 16-byte accesses (1 KB per wave instruction; the MFMA layout would give 4-byte ones and four times
    // as many memory instructions, which is what the epilogue spent its time on), requested before the barrier.
    const float* __restrict__ p_bias = J.bias;
    const float* __restrict__ p_bbias = J.bbias;
    const float* __restrict__ p_rcol = J.rcol;
    const float* __restrict__ p_rrow = J.rrow;
    const float* __restrict__ p_res = J.res;
    float* __restrict__ p_out = J.out;
    double* p_gn = J.gn_part;
    float* img0 = smem;                                  // shares of the even output rows' owner ... indexed [row parity][pixel][64]
    __syncthreads();                                     // all patch reads of the last step are done
    W_STAMP(5)
    {
        // lane (i, half) holds channel wn*32 + i of Winograd tiles ti(r) = (r&3) + 8*(r>>2) + 4*half; pixel (2*(ti>>3) + y, 2*(ti&7) + x)
        float* mine = img0 + wn * 32 + i;                // [share kind][pixel][64]: kind 0 = shares of finished rows, 1 = shares sent
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float kp[4], sd[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float ma = acc[v][r], mb = acc[4 + v][r];              // fh = 0: M0, M1 ; fh = 1: M2, M3
                const float sm = ma + mb;
                kp[v] = fh == 0 ? sm : -sm;                                   // share of P[fh]
                sd[v] = fh == 0 ? mb : ma;                                    // share of P[1 - fh]
            }
            const int ti = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int pk = ((2 * (ti >> 3) + fh) * W2_TW + 2 * (ti & 7)) * 64;          // own row (parity fh)
            const int ps = ((2 * (ti >> 3) + 1 - fh) * W2_TW + 2 * (ti & 7)) * 64;      // partner's row
            mine[pk] = kp[0] + kp[1] + kp[2];
            mine[pk + 64] = kp[1] - kp[2] - kp[3];
            mine[W2_TH * W2_TW * 64 + ps] = sd[0] + sd[1] + sd[2];
            mine[W2_TH * W2_TW * 64 + ps + 64] = sd[1] - sd[2] - sd[3];
        }
    }
    // finishing thread: channels co4..co4+3 of pixel column xl, rows 0..7
    const int quad = tid & 15, xl = tid >> 4;
    const int co4 = ntile * 64 + quad * 4;
    const bool c_ok = co4 < cout;
    const int coc = c_ok ? co4 : 0;
    const int x = tx0 + xl;
    const bool x_ok = x < w && c_ok;
    const int xc = x < w ? x : 0;
    f32x4 base4 = p_bias ? *reinterpret_cast<const f32x4*>(p_bias + coc) : zero4;
    if (p_bbias) base4 += *reinterpret_cast<const f32x4*>(p_bbias + size_t(b) * J.bbias_stride + coc);
    f32x4 tcol[W2_TH], trow[W2_TH], tres[W2_TH];
#pragma unroll
    for (int yl = 0; yl < W2_TH; ++yl) { tcol[yl] = zero4; trow[yl] = zero4; tres[yl] = zero4; }
    if (p_rcol) {
#pragma unroll
        for (int yl = 0; yl < W2_TH; ++yl) {
            const int y = ty0 + yl;
            tcol[yl] = *reinterpret_cast<const f32x4*>(p_rcol + ((size_t(b) * w + xc) * 4 + w_edge_variant(y < h ? y : 0, h)) * cout + coc);
        }
    }
    if (p_rrow) {
        const int vx = w_edge_variant(xc, w);
#pragma unroll
        for (int yl = 0; yl < W2_TH; ++yl) {
            const int y = ty0 + yl;
            trow[yl] = *reinterpret_cast<const f32x4*>(p_rrow + ((size_t(b) * h + (y < h ? y : 0)) * 4 + vx) * cout + coc);
        }
    }
    if (p_res) {
#pragma unroll
        for (int yl = 0; yl < W2_TH; ++yl) {
            const int y = ty0 + yl;
            tres[yl] = *reinterpret_cast<const f32x4*>(p_res + ((size_t(b) * h + (y < h ? y : 0)) * w + xc) * cout + coc);
        }
    }
    __syncthreads();                                     // both share images are complete
    W_STAMP(6)
    f32x4 gs4 = zero4, gss4 = zero4;
#pragma unroll
    for (int yl = 0; yl < W2_TH; ++yl) {
        const int y = ty0 + yl;
        const float* sp = img0 + (yl * W2_TW + xl) * 64 + quad * 4;
        const f32x4 ka = *reinterpret_cast<const f32x4*>(sp), kb = *reinterpret_cast<const f32x4*>(sp + W2_TH * W2_TW * 64);
        const f32x4 v = ((ka + kb) + base4) + ((tcol[yl] + trow[yl]) + tres[yl]);
        if (x_ok && y < h) {
            *reinterpret_cast<f32x4*>(p_out + ((size_t(b) * h + y) * w + x) * cout + co4) = v;
            gs4 += v; gss4 += v * v;
        }
    }
    if (p_gn) {
        // per wave: 4 pixel columns (lanes l, l+16, l+32, l+48) x 8 rows of 16 channel quads; one part per wave
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gs4[e] += __shfl_xor(gs4[e], 16, 64); gss4[e] += __shfl_xor(gss4[e], 16, 64);
            gs4[e] += __shfl_xor(gs4[e], 32, 64); gss4[e] += __shfl_xor(gss4[e], 32, 64);
        }
        const int sg = args.gn_sg;
        const int part = tile_idx * 4 + wid;
        double* dst = p_gn + (size_t(b) * 3 * args.gn_maxparts + part) * args.gn_nsub * 2;
        if (sg >= 4) {
            float s = (gs4[0] + gs4[1]) + (gs4[2] + gs4[3]), ss = (gss4[0] + gss4[1]) + (gss4[2] + gss4[3]);
            for (int off = 1; off < (sg >> 2); off <<= 1) { s += __shfl_xor(s, off, 64); ss += __shfl_xor(ss, off, 64); }
            if (lane < 16 && c_ok && (co4 % sg) == 0) { dst[(co4 / sg) * 2] = double(s); dst[(co4 / sg) * 2 + 1] = double(ss); }
        } else if (lane < 16 && c_ok) {
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                if (sg == 2) { dst[((co4 + e) / 2) * 2] = double(gs4[e] + gs4[e + 1]); dst[((co4 + e) / 2) * 2 + 1] = double(gss4[e] + gss4[e + 1]); }
                else { dst[(co4 + e) * 2] = double(gs4[e]); dst[(co4 + e) * 2 + 1] = double(gss4[e]);
                       dst[(co4 + e + 1) * 2] = double(gs4[e + 1]); dst[(co4 + e + 1) * 2 + 1] = double(gss4[e + 1]); }
            }
        }
    }
    W_STAMP(3)
#ifdef W_TIMING
#endif
}

// ------------------------------------------------------------------ host side
static int wino_variant() {          // 2: two waves per SIMD (default), 1: one wave per SIMD, 0: direct kernel
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("S3D_WINO");
        v = e ? atoi(e) : 2;
        if (v < 0 || v > 2) v = 2;
    }
    return v;
}
bool conv_use_wino() { return wino_variant() != 0 && !conv_use_naive(); }

void wino_gn_parts(const Geo& g, int nparts[3]) {
    for (int p = 0; p < 3; ++p) {
        if (wino_variant() == 2) nparts[p] = ((g.w[p] + W2_TW - 1) / W2_TW) * ((g.h[p] + W2_TH - 1) / W2_TH) * 4;   // one per wave
        else nparts[p] = ((g.w[p] + W_T - 1) / W_T) * ((g.h[p] + W_T - 1) / W_T) * 2;
    }
}

// U = G g G^T in double, stored in MFMA fragment order [n32][k8][16][64 lanes][4]; W is OIHW [cout][ctot][3][3],
// only input channels [0, cin) are used (the plane's own channels).
size_t pack_wino_weights(std::vector<float>& stage, const float* W, int cout, int ctot, int cin) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int n32 = (cout + 31) / 32, k8t = cin / 8;
    const size_t total = size_t(n32) * k8t * 16 * 256;
    const size_t off = push(stage, nullptr, total);
    float* d = stage.data() + off;
    std::fill(d, d + total, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int c = 0; c < cin; ++c) {
            const float* g = W + (size_t(co) * ctot + c) * 9;
            double t[4][3];
            for (int u = 0; u < 4; ++u)
                for (int k = 0; k < 3; ++k) t[u][k] = G[u][0] * g[0 * 3 + k] + G[u][1] * g[1 * 3 + k] + G[u][2] * g[2 * 3 + k];
            const int nt = co >> 5, jn = co & 31, k8 = c >> 3, hf = (c >> 2) & 1, e = c & 3;
            for (int u = 0; u < 4; ++u)
                for (int v = 0; v < 4; ++v) {
                    const double uv = t[u][0] * G[v][0] + t[u][1] * G[v][1] + t[u][2] * G[v][2];
                    d[(((size_t(nt) * k8t + k8) * 16 + (u * 4 + v)) * 64 + (hf * 32 + jn)) * 4 + e] = float(uv);
                }
        }
    return off;
}

int launch_conv_wino(ConvArgs& a, hipStream_t st) {
    S3D_CHECK(a.njobs >= 1 && a.njobs <= kMaxConvJobs && a.cin % W_KC == 0 && a.cout % 4 == 0, S3D_ERR_INVALID, "wino conv: bad arguments");
    const bool two = wino_variant() == 2;
    const int th = two ? W2_TH : W_T, tw = two ? W2_TW : W_T;
    int blocks = 0;
    for (int j = 0; j < a.njobs; ++j) {
        ConvJob& J = a.job[j];
        S3D_CHECK(size_t(J.h) * J.w * a.cin * 4 < (size_t(1) << 31), S3D_ERR_INVALID, "wino conv: a plane of one sample must stay below 2 GiB");
        J.tiles_x = (J.w + tw - 1) / tw;
        J.tiles_per_img = J.tiles_x * ((J.h + th - 1) / th);
        J.n_tiles_n = (a.cout + 63) / 64;
        J.block_begin = blocks;
        blocks += J.tiles_per_img * J.n_tiles_n * a.B;
    }
    if (!blocks) return 0;
    static const int xcd = (getenv("S3D_XCD") ? atoi(getenv("S3D_XCD")) : 1) | (getenv("S3D_PRIO") ? atoi(getenv("S3D_PRIO")) * 2 : 2);
    a.xcd_swizzle = xcd;
    if (two) hipLaunchKernelGGL(k_conv_wino2, dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_conv_wino, dim3(blocks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

}  // namespace s3d
