// s3d_wino24.hip — 3x3 TriplaneConv as a fused MIXED Winograd convolution F(2x4, 3x3) on the fp32 matrix cores:
// F(2,3) along the rows, F(4,3) along the columns.  A 4x6 input patch gives a 2x4 output tile through 24 "frequency"
// GEMMs: 3 multiplies per output instead of 4 (F(2x2,3x3), s3d_wino.hip) or 9 (direct) — 25 % fewer MFMA flops than
// k_conv_wino4 in a k-loop that was already matrix-pipe bound.
//
//   Y = A2^T [ (G2 g G4^T) .* (B2^T d B4) ] A4        B2/G2/A2: F(2,3) (points 0, +-1, inf),  B4/G4/A4: F(4,3) (0, +-1, +-2, inf)
//
// fp32 error against an fp64 direct convolution, measured on this network's layer shapes: 0.9-1.2e-6 relative
// (F(2x2): 2-3e-7, direct fp32: 2.5-4.7e-7, full F(4x4): 3.6-4.9e-6) — three decades inside the 1e-3 gate.
//
// Two kernels, same arithmetic in the same order (bit-identical results): k_conv_wino24s (8x16 pixels x 32 output channels per
// block, three blocks per CU — launches of one or two rounds of blocks) and k_conv_wino24w (x 64 output channels, two blocks per
// CU: one halo fetch + input transform feeds twice the MFMAs — multi-round launches).  Round 2's 16x16-pixel form on
// v_mfma_f32_32x32x2_f32, round 3's persistent multi-tile form and the in-launch rank-1 producers were measured slower and are
// gone (profiles/r02_wino_ubench.txt, r03_wino_persistent.txt, r03_rank1_inline.txt).
#include "s3d_common.h"
#include "s3d_rank1.h"

namespace s3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTabAux = 16;                         // buffer-instruction aux bit 4 = sc1 on the rank-1 table loads
__device__ __forceinline__ int x_edge_variant(int idx, int n) { return n == 1 ? 3 : (idx == 0 ? 1 : (idx == n - 1 ? 2 : 0)); }

// ------------------------------------------------------------------ the small-block form: three waves per SIMD
// Same arithmetic (bit-identical results: the same products are added in the same order), blocked like k_conv_wino4: a block
// owns 8x16 output pixels = 4x4 tiles of 2x4 and 32 output channels, on v_mfma_f32_16x16x4_f32: a lane owns one tile AND one
// channel quad of a 16-channel k-step (lane = 16 * quad + tile); the accumulators are 6 frequencies x 2 blocks of 16 output
// channels x 4 registers = 48 -> <= 168 VGPRs, three blocks per CU, and a launch has as many blocks as with the F(2x2)
// kernel (768 on the half-resolution layers of a batch-1 step: one full round of the chip, where the 16x16-pixel form above
// leaves a quarter of it idle).
// LDS: two halo buffers of 10 x 18 pixels x 36 floats (a 32-channel chunk = two k-steps; one barrier per chunk — with
// 16-channel chunks and a barrier per k-step the loop ran at 80 % of the matrix pipe); the channel quad q of pixel row r
// is stored at quad q ^ 2*((r >> 1) & 3): the patch reads of every ds_read_b128 lane group hit 16 distinct 16-byte slots.
// Weights: [n32][k16][24 freq][2 x 16 couts][64 lanes][4], a six-deep register ring of 1 KB fragments, each requested half a
// step before its use.  The A operands of the next step overwrite the current ones frequency by frequency as soon as their
// MFMAs have been issued.
#ifdef W24_TIMING
__device__ unsigned long long* g_w24time;     // tools/wino24_ubench.hip: per-block wall-clock stamps (entry, halo in LDS, first MFMA, last MFMA, images written, exit)
                                              // + the shader-clock counter at the two ends of the k-loop (slots 6, 7; tools/clock_probe.hip)
__device__ unsigned* g_w24id;                 // ... and where the block ran: XCC_ID << 16 | (HW_ID: CU 11:8, SH 12, SE 15:13)
#define W24_WHERE() ((__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 16) | (__builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4) & 0xFFFF))
#ifdef W24_WHERE_ID                           // (its own build: the extra registers at a block's start make k_conv_wino24s spill, 1.5x slower)
#define W24_NOTE_WHERE(k) if ((k) == 0 && g_w24id) g_w24id[blockIdx.x] = W24_WHERE();
#else
#define W24_NOTE_WHERE(k)
#endif
#define W24_STAMP(k) if (threadIdx.x == 0) { W24_NOTE_WHERE(k) g_w24time[size_t(blockIdx.x) * 8 + (k)] = wall_clock64(); \
        if ((k) == 2) g_w24time[size_t(blockIdx.x) * 8 + 6] = clock64(); if ((k) == 3) g_w24time[size_t(blockIdx.x) * 8 + 7] = clock64(); }
#else
#define W24_STAMP(k)
#endif
constexpr int C_KC = 32, C_LD = C_KC + 4;
constexpr int C_TH = 8, C_TW = 16;
constexpr int C_HH = C_TH + 2, C_HW = C_TW + 2;
constexpr int C_ITEMS = C_HH * C_HW * (C_KC / 4);            // 1440 float4 items per chunk
constexpr int C_ITEMS_PT = (C_ITEMS + 255) / 256;            // 6 (the sixth round covers 160 items)
constexpr int C_ABUF = C_HH * C_HW * C_LD;                   // floats per halo buffer (6480)
constexpr int C_IMG = (C_TH / 2) * C_TW * 32;                // one share image [4 tile rows][16 columns][32 channels]

#define W24S_GNB 0
#include "s3d_wino24s_body.h"
#undef W24S_GNB
#define W24S_GNB 1
#include "s3d_wino24s_body.h"
#undef W24S_GNB

// ------------------------------------------------------------------ the LDS-DMA form: halo by buffer_load ... lds, persistent blocks
// (see s3d_wino24g_body.h)
constexpr int G_SLOT = 12288;                                  // bytes of one ring slot: 12 DMA pieces (180 pixels x 64 B = 11 520 used)
constexpr int G_GRED_FLOAT = 4 * G_SLOT / 4;                   // the GroupNorm cross-wave scratch behind the ring
constexpr int G_SMEM_FLOATS = G_GRED_FLOAT + 256;
#ifdef W24_TIMING
#define W24G_STAMP(k) if (threadIdx.x == 0) { g_w24time[size_t(item) * 8 + (k)] = wall_clock64(); \
        if ((k) == 2) g_w24time[size_t(item) * 8 + 6] = clock64(); if ((k) == 3) g_w24time[size_t(item) * 8 + 7] = clock64(); }
#else
#define W24G_STAMP(k)
#endif
#define W24G_GNB 0
#include "s3d_wino24g_body.h"
#undef W24G_GNB
#define W24G_GNB 1
#include "s3d_wino24g_body.h"
#undef W24G_GNB

// ------------------------------------------------------------------ the wide-block form: 64 output channels per block
// VERDICT r3 item 1.  k_conv_wino24s repeats a pixel tile's halo fetch and B^T d B input transform in every one of its cout / 32
// blocks, and its k-loop has no issue slack left (profiles/r03_gn_in_halo_price.txt); in steady state (launches of many rounds
// of blocks: batch 8, the (256,256,128) planes, the training tier) one of a CU's three blocks is always in its ~10 us prologue /
// epilogue and the other two do not fill the matrix pipe (0.45-0.50 of peak, profiles/r03_wino_ubench.txt).  Here a block owns
// the same 8x16 pixels and TWO n32 sub-blocks: every A operand built from LDS feeds 8 MFMAs instead of 4 — half the
// ds_read_b128, transform VALU, halo loads, prologues and epilogues per MFMA.  96 accumulator registers -> two blocks per CU
// (launch bound 2: 256 registers per lane), 64 KB of LDS (the four share images are 64 channels wide).
// SAME arithmetic in the SAME order per output element and per GroupNorm partial as k_conv_wino24s: each accumulator still sees
// k-steps 0, 1, ... with products j = 0..3 in order (the two sub-blocks' chains are interleaved MFMA by MFMA, which only separates
// dependent instructions), the share-image / finishing pass runs once per n32 half with the thread mapping of k_conv_wino24s,
// and the partial-sum tree is the same.  Results are bit-identical (tests/test_hip_parity.py::test_switched_conv_forms_...),
// so the launcher may choose by launch size — including the batch.
// Weights: the image of k_conv_wino24s, [n32][k16][24 freq][2 x 16 couts][64 lanes][4]; sub-block n2 reads n32 = 2 * n64 + n2.
#ifndef W24W_ABL
#define W24W_ABL 0                                 // tools/wino24_ubench.hip ablation builds (results meaningless): 1 no weight loads in the loop, 2 no halo loads / stores, 4 no patch reads + transform
#endif
#ifndef W24W_RING
#define W24W_RING 8                                // weight fragments in flight per wave (pairs: one per sub-block), each requested W24W_RING / 2 groups of 8 MFMAs ahead
#endif
constexpr int D_IMG = (C_TH / 2) * C_TW * 64;      // one share image [4 tile rows][16 columns][64 channels]

__global__ __launch_bounds__(256, 2) void k_conv_wino24w(ConvArgs args) {
    __shared__ __attribute__((aligned(16))) float smem[4 * D_IMG];              // 64 KB: two halo buffers (51.8 KB) in the k-loop, four share images after it
    static_assert(2 * C_ABUF <= 4 * D_IMG, "LDS plan");
    W24_STAMP(0)
    int bid = blockIdx.x;
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    if (args.xcd_swizzle & 1) {
        const int chunk = int(gridDim.x) >> 3;
        if (bid < (chunk << 3)) bid = (bid & 7) * chunk + (bid >> 3);
    }
    int j = 0;
#pragma unroll
    for (int k = 1; k < kMaxConvJobs; ++k) j += (k < args.njobs && bid >= args.job[k].block_begin) ? 1 : 0;
    const ConvJob& J = args.job[j];
    int local = bid - J.block_begin;
    const int n64 = local % J.n_tiles_n; local /= J.n_tiles_n;
    const int b = local / J.tiles_per_img; local %= J.tiles_per_img;
    const int tile_idx = local;
    const int ty0 = (local / J.tiles_x) * C_TH, tx0 = (local % J.tiles_x) * C_TW;
    const int h = J.h, w = J.w, cin = args.cin, cout = args.cout;

    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);                 // row frequency of this wave
    const int t16 = lane & 15, g = lane >> 4;                               // tile of the lane, channel quad of the lane
    const int tr = t16 >> 2, tc = t16 & 3;
    const int xrow = u == 0 ? 0 : (u == 2 ? 2 : 1), yrow = u == 2 ? 1 : (u == 3 ? 3 : 2);
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(u == 1 ? 0x3F800000 : 0xBF800000));
    const int qx = g ^ (((tr + (xrow >> 1)) & 3) << 1), qy = g ^ (((tr + (yrow >> 1)) & 3) << 1);
    const int bx = ((2 * tr + xrow) * C_HW + 4 * tc) * C_LD * 4, by = ((2 * tr + yrow) * C_HW + 4 * tc) * C_LD * 4;
    const int ax0 = bx + 16 * qx, ax1 = bx + 16 * (qx ^ 4), ay0 = by + 16 * qy, ay1 = by + 16 * (qy ^ 4);

    const int k16_total = cin / 16;
    const int n2stride = k16_total * 48 * 1024;                             // bytes between the images of two n32 sub-blocks
    const float* ub = J.wgt + ((size_t(2 * n64) * k16_total) * 48 + u * 12) * 256;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, 2 * n2stride, 0x00020000);
    const int wlane = lane * 16;
    auto wfrag = [&](int step, int s, int n2) -> f32x4 {                    // s = 2 * frequency + cout block of 16
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (step * 48 + s) * 1024 + n2 * n2stride, 0));
    };
    const float* inb = J.in + size_t(b) * h * w * cin;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inb), 0, h * w * cin * 4, 0x00020000);
    constexpr int R = W24W_RING, RG = R / 2;                                // ring slots / groups of lead
    static_assert(R % 2 == 0 && RG >= 1 && RG <= 12, "weight ring");
    f32x4 ring[R];
#pragma unroll
    for (int s = 0; s < R; ++s) { ring[s] = wfrag(0, s >> 1, s & 1); __builtin_amdgcn_sched_barrier(0); }
    unsigned goff[C_ITEMS_PT];
    int loff[C_ITEMS_PT];
    const unsigned rowstride = unsigned(w) * unsigned(cin) * 4u, pixstride = unsigned(cin) * 4u;
    const bool small_strides = rowstride < (1u << 24) && h < (1 << 24);
#pragma unroll
    for (int it = 0; it < C_ITEMS_PT; ++it) {
        const int item = it * 256 + tid;
        const int pix = item >> 3, q = item & 7;
        const int hy = (pix * 57) >> 10, hx = pix - hy * C_HW;             // pix / 18 for pix < 192
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        const bool ok = item < C_ITEMS && gy >= 0 && gy < h && gx >= 0 && gx < w;
        const unsigned off = small_strides ? __umul24(unsigned(gy), rowstride) + __umul24(unsigned(gx), pixstride) + unsigned(q) * 16u
                                           : unsigned((gy * w + gx) * cin + q * 4) * 4u;
        goff[it] = ok ? off : 0x80000000u;
        loff[it] = pix * C_LD + ((q ^ (((hy >> 1) & 3) << 1)) << 2);
    }
    const bool last_ok = (C_ITEMS_PT - 1) * 256 + tid < C_ITEMS;
    auto item_load = [&](int it, int ch) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[it], ch * (C_KC * 4), 0));
    };
    auto item_store = [&](int it, int buf, f32x4 v) {
        if (it < C_ITEMS_PT - 1 || last_ok) *reinterpret_cast<f32x4*>(smem + buf * C_ABUF + loff[it]) = v;
    };

    f32x4 acc[6][2][2];                                                     // [frequency][n32 sub-block][16-cout block]
#pragma unroll
    for (int f = 0; f < 6; ++f)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[f][n2][nb] = zero4;

    const int nchunks = cin / C_KC;
    f32x4 V[6];
    f32x4 pre[3];
    {
        const int c1 = nchunks > 1 ? 1 : 0;
        f32x4 h0[C_ITEMS_PT];
#pragma unroll
        for (int it = 0; it < C_ITEMS_PT; ++it) h0[it] = item_load(it, 0);
#pragma unroll
        for (int it = 0; it < 3; ++it) pre[it] = item_load(it, c1);
#pragma unroll
        for (int it = 0; it < C_ITEMS_PT; ++it) item_store(it, 0, h0[it]);
    }
    __syncthreads();
    W24_STAMP(1)
#define C_LDS4(off) (*static_cast<const f32x4*>(__builtin_assume_aligned(reinterpret_cast<const char*>(smem) + (off), 16)))
#define C_PIN(v) asm volatile("" : "+v"(v))
    {
        f32x4 t[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const f32x4 x = C_LDS4(ax0 + c * (C_LD * 4)), y = C_LDS4(ay0 + c * (C_LD * 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) t[c][e] = fmaf(sgn, y[e], x[e]);
        }
        const f32x4 s1 = t[4] - 4.f * t[2], s2 = t[3] - 4.f * t[1], s3 = t[4] - t[2], s4 = t[3] - t[1];
        V[0] = 4.f * t[0] + (t[4] - 5.f * t[2]);
        V[1] = s1 + s2; V[2] = s1 - s2;
        V[3] = s3 + 2.f * s4; V[4] = s3 - 2.f * s4;
        V[5] = 4.f * t[1] + (t[5] - 5.f * t[3]);
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) item_store(it, 1, pre[it]);

    // One k-step (16 channels) = 12 groups {eight MFMAs on one A operand: the two sub-blocks' chains interleaved + a piece of the
    // other work}, pinned.  Group g_ = 2 * F + NB consumes the fragment pair (g_, n2 = 0 / 1) and requests the pair RG groups on.
#define D_HAS_NEXT 1                                  /* 0 in a tile's last k-step: no next step's weight fragments to request */
#define D_GROUP(F, NB, WORK)                                                                                          \
    {                                                                                                                 \
        constexpr int g_ = 2 * (F) + (NB);                                                                            \
        const f32x4 bqa = ring[(2 * g_) % R], bqb = ring[(2 * g_ + 1) % R];                                           \
        acc[F][0][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][0], bqa[0], acc[F][0][NB], 0, 0, 0);                \
        acc[F][1][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][0], bqb[0], acc[F][1][NB], 0, 0, 0);                \
        if (W24W_ABL & 1) {} else if (g_ + RG < 12) { ring[(2 * g_) % R] = wfrag(step, g_ + RG, 0); ring[(2 * g_ + 1) % R] = wfrag(step, g_ + RG, 1); } \
        else if (D_HAS_NEXT) { ring[(2 * g_) % R] = wfrag(nstep, g_ + RG - 12, 0); ring[(2 * g_ + 1) % R] = wfrag(nstep, g_ + RG - 12, 1); } \
        acc[F][0][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][1], bqa[1], acc[F][0][NB], 0, 0, 0);                \
        acc[F][1][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][1], bqb[1], acc[F][1][NB], 0, 0, 0);                \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        if (!(W24W_ABL & 4)) { WORK }                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        acc[F][0][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][2], bqa[2], acc[F][0][NB], 0, 0, 0);                \
        acc[F][1][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][2], bqb[2], acc[F][1][NB], 0, 0, 0);                \
        acc[F][0][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][3], bqa[3], acc[F][0][NB], 0, 0, 0);                \
        acc[F][1][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][3], bqb[3], acc[F][1][NB], 0, 0, 0);                \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
#define C_COMB(T, X, Y) { _Pragma("unroll") for (int e = 0; e < 4; ++e) T[e] = fmaf(sgn, Y[e], X[e]); C_PIN(T); }
#define D_BUILD_GROUPS(LOADS)                                                                                         \
        D_GROUP(0, 0, cx0 = C_LDS4(rx); cy0 = C_LDS4(ry); cx1 = C_LDS4(rx + C_LD * 4); cy1 = C_LDS4(ry + C_LD * 4);)  \
        D_GROUP(0, 1, C_COMB(t0, cx0, cy0) C_COMB(t1, cx1, cy1) LOADS)                                                \
        D_GROUP(1, 0, cx0 = C_LDS4(rx + 2 * C_LD * 4); cy0 = C_LDS4(ry + 2 * C_LD * 4); cx1 = C_LDS4(rx + 3 * C_LD * 4); cy1 = C_LDS4(ry + 3 * C_LD * 4);) \
        D_GROUP(1, 1, C_COMB(t2, cx0, cy0) C_COMB(t3, cx1, cy1))                                                      \
        D_GROUP(2, 0, cx0 = C_LDS4(rx + 4 * C_LD * 4); cy0 = C_LDS4(ry + 4 * C_LD * 4); cx1 = C_LDS4(rx + 5 * C_LD * 4); cy1 = C_LDS4(ry + 5 * C_LD * 4);) \
        D_GROUP(2, 1, C_COMB(t4, cx0, cy0) C_COMB(t5, cx1, cy1))                                                      \
        /* from here on V[0..2] are free: their MFMAs have been issued */                                             \
        D_GROUP(3, 0, s1 = t4 - 4.f * t2; C_PIN(s1); s2 = t3 - 4.f * t1; C_PIN(s2); V[1] = s1 + s2; C_PIN(V[1]); V[2] = s1 - s2; C_PIN(V[2]);) \
        D_GROUP(3, 1, V[0] = 4.f * t0 + (t4 - 5.f * t2); C_PIN(V[0]); s3 = t4 - t2; C_PIN(s3); s4 = t3 - t1; C_PIN(s4);) \
        D_GROUP(4, 0, V[3] = s3 + 2.f * s4; C_PIN(V[3]);)                                                             \
        D_GROUP(4, 1, v5n = 4.f * t1 + (t5 - 5.f * t3); C_PIN(v5n);)                                                  \
        D_GROUP(5, 0, V[4] = s3 - 2.f * s4; C_PIN(V[4]);)                                                             \
        D_GROUP(5, 1, ;)                                                                                              \
        if (!(W24W_ABL & 4)) V[5] = v5n;
#define D_STEP(KK)                                                                                                    \
    {                                                                                                                 \
        const int step = chunk * 2 + (KK);                                                                            \
        const int nstep = step + 1 < k16_total ? step + 1 : step;                                                     \
        const int rx = ((KK) == 0 ? ax1 + cur : ax0 + (cur ^ tog)), ry = ((KK) == 0 ? ay1 + cur : ay0 + (cur ^ tog)); \
        constexpr int it0 = (KK) == 0 ? 3 : 0;                                                                        \
        const int lch = (KK) == 0 ? cn1 : cn2;                                                                        \
        f32x4 pf[3], cx0, cy0, cx1, cy1, t0, t1, t2, t3, t4, t5, s1, s2, s3, s4, v5n;                                 \
        if (W24W_ABL & 2) { D_BUILD_GROUPS(;) } else {                                                                \
        D_BUILD_GROUPS(pf[0] = item_load(it0, lch); pf[1] = item_load(it0 + 1, lch); pf[2] = item_load(it0 + 2, lch);) \
        _Pragma("unroll") for (int t = 0; t < 3; ++t) item_store(it0 + t, (KK) == 0 ? (chunk + 1) & 1 : chunk & 1, pf[t]); } \
        if ((KK) == 0) __syncthreads();                                                                               \
    }

    int cur = 0;                                  // byte offset of the buffer that holds the current chunk
    const int tog = C_ABUF * 4;
    W24_STAMP(2)
    __builtin_amdgcn_s_setprio(0);
    for (int chunk = 0; chunk < nchunks - 1; ++chunk) {
        const int cn1 = chunk + 1;
        const int cn2 = chunk + 2 < nchunks ? chunk + 2 : nchunks - 1;
        D_STEP(0)
        D_STEP(1)
        cur ^= tog;
    }
    // the last chunk is peeled as in k_conv_wino24s: no successor halo, and its second step builds no operands
    {
        const int chunk = nchunks - 1;
        const int step = chunk * 2, nstep = step + 1;
        const int rx = ax1 + cur, ry = ay1 + cur;
        f32x4 cx0, cy0, cx1, cy1, t0, t1, t2, t3, t4, t5, s1, s2, s3, s4, v5n;
        D_BUILD_GROUPS(;)
    }
    const float* __restrict__ p_bias = J.bias;
    const float* __restrict__ p_bbias = J.bbias;
    const float* __restrict__ p_rcol = J.rcol;
    const float* __restrict__ p_rrow = J.rrow;
    const float* __restrict__ p_res = J.res;
    float* __restrict__ p_out = J.out;
    double* p_gn = J.gn_part;
    // finishing threads: the mapping of k_conv_wino24s (8 channel quads x 16 columns x 2 row halves), once per n32 half
    const int quad = tid & 7, xl = (tid >> 3) & 15, rsel = tid >> 7;
    const int x = tx0 + xl;
    const int xc = x < w ? x : 0;
    const bool two = args.r1_slices == 2;
    auto tload = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, kTabAux)); };
    f32x4 tres[2][4];
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
        for (int k = 0; k < 4; ++k) tres[n2][k] = zero4;
    auto request_residual = [&]() {
        if (p_res) {
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = ty0 + rsel * 4 + k;
                    tres[n2][k] = *reinterpret_cast<const f32x4*>(p_res + ((size_t(b) * h + (y < h ? y : 0)) * w + xc) * cout + n64 * 64 + n2 * 32 + quad * 4);
                }
        }
    };
#undef D_HAS_NEXT
#define D_HAS_NEXT 0
    {   // the tile's last k-step: MFMAs only; the residual request goes out behind its first groups
        const int step = (nchunks - 1) * 2 + 1, nstep = step; (void)nstep;
        D_GROUP(0, 0, ;) D_GROUP(0, 1, request_residual();) D_GROUP(1, 0, ;) D_GROUP(1, 1, ;) D_GROUP(2, 0, ;) D_GROUP(2, 1, ;)
        D_GROUP(3, 0, ;) D_GROUP(3, 1, ;) D_GROUP(4, 0, ;) D_GROUP(4, 1, ;) D_GROUP(5, 0, ;) D_GROUP(5, 1, ;)
    }
#undef D_HAS_NEXT
#undef D_STEP
#undef D_BUILD_GROUPS
#undef D_GROUP
#undef C_COMB
#undef C_LDS4
#undef C_PIN
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    W24_STAMP(3)
    // rank-1 tables (L2-hot, written by the launch before) + bias: half 0 is requested before the barrier, half 1 once the
    // accumulators have been written out (the registers of both halves' operands + 96 accumulators would not fit)
    f32x4 base4[2], tcol[2][4], trow[2][4], tcol2[2][4], trow2[2][4];
    auto request_tables = [&](int n2) {
        const int coc = n64 * 64 + n2 * 32 + quad * 4;
        base4[n2] = p_bias ? *reinterpret_cast<const f32x4*>(p_bias + coc) : zero4;
        if (p_bbias) base4[n2] += *reinterpret_cast<const f32x4*>(p_bbias + size_t(b) * J.bbias_stride + coc);
#pragma unroll
        for (int k = 0; k < 4; ++k) { tcol[n2][k] = zero4; trow[n2][k] = zero4; tcol2[n2][k] = zero4; trow2[n2][k] = zero4; }
        if (p_rrow) {
            const float* base = p_rrow + size_t(b) * h * 4 * cout;
            const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, h * 4 * cout * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t trs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (two ? size_t(args.B) * h * 4 * cout : 0)), 0, h * 4 * cout * 4, 0x00020000);
            const int vx = x_edge_variant(xc, w);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = ty0 + rsel * 4 + k;
                const unsigned off = unsigned((((y < h ? y : 0) * 4 + vx) * cout + coc) * 4);
                trow[n2][k] = tload(trs, off);
                if (two) trow2[n2][k] = tload(trs2, off);
            }
        }
        if (p_rcol) {
            const float* base = p_rcol + size_t(b) * w * 4 * cout;
            const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, w * 4 * cout * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t trs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (two ? size_t(args.B) * w * 4 * cout : 0)), 0, w * 4 * cout * 4, 0x00020000);
            if (ty0 > 0 && ty0 + C_TH < h) {
                const unsigned off = unsigned(((xc * 4 + 0) * cout + coc) * 4);
                const f32x4 v0 = tload(trs, off);
                f32x4 v1 = zero4;
                if (two) v1 = tload(trs2, off);
#pragma unroll
                for (int k = 0; k < 4; ++k) { tcol[n2][k] = v0; tcol2[n2][k] = v1; }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = ty0 + rsel * 4 + k;
                    const unsigned off = unsigned(((xc * 4 + x_edge_variant(y < h ? y : 0, h)) * cout + coc) * 4);
                    tcol[n2][k] = tload(trs, off);
                    if (two) tcol2[n2][k] = tload(trs2, off);
                }
            }
        }
    };
    request_tables(0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                     // all patch reads and halo stores of the last step are done
    {
        float* img = smem + u * D_IMG + t16;             // lane (g, t16) holds output channel n2*32 + nb*16 + t16 of the tiles (tile row g, tile column r)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m0 = acc[0][n2][nb][r], m1 = acc[1][n2][nb][r], m2 = acc[2][n2][nb][r], m3 = acc[3][n2][nb][r], m4 = acc[4][n2][nb][r], m5 = acc[5][n2][nb][r];
                    const float p = m1 + m2, q = m1 - m2, rr = m3 + m4, s = m3 - m4;
                    const int pp = (g * C_TW + 4 * r) * 64 + n2 * 32 + nb * 16;
                    img[pp] = (m0 + p) + rr; img[pp + 64] = fmaf(2.f, s, q); img[pp + 128] = fmaf(4.f, rr, p); img[pp + 192] = fmaf(8.f, s, q) + m5;
                }
    }
    __builtin_amdgcn_sched_barrier(0);
    request_tables(1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                     // the share images are complete
    W24_STAMP(4)
    __shared__ float gred[2][4][8][8];
    const bool x_ok = x < w;                             // (cout % 64 == 0: every channel quad of the block exists)
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
        const int co4 = n64 * 64 + n2 * 32 + quad * 4;
        f32x4 gs4 = zero4, gss4 = zero4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yl = rsel * 4 + k, y = ty0 + yl;
            const float* sp = smem + (yl & 1) * D_IMG + ((yl >> 1) * C_TW + xl) * 64 + n2 * 32 + quad * 4;
            const f32x4 ka = *reinterpret_cast<const f32x4*>(sp), kb = *reinterpret_cast<const f32x4*>(sp + D_IMG),
                        kc = *reinterpret_cast<const f32x4*>(sp + 2 * D_IMG);
            const f32x4 sum3 = (yl & 1) ? (ka - kb) - kc : (ka + kb) + kc;
            const f32x4 v = (sum3 + base4[n2]) + (((tcol[n2][k] + tcol2[n2][k]) + (trow[n2][k] + trow2[n2][k])) + tres[n2][k]);
            if (x_ok && y < h) {
                *reinterpret_cast<f32x4*>(p_out + ((size_t(b) * h + y) * w + x) * cout + co4) = v;
                gs4 += v; gss4 += v * v;
            }
        }
        if (p_gn) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) { gs4[e] += __shfl_xor(gs4[e], off, 64); gss4[e] += __shfl_xor(gss4[e], off, 64); }
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { gred[n2][u][lane][e] = gs4[e]; gred[n2][u][lane][4 + e] = gss4[e]; }
            }
        }
    }
    if (p_gn) {
        // one partial per TILE and sub-group, formed exactly as in k_conv_wino24s (four waves in order, double); wave n2 writes half n2
        __syncthreads();
        if (tid < 128) {
            const int n2 = tid >> 6;
            const int co4 = n64 * 64 + n2 * 32 + (lane & 7) * 4;
            double ds[4], dss[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int l8 = lane & 7;
                ds[e] = ((double(gred[n2][0][l8][e]) + double(gred[n2][1][l8][e])) + double(gred[n2][2][l8][e])) + double(gred[n2][3][l8][e]);
                dss[e] = ((double(gred[n2][0][l8][4 + e]) + double(gred[n2][1][l8][4 + e])) + double(gred[n2][2][l8][4 + e])) + double(gred[n2][3][l8][4 + e]);
            }
            const int sg = args.gn_sg;
            const int part = tile_idx;
            auto put = [&](int sub, double sv, double ssv) {
                double* dst = p_gn + ((size_t(b) * 3 * args.gn_nsub + sub) * args.gn_maxparts + part) * 2;
                dst[0] = sv; dst[1] = ssv;
            };
            if (sg >= 4) {
                double sv = (ds[0] + ds[1]) + (ds[2] + ds[3]), ssv = (dss[0] + dss[1]) + (dss[2] + dss[3]);
                for (int off = 1; off < (sg >> 2); off <<= 1) { sv += __shfl_xor(sv, off, 64); ssv += __shfl_xor(ssv, off, 64); }
                if (lane < 8 && (co4 % sg) == 0) put(co4 / sg, sv, ssv);
            } else if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    if (sg == 2) put((co4 + e) / 2, ds[e] + ds[e + 1], dss[e] + dss[e + 1]);
                    else { put(co4 + e, ds[e], dss[e]); put(co4 + e + 1, ds[e + 1], dss[e + 1]); }
                }
            }
        }
    }
    W24_STAMP(5)
}

// ------------------------------------------------------------------ host side
static const double kG2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
static const double kG4[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};

size_t wino24_packed_floats(int cout, int cin) { return size_t((cout + 31) / 32) * (cin / 8) * 24 * 256; }

// U = G2 g G4^T in double (explicit fma: host and device — s3d_pack.hip — round alike); W is OIHW [cout][ctot][3][3], only input
// channels [0, cin) are used (the plane's own channels).
// the small-block kernel's image: [n32][k16][24 freq][2 x 16 couts][64 lanes = 16 * channel quad + cout][4 channels]
size_t pack_wino24s_weights(std::vector<float>& stage, const float* W, int cout, int ctot, int cin) {
    const int k16t = cin / 16;
    const size_t total = wino24_packed_floats(cout, cin);
    const size_t off = push(stage, nullptr, total);
    float* d = stage.data() + off;
    std::fill(d, d + total, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int c = 0; c < cin; ++c) {
            const float* g = W + (size_t(co) * ctot + c) * 9;
            double t[4][3];
            for (int u = 0; u < 4; ++u)
                for (int k = 0; k < 3; ++k) t[u][k] = kG2[u][0] * g[0 * 3 + k] + kG2[u][1] * g[1 * 3 + k] + kG2[u][2] * g[2 * 3 + k];
            const int nt = co >> 5, nb = (co >> 4) & 1, jn = co & 15, k16 = c >> 4, q = (c >> 2) & 3, e = c & 3;
            for (int u = 0; u < 4; ++u)
                for (int v = 0; v < 6; ++v) {
                    const double uv = fma(t[u][0], kG4[v][0], fma(t[u][1], kG4[v][1], t[u][2] * kG4[v][2]));    // explicit fma: host and device (s3d_pack.hip) round alike
                    d[((((size_t(nt) * k16t + k16) * 24 + (u * 6 + v)) * 2 + nb) * 64 + (q * 16 + jn)) * 4 + e] = float(uv);
                }
        }
    return off;
}



static int wino24s_layout(ConvArgs& a, int cout_per_block, const char* who) {
    int blocks = 0;
    for (int j = 0; j < a.njobs; ++j) {
        ConvJob& J = a.job[j];
        if (size_t(J.h) * J.w * a.cin * 4 >= (size_t(1) << 31)) { set_error("%s: a plane of one sample must stay below 2 GiB", who); return -1; }
        J.tiles_x = (J.w + C_TW - 1) / C_TW;
        J.tiles_per_img = J.tiles_x * ((J.h + C_TH - 1) / C_TH);
        J.n_tiles_n = (a.cout + cout_per_block - 1) / cout_per_block;
        J.block_begin = blocks;
        blocks += J.tiles_per_img * J.n_tiles_n * a.B;
    }
    return blocks;
}

// Which blocking a launch takes.  The two kernels are bit-identical, so the choice may depend on anything, the batch included.
// Measured in steady state on every layer shape x batch 1 / 2 / 4 / 8 (profiles/r04_wino_ubench.txt): the wide form wins from
// 1.5 rounds of its own blocks on (3 x CUs; two blocks per CU) when K >= 256 — that includes the 384 -> 128 layer of the batch-1
// step, 136 -> 127 us — and from 3 rounds on at K = 128 (a tie below); it loses on the one-round half-resolution launches of
// the batch-1 step (384 blocks on 512 slots) and at K = 64 (the k-loop is too short to amortise the 64-channel epilogue).
// S3D_WINO24W=0: never; =1: every launch whose cout is a multiple of 64.
static int conv_cus() { return device_cus(); }
static bool takes_wide(const ConvArgs& a) {
    const int mode = opt(OPT_WINO24W);
    if (mode == 0 || a.cout % 64 != 0) return false;
    if (mode == 1) return true;
    if (a.cin < 128) return false;
    long long blocks = 0;
    for (int j = 0; j < a.njobs; ++j) blocks += (long long)((a.job[j].w + C_TW - 1) / C_TW) * ((a.job[j].h + C_TH - 1) / C_TH) * (a.cout / 64) * a.B;
    const int cus = conv_cus();
    return blocks >= (long long)(a.cin >= 256 ? 3 : 6) * cus;
}

// (round 5: an XCD owning one half of the channel blocks of a quarter of the tiles instead — half the weight image per L2, every
// halo fetched twice — measured the same step time and 3 % less traffic on config 3; not kept: profiles/r05_xcd_mapping.txt)
int launch_conv_wino24_wide(ConvArgs& a, hipStream_t st) {
    S3D_CHECK(a.njobs >= 1 && a.njobs <= kMaxConvJobs && a.cin % C_KC == 0 && a.cout % 64 == 0, S3D_ERR_INVALID, "wino24w conv: bad arguments");
    const int blocks = wino24s_layout(a, 64, "wino24w conv");
    if (blocks < 0) return S3D_ERR_INVALID;
    if (!blocks) return 0;
    a.xcd_swizzle = 1 | 2;
    conv_note_kernel("k_conv_wino24w mixed Winograd F(2x4,3x3), 8x16-pixel x 64-cout blocks");
    hipLaunchKernelGGL(k_conv_wino24w, dim3(blocks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// k_conv_wino24g: at most three blocks per CU (a multiple of 8: whole XCD shares); more items than that -> persistent blocks
static int glds_grid(int items) {
    const int cap = (3 * conv_cus()) & ~7;
    return items > cap && cap >= 8 ? cap : items;
}
int launch_conv_wino24_glds(ConvArgs& a, hipStream_t st) {
    S3D_CHECK(a.njobs >= 1 && a.njobs <= kMaxConvJobs && a.cin % C_KC == 0 && a.cout % 4 == 0, S3D_ERR_INVALID, "wino24g conv: bad arguments");
    const int blocks = wino24s_layout(a, 32, "wino24g conv");
    if (blocks < 0) return S3D_ERR_INVALID;
    if (!blocks) return 0;
    a.xcd_swizzle = 1 | 2;
    conv_note_kernel("k_conv_wino24g mixed Winograd F(2x4,3x3), 8x16-pixel blocks, halo by LDS-DMA, persistent");
    hipLaunchKernelGGL(k_conv_wino24g, dim3(glds_grid(blocks)), dim3(256), 0, st, a, blocks);
    S3D_HIP(hipGetLastError());
    return 0;
}
int launch_conv_wino24_glds_gnb(ConvArgs& a, const GnbArgs& gb, hipStream_t st) {
    S3D_CHECK(a.njobs == 3 && a.cin % C_KC == 0 && a.cout % 32 == 0 && gb.groups >= 1 && a.cout % gb.groups == 0, S3D_ERR_INVALID, "wino24g gnb conv: bad arguments");
    const int blocks = wino24s_layout(a, 32, "wino24g gnb conv");
    if (blocks < 0) return S3D_ERR_INVALID;
    if (!blocks) return 0;
    a.xcd_swizzle = 1 | 2;
    conv_note_kernel("k_conv_wino24g_gnb mixed Winograd F(2x4,3x3) + GroupNorm-backward partial sums, halo by LDS-DMA, persistent");
    hipLaunchKernelGGL(k_conv_wino24g_gnb, dim3(glds_grid(blocks)), dim3(256), 0, st, a, gb, blocks);
    S3D_HIP(hipGetLastError());
    return 0;
}

int launch_conv_wino24_narrow(ConvArgs& a, hipStream_t st) {
    S3D_CHECK(a.njobs >= 1 && a.njobs <= kMaxConvJobs && a.cin % C_KC == 0 && a.cout % 4 == 0, S3D_ERR_INVALID, "wino24s conv: bad arguments");
    const int blocks = wino24s_layout(a, 32, "wino24s conv");
    if (blocks < 0) return S3D_ERR_INVALID;
    if (!blocks) return 0;
    a.xcd_swizzle = 1 | 2;                   // XCD-aware block order + raised priority outside the k-loop (were switchable in rounds 1-2: always wins)
    conv_note_kernel("k_conv_wino24s mixed Winograd F(2x4,3x3), 8x16-pixel blocks");
    hipLaunchKernelGGL(k_conv_wino24s, dim3(blocks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// the input-gradient convolution that also leaves the following GroupNorm backward's partial sums (k_conv_wino24s_gnb; always the
// 32-output-channel block: the sums' order must not depend on the launch size)
int launch_conv_wino24_gnb(ConvArgs& a, const GnbArgs& gb, hipStream_t st) {
    S3D_CHECK(a.njobs == 3 && a.cin % C_KC == 0 && a.cout % 32 == 0 && gb.groups >= 1 && a.cout % gb.groups == 0, S3D_ERR_INVALID, "wino24s gnb conv: bad arguments");
    const int blocks = wino24s_layout(a, 32, "wino24s gnb conv");
    if (blocks < 0) return S3D_ERR_INVALID;
    if (!blocks) return 0;
    a.xcd_swizzle = 1 | 2;
    conv_note_kernel("k_conv_wino24s_gnb mixed Winograd F(2x4,3x3) + GroupNorm-backward partial sums, 8x16-pixel blocks");
    hipLaunchKernelGGL(k_conv_wino24s_gnb, dim3(blocks), dim3(256), 0, st, a, gb);
    S3D_HIP(hipGetLastError());
    return 0;
}

// S3D_WINO24G=1: every launch of the mixed Winograd kernel takes the LDS-DMA / persistent form (bit-identical; measured on par with
// or behind the register-staged kernels on every shape: profiles/r06_wino_glds.txt — not a default).
int launch_conv_wino24s(ConvArgs& a, hipStream_t st) {
    const bool glds = opt(OPT_WINO24G) == 1;
    if (a.gnb) return glds ? launch_conv_wino24_glds_gnb(a, *a.gnb, st) : launch_conv_wino24_gnb(a, *a.gnb, st);
    if (glds) return launch_conv_wino24_glds(a, st);
    return takes_wide(a) ? launch_conv_wino24_wide(a, st) : launch_conv_wino24_narrow(a, st);
}

}  // namespace s3d
