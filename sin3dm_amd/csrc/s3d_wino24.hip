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
// Data flow (same skeleton as k_conv_wino4; what changes is noted):
//   * block = 16x16 output pixels = 8x4 tiles of 2x4 = one 32-row MFMA tile, x 32 output channels; four waves, wave u owns
//     row u of the 4x6 frequency grid: 6 accumulators of 32x32 (96 registers) -> two blocks per CU, two waves per SIMD;
//   * the 18x18-pixel input halo of a 16-channel chunk sits in LDS (20 floats per pixel, channel quads XOR-swizzled with
//     the tile row so that the 4x6-patch reads of a lane group hit 16 distinct 16-byte slots: conflict-free); two buffers.
//     A chunk is two k-steps of 8 channels; the halo of chunk c+1 is fetched half in step (c-1,1), half in step (c,0), with ONE
//     barrier per chunk (after step 0);
//   * a lane owns one tile: per k-step it reads 2 patch rows x 6 columns (ds_read_b128), combines them for its wave's row
//     frequency (t = x + s*y) and runs the 6-point column transform (12 vector operations) -> A operands of 6 frequencies x
//     4 channels, each used by four MFMAs;
//   * weights G2 g G4^T are packed (host, in double) in MFMA fragment order [n32][k8][24 freq][64 lanes][4]: one coalesced
//     1 KB load per (frequency, 8 channels), through a register ring, no LDS;
//   * the k-step is 24 pinned slots {one MFMA + a piece of the other work};
//   * epilogue: each wave applies the 6 -> 4 column pass to its frequency row and writes it as a share image
//     [8 tile rows][16 columns][32 channels]; pixel-quad threads combine three images (row pass), add bias / rank-1
//     rollout terms / residual, store 16 bytes and reduce the GroupNorm partial sums.
#include "s3d_common.h"
#include "s3d_rank1.h"

namespace s3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int X_KC = 16;                           // channels per chunk
constexpr int X_LD = X_KC + 4;                     // LDS floats per halo pixel
constexpr int X_TH = 16, X_TW = 16;                // output pixels of a block: 8 x 4 tiles of 2 x 4
constexpr int X_HH = X_TH + 2, X_HW = X_TW + 2;    // halo
constexpr int X_PIX = X_HH * X_HW;                 // 324
constexpr int X_ITEMS = X_PIX * (X_KC / 4);        // float4 items per chunk (1296)
constexpr int X_ITEMS_PT = (X_ITEMS + 255) / 256;  // per thread (6; the sixth round covers 16 items)
constexpr int X_ABUF = X_PIX * X_LD;               // floats per halo buffer
constexpr int X_IMG = (X_TH / 2) * X_TW * 32;      // one share image

__device__ __forceinline__ int x_edge_variant(int idx, int n) { return n == 1 ? 3 : (idx == 0 ? 1 : (idx == n - 1 ? 2 : 0)); }

__global__ __launch_bounds__(256, 2) void k_conv_wino24(ConvArgs args) {
    __shared__ __attribute__((aligned(16))) float smem[4 * X_IMG];              // 64 KB: two halo buffers in the k-loop, four share images after it
    static_assert(2 * X_ABUF <= 4 * X_IMG, "LDS plan");
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    int bid = blockIdx.x;
    if (args.xcd_swizzle & 1) {
        const int chunk = int(gridDim.x) >> 3;
        if (bid < (chunk << 3)) bid = (bid & 7) * chunk + (bid >> 3);
    }
    int j = 0;
#pragma unroll
    for (int k = 1; k < kMaxConvJobs; ++k) j += (k < args.njobs && bid >= args.job[k].block_begin) ? 1 : 0;
    const ConvJob& J = args.job[j];
    int local = bid - J.block_begin;
    const int n32 = local % J.n_tiles_n; local /= J.n_tiles_n;
    const int b = local / J.tiles_per_img; local %= J.tiles_per_img;
    const int tile_idx = local;
    const int ty0 = (local / J.tiles_x) * X_TH, tx0 = (local % J.tiles_x) * X_TW;
    const int h = J.h, w = J.w, cin = args.cin, cout = args.cout;

    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);                 // row frequency of this wave
    const int i = lane & 31, half = lane >> 5;
    const int tr = i >> 2, tc = i & 3;
    // wave u needs patch rows (x, y): t = x + s*y with (x, y, s) = (d0,d2,-), (d1,d2,+), (d2,d1,-), (d1,d3,-)
    const int xrow = u == 0 ? 0 : (u == 2 ? 2 : 1), yrow = u == 2 ? 1 : (u == 3 ? 3 : 2);
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(u == 1 ? 0x3F800000 : 0xBF800000));
    // byte addresses of the lane's patch rows in a halo buffer, for the first / second 8 channels of a chunk:
    // pixel (2*tr + row, 4*tc + c), logical quad q = 2*kk + half stored at quad q ^ ((2*tr + row) >> 1 & 3)
    const int ex = half ^ ((tr + (xrow >> 1)) & 3), ey = half ^ ((tr + (yrow >> 1)) & 3);
    const int bx = ((2 * tr + xrow) * X_HW + 4 * tc) * X_LD * 4, by = ((2 * tr + yrow) * X_HW + 4 * tc) * X_LD * 4;
    const int ax0 = bx + 16 * ex, ax1 = bx + 16 * (ex ^ 2), ay0 = by + 16 * ey, ay1 = by + 16 * (ey ^ 2);

    const int k8_total = cin / 8;
    const float* ub = J.wgt + ((size_t(n32) * k8_total) * 24 + u * 6) * 256;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, k8_total * 24 * 1024, 0x00020000);
    const int wlane = lane * 16;
    auto wfrag = [&](int step, int f) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (step * 24 + f) * 1024, 0));
    };
    const float* inb = J.in + size_t(b) * h * w * cin;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inb), 0, h * w * cin * 4, 0x00020000);
    // halo staging: item = it*256 + tid -> (pixel item>>2, channel quad item&3); out-of-image pixels (and items past the
    // halo) get an offset beyond the descriptor's range: the hardware returns zeros = the convolution's padding
    unsigned goff[X_ITEMS_PT];
    int loff[X_ITEMS_PT];
#pragma unroll
    for (int it = 0; it < X_ITEMS_PT; ++it) {
        const int item = it * 256 + tid;
        const int pix = item >> 2, q = item & 3;
        const int hy = pix / X_HW, hx = pix - hy * X_HW;
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        const bool ok = item < X_ITEMS && gy >= 0 && gy < h && gx >= 0 && gx < w;
        goff[it] = ok ? unsigned((gy * w + gx) * cin + q * 4) * 4u : 0x80000000u;
        loff[it] = pix * X_LD + ((q ^ ((hy >> 1) & 3)) << 2);
    }
    const bool last_ok = (X_ITEMS_PT - 1) * 256 + tid < X_ITEMS;
    auto item_load = [&](int it, int ch) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[it], ch * (X_KC * 4), 0));
    };
    auto item_store = [&](int it, int buf, f32x4 v) {
        if (it < X_ITEMS_PT - 1 || last_ok) *reinterpret_cast<f32x4*>(smem + buf * X_ABUF + loff[it]) = v;
    };

    f32x16 acc[6];
#pragma unroll
    for (int f = 0; f < 6; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    const int nchunks = cin / X_KC;
    f32x4 VA[6], VB[6], ring[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) { ring[f] = wfrag(0, f); __builtin_amdgcn_sched_barrier(0); }
    {
        const int c1 = nchunks > 1 ? 1 : 0;
#pragma unroll
        for (int it = 0; it < X_ITEMS_PT; ++it) item_store(it, 0, item_load(it, 0));
#pragma unroll
        for (int it = 0; it < 3; ++it) item_store(it, 1, item_load(it, c1));
    }
    __syncthreads();
#define X_LDS4(off) (*static_cast<const f32x4*>(__builtin_assume_aligned(reinterpret_cast<const char*>(smem) + (off), 16)))
#define X_PIN(v) asm volatile("" : "+v"(v))
    // column transform B4^T of the row-combined vector t[0..5] (F(4,3), points 0, +-1, +-2, inf):
    //   V0 = 4 t0 - 5 t2 + t4        V1 = s1 + s2, V2 = s1 - s2   with s1 = t4 - 4 t2, s2 = t3 - 4 t1
    //   V5 = 4 t1 - 5 t3 + t5        V3 = s3 + 2 s4, V4 = s3 - 2 s4 with s3 = t4 - t2,  s4 = t3 - t1
    {
        f32x4 t[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const f32x4 x = X_LDS4(ax0 + c * (X_LD * 4)), y = X_LDS4(ay0 + c * (X_LD * 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) t[c][e] = fmaf(sgn, y[e], x[e]);
        }
        const f32x4 s1 = t[4] - 4.f * t[2], s2 = t[3] - 4.f * t[1], s3 = t[4] - t[2], s4 = t[3] - t[1];
        VA[0] = 4.f * t[0] + (t[4] - 5.f * t[2]);
        VA[1] = s1 + s2; VA[2] = s1 - s2;
        VA[3] = s3 + 2.f * s4; VA[4] = s3 - 2.f * s4;
        VA[5] = 4.f * t[1] + (t[5] - 5.f * t[3]);
    }

    // One k-step = 24 slots of {one MFMA + a piece of the other work}.  For the NEXT step's operands: 12 patch reads
    // (slots of f = 0..2), the six row combinations (f = 0..2), the column transform (f = 3, 4); 6 weight fragments (one per
    // f), 3 halo loads whose data is only touched by the ds_write at the end of the step.
    //   RA/RB: byte address sets of the patch to read (same buffer, second channel half in step 0; other buffer, first half in step 1)
#define W24_STEP(Vc, Vn, KK)                                                                                          \
    {                                                                                                                 \
        const int step = chunk * 2 + (KK);                                                                            \
        const int nstep = step + 1 < k8_total ? step + 1 : step;                                                      \
        const int rx = ((KK) == 0 ? ax1 : ax0) + ((KK) == 0 ? cur : (cur ^ tog)) * 1;                                 \
        const int ry = ((KK) == 0 ? ay1 : ay0) + ((KK) == 0 ? cur : (cur ^ tog)) * 1;                                 \
        constexpr int it0 = (KK) == 0 ? 3 : 0;                                                                        \
        const int lch = (KK) == 0 ? cn1 : cn2;                                                                        \
        f32x4 pf[3], cx0, cy0, cx1, cy1, t0, t1, t2, t3, t4, t5, s1, s2, s3, s4;                                      \
        _Pragma("unroll") for (int f = 0; f < 6; ++f) {                                                               \
            const f32x4 bq = ring[f];                                                                                 \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][0], bq[0], acc[f], 0, 0, 0);                          \
            if (f < 3) {                                                                                              \
                cx0 = X_LDS4(rx + (2 * f) * (X_LD * 4)); cy0 = X_LDS4(ry + (2 * f) * (X_LD * 4));                     \
                cx1 = X_LDS4(rx + (2 * f + 1) * (X_LD * 4)); cy1 = X_LDS4(ry + (2 * f + 1) * (X_LD * 4));             \
            }                                                                                                         \
            if (f == 3) { s1 = t4 - 4.f * t2; X_PIN(s1); s2 = t3 - 4.f * t1; X_PIN(s2); }                             \
            if (f == 4) { s3 = t4 - t2; X_PIN(s3); s4 = t3 - t1; X_PIN(s4); }                                         \
            if (f == 5) { Vn[5] = 4.f * t1 + (t5 - 5.f * t3); X_PIN(Vn[5]); }                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][1], bq[1], acc[f], 0, 0, 0);                          \
            ring[f] = wfrag(nstep, f);                                                                                \
            if (f == 1) { pf[0] = item_load(it0, lch); pf[1] = item_load(it0 + 1, lch); pf[2] = item_load(it0 + 2, lch); } \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][2], bq[2], acc[f], 0, 0, 0);                          \
            if (f == 0) { _Pragma("unroll") for (int e = 0; e < 4; ++e) t0[e] = fmaf(sgn, cy0[e], cx0[e]); X_PIN(t0); } \
            if (f == 1) { _Pragma("unroll") for (int e = 0; e < 4; ++e) t2[e] = fmaf(sgn, cy0[e], cx0[e]); X_PIN(t2); } \
            if (f == 2) { _Pragma("unroll") for (int e = 0; e < 4; ++e) t4[e] = fmaf(sgn, cy0[e], cx0[e]); X_PIN(t4); } \
            if (f == 3) { Vn[1] = s1 + s2; X_PIN(Vn[1]); Vn[2] = s1 - s2; X_PIN(Vn[2]); }                             \
            if (f == 4) { Vn[3] = s3 + 2.f * s4; X_PIN(Vn[3]); Vn[4] = s3 - 2.f * s4; X_PIN(Vn[4]); }                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][3], bq[3], acc[f], 0, 0, 0);                          \
            if (f == 0) { _Pragma("unroll") for (int e = 0; e < 4; ++e) t1[e] = fmaf(sgn, cy1[e], cx1[e]); X_PIN(t1); } \
            if (f == 1) { _Pragma("unroll") for (int e = 0; e < 4; ++e) t3[e] = fmaf(sgn, cy1[e], cx1[e]); X_PIN(t3); } \
            if (f == 2) { _Pragma("unroll") for (int e = 0; e < 4; ++e) t5[e] = fmaf(sgn, cy1[e], cx1[e]); X_PIN(t5); } \
            if (f == 3) { Vn[0] = 4.f * t0 + (t4 - 5.f * t2); X_PIN(Vn[0]); }                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
        _Pragma("unroll") for (int t = 0; t < 3; ++t) item_store(it0 + t, (KK) == 0 ? (chunk + 1) & 1 : chunk & 1, pf[t]); \
        if ((KK) == 0) __syncthreads();                                                                               \
    }

    int cur = 0;                                  // byte offset of the buffer that holds the current chunk
    const int tog = X_ABUF * 4;
    __builtin_amdgcn_s_setprio(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cn1 = chunk + 1 < nchunks ? chunk + 1 : nchunks - 1;
        const int cn2 = chunk + 2 < nchunks ? chunk + 2 : nchunks - 1;
        W24_STEP(VA, VB, 0)
        W24_STEP(VB, VA, 1)
        cur ^= tog;
    }
#undef W24_STEP
#undef X_LDS4
#undef X_PIN
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);

    // ---- epilogue.  acc[v] = M[u][v].  Column pass A4^T (6 -> the 4 pixels of a tile row):
    //   c0 = m0 + p + r, c1 = q + 2 s, c2 = p + 4 r, c3 = q + 8 s + m5   with p = m1 + m2, q = m1 - m2, r = m3 + m4, s = m3 - m4
    // then the F(2,3) row pass over the waves: tile row 0 = c(u0) + c(u1) + c(u2), row 1 = c(u1) - c(u2) - c(u3).
    const float* __restrict__ p_bias = J.bias;
    const float* __restrict__ p_bbias = J.bbias;
    const float* __restrict__ p_rcol = J.rcol;
    const float* __restrict__ p_rrow = J.rrow;
    const float* __restrict__ p_res = J.res;
    float* __restrict__ p_out = J.out;
    double* p_gn = J.gn_part;
    __syncthreads();                                     // all patch reads and halo stores of the last step are done
    {
        float* img = smem + u * X_IMG + i;               // lane (i, half) holds output channel n32*32 + i of tiles ti(r)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
            const float p = m1 + m2, q = m1 - m2, rr = m3 + m4, s = m3 - m4;
            const int ti = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int pp = ((ti >> 2) * X_TW + 4 * (ti & 3)) * 32;
            img[pp] = (m0 + p) + rr; img[pp + 32] = fmaf(2.f, s, q); img[pp + 64] = fmaf(4.f, rr, p); img[pp + 96] = fmaf(8.f, s, q) + m5;
        }
    }
    // finishing thread: channels co4..co4+3 of pixel column xl, rows rsel*8 .. rsel*8+7 of the tile (two batches of four)
    const int quad = tid & 7, xl = (tid >> 3) & 15, rsel = tid >> 7;
    const int co4 = n32 * 32 + quad * 4;
    const bool c_ok = co4 < cout;
    const int coc = c_ok ? co4 : 0;
    const int x = tx0 + xl;
    const bool x_ok = x < w && c_ok;
    const int xc = x < w ? x : 0;
    f32x4 base4 = p_bias ? *reinterpret_cast<const f32x4*>(p_bias + coc) : zero4;
    if (p_bbias) base4 += *reinterpret_cast<const f32x4*>(p_bbias + size_t(b) * J.bbias_stride + coc);
    const bool interior_rows = ty0 > 0 && ty0 + X_TH < h;
    const int vx = x_edge_variant(xc, w);
    f32x4 tcol[4], trow[4], tres[4];
    auto fetch = [&](int batch) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { tcol[k] = zero4; trow[k] = zero4; tres[k] = zero4; }
        if (p_rcol) {
            if (interior_rows) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(p_rcol + ((size_t(b) * w + xc) * 4 + 0) * cout + coc);
#pragma unroll
                for (int k = 0; k < 4; ++k) tcol[k] = v0;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = ty0 + rsel * 8 + batch * 4 + k;
                    tcol[k] = *reinterpret_cast<const f32x4*>(p_rcol + ((size_t(b) * w + xc) * 4 + x_edge_variant(y < h ? y : 0, h)) * cout + coc);
                }
            }
        }
        if (p_rrow) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = ty0 + rsel * 8 + batch * 4 + k;
                trow[k] = *reinterpret_cast<const f32x4*>(p_rrow + ((size_t(b) * h + (y < h ? y : 0)) * 4 + vx) * cout + coc);
            }
        }
        if (p_res) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = ty0 + rsel * 8 + batch * 4 + k;
                tres[k] = *reinterpret_cast<const f32x4*>(p_res + ((size_t(b) * h + (y < h ? y : 0)) * w + xc) * cout + coc);
            }
        }
    };
    fetch(0);
    __syncthreads();                                     // the share images are complete
    f32x4 gs4 = zero4, gss4 = zero4;
#pragma unroll
    for (int batch = 0; batch < 2; ++batch) {
        if (batch == 1) fetch(1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yl = rsel * 8 + batch * 4 + k, y = ty0 + yl;
            const float* sp = smem + (yl & 1) * X_IMG + ((yl >> 1) * X_TW + xl) * 32 + quad * 4;
            const f32x4 ka = *reinterpret_cast<const f32x4*>(sp), kb = *reinterpret_cast<const f32x4*>(sp + X_IMG),
                        kc = *reinterpret_cast<const f32x4*>(sp + 2 * X_IMG);
            const f32x4 sum3 = (yl & 1) ? (ka - kb) - kc : (ka + kb) + kc;
            const f32x4 v = (sum3 + base4) + ((tcol[k] + trow[k]) + tres[k]);
            if (x_ok && y < h) {
                *reinterpret_cast<f32x4*>(p_out + ((size_t(b) * h + y) * w + x) * cout + co4) = v;
                gs4 += v; gss4 += v * v;
            }
        }
    }
    if (p_gn) {
        // per wave: 8 pixel columns (lanes l, l+8, ..) x 8 rows of 8 channel quads; one part per wave
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) { gs4[e] += __shfl_xor(gs4[e], off, 64); gss4[e] += __shfl_xor(gss4[e], off, 64); }
        const int sg = args.gn_sg;
        const int part = tile_idx * 4 + u;
        auto put = [&](int sub, float s, float ss) {            // partial layout [b][plane][sub][part][2]
            double* dst = p_gn + ((size_t(b) * 3 * args.gn_nsub + sub) * args.gn_maxparts + part) * 2;
            dst[0] = double(s); dst[1] = double(ss);
        };
        if (sg >= 4) {
            float s = (gs4[0] + gs4[1]) + (gs4[2] + gs4[3]), ss = (gss4[0] + gss4[1]) + (gss4[2] + gss4[3]);
            for (int off = 1; off < (sg >> 2); off <<= 1) { s += __shfl_xor(s, off, 64); ss += __shfl_xor(ss, off, 64); }
            if (lane < 8 && c_ok && (co4 % sg) == 0) put(co4 / sg, s, ss);
        } else if (lane < 8 && c_ok) {
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                if (sg == 2) put((co4 + e) / 2, gs4[e] + gs4[e + 1], gss4[e] + gss4[e + 1]);
                else { put(co4 + e, gs4[e], gss4[e]); put(co4 + e + 1, gs4[e + 1], gss4[e + 1]); }
            }
        }
    }
}

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
#define W24_STAMP(k) if (threadIdx.x == 0) g_w24time[size_t(blockIdx.x) * 8 + (k)] = wall_clock64();
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

__global__ __launch_bounds__(256, 3) void k_conv_wino24s(ConvArgs args, R1Inline r1) {
    __shared__ __attribute__((aligned(16))) float smem[2 * C_ABUF];             // 51.8 KB: two halo buffers; four share images after the loop
    static_assert(4 * C_IMG <= 2 * C_ABUF, "LDS plan");
    static_assert(kR1LdsFloats <= 2 * C_ABUF, "the rank-1 producer role fits the convolution's LDS");
    W24_STAMP(0)
    int bid = blockIdx.x;
    if (r1.nprod) {
        if (bid < r1.nprod) { r1_producer_role<true>(r1, bid, smem); return; }
        bid -= r1.nprod;
    }
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    if (args.xcd_swizzle & 1) {
        const int chunk = (int(gridDim.x) - r1.nprod) >> 3;
        if (bid < (chunk << 3)) bid = (bid & 7) * chunk + (bid >> 3);
    }
    int j = 0;
#pragma unroll
    for (int k = 1; k < kMaxConvJobs; ++k) j += (k < args.njobs && bid >= args.job[k].block_begin) ? 1 : 0;
    const ConvJob& J = args.job[j];
    int local = bid - J.block_begin;
    const int n32 = local % J.n_tiles_n; local /= J.n_tiles_n;
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
    // byte addresses of the lane's two patch rows for the first / second 16 channels of a chunk (logical quad kk*4 + g)
    const int qx = g ^ (((tr + (xrow >> 1)) & 3) << 1), qy = g ^ (((tr + (yrow >> 1)) & 3) << 1);
    const int bx = ((2 * tr + xrow) * C_HW + 4 * tc) * C_LD * 4, by = ((2 * tr + yrow) * C_HW + 4 * tc) * C_LD * 4;
    const int ax0 = bx + 16 * qx, ax1 = bx + 16 * (qx ^ 4), ay0 = by + 16 * qy, ay1 = by + 16 * (qy ^ 4);

    const int k16_total = cin / 16;
    const float* ub = J.wgt + ((size_t(n32) * k16_total) * 48 + u * 12) * 256;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, k16_total * 48 * 1024, 0x00020000);
    const int wlane = lane * 16;
    auto wfrag = [&](int step, int s) -> f32x4 {                            // s = 2 * frequency + cout block
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (step * 48 + s) * 1024, 0));
    };
    const float* inb = J.in + size_t(b) * h * w * cin;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inb), 0, h * w * cin * 4, 0x00020000);
    // (round 3) the first weight fragments leave before the halo addressing is worked out, and that addressing avoids the
    // quarter-rate 32-bit multiplies: the byte offset of a halo item is two 24-bit multiply-adds with wave-uniform strides (the
    // launcher guarantees a plane below 2 GiB, so row stride < 2^24 holds for every supported shape) and pix / 18 is a
    // multiply-shift, exact for the 180 pixels of a halo
    f32x4 ring[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) { ring[s] = wfrag(0, s); __builtin_amdgcn_sched_barrier(0); }
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

    f32x4 acc[6][2];
#pragma unroll
    for (int f = 0; f < 6; ++f)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[f][nb] = zero4;

    const int nchunks = cin / C_KC;
    f32x4 V[6];
    // chunk 0 into buffer 0, then the barrier; the first half of chunk 1 is requested with it but lands in buffer 1 only
    // after the first operands are built (the barrier of step (0,0) publishes it): it is off the prologue's critical path
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

    // VERDICT r2 item 4, priced before built (tools/wino24_ubench.hip -DW24_GNSILU_PRICE; profiles/r03_gn_in_halo_price.txt): what
    // GroupNorm-apply + FiLM + SiLU on the halo items between their load and their ds_write would cost the k-loop — two fmas, exp,
    // rcp, mul and the padding select per element, constants per channel quad fetched per chunk.  (Arithmetic stand-in: results
    // are meaningless in this build.)
#ifdef W24_GNSILU_PRICE
#define W24_PRICE_GNSILU                                                                                              \
    {                                                                                                                 \
        const f32x4 gA = *reinterpret_cast<const f32x4*>(J.wgt + ((lch * 32 + (tid & 7) * 4) % cout));                \
        const f32x4 gB = *reinterpret_cast<const f32x4*>(J.wgt + ((lch * 32 + (tid & 7) * 4 + 4) % cout));            \
        _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                               \
            const bool ok_ = goff[it0 + t] != 0x80000000u;                                                            \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                           \
                float v_ = fmaf(pf[t][e], gA[e], gB[e]);                                                              \
                v_ = fmaf(v_, gB[e], gA[e]);                                                                          \
                v_ = v_ * __builtin_amdgcn_rcpf(1.0f + __expf(-v_));                                                  \
                pf[t][e] = ok_ ? v_ : 0.f;                                                                            \
            }                                                                                                         \
        }                                                                                                             \
    }
#else
#define W24_PRICE_GNSILU
#endif
    // One k-step (16 channels) = 12 groups {four MFMAs on one A operand + a piece of the other work}, pinned.
#define C_HAS_NEXT 1                                  /* 0 in a tile's last k-step: no next step's weight fragments to request */
#define C_GROUP(F, NB, WORK)                                                                                          \
    {                                                                                                                 \
        constexpr int s_ = 2 * (F) + (NB);                                                                            \
        const f32x4 bq = ring[s_ % 6];                                                                                \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][0], bq[0], acc[F][NB], 0, 0, 0);                       \
        if (s_ + 6 < 12) ring[s_ % 6] = wfrag(step, s_ + 6); else if (C_HAS_NEXT) ring[s_ % 6] = wfrag(nstep, s_ - 6);     \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][1], bq[1], acc[F][NB], 0, 0, 0);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        WORK                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][2], bq[2], acc[F][NB], 0, 0, 0);                       \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][3], bq[3], acc[F][NB], 0, 0, 0);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
#define C_COMB(T, X, Y) { _Pragma("unroll") for (int e = 0; e < 4; ++e) T[e] = fmaf(sgn, Y[e], X[e]); C_PIN(T); }
    //   KK = 0: the next operands are the second 16 channels of the current buffer; halo items 3..5 of chunk c+1 go to the
    //           other buffer, then the chunk's barrier.   KK = 1: the next operands are the first 16 channels of the other
    //           buffer; halo items 0..2 of chunk c+2 go to the current buffer (whose last reads were before the barrier).
#define C_STEP(KK)                                                                                                    \
    {                                                                                                                 \
        const int step = chunk * 2 + (KK);                                                                            \
        const int nstep = step + 1 < k16_total ? step + 1 : step;                                                     \
        const int rx = ((KK) == 0 ? ax1 + cur : ax0 + (cur ^ tog)), ry = ((KK) == 0 ? ay1 + cur : ay0 + (cur ^ tog)); \
        constexpr int it0 = (KK) == 0 ? 3 : 0;                                                                        \
        const int lch = (KK) == 0 ? cn1 : cn2;                                                                        \
        f32x4 pf[3], cx0, cy0, cx1, cy1, t0, t1, t2, t3, t4, t5, s1, s2, s3, s4, v5n;                                 \
        C_GROUP(0, 0, cx0 = C_LDS4(rx); cy0 = C_LDS4(ry); cx1 = C_LDS4(rx + C_LD * 4); cy1 = C_LDS4(ry + C_LD * 4);)  \
        C_GROUP(0, 1, C_COMB(t0, cx0, cy0) C_COMB(t1, cx1, cy1)                                                       \
                      pf[0] = item_load(it0, lch); pf[1] = item_load(it0 + 1, lch); pf[2] = item_load(it0 + 2, lch);) \
        C_GROUP(1, 0, cx0 = C_LDS4(rx + 2 * C_LD * 4); cy0 = C_LDS4(ry + 2 * C_LD * 4); cx1 = C_LDS4(rx + 3 * C_LD * 4); cy1 = C_LDS4(ry + 3 * C_LD * 4);) \
        C_GROUP(1, 1, C_COMB(t2, cx0, cy0) C_COMB(t3, cx1, cy1))                                                      \
        C_GROUP(2, 0, cx0 = C_LDS4(rx + 4 * C_LD * 4); cy0 = C_LDS4(ry + 4 * C_LD * 4); cx1 = C_LDS4(rx + 5 * C_LD * 4); cy1 = C_LDS4(ry + 5 * C_LD * 4);) \
        C_GROUP(2, 1, C_COMB(t4, cx0, cy0) C_COMB(t5, cx1, cy1))                                                      \
        /* from here on V[0..2] are free: their MFMAs have been issued */                                             \
        C_GROUP(3, 0, s1 = t4 - 4.f * t2; C_PIN(s1); s2 = t3 - 4.f * t1; C_PIN(s2); V[1] = s1 + s2; C_PIN(V[1]); V[2] = s1 - s2; C_PIN(V[2]);) \
        C_GROUP(3, 1, V[0] = 4.f * t0 + (t4 - 5.f * t2); C_PIN(V[0]); s3 = t4 - t2; C_PIN(s3); s4 = t3 - t1; C_PIN(s4);) \
        C_GROUP(4, 0, V[3] = s3 + 2.f * s4; C_PIN(V[3]);)                                                             \
        C_GROUP(4, 1, v5n = 4.f * t1 + (t5 - 5.f * t3); C_PIN(v5n);)                                                  \
        C_GROUP(5, 0, V[4] = s3 - 2.f * s4; C_PIN(V[4]);)                                                             \
        C_GROUP(5, 1, ;)                                                                                              \
        V[5] = v5n;                                                                                                   \
        W24_PRICE_GNSILU                                                                                              \
        _Pragma("unroll") for (int t = 0; t < 3; ++t) item_store(it0 + t, (KK) == 0 ? (chunk + 1) & 1 : chunk & 1, pf[t]); \
        if ((KK) == 0) __syncthreads();                                                                               \
    }

    int cur = 0;                                  // byte offset of the buffer that holds the current chunk
    const int tog = C_ABUF * 4;
    // in-launch producers (s3d_rank1.h): this plane's tables are complete when its counter has reached the target; the first
    // poll is issued a chunk before the epilogue needs the answer (normally the only one: the producers finish well before)
    const unsigned* poll_ptr = r1.nprod ? r1.sync + (kSyncB + j) * kSyncStride : nullptr;
    const unsigned poll_target = r1.nprod ? r1.b_target[j] : 0u;
    unsigned poll_seen = poll_target;
    W24_STAMP(2)
    __builtin_amdgcn_s_setprio(0);
    for (int chunk = 0; chunk < nchunks - 1; ++chunk) {
        const int cn1 = chunk + 1;
        const int cn2 = chunk + 2 < nchunks ? chunk + 2 : nchunks - 1;
        C_STEP(0)
        C_STEP(1)
        cur ^= tog;
    }
    // The last chunk is peeled (round 3): it has no successor to fetch, and its second step has no operands to build — the
    // registers and issue slots that frees carry the EPILOGUE's residual request (the operand that comes from HBM / the MALL),
    // which used to go out only after the last MFMA and was waited for behind the share-image barrier; the rank-1 tables
    // (L2-hot, written by the launch before) are still requested after the loop: all of them early spills 32 registers.
    if (poll_ptr) poll_seen = sync_load(poll_ptr);
    {
        const int chunk = nchunks - 1;
        const int step = chunk * 2, nstep = step + 1;
        const int rx = ax1 + cur, ry = ay1 + cur;
        f32x4 cx0, cy0, cx1, cy1, t0, t1, t2, t3, t4, t5, s1, s2, s3, s4, v5n;
        C_GROUP(0, 0, cx0 = C_LDS4(rx); cy0 = C_LDS4(ry); cx1 = C_LDS4(rx + C_LD * 4); cy1 = C_LDS4(ry + C_LD * 4);)
        C_GROUP(0, 1, C_COMB(t0, cx0, cy0) C_COMB(t1, cx1, cy1))
        C_GROUP(1, 0, cx0 = C_LDS4(rx + 2 * C_LD * 4); cy0 = C_LDS4(ry + 2 * C_LD * 4); cx1 = C_LDS4(rx + 3 * C_LD * 4); cy1 = C_LDS4(ry + 3 * C_LD * 4);)
        C_GROUP(1, 1, C_COMB(t2, cx0, cy0) C_COMB(t3, cx1, cy1))
        C_GROUP(2, 0, cx0 = C_LDS4(rx + 4 * C_LD * 4); cy0 = C_LDS4(ry + 4 * C_LD * 4); cx1 = C_LDS4(rx + 5 * C_LD * 4); cy1 = C_LDS4(ry + 5 * C_LD * 4);)
        C_GROUP(2, 1, C_COMB(t4, cx0, cy0) C_COMB(t5, cx1, cy1))
        C_GROUP(3, 0, s1 = t4 - 4.f * t2; C_PIN(s1); s2 = t3 - 4.f * t1; C_PIN(s2); V[1] = s1 + s2; C_PIN(V[1]); V[2] = s1 - s2; C_PIN(V[2]);)
        C_GROUP(3, 1, V[0] = 4.f * t0 + (t4 - 5.f * t2); C_PIN(V[0]); s3 = t4 - t2; C_PIN(s3); s4 = t3 - t1; C_PIN(s4);)
        C_GROUP(4, 0, V[3] = s3 + 2.f * s4; C_PIN(V[3]);)
        C_GROUP(4, 1, v5n = 4.f * t1 + (t5 - 5.f * t3); C_PIN(v5n);)
        C_GROUP(5, 0, V[4] = s3 - 2.f * s4; C_PIN(V[4]);)
        C_GROUP(5, 1, ;)
        V[5] = v5n;
    }
    // ---- epilogue: as k_conv_wino24; lane (g, t16) holds output channel nb*16 + t16 of the tiles (tile row g, tile column r)
    const float* __restrict__ p_bias = J.bias;
    const float* __restrict__ p_bbias = J.bbias;
    const float* __restrict__ p_rcol = J.rcol;
    const float* __restrict__ p_rrow = J.rrow;
    const float* __restrict__ p_res = J.res;
    float* __restrict__ p_out = J.out;
    double* p_gn = J.gn_part;
    // the finishing threads' operands (bias, rank-1 tables, residual) are requested BEFORE the barrier and the share-image
    // writes: their L2/HBM latency runs beside both (2.3 -> 1.6 us for this phase)
    const int quad = tid & 7, xl = (tid >> 3) & 15, rsel = tid >> 7;
    const int co4 = n32 * 32 + quad * 4;
    const bool c_ok = co4 < cout;
    const int coc = c_ok ? co4 : 0;
    const int x = tx0 + xl;
    const bool x_ok = x < w && c_ok;
    const int xc = x < w ? x : 0;
    f32x4 base4 = p_bias ? *reinterpret_cast<const f32x4*>(p_bias + coc) : zero4;
    if (p_bbias) base4 += *reinterpret_cast<const f32x4*>(p_bbias + size_t(b) * J.bbias_stride + coc);
    f32x4 tcol[4], trow[4], tres[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { tcol[k] = zero4; trow[k] = zero4; tres[k] = zero4; }
    if (poll_ptr && !sync_wait(poll_ptr, poll_target, poll_seen))
        __hip_atomic_store(r1.sync + kSyncErr * kSyncStride, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the tables may have been written earlier in this launch by blocks on other XCDs (write-through stores): sc1 loads.
    // args.r1_slices == 2: each table is the sum of two K slices (s3d_rank1.h) — both requested now, added behind the barrier
    const bool two = args.r1_slices == 2;
    f32x4 tcol2[4], trow2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { tcol2[k] = zero4; trow2[k] = zero4; }
    auto tload = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, kAuxSc1)); };
    auto request_col_tables = [&]() {
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
                for (int k = 0; k < 4; ++k) { tcol[k] = v0; tcol2[k] = v1; }
            } else {
    #pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = ty0 + rsel * 4 + k;
                    const unsigned off = unsigned(((xc * 4 + x_edge_variant(y < h ? y : 0, h)) * cout + coc) * 4);
                    tcol[k] = tload(trs, off);
                    if (two) tcol2[k] = tload(trs2, off);
                }
            }
        }
    };
    auto request_row_tables = [&]() {
        if (p_rrow) {
            const float* base = p_rrow + size_t(b) * h * 4 * cout;
            const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, h * 4 * cout * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t trs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (two ? size_t(args.B) * h * 4 * cout : 0)), 0, h * 4 * cout * 4, 0x00020000);
            const int vx = x_edge_variant(xc, w);
    #pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = ty0 + rsel * 4 + k;
                const unsigned off = unsigned((((y < h ? y : 0) * 4 + vx) * cout + coc) * 4);
                trow[k] = tload(trs, off);
                if (two) trow2[k] = tload(trs2, off);
            }
        }
    };
    auto request_residual = [&]() {
        if (p_res) {
    #pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = ty0 + rsel * 4 + k;
                tres[k] = *reinterpret_cast<const f32x4*>(p_res + ((size_t(b) * h + (y < h ? y : 0)) * w + xc) * cout + coc);
            }
        }
    };
#undef C_HAS_NEXT
#define C_HAS_NEXT 0
    {   // the tile's last k-step: MFMAs only; the operand requests go out behind its first groups
        const int step = (nchunks - 1) * 2 + 1, nstep = step; (void)nstep;
        C_GROUP(0, 0, ;) C_GROUP(0, 1, request_residual();) C_GROUP(1, 0, ;) C_GROUP(1, 1, ;) C_GROUP(2, 0, ;) C_GROUP(2, 1, ;)
        C_GROUP(3, 0, ;) C_GROUP(3, 1, ;) C_GROUP(4, 0, ;) C_GROUP(4, 1, ;) C_GROUP(5, 0, ;) C_GROUP(5, 1, ;)
    }
#undef C_HAS_NEXT
#undef C_STEP
#undef C_GROUP
#undef C_COMB
#undef C_LDS4
#undef C_PIN
#undef W24_PRICE_GNSILU
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    W24_STAMP(3)
    request_row_tables();                                // (requesting the row tables a k-step early as well measured the same)
    request_col_tables();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                     // all patch reads and halo stores of the last step are done
    {
        float* img = smem + u * C_IMG + t16;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r], m4 = acc[4][nb][r], m5 = acc[5][nb][r];
                const float p = m1 + m2, q = m1 - m2, rr = m3 + m4, s = m3 - m4;
                const int pp = (g * C_TW + 4 * r) * 32 + nb * 16;
                img[pp] = (m0 + p) + rr; img[pp + 32] = fmaf(2.f, s, q); img[pp + 64] = fmaf(4.f, rr, p); img[pp + 96] = fmaf(8.f, s, q) + m5;
            }
    }
    __syncthreads();                                     // the share images are complete
    W24_STAMP(4)
    f32x4 gs4 = zero4, gss4 = zero4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yl = rsel * 4 + k, y = ty0 + yl;
        const float* sp = smem + (yl & 1) * C_IMG + ((yl >> 1) * C_TW + xl) * 32 + quad * 4;
        const f32x4 ka = *reinterpret_cast<const f32x4*>(sp), kb = *reinterpret_cast<const f32x4*>(sp + C_IMG),
                    kc = *reinterpret_cast<const f32x4*>(sp + 2 * C_IMG);
        const f32x4 sum3 = (yl & 1) ? (ka - kb) - kc : (ka + kb) + kc;
        const f32x4 v = (sum3 + base4) + (((tcol[k] + tcol2[k]) + (trow[k] + trow2[k])) + tres[k]);     // (x + 0 = x: one-slice tables add exact zeros)
        if (x_ok && y < h) {
            *reinterpret_cast<f32x4*>(p_out + ((size_t(b) * h + y) * w + x) * cout + co4) = v;
            gs4 += v; gss4 += v * v;
        }
    }
    if (p_gn) {
        // one partial per BLOCK: the four waves' sums meet through LDS (wave order, double) — a quarter of the records for
        // the statistics' readers (the GN-act kernel adds them itself, s3d_kernels.hip:k_gn_act)
        __shared__ float gred[4][8][8];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) { gs4[e] += __shfl_xor(gs4[e], off, 64); gss4[e] += __shfl_xor(gss4[e], off, 64); }
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { gred[u][lane][e] = gs4[e]; gred[u][lane][4 + e] = gss4[e]; }
        }
        __syncthreads();
        if (tid < 64) {                                   // lanes 0..7 of wave 0: channel quad `quad` = lane
            double ds[4], dss[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int l8 = lane & 7;
                ds[e] = ((double(gred[0][l8][e]) + double(gred[1][l8][e])) + double(gred[2][l8][e])) + double(gred[3][l8][e]);
                dss[e] = ((double(gred[0][l8][4 + e]) + double(gred[1][l8][4 + e])) + double(gred[2][l8][4 + e])) + double(gred[3][l8][4 + e]);
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
                if (lane < 8 && c_ok && (co4 % sg) == 0) put(co4 / sg, sv, ssv);
            } else if (lane < 8 && c_ok) {
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

// ------------------------------------------------------------------ the persistent form: a block owns SEVERAL tiles
// k_conv_wino24s pays ~4 us before a tile's first MFMA (halo chunk 0: HBM/L2 -> registers -> LDS -> barrier -> operands) and
// ~4 us after its last (operand loads, share images, finishing stores), and all co-resident blocks pay them together
// (profiles/r02_wino_ubench.txt): 8 of every ~27 us at K = 128.  Here a launch has at most one block per slot (three per CU) and
// block i walks tiles i, i + G, i + 2G, ... (G = grid size, a multiple of 8, so a block's tiles stay on its XCD's contiguous
// range).  Same arithmetic in the same order as k_conv_wino24s — results are bit-identical — but across a tile boundary:
//   * the NEXT tile's halo chunk 0 and first six weight fragments are requested during the LAST chunk of the current tile
//     (that chunk has no successor of its own to fetch) and wait in registers; its second chunk's first half is requested
//     with the epilogue operands;
//   * the epilogue operands (rank-1 tables, residual) are requested before the share images are written, as before;
//   * after the share images are consumed the parked halo goes to LDS and the next k-loop starts one barrier later; the
//     finishing stores drain beside it.
// What a boundary still exposes is LDS traffic and four barriers (~2 us) instead of two HBM round trips — per tile the phases
// add up to 29.6 us against 32.9 us for k_conv_wino24s (batch 8, K = 128) — and yet the launch is SLOWER (601 vs 547 us), so the
// form is off by default (S3D_WINO24_PERSIST=1 enables it; profiles/r03_wino_persistent.txt).  Two reasons, both structural:
//   * on gfx9 stores share the in-order vmcnt counter with loads: whatever waits for a load issued after a tile's output
//     stores also waits for their write acknowledgements.  A block of k_conv_wino24s ends behind its stores and the hardware
//     starts the next block in its slot at once; a persistent block meets them at its next vmcnt wait (a scratch reload of a
//     spilled value right behind the stores cost ~7 us per tile in the first version; with the stores moved last, the next
//     k-loop's first weight wait is ~1.3 us behind them).  Hiding that needs ~40 more live registers across the boundary
//     (parked halo, weight ring, second-chunk halo, finished outputs) than the 168 that three blocks per CU allow: every variant
//     built spilled 22-68 registers around the boundary, and each reload is a vmcnt(0);
//   * all blocks start together and own tiles of equal length, so every boundary — and its burst of operand loads and output
//     stores — hits every CU at the same moment, tile after tile; hardware-dispatched blocks drift apart and keep the k-loops
//     of their CU's neighbours running.  A deliberate one-third-tile start stagger of a CU's three blocks did not recover it (measured, removed).
#ifndef W24P_EARLY_RING
#define W24P_EARLY_RING 0          // 1: the next tile's first weight fragments + second-chunk halo are requested right after the share images are written (more live registers in the epilogue)
#endif
__global__ __launch_bounds__(256, 3) void k_conv_wino24p(ConvArgs args, int total_tiles) {
    __shared__ __attribute__((aligned(16))) float smem[2 * C_ABUF];
    static_assert(4 * C_IMG <= 2 * C_ABUF, "LDS plan");
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wlane = lane * 16;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);                 // row frequency of this wave
    const int t16 = lane & 15, g = lane >> 4;                               // tile of the lane, channel quad of the lane
    const int tr = t16 >> 2, tc = t16 & 3;
    const int xrow = u == 0 ? 0 : (u == 2 ? 2 : 1), yrow = u == 2 ? 1 : (u == 3 ? 3 : 2);
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(u == 1 ? 0x3F800000 : 0xBF800000));
    const int qx = g ^ (((tr + (xrow >> 1)) & 3) << 1), qy = g ^ (((tr + (yrow >> 1)) & 3) << 1);
    const int bx = ((2 * tr + xrow) * C_HW + 4 * tc) * C_LD * 4, by = ((2 * tr + yrow) * C_HW + 4 * tc) * C_LD * 4;
    const int ax0 = bx + 16 * qx, ax1 = bx + 16 * (qx ^ 4), ay0 = by + 16 * qy, ay1 = by + 16 * (qy ^ 4);
    const int cin = args.cin, cout = args.cout;
    const int k16_total = cin / 16, nchunks = cin / C_KC;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // tile-independent halo staging geometry: item = it*256 + tid -> (pixel item>>3, channel quad item&7)
    auto lds_off = [&](int it, int t) {
        const int item = it * 256 + t;
        const int pix = item >> 3, q = item & 7;
        const int hy = pix / C_HW;
        return pix * C_LD + ((q ^ (((hy >> 1) & 3) << 1)) << 2);
    };
    int loff[C_ITEMS_PT - 1];
#pragma unroll
    for (int it = 0; it < C_ITEMS_PT - 1; ++it) loff[it] = lds_off(it, tid);
    const bool last_ok = (C_ITEMS_PT - 1) * 256 + tid < C_ITEMS;
    auto item_store = [&](int it, int buf, f32x4 v) {
        if (it < C_ITEMS_PT - 1) *reinterpret_cast<f32x4*>(smem + buf * C_ABUF + loff[it]) = v;
        else if (last_ok) {            // the sixth round covers 160 items: its offset is recomputed (one register less across the k-loop; kept, it lived in scratch)
            int wl = wlane;                         // = 16 * lane, live across the k-loop anyway (the weight fragments' offset)
            asm volatile("" : "+v"(wl));
            *reinterpret_cast<f32x4*>(smem + buf * C_ABUF + lds_off(C_ITEMS_PT - 1, (wl >> 4) + u * 64)) = v;
        }
    };

    // per-tile context (wave-uniform except goff)
    struct Ctx {
        int j, n32, b, tile_idx, ty0, tx0, h, w;
        __amdgpu_buffer_rsrc_t rs, wrs;
        unsigned goff[C_ITEMS_PT];
    };
    const int G = int(gridDim.x);
    const int xchunk = total_tiles >> 3;
    auto setup = [&](int phys, Ctx& c) {
        int bid = phys;
        if ((args.xcd_swizzle & 1) && bid < (xchunk << 3)) bid = (bid & 7) * xchunk + (bid >> 3);
        int j = 0;
#pragma unroll
        for (int k = 1; k < kMaxConvJobs; ++k) j += (k < args.njobs && bid >= args.job[k].block_begin) ? 1 : 0;
        const ConvJob& J = args.job[j];
        int local = bid - J.block_begin;
        c.j = j;
        c.n32 = local % J.n_tiles_n; local /= J.n_tiles_n;
        c.b = local / J.tiles_per_img; local %= J.tiles_per_img;
        c.tile_idx = local;
        c.ty0 = (local / J.tiles_x) * C_TH; c.tx0 = (local % J.tiles_x) * C_TW;
        c.h = J.h; c.w = J.w;
        c.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(J.in + size_t(c.b) * c.h * c.w * cin), 0, c.h * c.w * cin * 4, 0x00020000);
        c.wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(J.wgt + ((size_t(c.n32) * k16_total) * 48 + u * 12) * 256), 0, k16_total * 48 * 1024, 0x00020000);
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));                  // recompute the halo geometry per tile: hoisted out of the tile loop it only lives in scratch
#pragma unroll
        for (int it = 0; it < C_ITEMS_PT; ++it) {
            const int item = it * 256 + tid_;
            const int pix = item >> 3, q = item & 7;
            const int hy = pix / C_HW, hx = pix - hy * C_HW;
            const int gy = c.ty0 - 1 + hy, gx = c.tx0 - 1 + hx;
            const bool ok = item < C_ITEMS && gy >= 0 && gy < c.h && gx >= 0 && gx < c.w;
            c.goff[it] = ok ? unsigned((gy * c.w + gx) * cin + q * 4) * 4u : 0x80000000u;
        }
    };
    auto wfrag = [&](const __amdgpu_buffer_rsrc_t& rs, int step, int s) -> f32x4 {          // s = 2 * frequency + cout block
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, wlane, (step * 48 + s) * 1024, 0));
    };
    auto item_load = [&](const Ctx& c, int it, int ch) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c.rs, c.goff[it], ch * (C_KC * 4), 0));
    };

    Ctx cx, nx;
    int phys = blockIdx.x;
    setup(phys, cx);
    f32x4 V[6], ring[6], hN[C_ITEMS_PT], pre[3];
#pragma unroll
    for (int s = 0; s < 6; ++s) { ring[s] = wfrag(cx.wrs, 0, s); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int it = 0; it < C_ITEMS_PT; ++it) hN[it] = item_load(cx, it, 0);
#pragma unroll
    for (int it = 0; it < 3; ++it) pre[it] = item_load(cx, it, nchunks > 1 ? 1 : 0);

#define C_LDS4(off) (*static_cast<const f32x4*>(__builtin_assume_aligned(reinterpret_cast<const char*>(smem) + (off), 16)))
#define C_PIN(v) asm volatile("" : "+v"(v))
#define C_COMB(T, X, Y) { _Pragma("unroll") for (int e = 0; e < 4; ++e) T[e] = fmaf(sgn, Y[e], X[e]); C_PIN(T); }
    // WF(s_): the fragment that refills ring slot s_ % 6 — this step's fragment s_ + 6, or the NEXT step's fragment s_ - 6 (the
    // next step is the next k-step of this tile, or step 0 of the next tile in the last step of a tile)
#define P_GROUP(F, NB, WORK)                                                                                          \
    {                                                                                                                 \
        constexpr int s_ = 2 * (F) + (NB);                                                                            \
        const f32x4 bq = ring[s_ % 6];                                                                                \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][0], bq[0], acc[F][NB], 0, 0, 0);                       \
        if (s_ + 6 < 12) ring[s_ % 6] = wfrag(cx.wrs, step, s_ + 6); else if (P_NEXTSTEP) ring[s_ % 6] = wfrag(cx.wrs, nstep, s_ - 6);                       \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][1], bq[1], acc[F][NB], 0, 0, 0);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        WORK                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][2], bq[2], acc[F][NB], 0, 0, 0);                       \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][3], bq[3], acc[F][NB], 0, 0, 0);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
    // a k-step that also builds the next step's operands from the patch at byte addresses (rx, ry); LOADS: three halo requests
#define P_STEP_BUILD(LOADS)                                                                                           \
    {                                                                                                                 \
        f32x4 cx0, cy0, cx1, cy1, t0, t1, t2, t3, t4, t5, s1, s2, s3, s4, v5n;                                        \
        P_GROUP(0, 0, cx0 = C_LDS4(rx); cy0 = C_LDS4(ry); cx1 = C_LDS4(rx + C_LD * 4); cy1 = C_LDS4(ry + C_LD * 4);)  \
        P_GROUP(0, 1, C_COMB(t0, cx0, cy0) C_COMB(t1, cx1, cy1) LOADS)                                                \
        P_GROUP(1, 0, cx0 = C_LDS4(rx + 2 * C_LD * 4); cy0 = C_LDS4(ry + 2 * C_LD * 4); cx1 = C_LDS4(rx + 3 * C_LD * 4); cy1 = C_LDS4(ry + 3 * C_LD * 4);) \
        P_GROUP(1, 1, C_COMB(t2, cx0, cy0) C_COMB(t3, cx1, cy1))                                                      \
        P_GROUP(2, 0, cx0 = C_LDS4(rx + 4 * C_LD * 4); cy0 = C_LDS4(ry + 4 * C_LD * 4); cx1 = C_LDS4(rx + 5 * C_LD * 4); cy1 = C_LDS4(ry + 5 * C_LD * 4);) \
        P_GROUP(2, 1, C_COMB(t4, cx0, cy0) C_COMB(t5, cx1, cy1))                                                      \
        P_GROUP(3, 0, s1 = t4 - 4.f * t2; C_PIN(s1); s2 = t3 - 4.f * t1; C_PIN(s2); V[1] = s1 + s2; C_PIN(V[1]); V[2] = s1 - s2; C_PIN(V[2]);) \
        P_GROUP(3, 1, V[0] = 4.f * t0 + (t4 - 5.f * t2); C_PIN(V[0]); s3 = t4 - t2; C_PIN(s3); s4 = t3 - t1; C_PIN(s4);) \
        P_GROUP(4, 0, V[3] = s3 + 2.f * s4; C_PIN(V[3]);)                                                             \
        P_GROUP(4, 1, v5n = 4.f * t1 + (t5 - 5.f * t3); C_PIN(v5n);)                                                  \
        P_GROUP(5, 0, V[4] = s3 - 2.f * s4; C_PIN(V[4]);)                                                             \
        P_GROUP(5, 1, ;)                                                                                              \
        V[5] = v5n;                                                                                                   \
    }
    // the last k-step of a tile: nothing to build
#define P_STEP_PLAIN(LOADS)                                                                                           \
    {                                                                                                                 \
        P_GROUP(0, 0, ;) P_GROUP(0, 1, LOADS) P_GROUP(1, 0, ;) P_GROUP(1, 1, ;) P_GROUP(2, 0, ;) P_GROUP(2, 1, ;)     \
        P_GROUP(3, 0, ;) P_GROUP(3, 1, ;) P_GROUP(4, 0, ;) P_GROUP(4, 1, ;) P_GROUP(5, 0, ;) P_GROUP(5, 1, ;)         \
    }

    const int tog = C_ABUF * 4;
#ifdef W24_TIMING
#define P_STAMP(k) if (threadIdx.x == 0) g_w24time[size_t(phys) * 8 + (k)] = wall_clock64();
#else
#define P_STAMP(k)
#endif
    for (;;) {
        const bool have_next = phys + G < total_tiles;
        P_STAMP(0)
        // ---- install the tile: its halo chunk 0 waits in hN, its first six weight fragments and the first half of chunk 1 were
        // requested during the previous tile's epilogue (chunk 1 lands in buffer 1 after the first operands are built; the
        // barrier of step (0,0) publishes it)
#pragma unroll
        for (int it = 0; it < C_ITEMS_PT; ++it) item_store(it, 0, hN[it]);
        __syncthreads();
        P_STAMP(1)
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

        f32x4 acc[6][2];
#pragma unroll
        for (int f = 0; f < 6; ++f)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[f][nb] = zero4;

        int cur = 0;                                  // byte offset of the buffer that holds the current chunk
        P_STAMP(2)
        __builtin_amdgcn_s_setprio(0);
#define P_NEXTSTEP 1
        for (int chunk = 0; chunk < nchunks - 1; ++chunk) {
            const int cn1 = chunk + 1;
            const int cn2 = chunk + 2 < nchunks ? chunk + 2 : nchunks - 1;
            f32x4 pf[3];
            {   // (chunk, 0): next operands = second 16 channels of the current buffer; halo items 3..5 of chunk c+1 -> other buffer; barrier
                const int step = chunk * 2, nstep = step + 1;
                const int rx = ax1 + cur, ry = ay1 + cur;
                P_STEP_BUILD(pf[0] = item_load(cx, 3, cn1); pf[1] = item_load(cx, 4, cn1); pf[2] = item_load(cx, 5, cn1);)
#pragma unroll
                for (int t = 0; t < 3; ++t) item_store(3 + t, (chunk + 1) & 1, pf[t]);
                __syncthreads();
            }
            {   // (chunk, 1): next operands = first 16 channels of the other buffer; halo items 0..2 of chunk c+2 -> current buffer
                const int step = chunk * 2 + 1, nstep = step + 1;
                const int rx = ax0 + (cur ^ tog), ry = ay0 + (cur ^ tog);
                P_STEP_BUILD(pf[0] = item_load(cx, 0, cn2); pf[1] = item_load(cx, 1, cn2); pf[2] = item_load(cx, 2, cn2);)
#pragma unroll
                for (int t = 0; t < 3; ++t) item_store(t, chunk & 1, pf[t]);
            }
            cur ^= tog;
        }
        // the last chunk has no successor in this tile: its two steps request the NEXT tile's chunk 0 (parked in hN)
        if (have_next) setup(phys + G, nx);
        else {
            nx.rs = cx.rs; nx.wrs = cx.wrs;
#pragma unroll
            for (int it = 0; it < C_ITEMS_PT; ++it) nx.goff[it] = 0x80000000u;     // out of range: the loads return zeros, no traffic
        }
        {
            const int step = (nchunks - 1) * 2, nstep = step + 1;
            const int rx = ax1 + cur, ry = ay1 + cur;
            P_STEP_BUILD(hN[0] = item_load(nx, 0, 0); hN[1] = item_load(nx, 1, 0); hN[2] = item_load(nx, 2, 0);)
        }
        {
            const int step = (nchunks - 1) * 2 + 1, nstep = 0;
#undef P_NEXTSTEP
#define P_NEXTSTEP 0                                  /* (the next tile's first fragments are requested when it is installed) */
            P_STEP_PLAIN(hN[3] = item_load(nx, 3, 0); hN[4] = item_load(nx, 4, 0); hN[5] = item_load(nx, 5, 0);)
        }
        if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
        P_STAMP(3)

        // ---- epilogue (as k_conv_wino24s): operands requested before the barrier and the share-image writes
        const ConvJob& J = args.job[cx.j];
        const float* __restrict__ p_bias = J.bias;
        const float* __restrict__ p_bbias = J.bbias;
        const float* __restrict__ p_rcol = J.rcol;
        const float* __restrict__ p_rrow = J.rrow;
        const float* __restrict__ p_res = J.res;
        float* __restrict__ p_out = J.out;
        double* p_gn = J.gn_part;
        const int h = cx.h, w = cx.w, b = cx.b, ty0 = cx.ty0, tx0 = cx.tx0;
        // the thread's epilogue geometry is re-derived per tile from a value that lives across the k-loop anyway (hoisted out of
        // the tile loop it would only live in scratch, and a scratch reload is a vmcnt(0) wait)
        int te = (wlane >> 4) + u * 64;
        asm volatile("" : "+v"(te));
        const int quad = te & 7, xl = (te >> 3) & 15, rsel = te >> 7;
        const int lane_e = te & 63, t16e = lane_e & 15, ge = lane_e >> 4;
        const int co4 = cx.n32 * 32 + quad * 4;
        const bool c_ok = co4 < cout;
        const int coc = c_ok ? co4 : 0;
        const int x = tx0 + xl;
        const bool x_ok = x < w && c_ok;
        const int xc = x < w ? x : 0;
        f32x4 base4 = p_bias ? *reinterpret_cast<const f32x4*>(p_bias + coc) : zero4;
        if (p_bbias) base4 += *reinterpret_cast<const f32x4*>(p_bbias + size_t(b) * J.bbias_stride + coc);
        f32x4 tcol[4], trow[4], tres[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { tcol[k] = zero4; trow[k] = zero4; tres[k] = zero4; }
        if (p_rcol) {
            if (ty0 > 0 && ty0 + C_TH < h) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(p_rcol + ((size_t(b) * w + xc) * 4 + 0) * cout + coc);
#pragma unroll
                for (int k = 0; k < 4; ++k) tcol[k] = v0;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = ty0 + rsel * 4 + k;
                    tcol[k] = *reinterpret_cast<const f32x4*>(p_rcol + ((size_t(b) * w + xc) * 4 + x_edge_variant(y < h ? y : 0, h)) * cout + coc);
                }
            }
        }
        if (p_rrow) {
            const int vx = x_edge_variant(xc, w);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = ty0 + rsel * 4 + k;
                trow[k] = *reinterpret_cast<const f32x4*>(p_rrow + ((size_t(b) * h + (y < h ? y : 0)) * 4 + vx) * cout + coc);
            }
        }
        if (p_res) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = ty0 + rsel * 4 + k;
                tres[k] = *reinterpret_cast<const f32x4*>(p_res + ((size_t(b) * h + (y < h ? y : 0)) * w + xc) * cout + coc);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                     // all patch reads of the last step are done
        {
            float* img = smem + u * C_IMG + t16e;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r], m4 = acc[4][nb][r], m5 = acc[5][nb][r];
                    const float p = m1 + m2, q = m1 - m2, rr = m3 + m4, s = m3 - m4;
                    const int pp = (ge * C_TW + 4 * r) * 32 + nb * 16;
                    img[pp] = (m0 + p) + rr; img[pp + 32] = fmaf(2.f, s, q); img[pp + 64] = fmaf(4.f, rr, p); img[pp + 96] = fmaf(8.f, s, q) + m5;
                }
        }
#if W24P_EARLY_RING
        if (have_next) {                                     // the accumulators are free: the next tile's first requests go out now
            const int c1 = nchunks > 1 ? 1 : 0;
#pragma unroll
            for (int s = 0; s < 6; ++s) { ring[s] = wfrag(nx.wrs, 0, s); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int it = 0; it < 3; ++it) pre[it] = item_load(nx, it, c1);
        }
#endif
        __syncthreads();                                     // the share images are complete
        P_STAMP(4)
        f32x4 gs4 = zero4, gss4 = zero4, vout[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yl = rsel * 4 + k;
            const float* sp = smem + (yl & 1) * C_IMG + ((yl >> 1) * C_TW + xl) * 32 + quad * 4;
            const f32x4 ka = *reinterpret_cast<const f32x4*>(sp), kb = *reinterpret_cast<const f32x4*>(sp + C_IMG),
                        kc = *reinterpret_cast<const f32x4*>(sp + 2 * C_IMG);
            const f32x4 sum3 = (yl & 1) ? (ka - kb) - kc : (ka + kb) + kc;
            vout[k] = (sum3 + base4) + ((tcol[k] + trow[k]) + tres[k]);
            if (x_ok && ty0 + yl < h) { gs4 += vout[k]; gss4 += vout[k] * vout[k]; }
        }
        __shared__ float gred[4][8][8];
        if (p_gn) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) { gs4[e] += __shfl_xor(gs4[e], off, 64); gss4[e] += __shfl_xor(gss4[e], off, 64); }
            if (lane_e < 8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { gred[u][lane_e][e] = gs4[e]; gred[u][lane_e][4 + e] = gss4[e]; }
            }
        }
        __syncthreads();                                     // the share images are consumed (and the waves' GroupNorm sums have met)
        // (the output stores come LAST, below: on gfx9 stores share the in-order vmcnt counter with loads, so anything that waits
        //  for a load issued after them also waits for their write acknowledgements — microseconds under load)
        float* const out_base = p_out + ((size_t(b) * h + ty0 + rsel * 4) * w + x) * cout + co4;
        const int out_rows = x_ok ? min(4, h - (ty0 + rsel * 4)) : 0;
        const size_t out_pitch = size_t(w) * cout;
        if (p_gn && te < 64) {                                // lanes 0..7 of wave 0: channel quad `quad` = lane
            const int lane = lane_e;
            double ds[4], dss[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int l8 = lane & 7;
                ds[e] = ((double(gred[0][l8][e]) + double(gred[1][l8][e])) + double(gred[2][l8][e])) + double(gred[3][l8][e]);
                dss[e] = ((double(gred[0][l8][4 + e]) + double(gred[1][l8][4 + e])) + double(gred[2][l8][4 + e])) + double(gred[3][l8][4 + e]);
            }
            const int sg = args.gn_sg;
            const int part = cx.tile_idx;
            auto put = [&](int sub, double sv, double ssv) {
                double* dst = p_gn + ((size_t(b) * 3 * args.gn_nsub + sub) * args.gn_maxparts + part) * 2;
                dst[0] = sv; dst[1] = ssv;
            };
            if (sg >= 4) {
                double sv = (ds[0] + ds[1]) + (ds[2] + ds[3]), ssv = (dss[0] + dss[1]) + (dss[2] + dss[3]);
                for (int off = 1; off < (sg >> 2); off <<= 1) { sv += __shfl_xor(sv, off, 64); ssv += __shfl_xor(ssv, off, 64); }
                if (lane < 8 && c_ok && (co4 % sg) == 0) put(co4 / sg, sv, ssv);
            } else if (lane < 8 && c_ok) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    if (sg == 2) put((co4 + e) / 2, ds[e] + ds[e + 1], dss[e] + dss[e + 1]);
                    else { put(co4 + e, ds[e], dss[e]); put(co4 + e + 1, ds[e + 1], dss[e + 1]); }
                }
            }
        }
        auto store_out = [&]() {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < out_rows) *reinterpret_cast<f32x4*>(out_base + k * out_pitch) = vout[k];
        };
        if (!have_next) { store_out(); P_STAMP(5) break; }
        phys += G;
        setup(phys, cx);                                     // (recomputed rather than carried: fewer live registers in the k-loop)
#if !W24P_EARLY_RING
        {
            const int c1 = nchunks > 1 ? 1 : 0;
#pragma unroll
            for (int s = 0; s < 6; ++s) { ring[s] = wfrag(cx.wrs, 0, s); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int it = 0; it < 3; ++it) pre[it] = item_load(cx, it, c1);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        store_out();
        __builtin_amdgcn_sched_barrier(0);
        phys -= G; P_STAMP(5) phys += G;

    }
#undef P_NEXTSTEP
#undef P_STEP_PLAIN
#undef P_STEP_BUILD
#undef P_GROUP
#undef C_COMB
#undef C_LDS4
#undef C_PIN
}

// ------------------------------------------------------------------ host side
void wino24_gn_parts(const Geo& g, int nparts[3]) {      // one part per wave of a tile's block
    for (int p = 0; p < 3; ++p) nparts[p] = ((g.w[p] + X_TW - 1) / X_TW) * ((g.h[p] + X_TH - 1) / X_TH) * 4;
}

static const double kG2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
static const double kG4[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};

size_t wino24_packed_floats(int cout, int cin) { return size_t((cout + 31) / 32) * (cin / 8) * 24 * 256; }

// U = G2 g G4^T in double, stored in MFMA fragment order [n32][k8][24][64 lanes][4]; W is OIHW [cout][ctot][3][3],
// only input channels [0, cin) are used (the plane's own channels).
size_t pack_wino24_weights(std::vector<float>& stage, const float* W, int cout, int ctot, int cin) {
    const int k8t = cin / 8;
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
            const int nt = co >> 5, jn = co & 31, k8 = c >> 3, hf = (c >> 2) & 1, e = c & 3;
            for (int u = 0; u < 4; ++u)
                for (int v = 0; v < 6; ++v) {
                    const double uv = fma(t[u][0], kG4[v][0], fma(t[u][1], kG4[v][1], t[u][2] * kG4[v][2]));    // explicit fma: host and device (s3d_pack.hip) round alike
                    d[(((size_t(nt) * k8t + k8) * 24 + (u * 6 + v)) * 64 + (hf * 32 + jn)) * 4 + e] = float(uv);
                }
        }
    return off;
}

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

bool conv_rank1_inline_enabled() {
    // default OFF: measured slower than the two stand-alone launches (profiles/r03_rank1_inline.txt) — a slot that hosts a
    // producer starts its convolution tiles late by the producers' whole critical path, and with exactly 1-2 rounds of 27-us
    // tiles per launch that delay is not absorbed
    static const bool on = getenv("S3D_RANK1_INLINE") && atoi(getenv("S3D_RANK1_INLINE")) != 0;
    return on;
}

int launch_conv_wino24s(ConvArgs& a, hipStream_t st) {
    R1Inline none;
    memset(&none, 0, sizeof none);
    return launch_conv_wino24s_r1(a, none, nullptr, st);
}

// r1.nprod != 0 on entry: r1.mf, r1.cin, r1.sync and job[].{vin, wgt, out, L} are set by the caller (Fwd::conv); the block
// layout and the cumulative counter targets are filled in here (expect = the host mirror of the counters).
int launch_conv_wino24s_r1(ConvArgs& a, R1Inline& r1, unsigned* expect, hipStream_t st) {
    S3D_CHECK(a.njobs >= 1 && a.njobs <= kMaxConvJobs && a.cin % C_KC == 0 && a.cout % 4 == 0, S3D_ERR_INVALID, "wino24s conv: bad arguments");
    if (r1.nprod) {
        S3D_CHECK(a.njobs == 3 && expect && r1.sync && a.cout % 8 == 0, S3D_ERR_INVALID, "wino24s conv: in-launch rank-1 producers need the three planes of one TriplaneConv");
        r1_layout(r1, a.cout, a.B);
    }
    int blocks = 0;
    for (int j = 0; j < a.njobs; ++j) {
        ConvJob& J = a.job[j];
        S3D_CHECK(size_t(J.h) * J.w * a.cin * 4 < (size_t(1) << 31), S3D_ERR_INVALID, "wino24s conv: a plane of one sample must stay below 2 GiB");
        J.tiles_x = (J.w + C_TW - 1) / C_TW;
        J.tiles_per_img = J.tiles_x * ((J.h + C_TH - 1) / C_TH);
        J.n_tiles_n = (a.cout + 31) / 32;
        J.block_begin = blocks;
        blocks += J.tiles_per_img * J.n_tiles_n * a.B;
    }
    if (!blocks) return 0;
    if (r1.nprod) r1_targets(r1, a.B, expect, true);    // what each counter will have reached when this launch's producers are done
    a.xcd_swizzle = 1 | 2;                   // XCD-aware block order + raised priority outside the k-loop (were switchable in rounds 1-2: always wins)
    conv_note_kernel(r1.nprod ? "k_conv_wino24s mixed Winograd F(2x4,3x3), 8x16-pixel blocks, rollout means + rank-1 tables as in-launch producer blocks"
                              : "k_conv_wino24s mixed Winograd F(2x4,3x3), 8x16-pixel blocks");
    hipLaunchKernelGGL(k_conv_wino24s, dim3(blocks + r1.nprod), dim3(256), 0, st, a, r1);
    S3D_HIP(hipGetLastError());
    return 0;
}

static int conv_slots() {                     // co-resident 256-thread blocks of the 3x3 kernels: three per CU
    static const int v = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 3 * (cus > 0 ? cus : 256);
    }();
    return v;
}
bool conv_wino24_persistent_enabled() {
    // default OFF: measured slower than one tile per block (profiles/r03_wino_persistent.txt)
    static const bool on = getenv("S3D_WINO24_PERSIST") && atoi(getenv("S3D_WINO24_PERSIST")) != 0;
    return on;
}
// The persistent form takes every launch that has more tiles than slots (each block then owns >= 2 tiles for some blocks);
// smaller launches are one round of k_conv_wino24s anyway.  Results are bit-identical, so the choice may depend on the batch.
int launch_conv_wino24p(ConvArgs& a, hipStream_t st) {
    S3D_CHECK(a.r1_slices <= 1, S3D_ERR_INVALID, "wino24p conv: sliced rank-1 tables are read by k_conv_wino24s only");
    S3D_CHECK(a.njobs >= 1 && a.njobs <= kMaxConvJobs && a.cin % C_KC == 0 && a.cout % 4 == 0, S3D_ERR_INVALID, "wino24p conv: bad arguments");
    int blocks = 0;
    for (int j = 0; j < a.njobs; ++j) {
        ConvJob& J = a.job[j];
        S3D_CHECK(size_t(J.h) * J.w * a.cin * 4 < (size_t(1) << 31), S3D_ERR_INVALID, "wino24p conv: a plane of one sample must stay below 2 GiB");
        J.tiles_x = (J.w + C_TW - 1) / C_TW;
        J.tiles_per_img = J.tiles_x * ((J.h + C_TH - 1) / C_TH);
        J.n_tiles_n = (a.cout + 31) / 32;
        J.block_begin = blocks;
        blocks += J.tiles_per_img * J.n_tiles_n * a.B;
    }
    if (!blocks) return 0;
    a.xcd_swizzle = 1 | 2;                   // XCD-aware block order + raised priority outside the k-loop (were switchable in rounds 1-2: always wins)
    const int grid = std::min(blocks, conv_slots() & ~7);
    conv_note_kernel("k_conv_wino24p mixed Winograd F(2x4,3x3), 8x16-pixel tiles, persistent blocks (next tile's halo + weights prefetched across the tile boundary)");
    hipLaunchKernelGGL(k_conv_wino24p, dim3(grid), dim3(256), 0, st, a, blocks);
    S3D_HIP(hipGetLastError());
    return 0;
}
long long conv_wino24s_tiles(const ConvArgs& a) {
    long long t = 0;
    for (int j = 0; j < a.njobs; ++j) t += (long long)((a.job[j].w + C_TW - 1) / C_TW) * ((a.job[j].h + C_TH - 1) / C_TH) * ((a.cout + 31) / 32) * a.B;
    return t;
}
bool conv_wino24_takes_persistent(const ConvArgs& a) { return conv_wino24_persistent_enabled() && conv_wino24s_tiles(a) > conv_slots(); }

int launch_conv_wino24(ConvArgs& a, hipStream_t st) {
    S3D_CHECK(a.r1_slices <= 1, S3D_ERR_INVALID, "wino24 conv: sliced rank-1 tables are read by k_conv_wino24s only");
    S3D_CHECK(a.njobs >= 1 && a.njobs <= kMaxConvJobs && a.cin % X_KC == 0 && a.cout % 4 == 0, S3D_ERR_INVALID, "wino24 conv: bad arguments");
    int blocks = 0;
    for (int j = 0; j < a.njobs; ++j) {
        ConvJob& J = a.job[j];
        S3D_CHECK(size_t(J.h) * J.w * a.cin * 4 < (size_t(1) << 31), S3D_ERR_INVALID, "wino24 conv: a plane of one sample must stay below 2 GiB");
        J.tiles_x = (J.w + X_TW - 1) / X_TW;
        J.tiles_per_img = J.tiles_x * ((J.h + X_TH - 1) / X_TH);
        J.n_tiles_n = (a.cout + 31) / 32;
        J.block_begin = blocks;
        blocks += J.tiles_per_img * J.n_tiles_n * a.B;
    }
    if (!blocks) return 0;
    a.xcd_swizzle = 1 | 2;                   // XCD-aware block order + raised priority outside the k-loop (were switchable in rounds 1-2: always wins)
    conv_note_kernel("k_conv_wino24 mixed Winograd F(2x4,3x3), 16x16-pixel blocks");
    hipLaunchKernelGGL(k_conv_wino24, dim3(blocks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

}  // namespace s3d
