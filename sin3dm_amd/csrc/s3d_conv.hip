// s3d_conv.hip — the hot kernel: im2col-free NHWC direct convolution on the fp32 matrix cores.
//
// Each TriplaneConv of the reference (src/diffusion/unet_triplane.py:21-60) is three independent
// Conv2d's whose input is [own C | axis-mean A (C) | axis-mean B (C)].  The two mean blocks are constant
// along one spatial axis, so their contribution is a pair of 1-D convolutions (rank-1 "rollout" terms)
// that this file evaluates as tiny vector jobs (CONV_1x3_VEC) and then adds in the epilogue of the dense
// convolution over the plane's own C channels.  See DESIGN.md §3.
//
// Dense part = implicit GEMM: M = pixels of a TH x TW tile, N = output channels, K = taps x C.
//   * the (TH+KH-1) x (TW+KW-1) input halo tile of one 32-channel chunk is staged once in LDS and reused by
//     all KH*KW taps (a tap is just a different LDS base offset: no im2col buffer anywhere);
//   * weights are pre-packed [tap][cout][cin] so a stage's B tile is BN rows of 128 contiguous bytes;
//   * the contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32, same rate as the fp32 vector peak but one
//     operand VGPR per lane); the K order inside an 8-wide step is permuted (lane half 0 takes k0..3, half 1
//     k4..7) so both operands are fetched with one ds_read_b128 per 4 MFMAs;
//   * LDS rows are padded to 36 floats: the 32-lane column read of the B tile is conflict-free, the A tile's
//     2-D pixel patch costs at most 3x on a read that is issued once per 256+ MFMA cycles (tools/lds_bank_check.py);
//   * global->LDS staging is register-prefetched one stage ahead and double-buffered: one barrier per stage.
//   * epilogue: + bias (+ per-sample bias) + rank-1 row/column terms + residual, 128-byte runs per pixel.
#include "s3d_common.h"
#include "s3d_rank1.h"

namespace s3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef S3D_ABLATE
#define S3D_ABLATE 0            // tools/conv_ubench.hip only: 1 no prefetch loads, 2 no LDS stores, 4 no barrier, 8 no ds_reads
#endif
constexpr int KC = 32;          // channels per K chunk
constexpr int LDP = KC + 4;     // padded LDS row (floats)

// ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11) of an upsampled residual with the multiply-adds spelled out: the
// two epilogues of k_conv_mfma (plain / transposed accumulators) are chosen by launch size, and left to the compiler the
// expression was contracted into different fma chains in them — a sample's bits then depended on the batch it ran in
// (test_unet_properties_full_size).  Three fmas, two products, no free choice left.
__device__ __forceinline__ float up2x_blend(float ly0, float ly1, float lx0, float lx1, float v00, float v01, float v10, float v11) {
    const float top = __fmaf_rn(lx1, v01, __fmul_rn(lx0, v00));
    const float bot = __fmaf_rn(lx1, v11, __fmul_rn(lx0, v10));
    return __fmaf_rn(ly1, bot, __fmul_rn(ly0, top));
}

template <int TH_, int TW_, int KH_, int KW_, int WM_, int WN_, int MTW_, int NTW_, int KS_ = 1, int PF_ = 0, bool TR_ = false>
struct ConvCfg {
    static constexpr int KS = KS_;       // independent partial accumulators over K (breaks the MFMA dependency chain)
    static constexpr int PF = PF_;       // 1x1 only: K chunks of global loads kept in flight (0: the one-stage-ahead pipeline)
    // 1x1 only (round 4): the MFMA operands are swapped, D^T[cout][pixel] = W X^T — a lane's accumulator registers are then
    // runs of FOUR CONSECUTIVE OUTPUT CHANNELS of one pixel ((r & 3) + 8 (r >> 2) + 4 (lane >> 5)): the epilogue (bias, residual,
    // the bilinear samples of a half-resolution residual, the store) moves 16 bytes per access instead of 4, a quarter of the
    // memory instructions of the launch's longest phase.  Same products, same k order per output: bit-identical to TR = false.
    static constexpr bool TR = TR_;
    static constexpr int TH = TH_, TW = TW_, KH = KH_, KW = KW_, WM = WM_, WN = WN_, MTW = MTW_, NTW = NTW_;
    static constexpr int BM = TH * TW, BN = WN * NTW * 32;
    static constexpr int HH = TH + KH - 1, HW = TW + KW - 1;
    static constexpr int A_ELEMS = HH * HW * LDP, B_ELEMS = BN * LDP;
    static constexpr int A_ITEMS = HH * HW * (KC / 4), B_ITEMS = BN * (KC / 4);
    static constexpr int NA = (A_ITEMS + 255) / 256, NB = (B_ITEMS + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(BM == WM * MTW * 32, "pixel tile must match the wave grid");
    static_assert(32 % TW == 0 || TW % 32 == 0, "an MFMA row block must cover whole tile rows");
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gf4ptr;      // global address space: global_load, not flat
__device__ __forceinline__ gf4ptr to_global4(const float* p) { return (gf4ptr)(uintptr_t)p; }

__device__ __forceinline__ int edge_variant(int idx, int n) {
    // which taps of the summed-out axis fall inside the image: 0 interior, 1 first, 2 last, 3 only element
    return n == 1 ? 3 : (idx == 0 ? 1 : (idx == n - 1 ? 2 : 0));
}

template <class CFG>
__global__ __launch_bounds__(256) void k_conv_mfma(ConvArgs args) {
    constexpr int TW = CFG::TW, KH = CFG::KH, KW = CFG::KW, MTW = CFG::MTW, NTW = CFG::NTW;
    constexpr int HW = CFG::HW, TAPS = KH * KW;
    __shared__ __attribute__((aligned(16))) float smem[2 * (CFG::A_ELEMS + CFG::B_ELEMS)];
    float* const Abase = smem;
    float* const Bbase = smem + 2 * CFG::A_ELEMS;

    // workgroup i runs on XCD i % 8: contiguous logical ranges per XCD keep a tile's column blocks and its halo
    // neighbours in one L2 (see k_conv_wino2)
    int bid = blockIdx.x;
    if (args.xcd_swizzle) {
        const int chunk = int(gridDim.x) >> 3;
        if (bid < (chunk << 3)) bid = (bid & 7) * chunk + (bid >> 3);
    }
    int j = 0;
#pragma unroll
    for (int k = 1; k < kMaxConvJobs; ++k) j += (k < args.njobs && bid >= args.job[k].block_begin) ? 1 : 0;   // independent kernarg loads
    const ConvJob& J = args.job[j];
    int local = bid - J.block_begin;
    const int ntile = local % J.n_tiles_n; local /= J.n_tiles_n;
    const int b = local / J.tiles_per_img; local %= J.tiles_per_img;
    const int ty0 = (local / J.tiles_x) * CFG::TH, tx0 = (local % J.tiles_x) * TW;
    const int n0 = ntile * CFG::BN;
    const int h = J.h, w = J.w, cin = args.cin, cout = args.cout;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / CFG::WN, wn = wid % CFG::WN;

    int offA[MTW], offB[NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int p = (wm * MTW + mt) * 32 + (lane & 31);
        offA[mt] = ((p / TW) * HW + (p % TW)) * LDP + (lane >> 5) * 4;
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) offB[nt] = ((wn * NTW + nt) * 32 + (lane & 31)) * LDP + (lane >> 5) * 4;

    // ---- global -> register staging descriptors (fixed per thread).  Out-of-image / out-of-range items keep a
    // valid (clamped) address and are zeroed after the load, so the pipeline below has no divergent control flow.
    gf4ptr aSrc[CFG::NA]; int aDst[CFG::NA]; bool aOk[CFG::NA];
#pragma unroll
    for (int it = 0; it < CFG::NA; ++it) {
        const int idx = min(it * 256 + tid, CFG::A_ITEMS - 1);      // tail threads duplicate the last item
        const int pix = idx >> 3, q = idx & 7;
        const int gy = ty0 + pix / HW - KH / 2, gx = tx0 + pix % HW - KW / 2;
        aOk[it] = gy >= 0 && gy < h && gx >= 0 && gx < w;
        aDst[it] = pix * LDP + q * 4;
        aSrc[it] = to_global4(J.in + ((size_t(b) * h + (aOk[it] ? gy : 0)) * w + (aOk[it] ? gx : 0)) * cin + q * 4);
    }
    gf4ptr bSrc[CFG::NB]; int bDst[CFG::NB]; bool bOk[CFG::NB];
#pragma unroll
    for (int it = 0; it < CFG::NB; ++it) {
        const int idx = min(it * 256 + tid, CFG::B_ITEMS - 1);
        const int n = idx >> 3, q = idx & 7;
        bOk[it] = n0 + n < cout;
        bDst[it] = n * LDP + q * 4;
        bSrc[it] = to_global4(J.wgt + size_t(bOk[it] ? n0 + n : 0) * cin + q * 4);
    }
    const size_t tapStride4 = size_t(cout) * cin / 4;               // in float4 units

    constexpr int KS = CFG::KS;
    f32x16 acc[KS][MTW][NTW];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ks][mt][nt][r] = 0.f;

    const int nchunks = cin / KC;
    f32x4 ra[CFG::NA], rb[CFG::NB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // prologue: stage 0 (chunk 0, tap 0) -> LDS buffers 0
#pragma unroll
    for (int it = 0; it < CFG::NA; ++it) { ra[it] = aSrc[it][0]; if (!aOk[it]) ra[it] = zero4; }
#pragma unroll
    for (int it = 0; it < CFG::NB; ++it) { rb[it] = bSrc[it][0]; if (!bOk[it]) rb[it] = zero4; }
#pragma unroll
    for (int it = 0; it < CFG::NA; ++it) *reinterpret_cast<f32x4*>(Abase + aDst[it]) = ra[it];
#pragma unroll
    for (int it = 0; it < CFG::NB; ++it) *reinterpret_cast<f32x4*>(Bbase + bDst[it]) = rb[it];
    __syncthreads();

    if constexpr (TAPS == 1 && CFG::PF > 0) {
        // 1x1 (round 3): a stage here is only KC/8 x MTW x NTW MFMAs (~0.3 us) — far less than a global round trip, so the
        // one-stage-ahead pipeline below waits ~1.5 us in EVERY stage (33.7 us for 128->128 @128^2, MFMA pipe 23 % busy).
        // Instead PF chunks of loads stay in flight in registers; a chunk is written to LDS when its turn comes, its register
        // slot is refilled with chunk c + PF, one barrier per chunk.  (The prologue above already staged chunk 0.)
        constexpr int PF = CFG::PF;
        f32x4 qa[PF][CFG::NA], qb[PF][CFG::NB];
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int ch = min(1 + d, nchunks - 1);
#pragma unroll
            for (int it = 0; it < CFG::NA; ++it) qa[d][it] = aSrc[it][ch * (KC / 4)];
#pragma unroll
            for (int it = 0; it < CFG::NB; ++it) qb[d][it] = bSrc[it][ch * (KC / 4)];
        }
        for (int c0 = 0; c0 < nchunks; c0 += PF) {
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                const int chunk = c0 + d;
                if (chunk < nchunks) {
                    const float* As = Abase + (chunk & 1) * CFG::A_ELEMS;
                    const float* Bs = Bbase + (chunk & 1) * CFG::B_ELEMS;
#pragma unroll
                    for (int k8 = 0; k8 < KC / 8; ++k8) {
                        f32x4 a4[MTW], b4[NTW];
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) a4[mt] = *reinterpret_cast<const f32x4*>(As + offA[mt] + k8 * 8);
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt) b4[nt] = *reinterpret_cast<const f32x4*>(Bs + offB[nt] + k8 * 8);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                                for (int nt = 0; nt < NTW; ++nt)
                                    acc[(k8 * 4 + e) % KS][mt][nt] = CFG::TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(b4[nt][e], a4[mt][e], acc[(k8 * 4 + e) % KS][mt][nt], 0, 0, 0)
                                                                             : __builtin_amdgcn_mfma_f32_32x32x2f32(a4[mt][e], b4[nt][e], acc[(k8 * 4 + e) % KS][mt][nt], 0, 0, 0);
                    }
                    // chunk + 1 (slot d) -> the other LDS buffer (its last readers passed the previous barrier), slot refilled
                    if (chunk + 1 < nchunks) {
                        float* Ad = Abase + ((chunk + 1) & 1) * CFG::A_ELEMS;
                        float* Bd = Bbase + ((chunk + 1) & 1) * CFG::B_ELEMS;
                        // (S3D_ABLATE & 2, tools/conv_ubench.hip only: no ds_write pass — what staging by LDS-DMA could remove at most)
#pragma unroll
                        for (int it = 0; it < CFG::NA; ++it) if (!(S3D_ABLATE & 2)) *reinterpret_cast<f32x4*>(Ad + aDst[it]) = aOk[it] ? qa[d][it] : zero4; else asm volatile("" :: "v"(qa[d][it]));
#pragma unroll
                        for (int it = 0; it < CFG::NB; ++it) if (!(S3D_ABLATE & 2)) *reinterpret_cast<f32x4*>(Bd + bDst[it]) = bOk[it] ? qb[d][it] : zero4; else asm volatile("" :: "v"(qb[d][it]));
                        const int ch = min(chunk + 1 + PF, nchunks - 1);
                        if (chunk + 1 + PF < nchunks) {
#pragma unroll
                            for (int it = 0; it < CFG::NA; ++it) qa[d][it] = aSrc[it][ch * (KC / 4)];
#pragma unroll
                            for (int it = 0; it < CFG::NB; ++it) qb[d][it] = bSrc[it][ch * (KC / 4)];
                        }
                    }
                    __syncthreads();
                }
            }
        }
    } else
    // Pipeline: chunk-outer, taps fully unrolled.  The next chunk's A halo tile is fetched at the top of a chunk and
    // parked in registers for all TAPS stages; each stage fetches the next stage's B tile before its MFMA block and
    // writes it to the other LDS buffer after it.  The final stage re-fetches a clamped (valid) tile that nobody reads.
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int nch = min(chunk + 1, nchunks - 1);
#pragma unroll
        for (int it = 0; it < CFG::NA; ++it) if (!(S3D_ABLATE & 1)) ra[it] = aSrc[it][nch * (KC / 4)];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            constexpr int kLast = TAPS - 1;
            const int ntap = tap == kLast ? 0 : tap + 1;
            const int bch = tap == kLast ? nch : chunk;
#pragma unroll
            for (int it = 0; it < CFG::NB; ++it) if (!(S3D_ABLATE & 1)) rb[it] = bSrc[it][ntap * tapStride4 + bch * (KC / 4)];
            __builtin_amdgcn_sched_barrier(0);
            const int stage = chunk * TAPS + tap;
            const float* As = Abase + (chunk & 1) * CFG::A_ELEMS + ((tap / KW) * HW + (tap % KW)) * LDP;
            const float* Bs = Bbase + (stage & 1) * CFG::B_ELEMS;
#pragma unroll
            for (int k8 = 0; k8 < KC / 8; ++k8) {
                f32x4 a4[MTW], b4[NTW];
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
                    a4[mt] = (S3D_ABLATE & 8) ? ra[0] + float(k8) : *reinterpret_cast<const f32x4*>(As + offA[mt] + k8 * 8);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    b4[nt] = (S3D_ABLATE & 8) ? rb[0] + float(k8) : *reinterpret_cast<const f32x4*>(Bs + offB[nt] + k8 * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)                    // consecutive MFMAs go to different accumulators
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt) {
                            constexpr int dummy = 0; (void)dummy;
                            const int ks = (k8 * 4 + e) % KS;
                            acc[ks][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[mt][e], b4[nt][e], acc[ks][mt][nt], 0, 0, 0);
                        }
            }
            __builtin_amdgcn_sched_barrier(0);
            float* Bd = Bbase + ((stage + 1) & 1) * CFG::B_ELEMS;
#pragma unroll
            for (int it = 0; it < CFG::NB; ++it) if (!(S3D_ABLATE & 2)) *reinterpret_cast<f32x4*>(Bd + bDst[it]) = bOk[it] ? rb[it] : zero4;
            if (tap == kLast && !(S3D_ABLATE & 2)) {
                float* Ad = Abase + ((chunk + 1) & 1) * CFG::A_ELEMS;
#pragma unroll
                for (int it = 0; it < CFG::NA; ++it) *reinterpret_cast<f32x4*>(Ad + aDst[it]) = aOk[it] ? ra[it] : zero4;
            }
            if (!(S3D_ABLATE & 4)) __syncthreads();
        }
    }

    if constexpr (CFG::TR) {
        // ---- transposed epilogue: lane = pixel (lane & 31) of an MFMA tile, registers 4q .. 4q+3 = output channels
        // 8q + 4 (lane >> 5) + {0..3} of the tile's 32: one 16-byte access per (pixel, q).  No rank-1 terms / GroupNorm partials
        // here (the launcher only sends plain 1x1 convolutions this way).
        static_assert(TAPS == 1 && CFG::PF > 0, "transposed accumulators: 1x1 only");
        const float* __restrict__ p_bias = J.bias;
        const float* __restrict__ p_bbias = J.bbias;
        const float* __restrict__ p_res = J.res;
        float* __restrict__ p_out = J.out;
        const size_t img = size_t(b) * h;
        const int hi = h >> 1, wi = w >> 1;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int p = (wm * MTW + mt) * 32 + (lane & 31);
            const int y = ty0 + p / TW, x = tx0 + p % TW;
            const bool pix_ok = y < h && x < w;
            const int yc = min(y, h - 1), xc = min(x, w - 1);
            const size_t obase = ((img + yc) * w + xc) * cout;
            // bilinear source of a half-resolution residual (F.interpolate arithmetic, as the plain epilogue below)
            float fy = 0.5f * (float(yc) + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
            float fx = 0.5f * (float(xc) + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
            int y0 = int(fy); y0 = y0 > hi - 1 ? hi - 1 : y0;
            int x0 = int(fx); x0 = x0 > wi - 1 ? wi - 1 : x0;
            y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
            const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
            const float ly1 = fy - float(y0), ly0 = 1.f - ly1, lx1 = fx - float(x0), lx0 = 1.f - lx1;
            const float* rb = p_res && J.res_up ? p_res + size_t(b) * hi * wi * cout : nullptr;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int cobase = n0 + (wn * NTW + nt) * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) {                // two channel quads at a time: 10 loads in flight, 40 registers (all four: 186 VGPRs, two waves per SIMD)
                f32x4 addv[2], r00[2], r01[2], r10[2], r11[2];
                bool ok[2];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {                // every load of the pair first ...
                    const int co4 = cobase + 8 * (2 * qh + qq);
                    ok[qq] = pix_ok && co4 < cout;
                    const int coc = co4 < cout ? co4 : 0;
                    addv[qq] = p_bias ? *reinterpret_cast<const f32x4*>(p_bias + coc) : zero4;
                    if (p_bbias) addv[qq] += *reinterpret_cast<const f32x4*>(p_bbias + size_t(b) * J.bbias_stride + coc);
                    if (p_res && !J.res_up) addv[qq] += *reinterpret_cast<const f32x4*>(p_res + obase + coc);
                    if (rb) {
                        r00[qq] = *reinterpret_cast<const f32x4*>(rb + (size_t(y0) * wi + x0) * cout + coc);
                        r01[qq] = *reinterpret_cast<const f32x4*>(rb + (size_t(y0) * wi + x1) * cout + coc);
                        r10[qq] = *reinterpret_cast<const f32x4*>(rb + (size_t(y1) * wi + x0) * cout + coc);
                        r11[qq] = *reinterpret_cast<const f32x4*>(rb + (size_t(y1) * wi + x1) * cout + coc);
                    }
                }
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {                // ... then the arithmetic and the stores
                    const int q = 2 * qh + qq;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[0][mt][nt][4 * q + e];
#pragma unroll
                        for (int ks = 1; ks < KS; ++ks) t += acc[ks][mt][nt][4 * q + e];
                        float ad = addv[qq][e];
                        if (rb) ad += up2x_blend(ly0, ly1, lx0, lx1, r00[qq][e], r01[qq][e], r10[qq][e], r11[qq][e]);
                        t += ad;
                        v[e] = args.relu ? fmaxf(t, 0.f) : t;
                    }
                    if (ok[qq]) *reinterpret_cast<f32x4*>(p_out + obase + cobase + 8 * q) = v;
                }
                }
            }
        }
        return;
    }
    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Structured for memory-level parallelism: per 32-channel column block, first issue every rank-1 / residual
    // load of the lane's 16 pixels (predicated, clamped addresses, no per-element branches), then add and store.
    const int tile_idx = local;                       // pixel-tile index inside the image
    const float* __restrict__ p_bias = J.bias;
    const float* __restrict__ p_bbias = J.bbias;
    const float* __restrict__ p_rcol = J.rcol;
    const float* __restrict__ p_rrow = J.rrow;
    const float* __restrict__ p_res = J.res;
    float* __restrict__ p_out = J.out;
    double* p_gn = J.gn_part;
    const size_t img = size_t(b) * h;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int co = n0 + (wn * NTW + nt) * 32 + (lane & 31);
        const bool co_ok = co < cout;
        const int coc = co_ok ? co : 0;
        float base = p_bias ? p_bias[coc] : 0.f;
        if (p_bbias) base += p_bbias[size_t(b) * J.bbias_stride + coc];
        float gs = 0.f, gss = 0.f;                    // GroupNorm partial sums of this lane's channel
        // eight of the lane's sixteen pixel rows at a time: all sixteen (addends, flags and 64-bit output indices) cost 184
        // VGPRs — two blocks per CU, 1.5 and 3 rounds of blocks for the 1x1 launches of the batch-1 step
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            float addv[8];
            bool ok[8];
            unsigned oidx[8];                         // element index inside the sample's plane (< 2^32: checked by the launcher)
            const size_t plane0 = size_t(b) * h * w * cout;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = rh * 8 + rr;
                const int p = (wm * MTW + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int y = ty0 + p / TW, x = tx0 + p % TW;
                ok[rr] = y < h && x < w && co_ok;
                oidx[rr] = ok[rr] ? unsigned(y * w + x) * unsigned(cout) + unsigned(co) : 0u;
                addv[rr] = base;
            }
            if (p_rcol) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = rh * 8 + rr;
                    const int p = (wm * MTW + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int y = ty0 + p / TW, x = tx0 + p % TW;
                    const float t = p_rcol[ok[rr] ? ((size_t(b) * w + x) * 4 + edge_variant(y, h)) * cout + co : 0];
                    addv[rr] += ok[rr] ? t : 0.f;
                }
            }
            if (p_rrow) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = rh * 8 + rr;
                    const int p = (wm * MTW + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int y = ty0 + p / TW, x = tx0 + p % TW;
                    const float t = p_rrow[ok[rr] ? ((img + y) * 4 + edge_variant(x, w)) * cout + co : 0];
                    addv[rr] += ok[rr] ? t : 0.f;
                }
            }
            if (p_res && !J.res_up) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) addv[rr] += p_res[plane0 + oidx[rr]];      // index 0 when !ok: finite, unused
            }
            if (p_res && J.res_up) {                  // residual = exact-2x bilinear upsample of a half-resolution tensor (F.interpolate arithmetic)
                const int hi = h >> 1, wi = w >> 1;
                const float* rb = p_res + size_t(b) * hi * wi * cout + coc;
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = rh * 8 + rr;
                    const int p = (wm * MTW + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int y = min(ty0 + p / TW, h - 1), x = min(tx0 + p % TW, w - 1);
                    float fy = 0.5f * (float(y) + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
                    float fx = 0.5f * (float(x) + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
                    int y0 = int(fy); y0 = y0 > hi - 1 ? hi - 1 : y0;
                    int x0 = int(fx); x0 = x0 > wi - 1 ? wi - 1 : x0;
                    const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
                    const float ly1 = fy - float(y0), ly0 = 1.f - ly1, lx1 = fx - float(x0), lx0 = 1.f - lx1;
                    const float v00 = rb[(size_t(y0) * wi + x0) * cout], v01 = rb[(size_t(y0) * wi + x1) * cout];
                    const float v10 = rb[(size_t(y1) * wi + x0) * cout], v11 = rb[(size_t(y1) * wi + x1) * cout];
                    addv[rr] += up2x_blend(ly0, ly1, lx0, lx1, v00, v01, v10, v11);
                }
            }
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = rh * 8 + rr;
                float v = acc[0][mt][nt][r];
#pragma unroll
                for (int ks = 1; ks < KS; ++ks) v += acc[ks][mt][nt][r];
                v += addv[rr];
                if (args.relu) v = fmaxf(v, 0.f);
                if (ok[rr]) {
                    p_out[plane0 + oidx[rr]] = v;
                    gs += v; gss = fmaf(v, v, gss);
                }
            }
        }
        if (p_gn) {
            // this wave's {sum, sumsq} per subgroup of sg consecutive channels: fold the two lane halves (the other
            // 16 pixel rows), then the sg channels; one part per (pixel tile, wave row) — see GnPartials
            gs += __shfl_xor(gs, 32, 64); gss += __shfl_xor(gss, 32, 64);
            for (int off = 1; off < args.gn_sg; off <<= 1) { gs += __shfl_xor(gs, off, 64); gss += __shfl_xor(gss, off, 64); }
            if (lane < 32 && co_ok && (co % args.gn_sg) == 0) {
                const int part = tile_idx * CFG::WM + wm;
                double* dst = p_gn + ((size_t(b) * 3 * args.gn_nsub + co / args.gn_sg) * args.gn_maxparts + part) * 2;
                dst[0] = double(gs); dst[1] = double(gss);
            }
        }
    }
}

// ------------------------------------------------------------------ naive direct convolution (debug only)
__global__ void k_conv_naive(ConvArgs args, int KH, int KW) {
    for (int j = 0; j < args.njobs; ++j) {
        const ConvJob& J = args.job[j];
        const long long n = (long long)args.B * J.h * J.w * args.cout;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const int co = int(i % args.cout);
            long long r = i / args.cout;
            const int x = int(r % J.w); r /= J.w;
            const int y = int(r % J.h);
            const int b = int(r / J.h);
            float acc = 0.f;
            for (int kh = 0; kh < KH; ++kh)
                for (int kw = 0; kw < KW; ++kw) {
                    const int gy = y + kh - KH / 2, gx = x + kw - KW / 2;
                    if (gy < 0 || gy >= J.h || gx < 0 || gx >= J.w) continue;
                    const float* xi = J.in + ((size_t(b) * J.h + gy) * J.w + gx) * args.cin;
                    const float* wr = J.wgt + (size_t(kh * KW + kw) * args.cout + co) * args.cin;
                    for (int c = 0; c < args.cin; ++c) acc = fmaf(xi[c], wr[c], acc);
                }
            float v = acc + (J.bias ? J.bias[co] : 0.f);
            if (J.bbias) v += J.bbias[size_t(b) * J.bbias_stride + co];
            if (J.rcol) v += J.rcol[((size_t(b) * J.w + x) * 4 + edge_variant(y, J.h)) * args.cout + co];
            if (J.rrow) v += J.rrow[((size_t(b) * J.h + y) * 4 + edge_variant(x, J.w)) * args.cout + co];
            if (J.res && !J.res_up) v += J.res[i];
            if (J.res && J.res_up) {
                const int hi = J.h / 2, wi = J.w / 2;
                float fy = 0.5f * (float(y) + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
                float fx = 0.5f * (float(x) + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
                int y0 = int(fy); y0 = y0 > hi - 1 ? hi - 1 : y0;
                int x0 = int(fx); x0 = x0 > wi - 1 ? wi - 1 : x0;
                const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
                const float ly1 = fy - float(y0), ly0 = 1.f - ly1, lx1 = fx - float(x0), lx0 = 1.f - lx1;
                const float* rb = J.res + size_t(b) * hi * wi * args.cout + co;
                v += ly0 * (lx0 * rb[(size_t(y0) * wi + x0) * args.cout] + lx1 * rb[(size_t(y0) * wi + x1) * args.cout]) +
                     ly1 * (lx0 * rb[(size_t(y1) * wi + x0) * args.cout] + lx1 * rb[(size_t(y1) * wi + x1) * args.cout]);
            }
            if (args.relu) v = fmaxf(v, 0.f);
            J.out[i] = v;
        }
    }
}

// ------------------------------------------------------------------ rank-1 rollout tables (skinny GEMMs)
// out[b][pos][n] = sum_{tap, c} W[tap][n][c] * v[b][pos + tap - 1][c]   (zero outside [0, L)), n = variant*cout + co.
// M = L is only ~128 rows, so the work is split along K instead.  One block = 32 positions x 32 columns.  K = 3 taps x C
// is walked in stages of (tap, <=128-channel chunk): whole
// 512-byte rows of the vector (with its +-1 halo, loaded once per chunk) and of the weights are staged in LDS with
// coalesced loads, register-prefetched one stage ahead; inside a stage the four waves each contract a quarter of the
// chunk, and their partial accumulators are added through LDS in wave order at the end.
// ROLL3 (CONV_1x3_ROLL, the forward rollout tables): the four edge variants of a table entry are sums over subsets of the
// three taps o of the SUMMED-OUT axis (interior o0+o1+o2, first o1+o2, last o0+o1, single o1), so only the three per-tap
// products U_o are contracted — weights [tap][n][cin] with n = (co / 8) * 24 + o * 8 + co % 8, a block owns 32 positions x
// 8 output channels = 24 weight rows (a quarter fewer weight bytes than four pre-summed variants: the kernel is bound by
// streaming them once) — and the variants are formed while the four waves' partials are added.
// NS = 2 (batch > 1): a block takes two samples of its (position tile, column tile) and stages each weight tile once for both
// (s3d_rank1.h); 69.7 KB of dynamic LDS, two blocks per CU.
template <bool ROLL3, int NS = 1>
__global__ __launch_bounds__(256) void k_rank1(ConvArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];          // NS = 2: kR1LdsFloats + 34 * kR1Ld floats
    __shared__ __attribute__((aligned(16))) float lds_static[NS == 1 ? kR1LdsFloats : 4];
    float* lds = NS == 1 ? lds_static : lds_dyn;
    const int bid = blockIdx.x;
    int j = 0;                                             // independent kernarg loads (a while loop chains up to five of them)
#pragma unroll
    for (int k = 1; k < kMaxConvJobs; ++k) j += (k < args.njobs && bid >= args.job[k].block_begin) ? 1 : 0;
    const ConvJob& J = args.job[j];
    int local = bid - J.block_begin;
    R1Block blk;
    const int nsl = ROLL3 && args.r1_slices > 1 ? args.r1_slices : 1;
    const int sl = local % nsl; local /= nsl;                 // (the slices of a tile are neighbouring blocks)
    blk.ntile = local % J.n_tiles_n; local /= J.n_tiles_n;
    blk.b = (local / J.tiles_per_img) * NS; blk.mtile = local % J.tiles_per_img;
    blk.bcount = min(NS, args.B - blk.b);
    blk.vin = J.in; blk.wgt = J.wgt; blk.L = J.w; blk.cin = args.cin; blk.cout4 = args.cout; blk.n_tiles_n = J.n_tiles_n;
    blk.out = J.out + size_t(sl) * args.B * J.w * 4 * args.cout;
    if (nsl > 1) {                                            // two slices: chunks [0, ceil(n/2)) and [ceil(n/2), n)
        const int nchunks = args.cin / kR1Chunk, first = (nchunks + 1) / 2;
        blk.chunk0 = sl ? first : 0; blk.nch = sl ? nchunks - first : first;
    }
    rank1_block<ROLL3, NS>(blk, lds);
}

// ------------------------------------------------------------------ rank-1 rollout tables, fragment-order form (own channels in whole 128-channel chunks)
// k_rank1 stages its weight tiles through LDS with two barriers per (tap, chunk) stage and 52-70 KB of LDS — two blocks per
// CU, a latency chain that every block of a batched launch repeats (batch 8: 37 / 68 / 114 us per launch at 128 / 256 / 384
// channels for 12-35 us of MFMA work; batch 1 on the (256,256,128) planes: 19.6 us).  k_rank1b keeps that kernel's split of K over
// the four waves of a block (wave w contracts channels [32w, 32w+32) of every 128-channel chunk) but
//  * a block owns 32 positions of one sample x 32 output channels x the three taps o of the summed-out axis — three 32x32
//    accumulators per wave, the four edge variants are formed from complete sums, no zero-padded weight rows;
//  * the weights come from a second image in MFMA fragment order (rank1_frag_index): a B operand is one coalesced 1-KB load
//    straight into a register ring three k-steps deep — no LDS, no barrier inside the K walk;
//  * the blocks that read the same weights (same vector, slice and channel group) are neighbours on one XCD: its L2 fetches
//    them once;
//  * 48 KB of LDS (the vector tile, then the four waves' partial sums) -> three blocks per CU.
// BIT-IDENTICAL to k_rank1<true>: same products in the same order per wave (chunk, tap, k8, e ascending), the four partial sums
// meet as ((q0 + q1) + q2) + q3, the variants are (u0 + u1) + u2, u1 + u2, u0 + u1, u1.
// cin must be a multiple of 128 (whole chunks); slices as in k_rank1.
constexpr int kR1bLdsFloats = 4 * 48 * 64;                 // the reduction image; the vector tile (34 x (slice channels + 4)) sits in the same bytes
__global__ __launch_bounds__(256) void k_rank1b(ConvArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * int(gridDim.x >> 3) + (bid >> 3);        // consecutive logical ids share an XCD
    int j = 0;
#pragma unroll
    for (int k = 1; k < kMaxConvJobs; ++k) j += (k < args.njobs && bid >= args.job[k].block_begin) ? 1 : 0;
    const ConvJob& J = args.job[j];
    int local = bid - J.block_begin;
    const int L = J.w, cin = args.cin, cout = args.cout, B = args.B;
    const int mtile = local % J.tiles_per_img; local /= J.tiles_per_img;     // (tile, sample) fastest: neighbours read the same weights
    const int b = local % B; local /= B;
    const int g = local % J.n_tiles_n, sl = local / J.n_tiles_n;            // 32-output-channel group, K slice
    const int nchunks = cin / kR1Chunk;
    int chunk0 = 0, nch = nchunks;
    if (args.r1_slices > 1) { const int first = (nchunks + 1) / 2; chunk0 = sl ? first : 0; nch = sl ? nchunks - first : first; }
    const int csl = nch * kR1Chunk, ld = csl + 4;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 31, half = lane >> 5;
    // this wave's fragment stream: steps (chunk, tap, k8) of (group g, quarter wid), three fragments (o = 0, 1, 2) per step
    const int nsteps = nch * 12;
    const r1_f32x4* pf = reinterpret_cast<const r1_f32x4*>(J.wgt_r1f) + (size_t(g * 4 + wid) * (nchunks * 12) + chunk0 * 12) * 192 + lane;
    constexpr int D = 3;                                    // (four deep: no faster)
    r1_f32x4 ring[D][3];
    int pf_s = 0;
    auto prefetch = [&](int slot) {                         // unconditional (past the end: the last step again, unused) — the
        ring[slot][0] = ((r1_gf4ptr)(uintptr_t)pf)[0];      // compiler can then count the loads in flight (vmcnt(6)) instead of
        ring[slot][1] = ((r1_gf4ptr)(uintptr_t)(pf + 64))[0];   // draining them at every step
        ring[slot][2] = ((r1_gf4ptr)(uintptr_t)(pf + 128))[0];
        ++pf_s;
        pf += pf_s < nsteps ? 192 : 0;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) prefetch(d);               // (the weights do not depend on the vector tile: requested first)
    {   // the sample's vector tile: rows = positions mtile*32 - 1 .. + 32, zero outside [0, L)
        const int q4 = csl / 4, items = 34 * q4;
        const float* vb = J.in + (size_t(b) * L) * cin + chunk0 * kR1Chunk;
        const r1_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        for (int it0 = tid; it0 < items; it0 += 256 * 5) {
            r1_f32x4 v[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int it = it0 + 256 * u, row = it / q4, q = it - row * q4, pos = mtile * 32 - 1 + row;
                const bool ok = it < items && pos >= 0 && pos < L;
                v[u] = ((r1_gf4ptr)(uintptr_t)(vb + size_t(ok ? pos : 0) * cin + (ok ? q : 0) * 4))[0];
                if (!ok) v[u] = zero4;
            }
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int it = it0 + 256 * u, row = it / q4, q = it - row * q4;
                if (it < items) *reinterpret_cast<r1_f32x4*>(lds_dyn + row * ld + q * 4) = v[u];
            }
        }
    }
    __syncthreads();
    r1_f32x16 q[3];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) q[o][r] = 0.f;
    const float* Ab = lds_dyn + i * ld + half * 4 + wid * 32;
    for (int ch = 0; ch < nch; ++ch) {
        const float* Ac = Ab + ch * kR1Chunk;
#pragma unroll
        for (int st = 0; st < 12; ++st) {
            const int tap = st >> 2, k8 = st & 3;
            const r1_f32x4 a4 = *reinterpret_cast<const r1_f32x4*>(Ac + tap * ld + k8 * 8);
            r1_f32x4 b4[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) b4[o] = ring[st % D][o];
            __builtin_amdgcn_sched_barrier(0);              // (left alone, the scheduler sinks the ring's loads to their use: one exposed L2 round trip per step)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int o = 0; o < 3; ++o) q[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[o][e], q[o], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            prefetch(st % D);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();                                        // every wave is done with the vector tile: the same bytes take the partial sums
    float* red = lds_dyn;                                   // [wave][o][r][lane]
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wid * 3 + o) * 16 + r) * 64 + lane] = q[o][r];
    __syncthreads();
    float* out = J.out + size_t(sl) * B * L * 4 * cout;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int it = tid + 256 * k, r = it >> 6, l = it & 63;
        float u[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float* p = red + (o * 16 + r) * 64 + l;
            u[o] = ((p[0] + p[3 * 1024]) + p[2 * 3 * 1024]) + p[3 * 3 * 1024];
        }
        const int row = mtile * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), co = g * 32 + (l & 31);
        if (row < L && co < cout) {
            float* o4 = out + (size_t(b) * L + row) * 4 * cout + co;        // [pos][variant][cout]
            o4[0] = (u[0] + u[1]) + u[2]; o4[cout] = u[1] + u[2]; o4[2 * cout] = u[0] + u[1]; o4[3 * cout] = u[1];
        }
    }
}
bool rank1b_takes(const ConvArgs& a) {
    if (a.cin % kR1Chunk != 0) return false;
    for (int j = 0; j < a.njobs; ++j) if (!a.job[j].wgt_r1f) return false;
    const int nchunks = a.cin / kR1Chunk, maxch = a.r1_slices > 1 ? (nchunks + 1) / 2 : nchunks;
    if (34 * (maxch * kR1Chunk + 4) > kR1bLdsFloats) return false;     // the vector tile must fit the reduction image's 48 KB
    return opt_on(OPT_RANK1_BATCH);
}

int conv_rank1_slices(int cin) {
    return opt_on(OPT_RANK1_SLICES) && cin >= 2 * kR1Chunk && cin % kR1Chunk == 0 ? 2 : 1;
}

// S3D_CONV_IMPL=naive counterpart of the ROLL3 form: one thread per table entry, plain loops over the same weight image
__global__ void k_rank1_roll_naive(ConvArgs args) {
    for (int j = 0; j < args.njobs; ++j) {
        const ConvJob& J = args.job[j];
        const int L = J.w, cout = args.cout, cin = args.cin, nt = (cout + 7) / 8;
        const long long n = (long long)args.B * L * cout;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const int co = int(i % cout), pos = int((i / cout) % L), b = int(i / cout / L);
            float u[3] = {0.f, 0.f, 0.f};
            for (int o = 0; o < 3; ++o)
                for (int t = 0; t < 3; ++t) {
                    const int q = pos + t - 1;
                    if (q < 0 || q >= L) continue;
                    const float* v = J.in + (size_t(b) * L + q) * cin;
                    const float* wr = J.wgt + (size_t(t) * nt * 24 + (co / 8) * 24 + o * 8 + (co & 7)) * cin;
                    for (int c = 0; c < cin; ++c) u[o] = fmaf(v[c], wr[c], u[o]);
                }
            float* o4 = J.out + (size_t(b) * L + pos) * 4 * cout + co;
            o4[0] = (u[0] + u[1]) + u[2]; o4[cout] = u[1] + u[2]; o4[2 * cout] = u[0] + u[1]; o4[3 * cout] = u[1];
        }
    }
}

int launch_rank1(ConvArgs& a, hipStream_t st, bool roll3) {
    S3D_CHECK(a.njobs >= 1 && a.njobs <= kMaxConvJobs && a.cin % KC == 0, S3D_ERR_INVALID, "rank1: bad arguments");
    S3D_CHECK(a.r1_slices <= 1 || (roll3 && !conv_use_naive() && a.r1_slices == 2 && a.cin % kR1Chunk == 0 && a.cin >= 2 * kR1Chunk), S3D_ERR_INVALID, "rank1: two K slices of whole 128-channel chunks");
    if (conv_use_naive()) {
        if (!roll3) return launch_conv_naive(CONV_1x3_VEC, a, st);
        hipLaunchKernelGGL(k_rank1_roll_naive, dim3(512), dim3(256), 0, st, a);
        S3D_HIP(hipGetLastError());
        return 0;
    }
    if (roll3 && rank1b_takes(a)) {                       // fragment-order weights at hand (S3D_RANK1_BATCH=0: k_rank1, one sample per block)
        int blocks = 0;
        for (int j = 0; j < a.njobs; ++j) {
            ConvJob& J = a.job[j];
            J.tiles_x = J.tiles_per_img = (J.w + 31) / 32;
            J.n_tiles_n = (a.cout + 31) / 32;
            J.block_begin = blocks;
            blocks += J.tiles_per_img * a.B * J.n_tiles_n * (a.r1_slices > 1 ? a.r1_slices : 1);
        }
        if (!blocks) return 0;
        conv_note_kernel("k_rank1b (three-tap rollout tables, fragment-order weights)");
        hipLaunchKernelGGL(k_rank1b, dim3(blocks), dim3(256), kR1bLdsFloats * sizeof(float), st, a);
        S3D_HIP(hipGetLastError());
        return 0;
    }
    // other widths: k_rank1, two samples per block from batch 2 on (S3D_RANK1_BATCH=0: one): same sums per sample, the weight tiles staged half as often
    const bool pair_ok = opt_on(OPT_RANK1_BATCH);
    const int ns = roll3 && pair_ok && a.B >= 2 ? 2 : 1;
    int blocks = 0;
    for (int j = 0; j < a.njobs; ++j) {
        ConvJob& J = a.job[j];
        J.tiles_x = J.tiles_per_img = (J.w + 31) / 32;
        J.n_tiles_n = roll3 ? (a.cout + 7) / 8 : (a.cout + 31) / 32;
        J.block_begin = blocks;
        blocks += J.tiles_per_img * J.n_tiles_n * ((a.B + ns - 1) / ns) * (roll3 && a.r1_slices > 1 ? a.r1_slices : 1);
    }
    if (!blocks) return 0;
    // (round 4, measured and removed: a wide form — 32 positions x 32 output channels x the three taps per block, three
    // accumulator tiles per wave, weight operands straight from L2 into registers; bit-identical tables — 0.117 -> 0.153 ms/step
    // at batch 1, 0.490 -> 0.500 at batch 8, 0.155 -> 0.196 on the (256,256,128) planes: a lane's 16-byte reads of 32 different
    // weight rows use a quarter of every cache line they touch, profiles/r04_rank1_wide.txt)
    // (measured in round 2 and dropped: whole-chunk stages with two chunks of loads in flight, and eight waves per block —
    // 9.4 / 14.7 / 18.7 us at 128 / 256 / 384 channels either way: the launch is bound by the ~21 MB per 128-channel chunk that
    // 384 blocks pull through L2 at once, not by its stage latency or its MFMA chains)
    conv_note_kernel(roll3 ? "k_rank1<true> (three-tap rollout tables)" : "k_rank1<false>");
    constexpr size_t lds1 = kR1LdsFloats * sizeof(float), lds2 = lds1 + 34 * kR1Ld * sizeof(float);
    if (ns == 2) {
        static std::atomic<unsigned long long> opted{0};
        S3D_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_rank1<true, 2>), int(lds2), opted));
        hipLaunchKernelGGL((k_rank1<true, 2>), dim3(blocks), dim3(256), lds2, st, a);
    } else if (roll3) hipLaunchKernelGGL(k_rank1<true>, dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_rank1<false>, dim3(blocks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

void conv_gn_parts(ConvKind kind, const Geo& g, int nparts[3], bool wino24) {
    // must mirror launch_conv's tile choice for the kinds whose epilogue emits GroupNorm partials (3x3 only)
    (void)kind;
    if (wino24 && conv_use_wino24()) {                            // k_conv_wino24s / k_conv_wino24w: one part per 8x16-pixel tile
        for (int p = 0; p < 3; ++p) nparts[p] = ((g.w[p] + 15) / 16) * ((g.h[p] + 7) / 8);
        return;
    }
    if (conv_use_wino()) { wino_gn_parts(g, nparts); return; }
    using CFG = ConvCfg<8, 8, 3, 3, 2, 2, 1, 1>;
    for (int p = 0; p < 3; ++p)
        nparts[p] = ((g.w[p] + CFG::TW - 1) / CFG::TW) * ((g.h[p] + CFG::TH - 1) / CFG::TH) * CFG::WM;
}

static thread_local const char* t_last_kernel = "";
void conv_note_kernel(const char* name) { t_last_kernel = name; }
const char* conv_last_kernel() { return t_last_kernel; }

bool conv_use_naive() {
    return opt(OPT_CONV_IMPL) == 1;
}

static void kind_taps(ConvKind kind, int& KH, int& KW) {
    switch (kind) {
        case CONV_3x3: KH = 3; KW = 3; break;
        case CONV_1x1: KH = 1; KW = 1; break;
        case CONV_1x3_VEC: case CONV_1x3_ROLL: KH = 1; KW = 3; break;
        default: KH = 5; KW = 5; break;
    }
}

int launch_conv_naive(ConvKind kind, ConvArgs& a, hipStream_t st) {
    int KH, KW;
    kind_taps(kind, KH, KW);
    conv_note_kernel("k_conv_naive (one thread per output, S3D_CONV_IMPL=naive)");
    hipLaunchKernelGGL(k_conv_naive, dim3(2048), dim3(256), 0, st, a, KH, KW);
    S3D_HIP(hipGetLastError());
    return 0;
}

template <class CFG>
static int launch_cfg(ConvArgs& a, hipStream_t st) {
    int blocks = 0;
    for (int j = 0; j < a.njobs; ++j) {
        ConvJob& J = a.job[j];
        J.tiles_x = (J.w + CFG::TW - 1) / CFG::TW;
        J.tiles_per_img = J.tiles_x * ((J.h + CFG::TH - 1) / CFG::TH);
        J.n_tiles_n = (a.cout + CFG::BN - 1) / CFG::BN;
        J.block_begin = blocks;
        blocks += J.tiles_per_img * J.n_tiles_n * a.B;
        S3D_CHECK((long long)J.h * J.w * a.cout < (1LL << 32), S3D_ERR_INVALID, "conv: a plane of one sample must stay below 2^32 elements");
    }
    if (!blocks) return 0;
    a.xcd_swizzle = 1;                       // XCD-aware block order (was switchable in rounds 1-2: always a win)
    conv_note_kernel(CFG::KH == 3 ? "k_conv_mfma<3x3> direct MFMA convolution" : (CFG::KH == 1 ? (CFG::TR ? "k_conv_mfma<1x1, transposed accumulators> direct MFMA convolution" : "k_conv_mfma<1x1> direct MFMA convolution") : "k_conv_mfma<5x5> direct MFMA convolution"));
    hipLaunchKernelGGL(k_conv_mfma<CFG>, dim3(blocks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

template <int TH, int TW>
static long long count_tiles(const ConvArgs& a) {
    long long t = 0;
    for (int j = 0; j < a.njobs; ++j) t += (long long)((a.job[j].w + TW - 1) / TW) * ((a.job[j].h + TH - 1) / TH);
    return t * a.B;
}

// Mixed Winograd F(2x4,3x3) takes every 3x3 launch of these widths (which of its two bit-identical kernels: launch_conv_wino24s)
bool conv_wino24_channels(int cin, int cout) { return conv_use_wino24() && cout % 4 == 0 && cin % 32 == 0; }
static bool takes_wino24(const ConvArgs& a) { return conv_wino24_channels(a.cin, a.cout) && a.job[0].wgt_wino24s; }
double conv_exec_fraction(ConvKind kind, const ConvArgs& a) {
    if (kind == CONV_3x3 && !conv_use_naive() && takes_wino24(a)) return 1.0 / 3.0;      // 24 multiplies per 2x4 outputs instead of 72
    if (kind == CONV_3x3 && !conv_use_naive() && conv_use_wino() && a.job[0].wgt_wino && a.cout % 4 == 0) return wino_exec_fraction();
    return 1.0;
}

int launch_conv(ConvKind kind, ConvArgs& a, hipStream_t st) {
    S3D_CHECK(a.njobs >= 1 && a.njobs <= kMaxConvJobs, S3D_ERR_INVALID, "conv: %d jobs", a.njobs);
    S3D_CHECK(a.cin % KC == 0 && a.cin > 0, S3D_ERR_INVALID, "conv: cin=%d must be a positive multiple of %d", a.cin, KC);
    if (conv_use_naive()) return kind == CONV_1x3_ROLL ? launch_rank1(a, st, true) : launch_conv_naive(kind, a, st);   // (the three-tap table form has its own plain kernel)
    // Tile choice (measured, tools/conv_ubench.hip and in the step; for 1x1 again after the Winograd rework): the 64-pixel x 64-cout tile wins at every
    // size of this network because three blocks stay resident per CU (47 KB LDS each) and their staggered barriers keep
    // the matrix pipe fed; the 128-pixel tiles halve the staging traffic but leave 1-2 blocks per CU (85 vs 102 TF, profiles/r01_tile_sweep.txt).
    switch (kind) {
        case CONV_3x3:
            if (takes_wino24(a)) {
                for (int j = 0; j < a.njobs; ++j) a.job[j].wgt = a.job[j].wgt_wino24s;
                return launch_conv_wino24s(a, st);
            }
            if (conv_use_wino() && a.job[0].wgt_wino && a.cout % 4 == 0) {   // (its epilogue moves channel quads; GroupNorm'd layers always qualify)
                for (int j = 0; j < a.njobs; ++j) a.job[j].wgt = a.job[j].wgt_wino;
                return launch_conv_wino(a, st);
            }
            return launch_cfg<ConvCfg<8, 8, 3, 3, 2, 2, 1, 1>>(a, st);
        case CONV_1x1:         // (round-2 sweep in the step: 128 px x 64 cout 0.081, 64 x 128 0.078, 128 x 128 0.218, two K accumulators 0.081,
                               //  64-channel K chunks (70 KB LDS, two blocks per CU) 0.084, 16-channel chunks 0.077 vs 0.072 ms/step)
            // two K chunks of global loads in flight (round 3: one stage ahead 0.0713 ms/step, 2 chunks 0.0686, whole K 0.0711 — the
            // stages' memory latency is NOT what these launches wait for, profiles/r03_conv1x1.txt)
            {
                // transposed accumulators (16-byte epilogue accesses) for plain 1x1 convolutions — no rank-1 terms, no GroupNorm
                // partials, channel quads — of at least four rounds of blocks: batch 8 0.383 -> 0.357 ms/step, the auto-encoder
                // iteration 6.31 -> 6.26 ms; the one-round launches of the batch-1 step are a latency chain either way
                // (0.0693 -> 0.0714 ms/step) and keep the 4-byte epilogue.  S3D_CONV1X1_T=0 never / =1 always (tests).
                const int tr_mode = opt(OPT_CONV1X1_T);
                bool plain = tr_mode != 0 && a.cout % 4 == 0;
                long long blocks = 0;
                for (int j = 0; j < a.njobs; ++j) {
                    plain = plain && !a.job[j].rrow && !a.job[j].rcol && !a.job[j].gn_part;
                    blocks += (long long)((a.job[j].w + 7) / 8) * ((a.job[j].h + 7) / 8) * ((a.cout + 63) / 64) * a.B;
                }
                if (plain && (tr_mode == 1 || blocks >= 4 * 768)) return launch_cfg<ConvCfg<8, 8, 1, 1, 2, 2, 1, 1, 1, 2, true>>(a, st);
            }
            return launch_cfg<ConvCfg<8, 8, 1, 1, 2, 2, 1, 1, 1, 2>>(a, st);
        case CONV_1x3_VEC:
            return launch_rank1(a, st, false);
        case CONV_1x3_ROLL:
            return launch_rank1(a, st, true);
        case CONV_5x5:
            return launch_cfg<ConvCfg<8, 8, 5, 5, 2, 2, 1, 1>>(a, st);
    }
    set_error("conv: unknown kind %d", int(kind));
    return S3D_ERR_INVALID;
}

}  // namespace s3d
