// s3d_wino.hip — 3x3 TriplaneConv as a fused Winograd F(2x2, 3x3) convolution on the fp32 matrix cores.
//
// Y = A^T [ (G g G^T) .* (B^T d B) ] A per 4x4 input patch d -> 2x2 outputs: 16 "frequency" GEMMs with 4 multiplies
// per output instead of 9 (2.25x fewer MFMA flops than the direct kernel in s3d_conv.hip).  fp32 error measured
// against an fp64 direct convolution: 4.5e-7 relative (direct fp32: 2.8e-7), far inside the 1e-3 gate.
//
// Everything happens inside one kernel — no transformed tensors ever touch HBM:
//   * a block owns an 8x16-pixel output tile = 4x8 Winograd tiles = one 32-row MFMA tile; the 10x18x32-channel input
//     halo of a K chunk sits in LDS; a lane owns ONE Winograd tile (MFMA row), reads its 4x4 patch with ds_read_b128,
//     applies B^T d B in registers and thereby holds the A operands for 4 channels — already in the "lane half 0 takes
//     k0..3, half 1 k4..7" order;
//   * the transformed weights U = G g G^T are pre-packed on the host in MFMA *fragment order*, so a B operand is one
//     fully coalesced 1 KB load per (frequency, 8 channels) — no LDS, no barrier for weights;
//   * the 16 frequencies of a (tile, 32 output channels) unit are split over several waves, so that two or three waves
//     fit on every SIMD and cover each other's barriers and latency-bound phases:
//       k_conv_wino4 (default)  four waves, one row of the 4x4 frequency grid each: 64 accumulator registers,
//                               three 32-channel blocks per CU;
//       k_conv_wino2 (S3D_WINO=2) two waves, two rows each: 128 accumulator registers, two 64-channel blocks per CU;
//   * software pipeline inside a wave: a k-step is a sequence of pinned slots {one MFMA + a small piece of the other
//     work} (next step's patch reads and transform, one weight fragment per frequency through a register ring, two halo
//     loads); one barrier per 32-channel chunk;
//   * the inverse transform is linear: every wave pushes its frequency rows through it alone and the shares meet in LDS,
//     where threads owning 4 consecutive channels of a pixel finish the tile (bias, rank-1 rollout terms, residual,
//     GroupNorm partial sums) with 16-byte accesses.
#include "s3d_common.h"

namespace s3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* wgf4;
__device__ __forceinline__ wgf4 wg4(const float* p) { return (wgf4)(uintptr_t)p; }

#ifndef W_ABL
#define W_ABL 0                        // tools/wino_ubench.hip only (k_conv_wino2): 8 weights from one hot 16 KB, 16 no halo loads
#endif
constexpr int kDefaultWino = 24;        // S3D_WINO default (see wino_variant)
constexpr int W_KC = 32;                // channels per chunk
constexpr int W_LD = W_KC + 4;          // padded LDS pixel row (floats)

__device__ __forceinline__ int w_edge_variant(int idx, int n) { return n == 1 ? 3 : (idx == 0 ? 1 : (idx == n - 1 ? 2 : 0)); }

__device__ __forceinline__ void wino_row_pass(f32x4* r) {      // one patch row, along b
    const f32x4 d0 = r[0], d1 = r[1], d2 = r[2], d3 = r[3];
    r[0] = d0 - d2; r[1] = d1 + d2; r[2] = d2 - d1; r[3] = d1 - d3;
}

// ------------------------------------------------------------------ two waves per SIMD: the frequencies split in halves
// A (tile, 32-channel) unit is shared by TWO waves that each own 8 of the 16 frequencies (rows u = 2*fh, 2*fh+1 of the 4x4
// frequency grid): 128 accumulator registers per wave, two 64-channel blocks per CU.  Slightly ahead of k_conv_wino4
// when a launch has many more blocks than the GPU has slots (batch 8: +0.7 %) because the halo is staged once per 64
// output channels; behind it everywhere else (batch 1: -6 %, small planes -15 %).
constexpr int W2_TH = 8, W2_TW = 16;                      // output tile: 4 x 8 Winograd tiles = one 32-row MFMA tile
constexpr int W2_HH = W2_TH + 2, W2_HW = W2_TW + 2;
constexpr int W2_ITEMS = W2_HH * W2_HW * (W_KC / 4);
constexpr int W2_ITEMS_PT = (W2_ITEMS + 255) / 256;       // halo float4 items per thread and chunk (6)
constexpr int W2_ABUF = W2_ITEMS_PT * 32 * W_LD;          // LDS buffer stride: 6 rounds x 32 pixels, so no store of a round needs a predicate
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef W_TIMING
__device__ unsigned long long* g_wtime;       // tools/wino_ubench.hip: eight wall-clock stamps per block (entry, halo landed, first MFMA, last MFMA, epilogue barriers, exit)
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
#pragma unroll
    for (int k = 1; k < kMaxConvJobs; ++k) j += (k < args.njobs && bid >= args.job[k].block_begin) ? 1 : 0;   // independent kernarg loads
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
        auto put = [&](int sub, float s, float ss) {            // partial layout [b][plane][sub][part][2]
            double* dst = p_gn + ((size_t(b) * 3 * args.gn_nsub + sub) * args.gn_maxparts + part) * 2;
            dst[0] = double(s); dst[1] = double(ss);
        };
        if (sg >= 4) {
            float s = (gs4[0] + gs4[1]) + (gs4[2] + gs4[3]), ss = (gss4[0] + gss4[1]) + (gss4[2] + gss4[3]);
            for (int off = 1; off < (sg >> 2); off <<= 1) { s += __shfl_xor(s, off, 64); ss += __shfl_xor(ss, off, 64); }
            if (lane < 16 && c_ok && (co4 % sg) == 0) put(co4 / sg, s, ss);
        } else if (lane < 16 && c_ok) {
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                if (sg == 2) put((co4 + e) / 2, gs4[e] + gs4[e + 1], gss4[e] + gss4[e + 1]);
                else { put(co4 + e, gs4[e], gss4[e]); put(co4 + e + 1, gs4[e + 1], gss4[e + 1]); }
            }
        }
    }
    W_STAMP(3)
}

// ------------------------------------------------------------------ three waves per SIMD: one frequency row per wave
// Same tile (8x16 pixels) and the same data flow as k_conv_wino2, but a block owns 32 output channels and its four waves
// own one row u of the 4x4 frequency grid each: 64 accumulator registers, <= 168 VGPRs, 52 KB of LDS -> THREE blocks per
// CU, three waves on every SIMD.  The half-resolution layers at batch 1 become 768 blocks on 768 slots (k_conv_wino2:
// 384 blocks, half the CUs with one block and the other half with two), the full-resolution ones two full rounds instead
// of one and a half; small planes fill more of the GPU.  The price: the halo is staged per 32 output channels.
//   wave u needs two rows of the 4x4 patch: t = x + s*y with (x, y, s) = (d0, d2, -), (d1, d2, +), (d2, d1, -), (d1, d3, -)
constexpr int W4_ABUF = W2_HH * W2_HW * W_LD;             // halo buffer stride (unpadded: the last staging round is predicated)
constexpr int W4_IMG = (W2_TH / 2) * W2_TW * 32;          // one share image: [4 tile rows][16 columns][32 channels]
__global__ __launch_bounds__(256, 3) void k_conv_wino4(ConvArgs args) {
    // k-loop: two halo buffers; epilogue: four share images (one per wave) over the same memory
    __shared__ __attribute__((aligned(16))) float smem[2 * W4_ABUF];
    static_assert(4 * W4_IMG <= 2 * W4_ABUF, "LDS plan");
    W_STAMP(0)
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    int bid = blockIdx.x;
    if (args.xcd_swizzle & 1) {
        const int chunk = int(gridDim.x) >> 3;
        if (bid < (chunk << 3)) bid = (bid & 7) * chunk + (bid >> 3);
    }
    int j = 0;
#pragma unroll
    for (int k = 1; k < kMaxConvJobs; ++k) j += (k < args.njobs && bid >= args.job[k].block_begin) ? 1 : 0;   // independent kernarg loads
    const ConvJob& J = args.job[j];
    int local = bid - J.block_begin;
    const int n32 = local % J.n_tiles_n; local /= J.n_tiles_n;
    const int b = local / J.tiles_per_img; local %= J.tiles_per_img;
    const int tile_idx = local;
    const int ty0 = (local / J.tiles_x) * W2_TH, tx0 = (local % J.tiles_x) * W2_TW;
    const int h = J.h, w = J.w, cin = args.cin, cout = args.cout;

    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);             // frequency row of this wave
    const int i = lane & 31, half = lane >> 5;
    const int tr = i >> 3, tc = i & 7;
    const int prow = (2 * tr * W2_HW + 2 * tc) * W_LD + half * 4;
    const int xrow = u == 0 ? 0 : (u == 2 ? 2 : 1), yrow = u == 2 ? 1 : (u == 3 ? 3 : 2);
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(u == 1 ? 0x3F800000 : 0xBF800000));

    const int k8_total = cin / 8;
    const float* ub = J.wgt + ((size_t(n32) * k8_total) * 16 + u * 4) * 256;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, k8_total * 16 * 1024, 0x00020000);
    const int wlane = lane * 16;
    auto wfrag = [&](int step, int f) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (step * 16 + f) * 1024, 0));
    };
    const float* inb = J.in + size_t(b) * h * w * cin;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inb), 0, h * w * cin * 4, 0x00020000);
    unsigned goff[W2_ITEMS_PT];                                         // (this kernel has the registers for them)
#pragma unroll
    for (int it = 0; it < W2_ITEMS_PT; ++it) {
        const int pix = it * 32 + (tid >> 3);
        const int hy = pix / W2_HW, hx = pix - hy * W2_HW;
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        const bool ok = pix < W2_HH * W2_HW && gy >= 0 && gy < h && gx >= 0 && gx < w;
        goff[it] = ok ? unsigned((gy * w + gx) * cin + (tid & 7) * 4) * 4u : 0x80000000u;
    }
    const int lds_w = (tid >> 3) * W_LD + (tid & 7) * 4;
    const bool last_ok = (W2_ITEMS_PT - 1) * 32 + (tid >> 3) < W2_HH * W2_HW;   // the sixth round covers pixels 160..191 of 180
    auto item_load = [&](int it, int ch) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[it], ch * (W_KC * 4), 0));
    };
    auto item_store = [&](int it, int buf, f32x4 v) {
        if (it < W2_ITEMS_PT - 1 || last_ok) *reinterpret_cast<f32x4*>(smem + buf * W4_ABUF + lds_w + it * (32 * W_LD)) = v;
    };

    f32x16 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    const int nchunks = cin / W_KC;
    f32x4 VA[4], VB[4], ring[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { ring[f] = wfrag(0, f); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int it = 0; it < W2_ITEMS_PT; ++it) item_store(it, 0, item_load(it, 0));
    W_STAMP(7)
    __syncthreads();
    W_STAMP(4)
#define W4_LDS4(off) (*static_cast<const f32x4*>(__builtin_assume_aligned(reinterpret_cast<const char*>(smem) + (off), 16)))
#define W4_PIN(v) asm volatile("" : "+v"(v))
    int ax = (prow + xrow * W2_HW * W_LD) * 4, ay = (prow + yrow * W2_HW * W_LD) * 4, tog = W4_ABUF * 4;
    {
        f32x4 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 x = W4_LDS4(ax + c * (W_LD * 4)), y = W4_LDS4(ay + c * (W_LD * 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) t[c][e] = fmaf(sgn, y[e], x[e]);
        }
        VA[0] = t[0] - t[2]; VA[1] = t[1] + t[2]; VA[2] = t[2] - t[1]; VA[3] = t[1] - t[3];
    }

    // One k-step = 16 slots of {one MFMA + a small piece of the other work}: 8 patch reads + column pass for the next
    // step's operands (slots of f = 0, 1), its row pass (f = 2, 3), 4 weight fragments, 2 halo loads.
#define WINO4_STEP(Vc, Vn, K8)                                                                                        \
    {                                                                                                                 \
        const int step = chunk * 4 + (K8);                                                                            \
        const int nstep = (K8) < 3 ? step + 1 : gnext * 4;                                                            \
        if ((K8) == 3) { ax += tog; ay += tog; tog = -tog; }                 /* the next patch is in the other buffer */ \
        constexpr int koff = (((K8) + 1) & 3) * 32;                                                                   \
        constexpr int it0 = (K8) * 2, itn = (K8) == 3 ? 0 : 2;                                                        \
        f32x4 pf[2], cx0, cy0, cx1, cy1, t0, t1, t2, t3;                                                              \
        _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                               \
            const f32x4 bq = ring[f];                                                                                 \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][0], bq[0], acc[f], 0, 0, 0);                          \
            if (f < 2) {                                                                                              \
                cx0 = W4_LDS4(ax + koff + (2 * f) * (W_LD * 4)); cy0 = W4_LDS4(ay + koff + (2 * f) * (W_LD * 4));     \
                cx1 = W4_LDS4(ax + koff + (2 * f + 1) * (W_LD * 4)); cy1 = W4_LDS4(ay + koff + (2 * f + 1) * (W_LD * 4)); \
            }                                                                                                         \
            if (f == 2) { Vn[0] = t0 - t2; W4_PIN(Vn[0]); }                                                           \
            if (f == 3) { Vn[3] = t1 - t3; W4_PIN(Vn[3]); }                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][1], bq[1], acc[f], 0, 0, 0);                          \
            ring[f] = wfrag(nstep, f);                                                                                \
            if (f == 1 && itn) { pf[0] = item_load(it0, gnext); pf[1] = item_load(it0 + 1, gnext); }                  \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][2], bq[2], acc[f], 0, 0, 0);                          \
            if (f == 0) { _Pragma("unroll") for (int e = 0; e < 4; ++e) t0[e] = fmaf(sgn, cy0[e], cx0[e]); W4_PIN(t0); } \
            if (f == 1) { _Pragma("unroll") for (int e = 0; e < 4; ++e) t2[e] = fmaf(sgn, cy0[e], cx0[e]); W4_PIN(t2); } \
            if (f == 2) { Vn[1] = t1 + t2; W4_PIN(Vn[1]); }                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[f][3], bq[3], acc[f], 0, 0, 0);                          \
            if (f == 0) { _Pragma("unroll") for (int e = 0; e < 4; ++e) t1[e] = fmaf(sgn, cy1[e], cx1[e]); W4_PIN(t1); } \
            if (f == 1) { _Pragma("unroll") for (int e = 0; e < 4; ++e) t3[e] = fmaf(sgn, cy1[e], cx1[e]); W4_PIN(t3); } \
            if (f == 2) { Vn[2] = t2 - t1; W4_PIN(Vn[2]); }                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
        _Pragma("unroll") for (int t = 0; t < itn; ++t) item_store(it0 + t, (chunk + 1) & 1, pf[t]);                   \
        if ((K8) == 2) __syncthreads();                                                                               \
    }

    W_STAMP(1)
    __builtin_amdgcn_s_setprio(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int gnext = chunk + 1 < nchunks ? chunk + 1 : chunk;
        WINO4_STEP(VA, VB, 0)
        WINO4_STEP(VB, VA, 1)
        WINO4_STEP(VA, VB, 2)
        WINO4_STEP(VB, VA, 3)
    }
#undef WINO4_STEP
#undef W4_LDS4
#undef W4_PIN
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    W_STAMP(2)

    // ---- epilogue.  M[v] = acc[v] is row u of the frequency grid.  Column pass first (c0 = M0 + M1 + M2, c1 = M1 - M2 - M3
    // give the two pixels of a tile row); output row 0 of a tile is c(u0) + c(u1) + c(u2), row 1 is c(u1) - c(u2) - c(u3):
    // every wave writes its column-passed row as a share image [4 tile rows][16 columns][32 channels], then threads
    // owning 4 consecutive channels of a pixel combine three images, add the rank-1 tables, the residual and the bias
    // and store 16 bytes.
    const float* __restrict__ p_bias = J.bias;
    const float* __restrict__ p_bbias = J.bbias;
    const float* __restrict__ p_rcol = J.rcol;
    const float* __restrict__ p_rrow = J.rrow;
    const float* __restrict__ p_res = J.res;
    float* __restrict__ p_out = J.out;
    double* p_gn = J.gn_part;
    __syncthreads();                                     // all patch reads of the last step are done
    W_STAMP(5)
    {
        // wave u writes image u: its column-passed row, 32 values per lane (the LDS write port is what this phase waits for:
        // twelve waves of a CU arrive together)
        float* img = smem + u * W4_IMG + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r];
            const int ti = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int pp = ((ti >> 3) * W2_TW + 2 * (ti & 7)) * 32;
            img[pp] = m0 + m1 + m2; img[pp + 32] = m1 - m2 - m3;
        }
    }
    // finishing thread: channels co4..co4+3 of pixel column xl, rows rsel*4 .. rsel*4+3 of the tile
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
    if (p_rcol) {
        // the column table only depends on the row through its edge variant: away from the top / bottom edge of the image
        // (a block-uniform test) one load serves the four rows
        if (ty0 > 0 && ty0 + W2_TH < h) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p_rcol + ((size_t(b) * w + xc) * 4 + 0) * cout + coc);
#pragma unroll
            for (int k = 0; k < 4; ++k) tcol[k] = v0;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = ty0 + rsel * 4 + k;
                tcol[k] = *reinterpret_cast<const f32x4*>(p_rcol + ((size_t(b) * w + xc) * 4 + w_edge_variant(y < h ? y : 0, h)) * cout + coc);
            }
        }
    }
    if (p_rrow) {
        const int vx = w_edge_variant(xc, w);
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
    __syncthreads();                                     // the share images are complete
    W_STAMP(6)
    f32x4 gs4 = zero4, gss4 = zero4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yl = rsel * 4 + k, y = ty0 + yl;
        // output row 0 of a tile = c(u0) + c(u1) + c(u2), row 1 = c(u1) - c(u2) - c(u3)
        const float* sp = smem + (yl & 1) * W4_IMG + ((yl >> 1) * W2_TW + xl) * 32 + quad * 4;
        const f32x4 ka = *reinterpret_cast<const f32x4*>(sp), kb = *reinterpret_cast<const f32x4*>(sp + W4_IMG),
                    kc = *reinterpret_cast<const f32x4*>(sp + 2 * W4_IMG);
        const f32x4 sum3 = (yl & 1) ? (ka - kb) - kc : (ka + kb) + kc;
        const f32x4 v = (sum3 + base4) + ((tcol[k] + trow[k]) + tres[k]);
        if (x_ok && y < h) {
            *reinterpret_cast<f32x4*>(p_out + ((size_t(b) * h + y) * w + x) * cout + co4) = v;
            gs4 += v; gss4 += v * v;
        }
    }
    if (p_gn) {
        // per wave: 8 pixel columns (lanes l, l+8, ..) x 4 rows of 8 channel quads; one part per wave
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
    W_STAMP(3)
}

// ------------------------------------------------------------------ host side
static int wino_variant() {          // 24: mixed F(2x4,3x3) (s3d_wino24.hip), 4: F(2x2) one frequency row per wave, 2: two rows per wave, 0: direct kernel
    int v = opt(OPT_WINO);
    if (v == kOptUnset) v = kDefaultWino;
    if (v != 0 && v != 2 && v != 24) v = 4;
    return v;
}
double wino_exec_fraction() { return 4.0 / 9.0; }      // F(2x2,3x3): 16 multiplies per 2x2 outputs instead of 36
bool conv_use_wino() { return wino_variant() != 0 && !conv_use_naive(); }
bool conv_use_wino24() { return wino_variant() == 24 && !conv_use_naive(); }

void wino_gn_parts(const Geo& g, int nparts[3]) {    // one part per wave of a tile's block(s), both kernels
    for (int p = 0; p < 3; ++p) nparts[p] = ((g.w[p] + W2_TW - 1) / W2_TW) * ((g.h[p] + W2_TH - 1) / W2_TH) * 4;
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
    // one kernel for every shape and batch size: a sample's result must not depend on what it is batched with
    const bool four = wino_variant() != 2;          // (24: layers the mixed kernel does not take, and the training tier)
    int blocks = 0;
    for (int j = 0; j < a.njobs; ++j) {
        ConvJob& J = a.job[j];
        S3D_CHECK(size_t(J.h) * J.w * a.cin * 4 < (size_t(1) << 31), S3D_ERR_INVALID, "wino conv: a plane of one sample must stay below 2 GiB");
        J.tiles_x = (J.w + W2_TW - 1) / W2_TW;
        J.tiles_per_img = J.tiles_x * ((J.h + W2_TH - 1) / W2_TH);
        J.n_tiles_n = four ? (a.cout + 31) / 32 : (a.cout + 63) / 64;
        J.block_begin = blocks;
        blocks += J.tiles_per_img * J.n_tiles_n * a.B;
    }
    if (!blocks) return 0;
    a.xcd_swizzle = 1 | 2;                   // XCD-aware block order + raised priority outside the k-loop (were switchable in rounds 1-2: always wins)
    conv_note_kernel(four ? "k_conv_wino4 Winograd F(2x2,3x3), one frequency row per wave" : "k_conv_wino2 Winograd F(2x2,3x3), two waves per SIMD");
    if (four) hipLaunchKernelGGL(k_conv_wino4, dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_conv_wino2, dim3(blocks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

}  // namespace s3d
