// s3d_bwd.hip — backward kernels of the triplane UNet (training tier).
//
// Everything the reference gets from autograd for TriplaneUNetModelSmall (src/diffusion/unet_triplane.py), written
// against the same NHWC-per-plane layout as the forward kernels.  The rollout structure is exploited in the backward
// pass too: the two mean channel blocks of a TriplaneConv are constant along one image axis, so
//   * their input gradient is a 1-D transposed convolution of row / column sums of dy   (edge sums -> k_rank1),
//   * their weight gradient is a small GEMM between those sums and the mean vectors      (k_slot_wgrad),
// and only the plane's own channels need the dense dgrad (the forward conv kernels with a transposed operator) and
// the dense wgrad (k_wgrad_mfma).  All reductions are two-stage with a fixed order: bit-repeatable, no float atomics.
#include "s3d_bwd.h"
#include <hip/hip_ext.h>

namespace s3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// ------------------------------------------------------------------ row / column sums of dy with edge variants
// R[b][r][j][co]  = sum over columns c of dy[b][r][c][co] restricted by the horizontal tap j: j=0 -> c >= 1,
//                   j=1 -> all c, j=2 -> c <= w-2   (the columns whose tap j stays inside the image)
// Cs[b][c][j][co] = the same with rows and columns exchanged.
struct EdgeArgs {
    const float* dy[3]; float* R[3]; float* Cs[3];
    int h[3], w[3];
    int B, C;
    int begin[7];             // block prefix over the 6 (plane, row|col) jobs, per sample
};
__global__ __launch_bounds__(256) void k_edge_sums(EdgeArgs a) {
    // one block per (sample, line); a thread owns a float4 of channels and every pl-th interior element of the line
    extern __shared__ __attribute__((aligned(16))) float sm_es[];       // [pl][C]
    const int per = a.begin[6];
    const int b = blockIdx.x / per;
    int r = blockIdx.x % per, v = 0;
    while (r >= a.begin[v + 1]) ++v;
    r -= a.begin[v];
    const int p = v >> 1, is_col = v & 1;
    const int h = a.h[p], w = a.w[p], C = a.C, cq = C / 4;
    const int n = is_col ? h : w;                     // length of the summed axis
    const size_t stride = (is_col ? size_t(w) * C : size_t(C)) / 4;
    const float4* base = reinterpret_cast<const float4*>(a.dy[p] + size_t(b) * h * w * C + (is_col ? size_t(r) * C : size_t(r) * w * C));
    float* out = (is_col ? a.Cs[p] + (size_t(b) * w + r) * 3 * C : a.R[p] + (size_t(b) * h + r) * 3 * C);
    const int pl = blockDim.x / cq, q = threadIdx.x % cq, l = threadIdx.x / cq;
    float4 mid = make_float4(0, 0, 0, 0);
    for (int k = 1 + l; k < n - 1; k += pl) {
        const float4 t = base[size_t(k) * stride + q];
        mid.x += t.x; mid.y += t.y; mid.z += t.z; mid.w += t.w;
    }
    reinterpret_cast<float4*>(sm_es)[l * cq + q] = mid;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float m = 0.f;
        for (int ll = 0; ll < pl; ++ll) m += sm_es[ll * C + c];
        const float first = reinterpret_cast<const float*>(base)[c];
        const float last = reinterpret_cast<const float*>(base + size_t(n - 1) * stride)[c];
        float s0, s1, s2;
        if (n >= 2) { s0 = m + last; s2 = first + m; s1 = first + m + last; }
        else { s0 = 0.f; s2 = 0.f; s1 = first; }
        out[c] = s0; out[C + c] = s1; out[2 * C + c] = s2;
    }
}
int launch_edge_sums(const Tri& dy, int B, float* const R[3], float* const Cs[3], hipStream_t st, hipEvent_t stop) {
    EdgeArgs a;
    a.B = B; a.C = dy.C; a.begin[0] = 0;
    S3D_CHECK(dy.C % 4 == 0 && dy.C <= 1024, S3D_ERR_INVALID, "edge_sums: C=%d", dy.C);
    for (int p = 0; p < 3; ++p) {
        a.dy[p] = dy.p[p]; a.R[p] = R[p]; a.Cs[p] = Cs[p]; a.h[p] = dy.g.h[p]; a.w[p] = dy.g.w[p];
        a.begin[2 * p + 1] = a.begin[2 * p] + dy.g.h[p];
        a.begin[2 * p + 2] = a.begin[2 * p + 1] + dy.g.w[p];
    }
    if (!B || !a.begin[6]) return 0;
    const int cq = dy.C / 4, pl = std::max(1, 256 / cq);
    // stop: an event tied to THIS launch's completion signal (hipExtLaunchKernelGGL) — a separate hipEventRecord behind the launch is
    // a barrier packet of its own, which opens a 6-8-us gap in front of the stream's next kernel
    if (stop) hipExtLaunchKernelGGL(k_edge_sums, dim3(a.begin[6] * B), dim3(cq * pl), size_t(pl) * dy.C * sizeof(float), st, nullptr, stop, 0, a);
    else hipLaunchKernelGGL(k_edge_sums, dim3(a.begin[6] * B), dim3(cq * pl), size_t(pl) * dy.C * sizeof(float), st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ bias gradients from the row sums
// dbias[p][co] = sum_b sum_r R[p][b][r][1][co]; optionally the per-sample sums over all planes (the gradient of
// emb_out when it is added to h, use_scale_shift_norm=False, src/diffusion/unet_triplane.py:298-303).
struct BiasArgs { const float* R[3]; float* dbias[3]; float* per_sample; int per_sample_stride; int h[3]; int B, C; };
constexpr int kBiasCh = 16, kBiasLanes = 1024 / kBiasCh;   // channels per block x row lanes (few channels: more blocks for a latency-bound sum)
__device__ __forceinline__ void bias_grad_block(const BiasArgs& a, int block) {
    // All (sample, plane) partial sums are accumulated first (independent loads in flight), then reduced over the lanes
    // through LDS in lane order, four samples at a time.
    __shared__ float sm[12][kBiasLanes][kBiasCh];
    __shared__ float sm2[12][kBiasCh];
    const int cl = threadIdx.x % kBiasCh, lane = threadIdx.x / kBiasCh, co = block * kBiasCh + cl;
    float tot_p[3] = {0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < a.B; b0 += 4) {
        const int nb = min(4, a.B - b0);
        float s[4][3];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                float acc = 0.f;
                if (bb < nb && co < a.C) {
                    const float* R = a.R[p] + size_t(b0 + bb) * a.h[p] * 3 * a.C + a.C + co;
                    for (int r = lane; r < a.h[p]; r += kBiasLanes) acc += R[size_t(r) * 3 * a.C];
                }
                s[bb][p] = acc;
            }
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int p = 0; p < 3; ++p) sm[bb * 3 + p][lane][cl] = s[bb][p];
        __syncthreads();
        // one thread per (sample, plane, channel) adds the row lanes in lane order; then one per channel combines the planes
        if (threadIdx.x < 12 * kBiasCh) {
            const int bp = threadIdx.x / kBiasCh, c2 = threadIdx.x % kBiasCh;
            float t = 0.f;
#pragma unroll 8
            for (int k = 0; k < kBiasLanes; ++k) t += sm[bp][k][c2];
            sm2[bp][c2] = t;
        }
        __syncthreads();
        if (lane == 0 && co < a.C)
            for (int bb = 0; bb < nb; ++bb) {
                float tot = 0.f;
                for (int p = 0; p < 3; ++p) {
                    const float t = sm2[bb * 3 + p][cl];
                    tot += t; tot_p[p] += t;
                }
                if (a.per_sample) a.per_sample[size_t(b0 + bb) * a.per_sample_stride + co] = tot;
            }
    }
    if (lane == 0 && co < a.C)
        for (int p = 0; p < 3; ++p)
            if (a.dbias[p]) a.dbias[p][co] = tot_p[p];
}
__global__ __launch_bounds__(1024) void k_bias_grad(BiasArgs a) { bias_grad_block(a, blockIdx.x); }
// every queued job in one launch (DeferredTail): block -> (job, channel block) through a prefix table
constexpr int kTailJobs = 12;
struct BiasMultiArgs { BiasArgs job[kTailJobs]; int begin[kTailJobs + 1]; int njobs; };
__global__ __launch_bounds__(1024) void k_bias_grad_multi(BiasMultiArgs m) {
    int j = 0;
#pragma unroll
    for (int k = 1; k < kTailJobs; ++k) j += (k < m.njobs && int(blockIdx.x) >= m.begin[k]) ? 1 : 0;
    bias_grad_block(m.job[j], int(blockIdx.x) - m.begin[j]);
}
int launch_bias_grad_deferred(float* const R[3], const Geo& g, int C, int B, float* const dbias[3], float* per_sample,
                              int per_sample_stride, DeferredTail& tail) {
    DeferredTail::Bias b;
    for (int p = 0; p < 3; ++p) { b.R[p] = R[p]; b.dbias[p] = dbias ? dbias[p] : nullptr; b.h[p] = g.h[p]; }
    b.per_sample = per_sample; b.per_sample_stride = per_sample_stride; b.B = B; b.C = C;
    tail.bias.push_back(b);
    return 0;
}
int launch_bias_grad(float* const R[3], const Geo& g, int C, int B, float* const dbias[3], float* per_sample,
                     int per_sample_stride, hipStream_t st) {
    BiasArgs a;
    for (int p = 0; p < 3; ++p) { a.R[p] = R[p]; a.dbias[p] = dbias ? dbias[p] : nullptr; a.h[p] = g.h[p]; }
    a.per_sample = per_sample; a.per_sample_stride = per_sample_stride; a.B = B; a.C = C;
    hipLaunchKernelGGL(k_bias_grad, dim3(cdiv(C, kBiasCh)), dim3(1024), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ weight gradient of the two mean-channel slots
// row-varying slot:  dW[co][slot*C+ci][dr][dc] = sum_b sum_r' v[b][r'][ci] * R[b][r'-dr+1][dc][co]
// col-varying slot:  dW[co][slot*C+ci][dr][dc] = sum_b sum_c' v[b][c'][ci] * Cs[b][c'-dc+1][dr][co]
// One block = one 32(co) x 32(ci) tile of one slot and one tap offset t along the varying axis (three MFMA accumulators,
// one per tap of the other axis); the contraction over (b, position) is split across the block's sixteen waves and
// their partials are added through LDS in wave order.  The loop is bound by the latency of its operand loads (one
// step = 4 dwords from L2 per lane for 3 MFMAs), so what counts is waves: 3 x 16 per tile instead of 4.
struct SlotJob { const float* v; const float* S; float* dW; int L, slot, col_varying; };
struct SlotArgs { SlotJob job[6]; int B, C, cout, ctot, n_ci; };
constexpr int kSlotWaves = 16;
__global__ __launch_bounds__(64 * kSlotWaves) void k_slot_wgrad(SlotArgs a) {
    __shared__ float red[kSlotWaves][16][64];
    const SlotJob J = a.job[blockIdx.y];
    const int tci = blockIdx.x % a.n_ci, tco = blockIdx.x / a.n_ci, t = blockIdx.z;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, i = lane & 31, kk = lane >> 5;
    const int co = tco * 32 + i, ci = tci * 32 + i, L = J.L;
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int pairs = (L + 1) / 2, total = a.B * pairs;
    // one-deep software pipeline: the four operand loads of step n+1 are in flight during the MFMAs of step n (four-deep:
    // 18.0 us unchanged at 72 blocks, 21 against 18.6 at 288 — the launch is its 3 x 16-wave LDS reductions and stores)
    float bv = 0.f, av[3];
    auto load = [&](int it, float& b_out, float (&a_out)[3]) {
        const int b = it / pairs, pos = (it % pairs) * 2 + kk;
        const bool in = it < total && pos < L;
        b_out = in ? J.v[(size_t(b) * L + pos) * a.C + ci] : 0.f;
        const int q = pos - t + 1;
        const bool ok = in && q >= 0 && q < L;
        const float* S = J.S + (size_t(b) * L + (ok ? q : 0)) * 3 * a.cout + co;
#pragma unroll
        for (int j = 0; j < 3; ++j) a_out[j] = ok ? S[j * a.cout] : 0.f;
    };
    load(wid, bv, av);
    for (int it = wid; it < total; it += kSlotWaves) {
        float nb, na[3];
        load(it + kSlotWaves, nb, na);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bv = nb;
#pragma unroll
        for (int j = 0; j < 3; ++j) av[j] = na[j];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wid][r][lane] = acc[j][r];
        __syncthreads();
        const int dr = J.col_varying ? j : t, dc = J.col_varying ? t : j;
        for (int e = threadIdx.x; e < 1024; e += 64 * kSlotWaves) {
            const int r = e >> 6, l = e & 63;
            float v = red[0][r][l];
#pragma unroll
            for (int k = 1; k < kSlotWaves; ++k) v += red[k][r][l];
            const int oc = tco * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ic = tci * 32 + (l & 31);
            J.dW[(size_t(oc) * a.ctot + J.slot * a.C + ic) * 9 + dr * 3 + dc] = v;
        }
    }
}
int launch_slot_wgrad(const SlotWgradArgs& s, hipStream_t st) {
    S3D_CHECK(s.C % 32 == 0 && s.cout % 32 == 0, S3D_ERR_INVALID, "slot_wgrad: channels must be multiples of 32");
    SlotArgs a;
    a.B = s.B; a.C = s.C; a.cout = s.cout; a.ctot = 3 * s.C; a.n_ci = s.C / 32;
    for (int p = 0; p < 3; ++p) {
        const bool a_is_col = (p == 0);
        // row-varying vector (length h) and column-varying vector (length w) of this plane's conv
        SlotJob& jr = a.job[2 * p];
        jr.v = s.rowvec[p]; jr.S = s.R[p]; jr.dW = s.dW[p]; jr.L = s.g.h[p]; jr.slot = a_is_col ? 2 : 1; jr.col_varying = 0;
        SlotJob& jc = a.job[2 * p + 1];
        jc.v = s.colvec[p]; jc.S = s.Cs[p]; jc.dW = s.dW[p]; jc.L = s.g.w[p]; jc.slot = a_is_col ? 1 : 2; jc.col_varying = 1;
    }
    hipLaunchKernelGGL(k_slot_wgrad, dim3((s.cout / 32) * a.n_ci, 6, 3), dim3(64 * kSlotWaves), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ dense weight gradient on the fp32 MFMA
// dW[co][ci][dr][dc] = sum_{b,r,c} dy[b][r][c][co] * a[b][r+dr-1][c+dc-1][ci]          (3x3, pad 1;  1x1: no shift)
// GEMM view: M = co, N = ci (x taps), K = pixels.  v_mfma_f32_32x32x2_f32 takes K = 2 pixels per instruction:
// A[i][k] = dy[pixel k][co0+i], B[k][j] = a[pixel k + tap][ci0+j] — both operands are runs of 32 consecutive
// channels of one pixel, i.e. exactly what NHWC stores, so neither needs a transpose.  LDS holds the pixel tile
// as 32-channel panels [panel][pixel][32] so the two k-halves of a wave hit disjoint bank halves.
// One block = 64 co x 64 ci x all taps for a slice of the pixel tiles (split-K); partials are added in slice order
// by k_wgrad_reduce, which also scatters into the reference's OIHW layout.
constexpr int WG_TR = 4, WG_TC = 16;                 // pixel tile
// TAPS: 9 (3x3) or 1; TROWS: kernel rows handled by one block (3 = all nine taps, 1 = one row of three taps: three
// times the blocks and a third of the accumulators, for layers too small to fill the chip otherwise)
template <int TAPS, int TROWS, int WT>
struct WgSmem {
    static constexpr int KW = TAPS == 25 ? 5 : (TAPS == 9 ? 3 : 1), HALO = KW / 2;
    static constexpr int AR = WG_TR + (TROWS == 3 ? 2 * HALO : 0), AC = WG_TC + 2 * HALO;
    float dy[2 * WT][WG_TR * WG_TC][32];
    float a[2 * WT][AR * AC][32];
};
struct WgJob { const float* dy; const float* a; float* part; int h, w, tiles_x, tiles; int block_begin; };
struct WgArgs { WgJob job[3]; int B, cin, cout, a_cstride, ksplit, n_co, n_ci, njobs; };

// WT: 32x32 MFMA tiles per wave in each direction (block tile = 64*WT co x 64*WT ci).  The 1x1 case (the decoder MLPs'
// weight gradients, K = 65 536 points) uses WT = 2: with a single tap every staged element feeds only one MFMA per
// output tile, so the larger block tile halves the staging traffic per flop.
template <int TAPS, int TROWS, int WT>
__global__ __launch_bounds__(256) void k_wgrad_mfma(WgArgs args) {
    using SM = WgSmem<TAPS, TROWS, WT>;
    __shared__ __attribute__((aligned(16))) SM sm;
    constexpr int HALO = SM::HALO, AR = SM::AR, AC = SM::AC, KW = SM::KW, NP = 2 * WT, BT = 64 * WT;
    static_assert(TROWS == 1 || TAPS == 9, "all kernel rows in one block only for 3x3");
    static_assert(WT == 1 || TAPS == 1, "wide tiles only for 1x1");
    constexpr int NT = TAPS == 1 ? 1 : KW * TROWS;                    // taps of this block
    constexpr int NDY = (NP * WG_TR * WG_TC * 8) / 256, NA = (NP * AR * AC * 8 + 255) / 256;
    int p = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k) p += (k < args.njobs && int(blockIdx.x) >= args.job[k].block_begin) ? 1 : 0;   // independent kernarg loads
    const WgJob& J = args.job[p];
    int local = blockIdx.x - J.block_begin;
    int dr0 = 0;
    if (TAPS > 1 && TROWS == 1) { dr0 = local % KW; local /= KW; }
    const int ks = local % args.ksplit; local /= args.ksplit;
    const int tci = local % args.n_ci, tco = local / args.n_ci;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 31, kk = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int co0 = tco * BT, ci0 = tci * BT;
    f32x16 acc[NT][WT][WT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < WT; ++m)
#pragma unroll
            for (int n = 0; n < WT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][m][n][r] = 0.f;

    const long long total = (long long)J.tiles * args.B;
    const long long t_begin = total * ks / args.ksplit, t_end = total * (ks + 1) / args.ksplit;
    // the a-tile's first row relative to the dy tile: all three kernel rows need rows -1..TR, a single row dr0 needs dr0-1..
    const int arow0 = TROWS == 3 ? -HALO : dr0 - HALO;
    float4 rdy[NDY], ra[NA];
    auto load_tile = [&](long long tt) {
        const int b = int(tt / J.tiles), tile = int(tt % J.tiles);
        const int r0 = (tile / J.tiles_x) * WG_TR, c0 = (tile % J.tiles_x) * WG_TC;
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int it = k * 256 + tid;
            const int q = it & 7, px = (it >> 3) % (WG_TR * WG_TC), pan = it / (8 * WG_TR * WG_TC);
            const int r = r0 + px / WG_TC, c = c0 + px % WG_TC, ch = co0 + pan * 32 + q * 4;
            const bool ok = r < J.h && c < J.w && ch < args.cout;
            rdy[k] = ok ? *reinterpret_cast<const float4*>(J.dy + ((size_t(b) * J.h + r) * J.w + c) * args.cout + ch) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int it = k * 256 + tid;
            const int q = it & 7, px = (it >> 3) % (AR * AC), pan = it / (8 * AR * AC);
            const int r = r0 + arow0 + px / AC, c = c0 - HALO + px % AC, ch = ci0 + pan * 32 + q * 4;
            const bool ok = pan < NP && r >= 0 && r < J.h && c >= 0 && c < J.w && ch < args.cin;
            ra[k] = ok ? *reinterpret_cast<const float4*>(J.a + ((size_t(b) * J.h + r) * J.w + c) * args.a_cstride + ch) : make_float4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int it = k * 256 + tid;
            const int q = it & 7, px = (it >> 3) % (WG_TR * WG_TC), pan = it / (8 * WG_TR * WG_TC);
            *reinterpret_cast<float4*>(&sm.dy[pan][px][q * 4]) = rdy[k];
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int it = k * 256 + tid;
            const int q = it & 7, px = (it >> 3) % (AR * AC), pan = it / (8 * AR * AC);
            if (pan < NP) *reinterpret_cast<float4*>(&sm.a[pan][px][q * 4]) = ra[k];
        }
    };
    if (t_begin < t_end) load_tile(t_begin);
    for (long long tt = t_begin; tt < t_end; ++tt) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (tt + 1 < t_end) load_tile(tt + 1);            // in flight while the matrix cores work on this tile
        __builtin_amdgcn_sched_barrier(0);
        // one step = one pixel pair (lane half kk takes pixel c + kk); the operands of step s + 1 are read from LDS before the
        // MFMAs of step s are issued, so their latency runs under the matrix pipe (it used to sit in front of every MFMA group)
        constexpr int S = WG_TR * WG_TC / 2;
        float avc[WT], bvc[NT][WT], avn[WT], bvn[NT][WT];
        auto lds_step = [&](int st, float (&av)[WT], float (&bv)[NT][WT]) {
            const int r = st / (WG_TC / 2), c = (st % (WG_TC / 2)) * 2;
#pragma unroll
            for (int m = 0; m < WT; ++m) av[m] = sm.dy[wm * WT + m][r * WG_TC + c + kk][i];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int dr = TROWS == 3 ? t / KW : 0, dc = t % KW;
#pragma unroll
                for (int n = 0; n < WT; ++n) bv[t][n] = sm.a[wn * WT + n][(r + dr) * AC + c + kk + dc][i];
            }
        };
        lds_step(0, avc, bvc);
#pragma unroll
        for (int st = 0; st < S; ++st) {
            if (st + 1 < S) lds_step(st + 1, avn, bvn);
            __builtin_amdgcn_sched_barrier(0);                 // (left alone, the scheduler sinks the reads back to their uses)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int n = 0; n < WT; ++n)
#pragma unroll
                    for (int m = 0; m < WT; ++m) acc[t][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(avc[m], bvc[t][n], acc[t][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < WT; ++m) avc[m] = avn[m];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int n = 0; n < WT; ++n) bvc[t][n] = bvn[t][n];
        }
    }
    // partial [ks][co][ci][TAPS]
    float* part = J.part + size_t(ks) * args.cout * args.cin * TAPS;
#pragma unroll
    for (int m = 0; m < WT; ++m)
#pragma unroll
        for (int n = 0; n < WT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * WT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk, ci = ci0 + (wn * WT + n) * 32 + i;
                if (co >= args.cout || ci >= args.cin) continue;
                float* d = part + (size_t(co) * args.cin + ci) * TAPS + dr0 * KW;
#pragma unroll
                for (int t = 0; t < NT; ++t) d[t] = acc[t][m][n][r];
            }
}

// ------------------------------------------------------------------ Winograd weight gradient of the 3x3 convolutions
// Forward: Y = A^T [ (G g G^T) .* (B^T d B) ] A per 2x2 output tile (s3d_wino.hip).  Hence
//   dU[u][v][co][ci] = sum over tiles of (A dY A^T)[u][v][co] * (B^T d B)[u][v][ci]      (16 GEMMs, K = tiles)
//   dg = G^T dU G                                                                          (per slice, in the block's epilogue)
// 16 multiplies per tile and channel pair instead of 36: 2.25x fewer MFMA flops than the direct weight gradient.
// A block owns one 32(co) x 32(ci) pair and a slice of the pixel regions (8x16 pixels = 32 Winograd tiles, the
// forward kernel's tile); its four waves own one row u of the frequency grid each (four accumulators).  A region's
// input halo and dy tile are staged in LDS; a lane holds ONE channel (co for the
// A operand, ci for the B operand) and, per MFMA step, one of two tiles (lane half = k index): it reads the 2 x 4
// patch values and 2 x 2 dy values of its tile with ds_read_b32, transforms them in registers and feeds four MFMAs.
constexpr int WW_TH = 8, WW_TW = 16, WW_HH = WW_TH + 2, WW_HW = WW_TW + 2, WW_LD = 36;
struct WgwJob { const float* dy; const float* a; float* part; int h, w, tiles_x, regions; int block_begin; };
struct WgwArgs { WgwJob job[3]; int B, cin, cout, a_cstride, ksplit, n_co, n_ci, njobs; };
#ifdef WGW_TIMING
__device__ unsigned long long* g_wgwtime;   // tools/wgrad_ubench.hip: per block {entry, exit, sum over regions of (operands in LDS - request), sum of (last MFMA issued - operands in LDS)}, 10-ns ticks
#define WGW_NOW() wall_clock64()
#else
#define WGW_NOW() 0ull
#endif
__global__ __launch_bounds__(256, 3) void k_wgrad_wino(WgwArgs args) {
    __shared__ __attribute__((aligned(16))) float sx[WW_HH * WW_HW * WW_LD];
    __shared__ __attribute__((aligned(16))) float sdy[WW_TH * WW_TW * WW_LD];
    int p = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k) p += (k < args.njobs && int(blockIdx.x) >= args.job[k].block_begin) ? 1 : 0;   // independent kernarg loads
    const WgwJob& J = args.job[p];
    int local = blockIdx.x - J.block_begin;
    const int ks = local % args.ksplit; local /= args.ksplit;
    const int tci = local % args.n_ci, tco = local / args.n_ci;
    const int co0 = tco * 32, ci0 = tci * 32;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, half = lane >> 5;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);
    // row u of B^T d B combines two patch rows, t = x + s*y: (d0,d2,-), (d1,d2,+), (d2,d1,-), (d1,d3,-);
    // row u of A dY A^T combines the two dy rows with (c0, c1) = (1,0), (1,1), (1,-1), (0,-1)
    const int xrow = u == 0 ? 0 : (u == 2 ? 2 : 1), yrow = u == 2 ? 1 : (u == 3 ? 3 : 2);
    const float sgn = u == 1 ? 1.f : -1.f;
    const float c0 = u == 3 ? 0.f : 1.f, c1 = u == 0 ? 0.f : (u == 1 ? 1.f : -1.f);
    f32x16 acc[4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;

    const long long total = (long long)J.regions * args.B;
    const long long r_begin = total * ks / args.ksplit, r_end = total * (ks + 1) / args.ksplit;
    constexpr int NX = (WW_HH * WW_HW * 8 + 255) / 256, NDY = (WW_TH * WW_TW * 8) / 256;
    float4 rx[NX], rdy[NDY];
    const int q = tid & 7, prow = tid >> 3;
    // raw buffer loads: one byte offset per item, and an offset beyond the descriptor's range for everything outside the
    // image (the hardware answers with zeros): no predicated branches around ten loads, 32-bit addressing
    auto load_region = [&](long long rr) {
        const int b = int(rr / J.regions), reg = int(rr % J.regions);
        const int ty0 = (reg / J.tiles_x) * WW_TH, tx0 = (reg % J.tiles_x) * WW_TW;
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(J.a + size_t(b) * J.h * J.w * args.a_cstride), 0, J.h * J.w * args.a_cstride * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(J.dy + size_t(b) * J.h * J.w * args.cout), 0, J.h * J.w * args.cout * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int pix = k * 32 + prow, hy = pix / WW_HW, hx = pix - hy * WW_HW;
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
            const bool ok = pix < WW_HH * WW_HW && gy >= 0 && gy < J.h && gx >= 0 && gx < J.w;
            const unsigned off = ok ? unsigned((gy * J.w + gx) * args.a_cstride + ci0 + q * 4) * 4u : 0x80000000u;
            rx[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsa, off, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int pix = k * 32 + prow, y = ty0 + pix / WW_TW, x = tx0 + pix % WW_TW;
            const bool ok = y < J.h && x < J.w;
            const unsigned off = ok ? unsigned((y * J.w + x) * args.cout + co0 + q * 4) * 4u : 0x80000000u;
            rdy[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsd, off, 0, 0));
        }
    };
    auto store_region = [&]() {
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int pix = k * 32 + prow;
            if (pix < WW_HH * WW_HW) *reinterpret_cast<float4*>(sx + pix * WW_LD + q * 4) = rx[k];
        }
#pragma unroll
        for (int k = 0; k < NDY; ++k) *reinterpret_cast<float4*>(sdy + (k * 32 + prow) * WW_LD + q * 4) = rdy[k];
    };
    // operands of MFMA step st (two tiles, one per lane half): 8 patch values, 4 dy values
    struct Ops { float x[4], y[4], d0[2], d1[2]; };
    auto lds_step = [&](int st, Ops& o) {
        // the two lane halves take tiles FOUR columns apart: 8 pixels x 36 floats = 32 banks, so the 64 lanes of a ds_read_b32
        // cover all 64 banks once (adjacent tiles, 8 banks apart, made 24 of the 32 lanes of each half collide)
        const int tr = st >> 2, tc = (st & 3) + 4 * half;
        const float* px = sx + ((2 * tr + xrow) * WW_HW + 2 * tc) * WW_LD + i;
        const float* py = sx + ((2 * tr + yrow) * WW_HW + 2 * tc) * WW_LD + i;
        const float* pd = sdy + ((2 * tr) * WW_TW + 2 * tc) * WW_LD + i;
#pragma unroll
        for (int b = 0; b < 4; ++b) { o.x[b] = px[b * WW_LD]; o.y[b] = py[b * WW_LD]; }
#pragma unroll
        for (int x = 0; x < 2; ++x) { o.d0[x] = pd[x * WW_LD]; o.d1[x] = pd[(WW_TW + x) * WW_LD]; }
    };
    // (measured and dropped: starting the CU's three blocks a third of a region period apart with s_sleep — 60 us against 57)
    const unsigned long long wt_entry = WGW_NOW();
    unsigned long long wt_stage = 0, wt_mfma = 0;
    (void)wt_entry;
    for (long long rr = r_begin; rr < r_end; ++rr) {
        const unsigned long long wt0 = WGW_NOW();
        // no register prefetch of the next region (40 VGPRs that made the allocator spill): with three blocks per CU the
        // other waves of the SIMD keep the matrix pipe busy while this one waits for its loads
        load_region(rr);
        __syncthreads();                                  // everyone is done with the previous region
        store_region();
        __syncthreads();
        const unsigned long long wt1 = WGW_NOW();
        Ops cur, nxt;
        lds_step(0, cur);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            if (st + 1 < 16) lds_step(st + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            // B^T d B, row u: column pass then row pass
            float t[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) t[b] = fmaf(sgn, cur.y[b], cur.x[b]);
            const float V[4] = {t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]};
            // A dY A^T, row u
            const float r0 = c0 * cur.d0[0] + c1 * cur.d1[0], r1 = c0 * cur.d0[1] + c1 * cur.d1[1];
            const float M[4] = {r0, r0 + r1, r0 - r1, -r1};
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(M[v], V[v], acc[v], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        wt_stage += wt1 - wt0; wt_mfma += WGW_NOW() - wt1;
    }
#ifdef WGW_TIMING
    const unsigned long long wt_loop_end = WGW_NOW();
#endif
    // dg = G^T dU G of THIS slice before it leaves the block (9 values per channel pair instead of 16: the partials are what
    // this launch writes and the reduce kernel reads — 50 MB each way at 768 blocks, a fifth of the two kernels' time).
    // Column pass in registers (v: 4 -> 3, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]); row pass across the four waves
    // through LDS, one column j at a time: wave w finishes accumulator registers 4w .. 4w+3.
    float* red = sx;                                       // [4 waves][16 regs][64 lanes] = 16 KB of the 26 KB halo buffer
    static_assert(WW_HH * WW_HW * WW_LD >= 4 * 16 * 64, "row-pass scratch");
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        __syncthreads();                                   // the k-loop's last reads / the previous column's reads are done
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h1 = 0.5f * acc[1][r], h2 = 0.5f * acc[2][r];
            const float t = j == 0 ? acc[0][r] + (h1 + h2) : (j == 1 ? h1 - h2 : (h1 + h2) + acc[3][r]);
            red[(u * 16 + r) * 64 + lane] = t;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = u * 4 + rr;
            const float t0 = red[(0 * 16 + r) * 64 + lane], t1 = red[(1 * 16 + r) * 64 + lane];
            const float t2 = red[(2 * 16 + r) * 64 + lane], t3 = red[(3 * 16 + r) * 64 + lane];
            const float h1 = 0.5f * t1, h2 = 0.5f * t2;
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float* dst = J.part + ((size_t(ks) * 9 + j) * args.cout + co) * args.cin + ci0 + i;     // [ks][tap = 3*row + j][cout][cin]
            const size_t tap_stride = size_t(3) * args.cout * args.cin;
            dst[0] = t0 + (h1 + h2); dst[tap_stride] = h1 - h2; dst[2 * tap_stride] = (h1 + h2) + t3;
        }
    }
#ifdef WGW_TIMING
    if (tid == 0 && g_wgwtime) {
        unsigned long long* o = g_wgwtime + size_t(blockIdx.x) * 8;
        o[0] = wt_entry; o[1] = WGW_NOW(); o[2] = wt_stage; o[3] = wt_mfma; o[4] = wt_loop_end; o[5] = (unsigned long long)(r_end - r_begin);
    }
#endif
}
// ------------------------------------------------------------------ k_wgrad_wino with its operands staged by LDS-DMA (round 6)
// profiles/r06_wgrad.txt: a region of k_wgrad_wino is request -> wait -> LDS store -> barrier (1.8 us, nothing of this block on the
// matrix pipe) followed by 16 MFMA steps (4.9 us with the SIMD's three waves taking turns) — 0.77 of the pipe inside the loop.
// Here a region is cut into its two HALVES of tile rows (steps 0..7 / 8..15 of the same step order: the SAME products are added in
// the SAME order — bit-identical partials), each half has its own LDS buffer (6 halo rows + 4 dy rows, 23.5 KB; the two together are
// what one region took before), and half h + 1 arrives by LDS-DMA (buffer_load ... lds: no staging registers — the 40 that a
// register prefetch needs made the allocator spill —, no ds_write pass) while half h is multiplied.  The kernel has no other
// vector-memory traffic in its loop, so retiring a half is a plain vmcnt(0) + the barrier that also frees the other buffer.
// LDS image (lane-linear per DMA instruction, so no padding): pixel slot of halo column hx = 2 hx (hx < 8), 2 (hx - 8) + 1
// (8 <= hx < 16), 16 / 18 (hx = 16 / 17) in rows of 20 slots — the two lane halves of a ds_read_b32 take tiles 8 pixels apart, and
// slots of opposite parity put their 128-byte rows on the two halves of the 64 banks; dy columns likewise (x < 8: 2 x, else 2 (x - 8) + 1).
constexpr int WD_XROWS = 6, WD_XSLOTS = 20, WD_DYROWS = 4, WD_DYSLOTS = 16;
constexpr int WD_X_FLOATS = WD_XROWS * WD_XSLOTS * 32, WD_DY_FLOATS = WD_DYROWS * WD_DYSLOTS * 32, WD_HALF_FLOATS = WD_X_FLOATS + WD_DY_FLOATS;
constexpr int WD_X_INSTR = WD_XROWS * WD_XSLOTS / 8, WD_DY_INSTR = WD_DYROWS * WD_DYSLOTS / 8, WD_INSTR = WD_X_INSTR + WD_DY_INSTR;     // 15 + 8 DMA instructions of 1 KB
__global__ __launch_bounds__(256, 3) void k_wgrad_wino_dma(WgwArgs args) {
    __shared__ __attribute__((aligned(16))) float sm[2 * WD_HALF_FLOATS];           // 47 104 B: two half-region buffers; the epilogue's 16-KB scratch afterwards
    static_assert(2 * WD_HALF_FLOATS >= 4 * 16 * 64, "row-pass scratch");
    int p = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k) p += (k < args.njobs && int(blockIdx.x) >= args.job[k].block_begin) ? 1 : 0;
    const WgwJob& J = args.job[p];
    int local = blockIdx.x - J.block_begin;
    const int ks = local % args.ksplit; local /= args.ksplit;
    const int tci = local % args.n_ci, tco = local / args.n_ci;
    const int co0 = tco * 32, ci0 = tci * 32;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, half = lane >> 5;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xrow = u == 0 ? 0 : (u == 2 ? 2 : 1), yrow = u == 2 ? 1 : (u == 3 ? 3 : 2);
    const float sgn = u == 1 ? 1.f : -1.f;
    const float c0 = u == 3 ? 0.f : 1.f, c1 = u == 0 ? 0.f : (u == 1 ? 1.f : -1.f);
    f32x16 acc[4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;
    const long long total = (long long)J.regions * args.B;
    const long long r_begin = total * ks / args.ksplit, r_end = total * (ks + 1) / args.ksplit;
    const int n_halves = int(r_end - r_begin) * 2;
    const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) float*)sm));
    // request half hidx (region r_begin + hidx / 2, tile rows 2 (hidx & 1) .. + 1) into buffer hidx & 1: wave u issues DMA instructions u, u + 4, ...
    auto request_half = [&](int hidx) {
        const long long rr = r_begin + (hidx >> 1);
        const int hh = hidx & 1;
        const int b = int(rr / J.regions), reg = int(rr % J.regions);
        const int ty0 = (reg / J.tiles_x) * WW_TH, tx0 = (reg % J.tiles_x) * WW_TW;
        const i32x4 da = lds_dma_desc(J.a + size_t(b) * J.h * J.w * args.a_cstride, unsigned(J.h) * unsigned(J.w) * unsigned(args.a_cstride) * 4u);
        const i32x4 dd = lds_dma_desc(J.dy + size_t(b) * J.h * J.w * args.cout, unsigned(J.h) * unsigned(J.w) * unsigned(args.cout) * 4u);
        const unsigned buf = lds0 + unsigned((hidx & 1) * WD_HALF_FLOATS * 4);
#pragma unroll
        for (int k = 0; k < (WD_INSTR + 3) / 4; ++k) {
            const int n = u + 4 * k;                                  // wave-uniform
            if (n >= WD_INSTR) break;
            const int slot = 8 * (n < WD_X_INSTR ? n : n - WD_X_INSTR) + (lane >> 3), q = lane & 7;
            unsigned off;
            if (n < WD_X_INSTR) {
                const int row = slot / WD_XSLOTS, s = slot - row * WD_XSLOTS, kk = s >> 1;
                const int hx = (s & 1) ? (kk < 8 ? 8 + kk : -1) : (kk < 8 ? kk : 8 + kk);          // slots 17, 19: nothing
                const int gy = ty0 - 1 + 4 * hh + row, gx = tx0 - 1 + hx;
                const bool ok = hx >= 0 && gy >= 0 && gy < J.h && gx >= 0 && gx < J.w;
                off = ok ? unsigned((gy * J.w + gx) * args.a_cstride + ci0 + q * 4) * 4u : 0x80000000u;
                lds_dma16(buf + unsigned(n * 1024), off, da, 0u);
            } else {
                const int row = slot / WD_DYSLOTS, s = slot - row * WD_DYSLOTS;
                const int xx = (s & 1) ? 8 + (s >> 1) : (s >> 1);
                const int y = ty0 + 4 * hh + row, x = tx0 + xx;
                const bool ok = y < J.h && x < J.w;
                off = ok ? unsigned((y * J.w + x) * args.cout + co0 + q * 4) * 4u : 0x80000000u;
                lds_dma16(buf + unsigned(WD_X_FLOATS * 4 + (n - WD_X_INSTR) * 1024), off, dd, 0u);
            }
        }
    };
    struct Ops { float x[4], y[4], d0[2], d1[2]; };
    // operands of MFMA step st (0..7 within a half: tile row st >> 2, tile columns (st & 3) + 4 * lane half) from buffer `base`
    // Every read is one of two per-lane bases + a compile-time offset: columns hx0 < 8 sit at slot 2 hx0 + (lane half), i.e. at float
    // 64 hx0 + lane of their row; columns 8 / 9 (the last tile column's third and fourth patch column) at slot 1 / 3 for lane half 0
    // and 16 / 18 for lane half 1.
    const int lb = lane, lb2 = half ? i + 16 * 32 : i + 32;
    auto lds_step = [&](const float* base, int st, Ops& o) {
        const int trl = st >> 2, t4 = st & 3;
        const float* sxb = base;
        const float* sdb = base + WD_X_FLOATS;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int hx0 = 2 * t4 + b;                               // the column of lane half 0; half 1: hx0 + 8
            const int col = hx0 < 8 ? lb + 64 * hx0 : lb2 + 64 * (hx0 - 8);
            o.x[b] = sxb[(2 * trl + xrow) * WD_XSLOTS * 32 + col];
            o.y[b] = sxb[(2 * trl + yrow) * WD_XSLOTS * 32 + col];
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int col = lb + 64 * (2 * t4 + x);                   // column 2 t4 + x (+ 8 for lane half 1): slot 2 (2 t4 + x) + half
            o.d0[x] = sdb[(2 * trl) * WD_DYSLOTS * 32 + col];
            o.d1[x] = sdb[(2 * trl + 1) * WD_DYSLOTS * 32 + col];
        }
    };
    if (n_halves > 0) request_half(0);
    for (int hidx = 0; hidx < n_halves; ++hidx) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of half hidx have landed (nothing else is in flight)
        __syncthreads();                                    // ... everyone's have, and everyone is done with the other buffer
        if (hidx + 1 < n_halves) request_half(hidx + 1);
        const float* base = sm + (hidx & 1) * WD_HALF_FLOATS;
        Ops cur, nxt;
        lds_step(base, 0, cur);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            if (st + 1 < 8) lds_step(base, st + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            float t[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) t[b] = fmaf(sgn, cur.y[b], cur.x[b]);
            const float V[4] = {t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]};
            const float r0 = c0 * cur.d0[0] + c1 * cur.d1[0], r1 = c0 * cur.d0[1] + c1 * cur.d1[1];
            const float M[4] = {r0, r0 + r1, r0 - r1, -r1};
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(M[v], V[v], acc[v], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
    }
    // dg = G^T dU G of this slice: k_wgrad_wino's epilogue, verbatim
    float* red = sm;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h1 = 0.5f * acc[1][r], h2 = 0.5f * acc[2][r];
            const float t = j == 0 ? acc[0][r] + (h1 + h2) : (j == 1 ? h1 - h2 : (h1 + h2) + acc[3][r]);
            red[(u * 16 + r) * 64 + lane] = t;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = u * 4 + rr;
            const float t0 = red[(0 * 16 + r) * 64 + lane], t1 = red[(1 * 16 + r) * 64 + lane];
            const float t2 = red[(2 * 16 + r) * 64 + lane], t3 = red[(3 * 16 + r) * 64 + lane];
            const float h1 = 0.5f * t1, h2 = 0.5f * t2;
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float* dst = J.part + ((size_t(ks) * 9 + j) * args.cout + co) * args.cin + ci0 + i;
            const size_t tap_stride = size_t(3) * args.cout * args.cin;
            dst[0] = t0 + (h1 + h2); dst[tap_stride] = h1 - h2; dst[2 * tap_stride] = (h1 + h2) + t3;
        }
    }
}
// adds the slices' dg in slice order and scatters to OIHW
struct WgwRedArgs { const float* part[3]; float* dW[3]; int ksplit, cout, cin, ctot, cin_store; };
template <class Args>                                    // WgwRedArgs, or the WgRedArgs of a batched job (same fields)
__device__ __forceinline__ void wgrad_wino_reduce_block(const Args& a, int bx, int p) {
    // block = 64 (co, ci) pairs x 4 slice lanes: lane j adds slices j, j+4, ..., the four sums meet through LDS in lane order
    // (one thread per pair over all slices left 48 blocks walking 64 x 16 dependent loads each: 24 us at 64 channels)
    __shared__ float red[3][9][64];
    const long long n = (long long)a.cout * a.cin;
    const int l = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const long long idx = (long long)bx * 64 + l;
    float S[9];
#pragma unroll
    for (int f = 0; f < 9; ++f) S[f] = 0.f;
    if (idx < n)
        for (int k = kl; k < a.ksplit; k += 4)
#pragma unroll
            for (int f = 0; f < 9; ++f) S[f] += a.part[p][(size_t(k) * 9 + f) * n + idx];
    if (kl > 0) {
#pragma unroll
        for (int f = 0; f < 9; ++f) red[kl - 1][f][l] = S[f];
    }
    __syncthreads();
    if (kl != 0 || idx >= n) return;
    const int ci = int(idx % a.cin), co = int(idx / a.cin);
    if (ci >= a.cin_store) return;
    float* d = a.dW[p] + (size_t(co) * a.ctot + ci) * 9;
#pragma unroll
    for (int f = 0; f < 9; ++f) d[f] = ((S[f] + red[0][f][l]) + red[1][f][l]) + red[2][f][l];
}
__global__ __launch_bounds__(256) void k_wgrad_wino_reduce(WgwRedArgs a) { wgrad_wino_reduce_block(a, blockIdx.x, blockIdx.y); }
bool wgrad_uses_wino() { return opt_on(OPT_WGRAD_WINO); }
static bool wgrad_use_wino() {
    return opt_on(OPT_WGRAD_WINO);
}

struct WgRedArgs { const float* part[3]; float* dW[3]; int ksplit, cout, cin, ctot, taps, cin_store; };
__device__ __forceinline__ void wgrad_reduce_block(const WgRedArgs& a, int bx, int p) {
    // block = 64 items x 4 slice lanes (lane j adds slices j, j+4, ...; the four sums meet through LDS in lane order)
    __shared__ float red[4][64];
    const long long n = (long long)a.cout * a.cin * a.taps;
    const long long idx = (long long)bx * 64 + (threadIdx.x & 63);
    const int kl = threadIdx.x >> 6, l = threadIdx.x & 63;
    float s = 0.f;
    if (idx < n)
#pragma unroll 4
        for (int k = kl; k < a.ksplit; k += 4) s += a.part[p][size_t(k) * n + idx];
    red[kl][l] = s;
    __syncthreads();
    if (kl != 0 || idx >= n) return;
    s = ((red[0][l] + red[1][l]) + red[2][l]) + red[3][l];
    const int t = int(idx % a.taps);
    const long long r = idx / a.taps;
    const int ci = int(r % a.cin), co = int(r / a.cin);
    if (ci < a.cin_store) a.dW[p][(size_t(co) * a.ctot + ci) * a.taps + t] = s;
}
__global__ __launch_bounds__(256) void k_wgrad_reduce(WgRedArgs a) { wgrad_reduce_block(a, blockIdx.x, blockIdx.y); }
// every queued reduction in one launch (DeferredTail): block -> (job, plane, item block) through a prefix table
struct RedMultiJob { WgRedArgs r; int wino, bx, nplanes; };
struct RedMultiArgs { RedMultiJob job[kTailJobs]; int begin[kTailJobs + 1]; int njobs; };
__global__ __launch_bounds__(256) void k_wgrad_reduce_multi(RedMultiArgs m) {
    int j = 0;
#pragma unroll
    for (int k = 1; k < kTailJobs; ++k) j += (k < m.njobs && int(blockIdx.x) >= m.begin[k]) ? 1 : 0;
    const RedMultiJob& J = m.job[j];
    const int local = int(blockIdx.x) - m.begin[j], p = local / J.bx, bx = local - p * J.bx;
    if (J.wino) wgrad_wino_reduce_block(J.r, bx, p);
    else wgrad_reduce_block(J.r, bx, p);
}
int DeferredTail::flush(hipStream_t st) {
    for (size_t i0 = 0; i0 < red.size(); i0 += kTailJobs) {
        RedMultiArgs m;
        m.njobs = int(std::min(red.size() - i0, size_t(kTailJobs)));
        int blocks = 0;
        for (int j = 0; j < m.njobs; ++j) {
            const Red& r = red[i0 + j];
            RedMultiJob& J = m.job[j];
            for (int p = 0; p < 3; ++p) { J.r.part[p] = r.part[p]; J.r.dW[p] = r.dW[p]; }
            J.r.ksplit = r.ksplit; J.r.cout = r.cout; J.r.cin = r.cin; J.r.ctot = r.ctot; J.r.taps = r.taps; J.r.cin_store = r.cin_store;
            J.wino = r.wino; J.nplanes = r.nplanes;
            const long long n = (long long)r.cout * r.cin * (r.wino ? 1 : r.taps);
            J.bx = int((n + 63) / 64);
            m.begin[j] = blocks;
            blocks += J.bx * r.nplanes;
        }
        m.begin[m.njobs] = blocks;
        if (blocks) { hipLaunchKernelGGL(k_wgrad_reduce_multi, dim3(blocks), dim3(256), 0, st, m); S3D_HIP(hipGetLastError()); }
    }
    red.clear();
    for (size_t i0 = 0; i0 < bias.size(); i0 += kTailJobs) {
        BiasMultiArgs m;
        m.njobs = int(std::min(bias.size() - i0, size_t(kTailJobs)));
        int blocks = 0;
        for (int j = 0; j < m.njobs; ++j) {
            const Bias& b = bias[i0 + j];
            BiasArgs& a = m.job[j];
            for (int p = 0; p < 3; ++p) { a.R[p] = b.R[p]; a.dbias[p] = b.dbias[p]; a.h[p] = b.h[p]; }
            a.per_sample = b.per_sample; a.per_sample_stride = b.per_sample_stride; a.B = b.B; a.C = b.C;
            m.begin[j] = blocks;
            blocks += cdiv(b.C, kBiasCh);
        }
        m.begin[m.njobs] = blocks;
        if (blocks) { hipLaunchKernelGGL(k_bias_grad_multi, dim3(blocks), dim3(1024), 0, st, m); S3D_HIP(hipGetLastError()); }
    }
    bias.clear();
    return 0;
}

// 1x1: the 128 x 128 channel tile halves the staging traffic per flop, but a 64 -> 128 or 192 -> 64 skip convolution would
// fill 37-50 % of it (95 us for the 192 -> 64 one at towerruins size); take it only when the padding costs under a quarter
static bool wgrad_1x1_wide(int cin, int cout) {
    const long long wide = (long long)cdiv(cout, 128) * 128 * cdiv(cin, 128) * 128, narrow = (long long)cdiv(cout, 64) * 64 * cdiv(cin, 64) * 64;
    return 4 * wide <= 5 * narrow;
}
int wgrad_ksplit(const Geo& g, int B, int cin, int cout, int taps) {
    const int cus = device_cus();
    long long tiles = 0;
    int planes = 0;
    if (taps == 9 && wgrad_use_wino()) {
        // Winograd kernel: 32x32 channel pairs, 8x16-pixel regions, three blocks per CU; one balanced round of blocks
        for (int p = 0; p < 3; ++p) { tiles += (long long)cdiv(g.h[p], WW_TH) * cdiv(g.w[p], WW_TW); planes += g.h[p] > 0; }
        const long long base = (long long)(cout / 32) * (cin / 32) * std::max(planes, 1);
        long long ks = std::max<long long>(1, 3 * cus / base);
        const long long per_plane = std::max<long long>(1, tiles * B / std::max(planes, 1));
        return int(std::max<long long>(1, std::min(std::min(ks, per_plane), (long long)96)));
    }
    for (int p = 0; p < 3; ++p) { tiles += (long long)cdiv(g.h[p], WG_TR) * cdiv(g.w[p], WG_TC); planes += g.h[p] > 0; }
    const int bt = taps == 1 && wgrad_1x1_wide(cin, cout) ? 128 : 64;
    const int rows = taps == 9 ? 3 : (taps == 25 ? 5 : 1);           // blocks per (tile, slice): one kernel row each
    const long long base = (long long)cdiv(cout, bt) * cdiv(cin, bt) * std::max(planes, 1) * rows;
    // one balanced round of equal blocks: as many as are resident at once, never more - 516 blocks on 512 slots cost a
    // whole extra round (measured at 64 channels, batch 4: 2 per CU 4.90 ms/step, 3 per CU 4.79)
    const int slots = (taps == 9 || (taps == 1 && bt == 64) ? 3 : 2) * cus;   // blocks the kernel variant holds per CU (registers / LDS)
    long long ks = std::max<long long>(1, slots / base);
    const long long per_plane = std::max<long long>(1, tiles * B / std::max(planes, 1));
    ks = std::max<long long>(1, std::min(ks, per_plane));            // at least one pixel tile per slice (roughly)
    return int(std::min<long long>(ks, 128));                        // ... and bound the partial-sum traffic
}
size_t wgrad_part_floats(int ksplit, int cin, int cout, int taps) { return size_t(ksplit) * cin * cout * taps; }

static void queue_reduce(DeferredTail& tail, const WgradArgs& w, int wino) {
    DeferredTail::Red r;
    for (int p = 0; p < 3; ++p) { r.part[p] = p < w.nplanes ? w.part[p] : nullptr; r.dW[p] = p < w.nplanes ? w.dW[p] : nullptr; }
    r.ksplit = w.ksplit; r.cout = w.cout; r.cin = w.cin; r.ctot = w.ctot; r.taps = w.taps;
    r.cin_store = w.cin_store > 0 ? w.cin_store : w.cin; r.nplanes = w.nplanes; r.wino = wino;
    tail.red.push_back(r);
}
int launch_wgrad(const WgradArgs& w, hipStream_t st, DeferredTail* tail) {
    S3D_CHECK(w.taps == 9 || w.taps == 1 || w.taps == 25, S3D_ERR_INVALID, "wgrad: taps=%d", w.taps);
    S3D_CHECK(w.cin % 32 == 0 && w.cout % 32 == 0, S3D_ERR_INVALID, "wgrad: channels must be multiples of 32");
    if (w.taps == 9 && wgrad_use_wino()) {
        WgwArgs a;
        a.B = w.B; a.cin = w.cin; a.cout = w.cout; a.a_cstride = w.a.C; a.ksplit = w.ksplit;
        a.n_co = w.cout / 32; a.n_ci = w.cin / 32; a.njobs = w.nplanes;
        int blocks = 0;
        for (int p = 0; p < w.nplanes; ++p) {
            WgwJob& J = a.job[p];
            J.dy = w.dy.p[p]; J.a = w.a.p[p]; J.part = w.part[p]; J.h = w.dy.g.h[p]; J.w = w.dy.g.w[p];
            S3D_CHECK(size_t(J.h) * J.w * std::max(w.a.C, w.cout) * 4 < (size_t(1) << 31), S3D_ERR_INVALID, "wgrad: a plane of one sample must stay below 2 GiB");
            J.tiles_x = cdiv(J.w, WW_TW); J.regions = J.tiles_x * cdiv(J.h, WW_TH);
            J.block_begin = blocks;
            blocks += a.n_co * a.n_ci * a.ksplit;
        }
        if (!blocks || !w.B) return 0;
        // WGRAD_WINO = 2: operands by LDS-DMA, half regions double-buffered (bit-identical; measured 6-7 % SLOWER than the
        // register-staged kernel on every shape, profiles/r06_wgrad.txt: the launch is bound by the operand traffic, and the halves'
        // shared halo rows add a fifth to it) — not the default
        if (opt(OPT_WGRAD_WINO) == 2) hipLaunchKernelGGL(k_wgrad_wino_dma, dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_wgrad_wino, dim3(blocks), dim3(256), 0, st, a);
        S3D_HIP(hipGetLastError());
        if (tail) { queue_reduce(*tail, w, 1); return 0; }
        WgwRedArgs r;
        for (int p = 0; p < 3; ++p) { r.part[p] = p < w.nplanes ? w.part[p] : nullptr; r.dW[p] = p < w.nplanes ? w.dW[p] : nullptr; }
        r.ksplit = w.ksplit; r.cout = w.cout; r.cin = w.cin; r.ctot = w.ctot; r.cin_store = w.cin_store > 0 ? w.cin_store : w.cin;
        const long long n = (long long)w.cout * w.cin;
        hipLaunchKernelGGL(k_wgrad_wino_reduce, dim3((unsigned)((n + 63) / 64), w.nplanes), dim3(256), 0, st, r);
        S3D_HIP(hipGetLastError());
        return 0;
    }
    WgArgs a;
    a.B = w.B; a.cin = w.cin; a.cout = w.cout; a.a_cstride = w.a.C; a.ksplit = w.ksplit;
    const bool wide = w.taps == 1 && wgrad_1x1_wide(w.cin, w.cout);
    const int bt = wide ? 128 : 64;                                  // block tile (see WT)
    a.n_co = cdiv(w.cout, bt); a.n_ci = cdiv(w.cin, bt);
    // one kernel row per block (3x / 5x the blocks, a third of the accumulators -> three waves per SIMD)
    const bool split_rows = w.taps != 1;
    const int row_blocks = w.taps == 25 ? 5 : 3;
    int blocks = 0;
    a.njobs = w.nplanes;
    for (int p = 0; p < w.nplanes; ++p) {
        WgJob& J = a.job[p];
        J.dy = w.dy.p[p]; J.a = w.a.p[p]; J.part = w.part[p]; J.h = w.dy.g.h[p]; J.w = w.dy.g.w[p];
        J.tiles_x = cdiv(J.w, WG_TC); J.tiles = J.tiles_x * cdiv(J.h, WG_TR);
        J.block_begin = blocks;
        blocks += a.n_co * a.n_ci * a.ksplit * (split_rows ? row_blocks : 1);
    }
    if (!blocks || !w.B) return 0;
    if (w.taps == 25) hipLaunchKernelGGL((k_wgrad_mfma<25, 1, 1>), dim3(blocks), dim3(256), 0, st, a);
    else if (w.taps == 9) hipLaunchKernelGGL((k_wgrad_mfma<9, 1, 1>), dim3(blocks), dim3(256), 0, st, a);
    else if (wide) hipLaunchKernelGGL((k_wgrad_mfma<1, 1, 2>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_wgrad_mfma<1, 1, 1>), dim3(blocks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    if (tail) { queue_reduce(*tail, w, 0); return 0; }
    WgRedArgs r;
    for (int p = 0; p < 3; ++p) { r.part[p] = p < w.nplanes ? w.part[p] : nullptr; r.dW[p] = p < w.nplanes ? w.dW[p] : nullptr; }
    r.ksplit = w.ksplit; r.cout = w.cout; r.cin = w.cin; r.ctot = w.ctot; r.taps = w.taps;
    r.cin_store = w.cin_store > 0 ? w.cin_store : w.cin;
    const long long n = (long long)w.cout * w.cin * w.taps;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((n + 63) / 64), w.nplanes), dim3(256), 0, st, r);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ GroupNorm (+FiLM) + SiLU backward
// Forward (s3d_kernels.hip:k_gn_act): xh = (x-mean)*rstd ; u = xh*gamma+beta ; z = u*(1+s)+sh ; y = silu(z).
// With dz = dy*silu'(z), per (b, plane, channel):  A1 = sum_px dz,  A2 = sum_px dz*xh.  Then
//   dshift = A1, dscale = gamma*A2 + beta*A1, dbeta = sum_b (1+s)A1, dgamma = sum_b (1+s)A2,
//   dx = rstd * ( g*dz - mean_grp(g*dz) - xh * mean_grp(g*dz*xh) ),  g = gamma*(1+s).
// dy here is the dense dgrad of the plane's own channels plus the broadcast gradients of its two axis means
// (rowadd[r] * rowscale + coladd[c] * colscale, the scales being 1/length of the averaged axis).
// A thread owns one float4 of channels (cq = C/4 quads x pl pixel lanes per block), as in the forward kernel.
struct GnBwdArgs {
    const float* x[3]; const float* dy[3]; float* dx[3];
    const float* rowadd[3]; const float* coladd[3];      // [B][h][C], [B][w][C] or null
    float rowscale[3], colscale[3];
    const float* add[3];                                  // extra gradient path added to dx, or null
    const float* gamma[3]; const float* beta[3];
    const float* mr; const float* film; int film_stride;
    float* part;                                          // [B][3][nchunk][C][2]
    const float* coef;                                    // [B][3][C][8]: per-channel constants of the apply pass (k_gn_bwd_coefs)
    const float* A; float* dgamma[3]; float* dbeta[3]; float* dfilm;   // the apply launch's extra blocks: parameter / FiLM gradients from A[b][p][c][2]
    int h[3], w[3];
    int C, cq, pl, B, nchunk, ngroups;
};
constexpr int kGnBwdChunks = 64;
struct GnChan { float mean, rstd, gam, bet, sc, sh; };
__device__ __forceinline__ GnChan gn_chan(const GnBwdArgs& a, int p, int b, int ch) {
    const int cg = a.C / a.ngroups;
    const float* mr = a.mr + ((size_t(b) * 3 + p) * a.ngroups + ch / cg) * 2;
    GnChan c;
    c.mean = mr[0]; c.rstd = mr[1]; c.gam = a.gamma[p][ch]; c.bet = a.beta[p][ch];
    c.sc = a.film ? 1.0f + a.film[size_t(b) * a.film_stride + ch] : 1.0f;
    c.sh = a.film ? a.film[size_t(b) * a.film_stride + a.C + ch] : 0.0f;
    return c;
}
__device__ __forceinline__ float gn_dz(const GnChan& c, float x, float dy, float& xh) {
    xh = (x - c.mean) * c.rstd;
    const float z = (xh * c.gam + c.bet) * c.sc + c.sh;
    const float sg = sigmoid_f(z);
    return dy * sg * (1.0f + z * (1.0f - sg));
}
__device__ __forceinline__ float4 gn_dy4(const GnBwdArgs& a, int p, int b, int i, int w, size_t pix, int q) {
    float4 dy = reinterpret_cast<const float4*>(a.dy[p])[pix * a.cq + q];
    if (!a.rowadd[p] && !a.coladd[p]) return dy;
    const int r = i / w, c = i - r * w;
    if (a.rowadd[p]) {
        const float4 v = reinterpret_cast<const float4*>(a.rowadd[p])[(size_t(b) * a.h[p] + r) * a.cq + q];
        const float s = a.rowscale[p];
        dy.x = fmaf(v.x, s, dy.x); dy.y = fmaf(v.y, s, dy.y); dy.z = fmaf(v.z, s, dy.z); dy.w = fmaf(v.w, s, dy.w);
    }
    if (a.coladd[p]) {
        const float4 v = reinterpret_cast<const float4*>(a.coladd[p])[(size_t(b) * a.w[p] + c) * a.cq + q];
        const float s = a.colscale[p];
        dy.x = fmaf(v.x, s, dy.x); dy.y = fmaf(v.y, s, dy.y); dy.z = fmaf(v.z, s, dy.z); dy.w = fmaf(v.w, s, dy.w);
    }
    return dy;
}
__global__ __launch_bounds__(256) void k_gn_bwd_partials(GnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm_gn[];      // [pl][C][2]
    const int chunk = blockIdx.x, p = blockIdx.y, b = blockIdx.z;
    const int h = a.h[p], w = a.w[p], C = a.C;
    const int npix = h * w;                                 // (a plane of one sample: far below 2^31 pixels)
    const int p0 = int((long long)npix * chunk / a.nchunk), p1 = int((long long)npix * (chunk + 1) / a.nchunk);
    const int q = threadIdx.x % a.cq, l = threadIdx.x / a.cq;
    GnChan ch[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ch[k] = gn_chan(a, p, b, 4 * q + k);
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    const float4* xs = reinterpret_cast<const float4*>(a.x[p]);
    // (measured: issuing four pixels of loads per trip, or 192 chunks, is slower here — 22-25 us against 20.5)
    for (int i = p0 + l; i < p1; i += a.pl) {
        const size_t pix = size_t(b) * npix + i;
        const float4 x = xs[pix * a.cq + q];
        const float4 dy = gn_dy4(a, p, b, i, w, pix, q);
        const float xv[4] = {x.x, x.y, x.z, x.w}, dv[4] = {dy.x, dy.y, dy.z, dy.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float xh;
            const float dz = gn_dz(ch[k], xv[k], dv[k], xh);
            a1[k] += dz; a2[k] = fmaf(dz, xh, a2[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sm_gn[(size_t(l) * C + 4 * q + k) * 2] = a1[k]; sm_gn[(size_t(l) * C + 4 * q + k) * 2 + 1] = a2[k]; }
    __syncthreads();
    float* o = a.part + (((size_t(b) * 3 + p) * a.nchunk + chunk) * C) * 2;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float s = 0.f;
        for (int ll = 0; ll < a.pl; ++ll) s += sm_gn[size_t(ll) * C * 2 + i];
        o[i] = s;
    }
}
// A[b][p][c][2] = sum over chunks (double) ; then group coefficients and the parameter / FiLM gradients
struct GnBwdFinArgs {
    const double* dpart; int dmaxparts; int dnparts[3];   // != null: the sums come from the dgrad convolution's epilogue, [b][plane][channel][tile][2] doubles
    const float* part; float* A; float* coef;
    const float* gamma[3]; const float* beta[3];
    float* dgamma[3]; float* dbeta[3];
    const float* film; float* dfilm; int film_stride;    // dfilm [B][film_stride]: dscale at [0,C), dshift at [C,2C)
    const float* mr;
    double count[3];
    int C, B, nchunk, ngroups;
};
// One block per (sample, plane): A[c] = the chunk partials added in chunk order (double), then the group coefficients of the
// apply pass.  (One launch: the sums used to be a launch of their own ahead of a single-block coefficient kernel — 5.0 + 6.3 us
// of dependent launches per GroupNorm, ten per training step; folding the sums into that single block cost 16.5 us.)  The
// parameter and FiLM gradients, which add A over samples / planes, moved into extra blocks of the apply launch.
__global__ __launch_bounds__(256) void k_gn_bwd_fin(GnBwdFinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sA[];          // [C][2]
    const int C = a.C, G = a.ngroups, cg = C / G;
    const int bp = blockIdx.x, b = bp / 3, p = bp % 3;
    if (a.dpart) {
        // tile records of k_conv_wino24s_gnb: [channel][tile][2] doubles, a channel's records contiguous.  Eight lanes per channel —
        // lane l adds tiles l, l + 8, ... (its 16-byte records and its neighbours' form 128-byte runs), the eight sums meet in a fixed
        // xor tree: a wave reads whole lines instead of 64 scattered ones, and the order never depends on anything but the tile count
        const int l = threadIdx.x & 7, n = a.dnparts[p];
        for (int ch = threadIdx.x >> 3; ch < C; ch += blockDim.x >> 3) {
            const double2* q = reinterpret_cast<const double2*>(a.dpart + (size_t(bp) * C + ch) * a.dmaxparts * 2);
            double s1 = 0, s2 = 0;
#pragma unroll 4
            for (int k = l; k < n; k += 8) { const double2 v = q[k]; s1 += v.x; s2 += v.y; }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o, 8); s2 += __shfl_xor(s2, o, 8); }
            if (l == 0) {
                const float f1 = float(s1), f2 = float(s2);
                a.A[(size_t(bp) * C + ch) * 2] = f1; a.A[(size_t(bp) * C + ch) * 2 + 1] = f2;
                sA[2 * ch] = f1; sA[2 * ch + 1] = f2;
            }
        }
    } else
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
        double s1 = 0, s2 = 0;
        {
#pragma unroll 8
            for (int k = 0; k < a.nchunk; ++k) {
                const float2 q = *reinterpret_cast<const float2*>(a.part + ((size_t(bp) * a.nchunk + k) * C + ch) * 2);
                s1 += q.x; s2 += q.y;
            }
        }
        const float f1 = float(s1), f2 = float(s2);
        a.A[(size_t(bp) * C + ch) * 2] = f1; a.A[(size_t(bp) * C + ch) * 2 + 1] = f2;
        sA[2 * ch] = f1; sA[2 * ch + 1] = f2;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        double g1 = 0, g2 = 0;
        for (int k = 0; k < cg; ++k) {
            const int ch = g * cg + k;
            const double sc = a.film ? 1.0 + a.film[size_t(b) * a.film_stride + ch] : 1.0;
            const double gg = double(a.gamma[p][ch]) * sc;
            g1 += gg * sA[2 * ch]; g2 += gg * sA[2 * ch + 1];
        }
        // per-channel constants of the apply pass, 8 floats per channel:
        //   xh = x*c0 + c1 ; z = xh*c2 + c3 ; dx = dz*c4 - c5 - xh*c6        (c7 unused)
        const float mean = a.mr[(size_t(bp) * G + g) * 2], rstd = a.mr[(size_t(bp) * G + g) * 2 + 1];
        const float k1 = float(g1 / a.count[p]), k2 = float(g2 / a.count[p]);
        for (int k = 0; k < cg; ++k) {
            const int ch = g * cg + k;
            const float sc = a.film ? 1.0f + a.film[size_t(b) * a.film_stride + ch] : 1.0f;
            const float sh = a.film ? a.film[size_t(b) * a.film_stride + C + ch] : 0.0f;
            const float gam = a.gamma[p][ch], bet = a.beta[p][ch];
            float* o = a.coef + (size_t(bp) * C + ch) * 8;
            o[0] = rstd; o[1] = -mean * rstd; o[2] = gam * sc; o[3] = bet * sc + sh;
            o[4] = rstd * gam * sc; o[5] = rstd * k1; o[6] = rstd * k2; o[7] = 0.f;
        }
    }
}
// extra block (plane p, sample b) of the apply launch: dgamma / dbeta of plane p (sum over the batch; b == 0 only) and the FiLM
// gradients of sample b (sum over the planes, which share emb_out; p == 0 only)
__device__ __forceinline__ void gn_bwd_param_grads(const GnBwdArgs& a, int p, int b) {
    const int C = a.C;
    if (b == 0)
        for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
            double dg = 0, db = 0;
            for (int bb = 0; bb < a.B; ++bb) {
                const double sc = a.film ? 1.0 + a.film[size_t(bb) * a.film_stride + ch] : 1.0;
                const float* A = a.A + ((size_t(bb) * 3 + p) * C + ch) * 2;
                db += sc * A[0]; dg += sc * A[1];
            }
            a.dgamma[p][ch] = float(dg); a.dbeta[p][ch] = float(db);
        }
    if (p == 0 && a.dfilm)
        for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
            double ds = 0, dh = 0;
            for (int pp = 0; pp < 3; ++pp) {
                const float* A = a.A + ((size_t(b) * 3 + pp) * C + ch) * 2;
                ds += double(a.gamma[pp][ch]) * A[1] + double(a.beta[pp][ch]) * A[0];
                dh += A[0];
            }
            a.dfilm[size_t(b) * a.film_stride + ch] = float(ds);
            a.dfilm[size_t(b) * a.film_stride + C + ch] = float(dh);
        }
}
// block = (plane, sample, chunk of the plane's pixels); thread = (pixel lane, channel quad): the quad's constants sit in
// registers and four pixels are in flight per trip
__global__ __launch_bounds__(256) void k_gn_bwd_apply(GnBwdArgs a, int nchunk) {
    const int chunk = blockIdx.x, p = blockIdx.y, b = blockIdx.z;
    if (chunk == nchunk) { gn_bwd_param_grads(a, p, b); return; }
    const int w = a.w[p], h = a.h[p];
    const int npix = h * w;
    const int p0 = int((long long)npix * chunk / nchunk), p1 = int((long long)npix * (chunk + 1) / nchunk);
    const int q = threadIdx.x % a.cq, l = threadIdx.x / a.cq;
    const float4* ct = reinterpret_cast<const float4*>(a.coef + ((size_t(b) * 3 + p) * a.C + 4 * q) * 8);
    float4 c0[4], c1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { c0[k] = ct[2 * k]; c1[k] = ct[2 * k + 1]; }
    const float4* xs = reinterpret_cast<const float4*>(a.x[p]);
    const float4* as = reinterpret_cast<const float4*>(a.add[p]);
    float4* dxs = reinterpret_cast<float4*>(a.dx[p]);
    constexpr int U = 4;
    for (int i0 = p0 + l; i0 < p1; i0 += U * a.pl) {
        float4 x[U], dy[U], ad[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(i0 + u * a.pl, p1 - 1);
            const size_t pix = size_t(b) * npix + i;
            x[u] = xs[pix * a.cq + q];
            dy[u] = gn_dy4(a, p, b, i, w, pix, q);
            ad[u] = as ? as[pix * a.cq + q] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * a.pl;
            if (i >= p1) break;
            const float xv[4] = {x[u].x, x[u].y, x[u].z, x[u].w}, dv[4] = {dy[u].x, dy[u].y, dy[u].z, dy[u].w};
            const float av[4] = {ad[u].x, ad[u].y, ad[u].z, ad[u].w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = fmaf(xv[k], c0[k].x, c0[k].y);
                const float z = fmaf(xh, c0[k].z, c0[k].w);
                const float sg = sigmoid_f(z);
                const float dz = dv[k] * sg * (1.0f + z * (1.0f - sg));
                o[k] = dz * c1[k].x - c1[k].y - xh * c1[k].z + av[k];
            }
            dxs[(size_t(b) * npix + i) * a.cq + q] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}
size_t gn_bwd_ws_floats(int B, int C) { return size_t(B) * 3 * kGnBwdChunks * C * 2 + size_t(B) * 3 * C * 10; }
int launch_gn_act_bwd(const GnActBwd& s, hipStream_t st) {
    GnBwdArgs a;
    const Tri& x = s.x;
    S3D_CHECK(x.C % 32 == 0 && x.C <= 1024, S3D_ERR_INVALID, "gn_act_bwd: C=%d", x.C);
    long long begin[4] = {0, 0, 0, 0};
    a.C = x.C; a.cq = x.C / 4; a.pl = a.cq >= 256 ? 1 : 256 / a.cq;
    for (int p = 0; p < 3; ++p) {
        a.x[p] = x.p[p]; a.dy[p] = s.dy.p[p]; a.dx[p] = s.dx.p[p];
        a.rowadd[p] = s.rowadd ? s.rowadd[p] : nullptr; a.coladd[p] = s.coladd ? s.coladd[p] : nullptr;
        a.rowscale[p] = 1.0f / float(x.g.w[p]); a.colscale[p] = 1.0f / float(x.g.h[p]);   // d(mean over w) / d(mean over h)
        a.add[p] = s.add ? s.add->p[p] : nullptr;
        a.gamma[p] = s.gamma[p]; a.beta[p] = s.beta[p];
        a.h[p] = x.g.h[p]; a.w[p] = x.g.w[p];
        begin[p + 1] = begin[p] + (long long)x.g.h[p] * x.g.w[p] * a.cq;
    }
    a.mr = s.stats.mr; a.film = s.film; a.film_stride = s.film_stride;
    a.B = s.B; a.nchunk = kGnBwdChunks; a.ngroups = s.ngroups;
    S3D_CHECK(s.ngroups >= 1 && x.C % s.ngroups == 0, S3D_ERR_INVALID, "gn_act_bwd: %d groups for %d channels", s.ngroups, x.C);
    float* part = s.ws;
    float* A = part + size_t(s.B) * 3 * kGnBwdChunks * x.C * 2;
    float* coef = A + size_t(s.B) * 3 * x.C * 2;
    a.part = part; a.coef = coef;
    if (!s.B || !begin[3]) return 0;
    if (!s.conv_part) {
        hipLaunchKernelGGL(k_gn_bwd_partials, dim3(kGnBwdChunks, 3, s.B), dim3(a.cq * a.pl), size_t(a.pl) * x.C * 2 * sizeof(float), st, a);
        S3D_HIP(hipGetLastError());
    } else S3D_CHECK(s.conv_part->nsub == x.C && s.conv_part->p, S3D_ERR_INVALID, "gn_act_bwd: the convolution's partial records must be per channel");
    GnBwdFinArgs f;
    f.dpart = s.conv_part ? s.conv_part->p : nullptr;
    f.dmaxparts = s.conv_part ? s.conv_part->maxparts : 0;
    for (int p = 0; p < 3; ++p) f.dnparts[p] = s.conv_part ? s.conv_part->nparts[p] : 0;
    f.part = part; f.A = A; f.coef = coef; f.film = s.film; f.dfilm = s.dfilm; f.film_stride = s.film_stride; f.mr = s.stats.mr;
    for (int p = 0; p < 3; ++p) {
        f.gamma[p] = s.gamma[p]; f.beta[p] = s.beta[p]; f.dgamma[p] = s.dgamma[p]; f.dbeta[p] = s.dbeta[p];
        f.count[p] = double(x.C / s.ngroups) * x.g.h[p] * x.g.w[p];
    }
    f.C = x.C; f.B = s.B; f.nchunk = kGnBwdChunks; f.ngroups = s.ngroups;
    hipLaunchKernelGGL(k_gn_bwd_fin, dim3(s.B * 3), dim3(256), size_t(x.C) * 2 * sizeof(float), st, f);
    S3D_HIP(hipGetLastError());
    a.A = A; a.dfilm = s.dfilm;
    for (int p = 0; p < 3; ++p) { a.dgamma[p] = s.dgamma[p]; a.dbeta[p] = s.dbeta[p]; }
    long long maxpix = 0;
    for (int p = 0; p < 3; ++p) maxpix = std::max(maxpix, (long long)x.g.h[p] * x.g.w[p]);
    const int nchunk = int(std::max(1LL, (maxpix + a.pl * 8 - 1) / (a.pl * 8)));          // two trips of four pixels per thread
    hipLaunchKernelGGL(k_gn_bwd_apply, dim3(nchunk + 1, 3, s.B), dim3(a.cq * a.pl), 0, st, a, nchunk);   // (+1: the parameter-gradient blocks)
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ outer products with the composed 12-channel map
// The first and last TriplaneConv are 1x1 over the composed map's few channels (in_conv.0: 12 -> C, out.2: C -> 12).
// With s[px][o] a pixel of the composed NCHW tensor (decompose_featmaps views, src/utils/triplane_util.py:20-25)
// and v[px][c] an NHWC plane:
//   outer[p][o][c] = sum_{b,px} s[o]*v[c] ;  ssum[p][o] = sum s[o] ;  vsum[p][c] = sum v[c] ;
//   optional dv[px][c] = sum_o Wt[p][o][c]*s[px][o]                                      (out.2's dgrad)
// Two-stage: per-chunk partials (a thread owns a float4 of channels and every pl-th pixel of the chunk; the pixel
// lanes are added through LDS in lane order), then k_small_reduce adds the chunks in order.
constexpr int kSmallChunks = 256, kSmallMaxS = 12;
struct SmallArgs {
    const float* s;                 // composed [B][S][H+D][W+D]
    const float* v[3];              // NHWC [B][h][w][C]
    const float* Wt;                // [3][S][C] or null
    float* dv[3];                   // NHWC or null
    float* part;                    // [3][chunks][S+1][C] (row S = vsum) followed by [3][chunks][S] (ssum)
    int H, W, D, S, C, cq, pl, B;
};
__device__ __forceinline__ size_t composed_index(int p, int H, int W, int D, int S, int b, int o, int r, int c) {
    const int Y = p == 2 ? H + c : r, X = p == 0 ? c : (p == 1 ? W + c : r);
    return ((size_t(b) * S + o) * (H + D) + Y) * (W + D) + X;
}
__global__ __launch_bounds__(256) void k_small_outer(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm_so[];       // [pl][S+1][C] then [pl][S]
    constexpr int SM = kSmallMaxS;
    const int S = a.S;
    const int chunk = blockIdx.x, p = blockIdx.y;
    const int h = p == 2 ? a.W : a.H, w = p == 0 ? a.W : a.D;
    const int hw = h * w, npix = a.B * hw;               // (checked below 2^31 by the launcher)
    const int p0 = int((long long)npix * chunk / kSmallChunks), p1 = int((long long)npix * (chunk + 1) / kSmallChunks);
    const int C = a.C, q = threadIdx.x % a.cq, l = threadIdx.x / a.cq;
    float acc[SM][4], vs[4] = {0, 0, 0, 0}, wt[SM][4], ssum[SM];
#pragma unroll
    for (int o = 0; o < SM; ++o) {
        ssum[o] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[o][k] = 0.f;
            wt[o][k] = (a.Wt && o < S) ? a.Wt[(size_t(p) * S + o) * C + 4 * q + k] : 0.f;
        }
    }
    const size_t ostride = size_t(a.H + a.D) * (a.W + a.D);
    // one-deep software pipeline: the thirteen loads of the next pixel are in flight during this pixel's FMAs
    auto fetch = [&](int i, float (&sv)[SM], float4& v) {
        i = min(i, p1 - 1);
        const int b = i / hw, pq = i - b * hw;
        const int r = pq / w, c = pq - r * w;
        const size_t base = composed_index(p, a.H, a.W, a.D, S, b, 0, r, c);
#pragma unroll
        for (int o = 0; o < SM; ++o) sv[o] = o < S ? a.s[base + o * ostride] : 0.f;
        v = reinterpret_cast<const float4*>(a.v[p])[size_t(i) * a.cq + q];
    };
    float sv[SM];
    float4 v = make_float4(0, 0, 0, 0);
    if (p0 + l < p1) fetch(p0 + l, sv, v);
    for (int i = p0 + l; i < p1; i += a.pl) {
        float svn[SM];
        float4 vn;
        fetch(i + a.pl, svn, vn);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        float dv[4] = {0, 0, 0, 0};
#pragma unroll
        for (int o = 0; o < SM; ++o) {
            ssum[o] += sv[o];
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[o][k] = fmaf(sv[o], vv[k], acc[o][k]); dv[k] = fmaf(wt[o][k], sv[o], dv[k]); }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) vs[k] += vv[k];
        if (a.dv[p]) reinterpret_cast<float4*>(a.dv[p])[size_t(i) * a.cq + q] = make_float4(dv[0], dv[1], dv[2], dv[3]);
#pragma unroll
        for (int o = 0; o < SM; ++o) sv[o] = svn[o];
        v = vn;
    }
    // pixel lanes that share a wave meet by xor-shuffles first (a wave holds 64/cq of them), so LDS carries one partial per
    // wave (13 KB instead of 54 KB at 64 channels: the block count per CU was LDS-bound)
    const int lw = (a.cq < 64 && 64 % a.cq == 0) ? 64 / a.cq : 1, ng = a.pl / lw, grp = l / lw;
    for (int off = a.cq; lw > 1 && off < 64; off <<= 1) {
#pragma unroll
        for (int o = 0; o < SM; ++o) {
            ssum[o] += __shfl_xor(ssum[o], off, 64);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[o][k] += __shfl_xor(acc[o][k], off, 64);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) vs[k] += __shfl_xor(vs[k], off, 64);
    }
    float* sms = sm_so + size_t(ng) * (S + 1) * C;
    if (l % lw == 0) {
#pragma unroll
        for (int o = 0; o < SM; ++o)
            if (o < S) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sm_so[(size_t(grp) * (S + 1) + o) * C + 4 * q + k] = acc[o][k];
                if (q == 0) sms[grp * S + o] = ssum[o];
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) sm_so[(size_t(grp) * (S + 1) + S) * C + 4 * q + k] = vs[k];
    }
    __syncthreads();
    float* part = a.part + (size_t(p) * kSmallChunks + chunk) * (S + 1) * C;
    for (int i = threadIdx.x; i < (S + 1) * C; i += blockDim.x) {
        float s = 0.f;
        for (int g = 0; g < ng; ++g) s += sm_so[size_t(g) * (S + 1) * C + i];
        part[i] = s;
    }
    float* sp = a.part + size_t(3) * kSmallChunks * (S + 1) * C + (size_t(p) * kSmallChunks + chunk) * S;
    if (int(threadIdx.x) < S) {
        float s = 0.f;
        for (int g = 0; g < ng; ++g) s += sms[g * S + threadIdx.x];
        sp[threadIdx.x] = s;
    }
}
struct SmallRedArgs {
    const float* part; int S, C;
    float* outer[3]; int outer_transposed;   // 0: outer[p][o*C + c] ; 1: outer[p][c*S + o]
    float* ssum[3]; float* vsum[3];          // optional
};
__global__ __launch_bounds__(256) void k_small_reduce(SmallRedArgs a) {
    // block = 64 items x 4 chunk lanes: lane j adds chunks j, j+4, ... (double), the four sums meet through LDS in lane order
    __shared__ double red[4][64];
    const int p = blockIdx.y, S = a.S, C = a.C;
    const int idx = blockIdx.x * 64 + (threadIdx.x & 63), kl = threadIdx.x >> 6;
    const int n_outer = (S + 1) * C, n_all = n_outer + S;
    double s = 0;
    if (idx < n_outer) {
        const int o = idx / C, c = idx % C;
#pragma unroll 8
        for (int k = kl; k < kSmallChunks; k += 4) s += a.part[((size_t(p) * kSmallChunks + k) * (S + 1) + o) * C + c];
    } else if (idx < n_all) {
        const int o = idx - n_outer;
#pragma unroll 8
        for (int k = kl; k < kSmallChunks; k += 4)
            s += a.part[size_t(3) * kSmallChunks * n_outer + (size_t(p) * kSmallChunks + k) * S + o];
    }
    red[kl][threadIdx.x & 63] = s;
    __syncthreads();
    if (kl != 0 || idx >= n_all) return;
    const int l = threadIdx.x & 63;
    s = ((red[0][l] + red[1][l]) + red[2][l]) + red[3][l];
    if (idx < n_outer) {
        const int o = idx / C, c = idx % C;
        if (o < S) a.outer[p][a.outer_transposed ? size_t(c) * S + o : size_t(o) * C + c] = float(s);
        else if (a.vsum[p]) a.vsum[p][c] = float(s);
    } else if (a.ssum[p]) a.ssum[p][idx - n_outer] = float(s);
}
size_t small_outer_ws_floats(int S, int C) { return size_t(3) * kSmallChunks * ((S + 1) * C + S); }
int launch_small_outer(const SmallOuter& s, hipStream_t st) {
    S3D_CHECK(s.S >= 1 && s.S <= kSmallMaxS && s.C % 4 == 0 && s.C <= 1024, S3D_ERR_UNSUPPORTED,
              "training supports in/out_channels <= %d (the Sin3DM triplane feature width), got %d (C=%d)", kSmallMaxS, s.S, s.C);
    SmallArgs a;
    a.s = s.s; a.Wt = s.Wt; a.part = s.ws; a.H = s.H; a.W = s.W; a.D = s.D; a.S = s.S; a.C = s.C; a.B = s.B;
    a.cq = s.C / 4; a.pl = a.cq >= 256 ? 1 : 256 / a.cq;
    for (int p = 0; p < 3; ++p) { a.v[p] = s.v.p[p]; a.dv[p] = s.dv ? s.dv->p[p] : nullptr; }
    S3D_CHECK((long long)s.B * std::max(s.H, s.W) * std::max(s.W, s.D) < (1LL << 31), S3D_ERR_INVALID, "small_outer: batch x plane exceeds 2^31 pixels");
    const size_t shm = size_t((a.cq < 64 && 64 % a.cq == 0) ? a.pl / (64 / a.cq) : a.pl) * ((s.S + 1) * s.C + s.S) * sizeof(float);
    hipLaunchKernelGGL(k_small_outer, dim3(kSmallChunks, 3), dim3(a.cq * a.pl), shm, st, a);
    S3D_HIP(hipGetLastError());
    SmallRedArgs r;
    r.part = s.ws; r.S = s.S; r.C = s.C; r.outer_transposed = s.outer_transposed;
    for (int p = 0; p < 3; ++p) { r.outer[p] = s.outer[p]; r.ssum[p] = s.ssum ? s.ssum[p] : nullptr; r.vsum[p] = s.vsum ? s.vsum[p] : nullptr; }
    hipLaunchKernelGGL(k_small_reduce, dim3(cdiv((s.S + 1) * s.C + s.S, 64), 3), dim3(256), 0, st, r);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ resampling backward
// avg_pool2d(2,2) backward (+ the skip-connection slice of the concat gradient): one thread per float4
struct PoolBwdArgs {
    const float* dpool[3]; const float* dskip[3]; float* out[3];
    int h[3], w[3];
    int cq, skip_cq, skip_q0, B;
    long long begin[4];
};
__global__ void k_pool_bwd_add(PoolBwdArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.begin[3] * a.B) return;
    const int b = int(i / a.begin[3]);
    long long r = i % a.begin[3];
    const int p = r >= a.begin[2] ? 2 : (r >= a.begin[1] ? 1 : 0);
    r -= a.begin[p];
    const int q = int(r % a.cq);
    const long long pix = r / a.cq;
    const int h = a.h[p], w = a.w[p], hp = h / 2, wp = w / 2;
    const int y = int(pix / w), x = int(pix % w);
    float4 o = make_float4(0, 0, 0, 0);
    if (a.dpool[p] && y / 2 < hp && x / 2 < wp) {
        const float4 v = reinterpret_cast<const float4*>(a.dpool[p])[((size_t(b) * hp + y / 2) * wp + x / 2) * a.cq + q];
        o.x = 0.25f * v.x; o.y = 0.25f * v.y; o.z = 0.25f * v.z; o.w = 0.25f * v.w;
    }
    if (a.dskip[p]) {
        const float4 v = reinterpret_cast<const float4*>(a.dskip[p])[((size_t(b) * h + y) * w + x) * a.skip_cq + a.skip_q0 + q];
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    }
    reinterpret_cast<float4*>(a.out[p])[((size_t(b) * h + y) * w + x) * a.cq + q] = o;
}
int launch_pool_bwd_add(const Tri* dpool, const Tri* dskip, int skip_coff, int B, Tri& out, hipStream_t st) {
    PoolBwdArgs a;
    a.cq = out.C / 4; a.B = B; a.skip_cq = dskip ? dskip->C / 4 : 0; a.skip_q0 = skip_coff / 4; a.begin[0] = 0;
    for (int p = 0; p < 3; ++p) {
        a.dpool[p] = dpool ? dpool->p[p] : nullptr; a.dskip[p] = dskip ? dskip->p[p] : nullptr; a.out[p] = out.p[p];
        a.h[p] = out.g.h[p]; a.w[p] = out.g.w[p];
        a.begin[p + 1] = a.begin[p] + (long long)out.g.h[p] * out.g.w[p] * a.cq;
    }
    const long long n = a.begin[3] * B;
    if (!n) return 0;
    hipLaunchKernelGGL(k_pool_bwd_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// bilinear (align_corners=False) backward as a gather: d_in[y][x] = sum over the output pixels whose two source
// taps include (y, x), with the weights recomputed exactly as k_bilinear computes them.
__device__ __forceinline__ void bl_src(int o, int in, float scale, int& i0, int& i1, float& l0, float& l1) {
    float f = scale * (float(o) + 0.5f) - 0.5f; f = f < 0.f ? 0.f : f;
    i0 = int(f); i0 = i0 > in - 1 ? in - 1 : i0;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = f - float(i0); l0 = 1.f - l1;
}
// (blockIdx.y = plane: the three planes of a triplane in ONE launch — they were three dependent 16-us launches on the backward
// pass's main chain)
struct BlBwdArgs { const float* dout[3]; float* din[3]; int hi[3], wi[3], ho[3], wo[3]; int B, cq, out_cq, out_q0; };
__global__ void k_bilinear_bwd(BlBwdArgs a) {
    const int pl = blockIdx.y;
    const float* __restrict__ dout = a.dout[pl];
    float* __restrict__ din = a.din[pl];
    const int B = a.B, cq = a.cq, hi = a.hi[pl], wi = a.wi[pl], ho = a.ho[pl], wo = a.wo[pl], out_cq = a.out_cq, out_q0 = a.out_q0;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)B * hi * wi * cq;
    if (i >= n) return;
    const int q = int(i % cq);
    long long r = i / cq;
    const int x = int(r % wi); r /= wi;
    const int y = int(r % hi);
    const int b = int(r / hi);
    const float sh = float(hi) / float(ho), sw = float(wi) / float(wo);
    // candidate output rows: src in (y-1, y+1)  <=>  o in ((y-0.5)/s - 0.5, (y+1.5)/s - 0.5); widen by one and clamp
    int oy0 = int(floorf((float(y) - 0.5f) / sh - 0.5f)) - 1, oy1 = int(ceilf((float(y) + 1.5f) / sh - 0.5f)) + 1;
    int ox0 = int(floorf((float(x) - 0.5f) / sw - 0.5f)) - 1, ox1 = int(ceilf((float(x) + 1.5f) / sw - 0.5f)) + 1;
    oy0 = max(oy0, 0); oy1 = min(oy1, ho - 1); ox0 = max(ox0, 0); ox1 = min(ox1, wo - 1);
    if (y == 0) oy0 = 0;                     // clamped sources (src < 0) all land on index 0
    if (x == 0) ox0 = 0;
    float4 acc = make_float4(0, 0, 0, 0);
    const float4* src = reinterpret_cast<const float4*>(dout);
    for (int oy = oy0; oy <= oy1; ++oy) {
        int y0, y1; float ly0, ly1;
        bl_src(oy, hi, sh, y0, y1, ly0, ly1);
        const float wy = (y0 == y ? ly0 : 0.f) + (y1 == y ? ly1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            int x0, x1; float lx0, lx1;
            bl_src(ox, wi, sw, x0, x1, lx0, lx1);
            const float wx = (x0 == x ? lx0 : 0.f) + (x1 == x ? lx1 : 0.f);
            if (wx == 0.f) continue;
            const float4 v = src[((size_t(b) * ho + oy) * wo + ox) * out_cq + out_q0 + q];
            const float wgt = wy * wx;
            acc.x = fmaf(wgt, v.x, acc.x); acc.y = fmaf(wgt, v.y, acc.y); acc.z = fmaf(wgt, v.z, acc.z); acc.w = fmaf(wgt, v.w, acc.w);
        }
    }
    reinterpret_cast<float4*>(din)[i] = acc;
}
int launch_bilinear_bwd(const float* dout, int B, int C, int ho, int wo, int out_cstride, int out_coff, float* din, int hi,
                        int wi, hipStream_t st) {
    const long long n = (long long)B * hi * wi * (C / 4);
    if (!n) return 0;
    BlBwdArgs a{};
    a.dout[0] = dout; a.din[0] = din; a.hi[0] = hi; a.wi[0] = wi; a.ho[0] = ho; a.wo[0] = wo;
    a.B = B; a.cq = C / 4; a.out_cq = out_cstride / 4; a.out_q0 = out_coff / 4;
    hipLaunchKernelGGL(k_bilinear_bwd, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}
int launch_bilinear_bwd3(const float* const dout[3], int B, int C, const int ho[3], const int wo[3], int out_cstride, int out_coff,
                         float* const din[3], const int hi[3], const int wi[3], hipStream_t st) {
    BlBwdArgs a{};
    long long nmax = 0;
    for (int p = 0; p < 3; ++p) {
        a.dout[p] = dout[p]; a.din[p] = din[p]; a.hi[p] = hi[p]; a.wi[p] = wi[p]; a.ho[p] = ho[p]; a.wo[p] = wo[p];
        nmax = std::max(nmax, (long long)B * hi[p] * wi[p] * (C / 4));
    }
    if (!nmax) return 0;
    a.B = B; a.cq = C / 4; a.out_cq = out_cstride / 4; a.out_q0 = out_coff / 4;
    hipLaunchKernelGGL(k_bilinear_bwd, dim3((unsigned)((nmax + 255) / 256), 3), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ the timestep MLP backward (tiny)
// dW[o][i] = sum_b dy[b][o] * f(in[b][i]) ; db[o] = sum_b dy[b][o]        (f = identity, SiLU, or sinusoid of t)
__device__ __forceinline__ float lin_in(const float* in, int b, int i, int I, int mode) {
    if (mode == 2) {                                    // timestep_embedding (src/diffusion/nn.py:103-121): cos | sin
        const int half = I / 2;
        if (i >= 2 * half) return 0.f;
        const int k = i < half ? i : i - half;
        const float freq = expf(-logf(10000.0f) * float(k) / float(half));
        const float arg = in[b] * freq;
        return i < half ? cosf(arg) : sinf(arg);
    }
    const float v = in[size_t(b) * I + i];
    return mode == 1 ? v * (1.0f / (1.0f + expf(-v))) : v;
}
__global__ void k_linear_bwd_w(const float* __restrict__ dy, int dy_stride, const float* __restrict__ in, int B, int I, int O,
                               int in_mode, float* __restrict__ dW, float* __restrict__ db) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)O * I) return;
    const int i = int(idx % I), o = int(idx / I);
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) { const float d = dy[size_t(b) * dy_stride + o]; s = fmaf(d, lin_in(in, b, i, I, in_mode), s); sb += d; }
    dW[idx] = s;
    if (i == 0) db[o] = sb;
}
// The weight / bias gradients of several linears that read the same input and consecutive column ranges of one dy matrix (the
// residual blocks' emb_layers.1 on the stacked FiLM gradient), as ONE launch: row o of dy's columns belongs to the segment seg
// with seg_begin[seg] <= o (six 4.8-us launches per training step otherwise).  Same per-element arithmetic as k_linear_bwd_w.
constexpr int kLinSegs = 16;
struct LinSegArgs { float* dW[kLinSegs]; float* db[kLinSegs]; int seg_begin[kLinSegs + 1]; int nseg; };
__global__ void k_linear_bwd_w_multi(const float* __restrict__ dy, int dy_stride, const float* __restrict__ in, int B, int I, int in_mode,
                                     LinSegArgs sg) {
    const int O = sg.seg_begin[sg.nseg];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)O * I) return;
    const int i = int(idx % I), o = int(idx / I);
    int seg = 0;
#pragma unroll
    for (int k = 1; k < kLinSegs; ++k) seg += (k < sg.nseg && o >= sg.seg_begin[k]) ? 1 : 0;
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) { const float d = dy[size_t(b) * dy_stride + o]; s = fmaf(d, lin_in(in, b, i, I, in_mode), s); sb += d; }
    const int ol = o - sg.seg_begin[seg];
    sg.dW[seg][size_t(ol) * I + i] = s;
    if (i == 0) sg.db[seg][ol] = sb;
}
int launch_linear_bwd_w_multi(const float* dy, int dy_stride, const float* in, int B, int I, int in_mode, int nseg, const int* seg_begin,
                              float* const* dW, float* const* db, hipStream_t st) {
    S3D_CHECK(nseg >= 1 && nseg <= kLinSegs, S3D_ERR_INVALID, "linear_bwd_w_multi: %d segments", nseg);
    LinSegArgs sg;
    sg.nseg = nseg;
    for (int k = 0; k < nseg; ++k) { sg.dW[k] = dW[k]; sg.db[k] = db[k]; sg.seg_begin[k] = seg_begin[k]; }
    sg.seg_begin[nseg] = seg_begin[nseg];
    const long long n = (long long)seg_begin[nseg] * I;
    if (!n) return 0;
    hipLaunchKernelGGL(k_linear_bwd_w_multi, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, dy_stride, in, B, I, in_mode, sg);
    S3D_HIP(hipGetLastError());
    return 0;
}
// dx[b][i] = (sum_o dy[b][o] * W[o][i]) * (in_mode == 1 ? silu'(in[b][i]) : 1); block = (b, 32 inputs) x 32 output lanes
// (the FiLM projection has O = a few thousand outputs for 32 (b, input tile) blocks: the lanes split that loop, added in
// lane order through LDS)
constexpr int kLinBwdLanes = 32;
__global__ __launch_bounds__(32 * kLinBwdLanes) void k_linear_bwd_x(const float* __restrict__ dy, int dy_stride, const float* __restrict__ W,
                                                                    const float* __restrict__ in, int B, int I, int O, int in_mode,
                                                                    float* __restrict__ dx) {
    __shared__ float sm[kLinBwdLanes][32];
    const int il = threadIdx.x & 31, ol = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + il, b = blockIdx.y;
    float s = 0.f;
    if (i < I)
        for (int o = ol; o < O; o += kLinBwdLanes) s = fmaf(dy[size_t(b) * dy_stride + o], W[size_t(o) * I + i], s);
    sm[ol][il] = s;
    __syncthreads();
    if (ol == 0 && i < I) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < kLinBwdLanes; ++k) t += sm[k][il];
        if (in_mode == 1) {
            const float v = in[size_t(b) * I + i], sg = 1.0f / (1.0f + expf(-v));
            t *= sg * (1.0f + v * (1.0f - sg));
        }
        dx[size_t(b) * I + i] = t;
    }
}
int launch_linear_bwd(const float* dy, int dy_stride, const float* in, int B, int I, const float* W, int O, int in_mode, float* dW,
                      float* db, float* dx, hipStream_t st) {
    if (dW) {
        hipLaunchKernelGGL(k_linear_bwd_w, dim3((unsigned)(((long long)O * I + 255) / 256)), dim3(256), 0, st, dy, dy_stride, in, B, I,
                           O, in_mode, dW, db);
        S3D_HIP(hipGetLastError());
    }
    if (dx) {
        hipLaunchKernelGGL(k_linear_bwd_x, dim3(cdiv(I, 32), B), dim3(32 * kLinBwdLanes), 0, st, dy, dy_stride, W, in, B, I, O, in_mode, dx);
        S3D_HIP(hipGetLastError());
    }
    return 0;
}

// ------------------------------------------------------------------ diffusion training elementwise pieces
// q_sample (src/diffusion/gaussian_diffusion.py:189-207): x_t = sqrt(ac[t]) x0 + sqrt(1-ac[t]) eps
// (x0 rows are x0_bs elements apart: 0 for the single training triplane expanded to a batch, utils/triplane_util.py:64-69)
__global__ void k_q_sample(const float* __restrict__ x0, const float* __restrict__ eps, const float* __restrict__ sa,
                           const float* __restrict__ sb, const int64_t* __restrict__ t, long long per, int B,
                           float* __restrict__ xt, long long x0_bs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * B) return;
    const int b = int(i / per);
    xt[i] = sa[t[b]] * x0[b * x0_bs + (i - b * per)] + sb[t[b]] * eps[i];
}
// per-plane mean squared error of the composed maps (:838-851) and its gradient:
// terms[b][p] = mean over the plane's C*h*w elements of (target - out)^2 ; d_out = 2 (out - target) * wgt[b][p] / n_p
// (zero in the unused DxD corner).  Deterministic two-stage sum.
constexpr int kMseChunks = 32;
// tgt rows are tgt_bs elements apart (0: one target for the whole batch); terms [B][4]: xy, xz, yz and (xy + xz) + yz — the
// reference's terms["loss"] in its order (:851); wgt [B][wcols] (wcols 1: per sample, 3: per sample and plane) divided by wdiv
struct MseArgs { const float* out; const float* tgt; float* part; float* terms; const float* wgt; float* dout; int B, C, H, W, D;
                 long long tgt_bs; int wcols; float wdiv; };
__global__ __launch_bounds__(256) void k_mse_partials(MseArgs a) {
    __shared__ double red[256];
    const int chunk = blockIdx.x, p = blockIdx.y, b = blockIdx.z;
    const int h = p == 2 ? a.W : a.H, w = p == 0 ? a.W : a.D;
    const long long n = (long long)a.C * h * w;
    const long long i0 = n * chunk / kMseChunks, i1 = n * (chunk + 1) / kMseChunks;
    double s = 0;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const int c = int(i % w); long long r = i / w;
        const int rr = int(r % h), ch = int(r / h);
        const size_t k = composed_index(p, a.H, a.W, a.D, a.C, b, ch, rr, c);
        const size_t k0 = composed_index(p, a.H, a.W, a.D, a.C, 0, ch, rr, c);
        const float d = a.tgt[size_t(b) * a.tgt_bs + k0] - a.out[k];
        s += double(d * d);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (int(threadIdx.x) < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) a.part[(size_t(b) * 3 + p) * kMseChunks + chunk] = float(red[0]);
}
__global__ void k_mse_finalize(MseArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    float t3[3];
    for (int p = 0; p < 3; ++p) {
        const int h = p == 2 ? a.W : a.H, w = p == 0 ? a.W : a.D;
        double s = 0;
        for (int k = 0; k < kMseChunks; ++k) s += a.part[(size_t(b) * 3 + p) * kMseChunks + k];
        t3[p] = float(s / (double(a.C) * h * w));
        a.terms[b * 4 + p] = t3[p];
    }
    a.terms[b * 4 + 3] = (t3[0] + t3[1]) + t3[2];
}
__global__ void k_mse_grad(MseArgs a) {
    const long long per = (long long)a.C * (a.H + a.D) * (a.W + a.D);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * a.B) return;
    const int b = int(i / per);
    const int X = int(i % (a.W + a.D)), Y = int((i / (a.W + a.D)) % (a.H + a.D));
    float g = 0.f;
    if (!(Y >= a.H && X >= a.W)) {
        const int p = Y >= a.H ? 2 : (X >= a.W ? 1 : 0);
        const int h = p == 2 ? a.W : a.H, w = p == 0 ? a.W : a.D;
        const float wv = a.wgt[b * a.wcols + (a.wcols == 3 ? p : 0)] / a.wdiv;
        g = 2.0f * (a.out[i] - a.tgt[(long long)b * a.tgt_bs + (i - (long long)b * per)]) * wv / float((long long)a.C * h * w);
    }
    a.dout[i] = g;
}
int launch_q_sample(const float* x0, const float* eps, const float* sa, const float* sb, const int64_t* t, long long per, int B,
                    float* xt, hipStream_t st, long long x0_bs) {
    const long long n = per * B;
    if (!n) return 0;
    hipLaunchKernelGGL(k_q_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x0, eps, sa, sb, t, per, B, xt, x0_bs < 0 ? per : x0_bs);
    S3D_HIP(hipGetLastError());
    return 0;
}
int launch_mse_terms(const float* out, const float* tgt, int B, int C, int H, int W, int D, float* ws, float* terms, hipStream_t st, long long tgt_bs) {
    const long long per = (long long)C * (H + D) * (W + D);
    MseArgs a{out, tgt, ws, terms, nullptr, nullptr, B, C, H, W, D, tgt_bs < 0 ? per : tgt_bs, 3, 1.0f};
    if (!B) return 0;
    hipLaunchKernelGGL(k_mse_partials, dim3(kMseChunks, 3, B), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_mse_finalize, dim3(cdiv(B, 64)), dim3(64), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}
int launch_mse_grad(const float* out, const float* tgt, const float* wgt, int B, int C, int H, int W, int D, float* dout, hipStream_t st,
                    long long tgt_bs, int wcols, float wdiv) {
    const long long per = (long long)C * (H + D) * (W + D);
    MseArgs a{out, tgt, nullptr, nullptr, wgt, dout, B, C, H, W, D, tgt_bs < 0 ? per : tgt_bs, wcols, wdiv};
    const long long n = (long long)B * C * (H + D) * (W + D);
    if (!n) return 0;
    hipLaunchKernelGGL(k_mse_grad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ AdamW + EMA on the flat parameter vector
// torch.optim.AdamW (decoupled weight decay, bias correction) followed by update_ema (src/diffusion/nn.py:55-65)
// for up to 4 EMA copies, one pass over the parameters.
struct AdamArgs {
    float* p; const float* g; float* m; float* v; float* ema[4]; float ema_rate[4]; int n_ema;
    long long n; float lr, b1, b2, eps, wd, c1, c2s;          // c1 = 1-b1^t, c2s = sqrt(1-b2^t)
};
__global__ void k_adamw_ema(AdamArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    float p = a.p[i];
    const float g = a.g[i];
    p *= 1.0f - a.lr * a.wd;
    const float m = a.m[i] * a.b1 + (1.0f - a.b1) * g;             // exp_avg.lerp_(grad, 1-b1) == mul_(b1).add_(g, 1-b1) to 1 ulp
    const float v = a.v[i] * a.b2 + (1.0f - a.b2) * g * g;
    a.m[i] = m; a.v[i] = v;
    const float denom = sqrtf(v) / a.c2s + a.eps;
    p -= (a.lr / a.c1) * (m / denom);
    a.p[i] = p;
    for (int k = 0; k < a.n_ema; ++k) a.ema[k][i] = a.ema[k][i] * a.ema_rate[k] + (1.0f - a.ema_rate[k]) * p;
}
int launch_adamw_ema(float* p, const float* g, float* m, float* v, float* const* ema, const float* ema_rate, int n_ema, long long n,
                     float lr, float b1, float b2, float eps, float wd, int step, hipStream_t st) {
    S3D_CHECK(n_ema >= 0 && n_ema <= 4 && step >= 1, S3D_ERR_INVALID, "adamw_ema: n_ema=%d step=%d", n_ema, step);
    AdamArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n_ema = n_ema; a.n = n;
    for (int k = 0; k < 4; ++k) { a.ema[k] = k < n_ema ? ema[k] : nullptr; a.ema_rate[k] = k < n_ema ? ema_rate[k] : 0.f; }
    a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = wd;
    a.c1 = float(1.0 - pow(double(b1), step)); a.c2s = float(sqrt(1.0 - pow(double(b2), step)));
    if (!n) return 0;
    hipLaunchKernelGGL(k_adamw_ema, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

}  // namespace s3d
