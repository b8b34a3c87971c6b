// s3d_decoder.hip — AutoEncoderGroupSkip.decode on MI355X (src/encoding/networks.py:192-220).
//
// Two stages:
//  (1) prepare_triplane: the per-plane TriplaneGroupResnetBlock (conv5x5 -> InstanceNorm -> SiLU -> conv5x5 + 1x1
//      shortcut, src/encoding/blocks.py:189-256) for the geo and tex feature groups, run ONCE per triplane on the
//      MFMA convolution of s3d_conv.hip (the reference recomputes it for every 16 384-point chunk,
//      src/encoding/model.py:327-330; the result does not depend on the points).
//  (2) decode: one fused kernel per 128 points — bilinear border-clamped gather of the three feature planes
//      (F.grid_sample semantics, networks.py:182-190) straight into MFMA operand registers, then both
//      DecoderMLPSkipConcat chains (blocks.py:65-91) on the fp32 matrix cores without the activations ever
//      leaving the register file:
//        * the GEMM is evaluated transposed, D[hidden unit][point] = W · Xᵀ, so a layer's accumulator (lane =
//          point, registers = 16 hidden rows of a 32-row tile) IS the next layer's B operand: row order
//          (r&3) + 8(r>>2) + 4(lane>>5) is exactly the k-permutation "lane half 0 takes k0..3, half 1 k4..7" that
//          lets the weight operand be fetched with one ds_read_b128 per 4 MFMAs;
//        * weights stream through LDS in [rows][32 k] slabs shared by the block's 4 waves (each wave = 32 points),
//          register-prefetched one slab ahead, one barrier per slab.
//      Bound: fp32 MFMA (1.18 MFLOP per point, 16 B written per point).
#include <algorithm>
#include <cmath>
#include <memory>

#include "s3d_ae.h"

namespace s3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gf4p;
__device__ __forceinline__ gf4p g4(const float* p) { return (gf4p)(uintptr_t)p; }

static const char* kPl[3] = {"xy", "xz", "yz"};
static inline int rup32(int v) { return (v + 31) / 32 * 32; }

// ------------------------------------------------------------------ small kernels of the plane block
// channels [c0, c0+cin) of an NCHW [1][Ctot][h][w] plane -> NHWC [h][w][32] zero-padded
__global__ void k_slice_pad(const float* __restrict__ in, float* __restrict__ out, int hw, int c0, int cin) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)hw * 32) return;
    const int c = int(i & 31);
    const long long pix = i >> 5;
    out[i] = c < cin ? in[size_t(c0 + c) * hw + pix] : 0.f;
}

// InstanceNorm2d(C, eps=1e-6, affine) + SiLU over one NHWC plane [hw][C]  (src/encoding/blocks.py:219-221, 94-96)
constexpr int kInChunks = 64;
__global__ void k_chan_partials(const float* __restrict__ x, double* __restrict__ part, int hw, int C) {
    // grid (kInChunks), block: C/4 quads x pl pixel lanes; part[chunk][C][2]
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sm = reinterpret_cast<double*>(smem_raw);
    const int cq = C / 4, pl = blockDim.x / cq;
    const int q = threadIdx.x % cq, l = threadIdx.x / cq;
    const int per = (hw + kInChunks - 1) / kInChunks;
    const int p0 = blockIdx.x * per, p1 = min(hw, p0 + per);
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (int pix = p0 + l; pix < p1; pix += pl) {
        const float4 v = reinterpret_cast<const float4*>(x)[size_t(pix) * cq + q];
        s[0] += v.x; ss[0] += double(v.x) * v.x; s[1] += v.y; ss[1] += double(v.y) * v.y;
        s[2] += v.z; ss[2] += double(v.z) * v.z; s[3] += v.w; ss[3] += double(v.w) * v.w;
    }
    for (int k = 0; k < 4; ++k) { sm[(size_t(l) * C + 4 * q + k) * 2] = s[k]; sm[(size_t(l) * C + 4 * q + k) * 2 + 1] = ss[k]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double S = 0, SS = 0;
        for (int ll = 0; ll < pl; ++ll) { S += sm[(size_t(ll) * C + c) * 2]; SS += sm[(size_t(ll) * C + c) * 2 + 1]; }
        part[(size_t(blockIdx.x) * C + c) * 2] = S; part[(size_t(blockIdx.x) * C + c) * 2 + 1] = SS;
    }
}
__global__ void k_inorm_silu(const float* __restrict__ x, const double* __restrict__ part, const float* __restrict__ gamma,
                             const float* __restrict__ beta, float* __restrict__ y, int hw, int C, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* A = reinterpret_cast<float*>(smem_raw); float* Bc = A + C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double S = 0, SS = 0;
        for (int k = 0; k < kInChunks; ++k) { S += part[(size_t(k) * C + c) * 2]; SS += part[(size_t(k) * C + c) * 2 + 1]; }
        const double m = S / hw;
        double var = SS / hw - m * m; if (var < 0) var = 0;
        const float scale = float(1.0 / sqrt(var + double(eps))) * gamma[c];
        A[c] = scale; Bc[c] = beta[c] - scale * float(m);
    }
    __syncthreads();
    const int cq = C / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)hw * cq; i += (long long)gridDim.x * blockDim.x) {
        const int q = int(i % cq);
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 a = reinterpret_cast<const float4*>(A)[q], b = reinterpret_cast<const float4*>(Bc)[q];
        float4 o;
        o.x = fmaf(v.x, a.x, b.x); o.y = fmaf(v.y, a.y, b.y); o.z = fmaf(v.z, a.z, b.z); o.w = fmaf(v.w, a.w, b.w);
        o.x = o.x / (1.f + expf(-o.x)); o.y = o.y / (1.f + expf(-o.y)); o.z = o.z / (1.f + expf(-o.z)); o.w = o.w / (1.f + expf(-o.w));
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

int launch_inorm_silu(const float* x, double* part, const float* gamma, const float* beta, float* y, int hw, int C, float eps,
                      hipStream_t st) {
    static_assert(kInNormChunks == kInChunks, "chunk count");
    const int cq = C / 4, pl = std::max(1, 256 / cq);
    hipLaunchKernelGGL(k_chan_partials, dim3(kInChunks), dim3(cq * pl), size_t(pl) * C * 2 * sizeof(double), st, x, part, hw, C);
    hipLaunchKernelGGL(k_inorm_silu, dim3(std::min(1024, (hw * cq + 255) / 256)), dim3(256), size_t(2) * C * sizeof(float), st, x, part,
                       gamma, beta, y, hw, C, eps);
    S3D_HIP(hipGetLastError());
    return 0;
}

// The auto-encoder training tier's form (round 5): three planes per launch (blockIdx.y), the statistics finished by their own
// small launch between the two (launch_mr_from_partials3, s3d_ae_kernels.hip — every block of the apply pass used to walk the 64
// chunk records of its channels itself, 10 of its 18 us)
struct InNorm3Args { const float* x[3]; float* y[3]; double* part[3]; const float* gamma[3]; const float* beta[3]; const float* mr; int hw[3]; int C; };
__global__ void k_chan_partials3(InNorm3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sm = reinterpret_cast<double*>(smem_raw);
    const int p = blockIdx.y, C = a.C, hw = a.hw[p];
    const float* x = a.x[p];
    double* part = a.part[p];
    const int cq = C / 4, pl = blockDim.x / cq;
    const int q = threadIdx.x % cq, l = threadIdx.x / cq;
    const int per = (hw + kInChunks - 1) / kInChunks;
    const int p0 = blockIdx.x * per, p1 = min(hw, p0 + per);
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (int pix = p0 + l; pix < p1; pix += pl) {
        const float4 v = reinterpret_cast<const float4*>(x)[size_t(pix) * cq + q];
        s[0] += v.x; ss[0] += double(v.x) * v.x; s[1] += v.y; ss[1] += double(v.y) * v.y;
        s[2] += v.z; ss[2] += double(v.z) * v.z; s[3] += v.w; ss[3] += double(v.w) * v.w;
    }
    for (int k = 0; k < 4; ++k) { sm[(size_t(l) * C + 4 * q + k) * 2] = s[k]; sm[(size_t(l) * C + 4 * q + k) * 2 + 1] = ss[k]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double S = 0, SS = 0;
        for (int ll = 0; ll < pl; ++ll) { S += sm[(size_t(ll) * C + c) * 2]; SS += sm[(size_t(ll) * C + c) * 2 + 1]; }
        part[(size_t(blockIdx.x) * C + c) * 2] = S; part[(size_t(blockIdx.x) * C + c) * 2 + 1] = SS;
    }
}
__global__ void k_inorm_silu3(InNorm3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int p = blockIdx.y, C = a.C, hw = a.hw[p];
    float* A = reinterpret_cast<float*>(smem_raw); float* Bc = A + C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float m = a.mr[(size_t(p) * C + c) * 2], scale = a.mr[(size_t(p) * C + c) * 2 + 1] * a.gamma[p][c];
        A[c] = scale; Bc[c] = a.beta[p][c] - scale * m;
    }
    __syncthreads();
    const int cq = C / 4;
    const float* x = a.x[p]; float* y = a.y[p];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)hw * cq; i += (long long)gridDim.x * blockDim.x) {
        const int q = int(i % cq);
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 s = reinterpret_cast<const float4*>(A)[q], b = reinterpret_cast<const float4*>(Bc)[q];
        float4 o;
        o.x = fmaf(v.x, s.x, b.x); o.y = fmaf(v.y, s.y, b.y); o.z = fmaf(v.z, s.z, b.z); o.w = fmaf(v.w, s.w, b.w);
        o.x = o.x / (1.f + expf(-o.x)); o.y = o.y / (1.f + expf(-o.y)); o.z = o.z / (1.f + expf(-o.z)); o.w = o.w / (1.f + expf(-o.w));
        reinterpret_cast<float4*>(y)[i] = o;
    }
}
int launch_mr_from_partials3(double* const part[3], int nchunks, int C, const size_t hw[3], float eps, float* mr, hipStream_t st);
int launch_inorm_silu3(float* const x[3], double* const part[3], const float* const gamma[3], const float* const beta[3], float* const y[3],
                       const size_t hw[3], int C, float eps, float* mr, hipStream_t st) {
    InNorm3Args a; a.C = C; a.mr = mr;
    int mx = 0;
    for (int p = 0; p < 3; ++p) {
        a.x[p] = x[p]; a.y[p] = y[p]; a.part[p] = part[p]; a.gamma[p] = gamma[p]; a.beta[p] = beta[p]; a.hw[p] = int(hw[p]);
        mx = std::max(mx, a.hw[p]);
    }
    const int cq = C / 4, pl = std::max(1, 256 / cq);
    hipLaunchKernelGGL(k_chan_partials3, dim3(kInChunks, 3), dim3(cq * pl), size_t(pl) * C * 2 * sizeof(double), st, a);
    S3D_HIP(hipGetLastError());
    S3D_TRY(launch_mr_from_partials3(part, kInChunks, C, hw, eps, mr, st));
    hipLaunchKernelGGL(k_inorm_silu3, dim3(std::min(1024, (mx * cq + 255) / 256), 3), dim3(256), size_t(2) * C * sizeof(float), st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ fused gather + MLP
struct MlpW {                 // device pointers of one DecoderMLPSkipConcat, padded to multiples of 32
    const float* w[6]; const float* b[6];
};
struct DecodeArgs {
    const float* pts;         // [N][3] or null -> cell-centred grid points generated on the fly
    long long N;
    float amin[3];            // aabb min
    int gdim[3]; float gsize[3];   // decode_grid: resolutions and aabb size
    const float* feat[2][3];  // [geo|tex][plane] NHWC [h][w][UP]
    int ph[3], pw[3];
    MlpW mlp[2];
    int nout[2];              // 1, tex_channels
    int clamp_color;
    float* out;               // [N][1 + tex_channels]
    int out_stride;
};

constexpr int kSlabLd = 36;   // padded slab row (floats)

// One layer on the matrix cores: hout[m] (MT tiles of 32 rows) = W[MT*32][K] x [in0 | in1] + bias, optional ReLU.
// in0 has KT0 tiles of 32 rows, in1 KT1.  All four waves run it in lockstep (they share the LDS weight slabs).
template <int KT0, int KT1, int MT>
__device__ __forceinline__ void mlp_layer(const float* __restrict__ Wg, const float* __restrict__ bias,
                                          const f32x16* in0, const f32x16* in1, f32x16* hout, float* lds, bool relu) {
    constexpr int KT = KT0 + KT1, K = KT * 32, M = MT * 32;
    constexpr int ITEMS = M * 8, NI = (ITEMS + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, half = lane >> 5;
    // accumulators start at the bias of their rows
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) hout[m][r] = bias[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
    // slab staging descriptors: item -> (row, float4 q)
    gf4p src[NI]; int dst[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int idx = min(it * 256 + tid, ITEMS - 1);
        const int row = idx >> 3, q = idx & 7;
        src[it] = g4(Wg + size_t(row) * K + q * 4);
        dst[it] = row * kSlabLd + q * 4;
    }
    f32x4 rg[NI];
    __syncthreads();                                   // previous layer's last slab reads are done
#pragma unroll
    for (int it = 0; it < NI; ++it) rg[it] = src[it][0];
#pragma unroll
    for (int it = 0; it < NI; ++it) *reinterpret_cast<f32x4*>(lds + dst[it]) = rg[it];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        constexpr int kLastT = KT - 1;
        const int nt = t == kLastT ? t : t + 1;        // last slab re-fetches itself (harmless)
#pragma unroll
        for (int it = 0; it < NI; ++it) rg[it] = src[it][nt * 8];
        __builtin_amdgcn_sched_barrier(0);
        const float* slab = lds + (t & 1) * (M * kSlabLd);
        const f32x16& hin = t < KT0 ? in0[t < KT0 ? t : 0] : in1[t >= KT0 ? t - KT0 : 0];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a4[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a4[m] = *reinterpret_cast<const f32x4*>(slab + (m * 32 + j) * kSlabLd + q * 8 + half * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    hout[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m][e], hin[q * 4 + e], hout[m], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        float* nxt = lds + ((t + 1) & 1) * (M * kSlabLd);
#pragma unroll
        for (int it = 0; it < NI; ++it) *reinterpret_cast<f32x4*>(nxt + dst[it]) = rg[it];
        __syncthreads();
    }
    if (relu) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) hout[m][r] = fmaxf(hout[m][r], 0.f);
    }
}

// bilinear, padding_mode='border', align_corners=False sample of plane `fm` [h][w][UPT*32] at (u -> rows, v -> cols),
// accumulated into the lane's operand registers: x[t][4q+e] is channel 32t + 8q + 4*half + e.
template <int UPT>
__device__ __forceinline__ void gather_plane(const float* __restrict__ fm, int h, int w, float u, float v, int half,
                                             f32x16* x) {
    constexpr int C = UPT * 32;
    float fy = ((u + 1.f) * float(h) - 1.f) * 0.5f, fx = ((v + 1.f) * float(w) - 1.f) * 0.5f;
    fy = fminf(fmaxf(fy, 0.f), float(h - 1)); fx = fminf(fmaxf(fx, 0.f), float(w - 1));
    const int y0 = int(floorf(fy)), x0 = int(floorf(fx));
    const float ty = fy - float(y0), tx = fx - float(x0);
    const float w00 = (1.f - ty) * (1.f - tx), w01 = (1.f - ty) * tx, w10 = ty * (1.f - tx), w11 = ty * tx;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float* p00 = fm + (size_t(y0) * w + x0) * C + half * 4;
    const float* p01 = fm + (size_t(y0) * w + x1) * C + half * 4;
    const float* p10 = fm + (size_t(y1) * w + x0) * C + half * 4;
    const float* p11 = fm + (size_t(y1) * w + x1) * C + half * 4;
#pragma unroll
    for (int t = 0; t < UPT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = t * 32 + q * 8;
            const f32x4 a = g4(p00 + c)[0], b = g4(p01 + c)[0], cc = g4(p10 + c)[0], d = g4(p11 + c)[0];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[t][q * 4 + e] += w00 * a[e] + w01 * b[e] + w10 * cc[e] + w11 * d[e];
        }
}

template <int UPT, int HIDT>
__global__ __launch_bounds__(256, 1) void k_decode(DecodeArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * HIDT * 32 * kSlabLd];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, half = lane >> 5;
    const long long pt = (long long)blockIdx.x * 128 + wave * 32 + j;
    const bool live = pt < a.N;
    float p[3] = {0.f, 0.f, 0.f};
    if (live) {
        if (a.pts) { p[0] = a.pts[pt * 3]; p[1] = a.pts[pt * 3 + 1]; p[2] = a.pts[pt * 3 + 2]; }
        else {        // sample_grid_points_aabb (src/encoding/utils3d.py:13-25): cell centres, 'ij' order
            const long long iz = pt % a.gdim[2], iy = (pt / a.gdim[2]) % a.gdim[1], ix = pt / ((long long)a.gdim[2] * a.gdim[1]);
            // linspace(0.5, r-0.5, r) / r * size + min, one rounding per torch op (no fma contraction)
            p[0] = __fadd_rn(__fmul_rn(__fdiv_rn(0.5f + float(ix), float(a.gdim[0])), a.gsize[0]), a.amin[0]);
            p[1] = __fadd_rn(__fmul_rn(__fdiv_rn(0.5f + float(iy), float(a.gdim[1])), a.gsize[1]), a.amin[1]);
            p[2] = __fadd_rn(__fmul_rn(__fdiv_rn(0.5f + float(iz), float(a.gdim[2])), a.gsize[2]), a.amin[2]);
        }
    }
    float qn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)                                                    // x = 2 * (x - min) / (max - min) - 1  (:196)
        qn[k] = __fsub_rn(__fdiv_rn(2.f * __fsub_rn(p[k], a.amin[k]), a.gsize[k]), 1.f);
    const float uu[3] = {qn[0], qn[0], qn[1]}, vv[3] = {qn[1], qn[2], qn[2]};      // coords_list [[0,1],[0,2],[1,2]] (:201)

    float result[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        f32x16 x[UPT], hA[HIDT], hB[HIDT], ho[1];
#pragma unroll
        for (int t = 0; t < UPT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[t][r] = 0.f;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) gather_plane<UPT>(a.feat[net][pl], a.ph[pl], a.pw[pl], uu[pl], vv[pl], half, x);
        const MlpW& M = a.mlp[net];
        mlp_layer<UPT, 0, HIDT>(M.w[0], M.b[0], x, x, hA, lds, true);          // first_layers
        mlp_layer<HIDT, 0, HIDT>(M.w[1], M.b[1], hA, hA, hB, lds, true);
        mlp_layer<HIDT, 0, HIDT>(M.w[2], M.b[2], hB, hB, hA, lds, true);
        mlp_layer<UPT, HIDT, HIDT>(M.w[3], M.b[3], x, hA, hB, lds, true);        // second_layers on cat([x, h])
        mlp_layer<HIDT, 0, HIDT>(M.w[4], M.b[4], hB, hB, hA, lds, true);
        mlp_layer<HIDT, 0, 1>(M.w[5], M.b[5], hA, hA, ho, lds, false);
        // rows 0..3 of the output tile live in registers 0..3 of lane half 0
        if (net == 0) result[0] = ho[0][0];
        else { result[1] = ho[0][0]; result[2] = ho[0][1]; result[3] = ho[0][2]; }
    }
    if (live && half == 0) {
        float* o = a.out + pt * a.out_stride;
        o[0] = result[0];
        for (int k = 0; k < a.nout[1]; ++k) {
            float c = 1.f / (1.f + expf(-result[1 + k]));                        // .sigmoid() (:216)
            if (a.clamp_color) c = fminf(fmaxf(c, 0.f), 1.f);                    // decode_batch clamp (model.py:332)
            o[1 + k] = c;
        }
    }
}

}  // namespace s3d

using namespace s3d;

struct s3d_decoder {
    s3d_decoder_cfg cfg;
    struct Spec { std::string name; std::vector<int64_t> shape; };
    std::vector<Spec> specs;
    std::map<std::string, std::vector<float>> host;
    bool packed = false;
    int up_p = 0, hid_p = 0;
    DevBuf wbuf;
    struct Net { ConvW cin, cout_, sc; size_t gamma[3], beta[3]; size_t mw[6], mb[6]; int cin_ch; } net[2];
    // prepared features
    DevBuf feat;
    float* featp[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    int ph[3] = {0, 0, 0}, pw[3] = {0, 0, 0};
    bool prepared = false;
    Arena arena;
    const float* dev(size_t off) const { return static_cast<const float*>(wbuf.p) + off; }
};

namespace s3d {

static void dec_specs(s3d_decoder* d) {
    const s3d_decoder_cfg& c = d->cfg;
    const int up = c.feat_channel_up, hid = c.mlp_hidden_channels, n = c.mlp_hidden_layers / 2;
    auto add = [&](const std::string& nme, std::vector<int64_t> sh) { d->specs.push_back({nme, sh}); };
    const char* pre[2] = {"geo", "tex"};
    const int cin[2] = {c.geo_feat_channels, c.tex_feat_channels}, cout[2] = {1, c.tex_channels};
    for (int k = 0; k < 2; ++k) {
        const std::string cv = std::string(pre[k]) + "_convs";
        add(cv + ".in_layers.0.weight", {3 * up, cin[k], 5, 5}); add(cv + ".in_layers.0.bias", {3 * up});
        for (int p = 0; p < 3; ++p) { add(cv + ".norm_" + kPl[p] + ".weight", {up}); add(cv + ".norm_" + kPl[p] + ".bias", {up}); }
        add(cv + ".out_layers.1.weight", {3 * up, up, 5, 5}); add(cv + ".out_layers.1.bias", {3 * up});
        add(cv + ".shortcut.weight", {3 * up, cin[k], 1, 1}); add(cv + ".shortcut.bias", {3 * up});
        const std::string ml = std::string(pre[k]) + "_decoder";
        add(ml + ".first_layers.0.weight", {hid, up}); add(ml + ".first_layers.0.bias", {hid});
        for (int i = 0; i < n; ++i) {
            add(ml + ".first_layers." + std::to_string(2 * (i + 1)) + ".weight", {hid, hid});
            add(ml + ".first_layers." + std::to_string(2 * (i + 1)) + ".bias", {hid});
        }
        add(ml + ".second_layers.0.weight", {hid, up + hid}); add(ml + ".second_layers.0.bias", {hid});
        for (int i = 0; i < n - 1; ++i) {
            add(ml + ".second_layers." + std::to_string(2 * (i + 1)) + ".weight", {hid, hid});
            add(ml + ".second_layers." + std::to_string(2 * (i + 1)) + ".bias", {hid});
        }
        add(ml + ".second_layers." + std::to_string(2 * n) + ".weight", {cout[k], hid});
        add(ml + ".second_layers." + std::to_string(2 * n) + ".bias", {cout[k]});
    }
}

// dense [out][in] -> zero-padded [outp][inp] with column segments (in = seg0 | seg1 -> seg0p | seg1p)
static size_t pack_linear(std::vector<float>& st, const std::vector<float>& W, int out, int seg0, int seg1, int outp,
                          int seg0p, int seg1p) {
    const int in = seg0 + seg1, inp = seg0p + seg1p;
    size_t off = push(st, nullptr, size_t(outp) * inp);
    float* d = st.data() + off;
    std::fill(d, d + size_t(outp) * inp, 0.f);
    for (int o = 0; o < out; ++o) {
        for (int i = 0; i < seg0; ++i) d[size_t(o) * inp + i] = W[size_t(o) * in + i];
        for (int i = 0; i < seg1; ++i) d[size_t(o) * inp + seg0p + i] = W[size_t(o) * in + seg0 + i];
    }
    return off;
}
static size_t pack_vec(std::vector<float>& st, const float* v, int n, int np) {
    size_t off = push(st, nullptr, np);
    std::fill(st.begin() + off, st.begin() + off + np, 0.f);
    std::copy(v, v + n, st.begin() + off);
    return off;
}

static int dec_pack(s3d_decoder* d) {
    for (const auto& sp : d->specs)
        S3D_CHECK(d->host.count(sp.name), S3D_ERR_MISSING, "decoder parameter '%s' was never set", sp.name.c_str());
    const s3d_decoder_cfg& c = d->cfg;
    const int up = c.feat_channel_up, hid = c.mlp_hidden_channels, upp = d->up_p, hidp = d->hid_p;
    std::vector<float> st;
    const char* pre[2] = {"geo", "tex"};
    const int cin[2] = {c.geo_feat_channels, c.tex_feat_channels}, cout[2] = {1, c.tex_channels};
    for (int k = 0; k < 2; ++k) {
        auto& N = d->net[k];
        N.cin_ch = cin[k];
        const std::string cv = std::string(pre[k]) + "_convs";
        const auto& Wi = d->host.at(cv + ".in_layers.0.weight"); const auto& bi = d->host.at(cv + ".in_layers.0.bias");
        const auto& Wo = d->host.at(cv + ".out_layers.1.weight"); const auto& bo = d->host.at(cv + ".out_layers.1.bias");
        const auto& Ws = d->host.at(cv + ".shortcut.weight"); const auto& bs = d->host.at(cv + ".shortcut.bias");
        // pad every conv to (cin 32, cout upp): [tap][coutp][cinp]
        auto pack_conv = [&](const std::vector<float>& W, const std::vector<float>& b, int ci, int cip, int kk, ConvW& cw) {
            cw.cin = cip; cw.cout = upp; cw.k = kk; cw.rollout = false;
            const int taps = kk * kk;
            for (int p = 0; p < 3; ++p) {
                cw.bias[p] = pack_vec(st, b.data() + size_t(p) * up, up, upp);
                cw.dense[p] = push(st, nullptr, size_t(taps) * upp * cip);
                float* dd = st.data() + cw.dense[p];
                std::fill(dd, dd + size_t(taps) * upp * cip, 0.f);
                for (int t = 0; t < taps; ++t)
                    for (int co = 0; co < up; ++co)
                        for (int ch = 0; ch < ci; ++ch)
                            dd[(size_t(t) * upp + co) * cip + ch] = W[((size_t(p) * up + co) * ci + ch) * taps + t];
            }
        };
        pack_conv(Wi, bi, cin[k], 32, 5, N.cin);
        pack_conv(Wo, bo, up, upp, 5, N.cout_);
        pack_conv(Ws, bs, cin[k], 32, 1, N.sc);
        for (int p = 0; p < 3; ++p) {
            N.gamma[p] = pack_vec(st, d->host.at(cv + ".norm_" + kPl[p] + ".weight").data(), up, upp);
            N.beta[p] = pack_vec(st, d->host.at(cv + ".norm_" + kPl[p] + ".bias").data(), up, upp);
        }
        const std::string ml = std::string(pre[k]) + "_decoder";
        const char* ln[6] = {".first_layers.0", ".first_layers.2", ".first_layers.4", ".second_layers.0", ".second_layers.2", ".second_layers.4"};
        const int outs[6] = {hid, hid, hid, hid, hid, cout[k]};
        const int outp[6] = {hidp, hidp, hidp, hidp, hidp, 32};
        const int s0[6] = {up, hid, hid, up, hid, hid}, s1[6] = {0, 0, 0, hid, 0, 0};
        const int s0p[6] = {upp, hidp, hidp, upp, hidp, hidp}, s1p[6] = {0, 0, 0, hidp, 0, 0};
        for (int l = 0; l < 6; ++l) {
            N.mw[l] = pack_linear(st, d->host.at(ml + ln[l] + ".weight"), outs[l], s0[l], s1[l], outp[l], s0p[l], s1p[l]);
            N.mb[l] = pack_vec(st, d->host.at(ml + ln[l] + ".bias").data(), outs[l], outp[l]);
        }
    }
    S3D_TRY(upload(d->wbuf, st.data(), st.size() * sizeof(float)));
    d->packed = true;
    return 0;
}

template <int UPT, int HIDT>
static int launch_decode(const DecodeArgs& a, hipStream_t st) {
    const long long blocks = (a.N + 127) / 128;
    if (!blocks) return 0;
    hipLaunchKernelGGL((k_decode<UPT, HIDT>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

static int run_decode(s3d_decoder* d, const float* pts, long long N, const float aabb[6], const int* gdim, int clamp,
                      float* out, hipStream_t st) {
    S3D_CHECK(d->prepared, S3D_ERR_INVALID, "decoder: call s3d_decoder_prepare_triplane first");
    DecodeArgs a; memset(&a, 0, sizeof a);
    a.pts = pts; a.N = N; a.clamp_color = clamp; a.out = out; a.out_stride = 1 + d->cfg.tex_channels;
    for (int k = 0; k < 3; ++k) {
        a.amin[k] = aabb[k]; a.gsize[k] = aabb[3 + k] - aabb[k];
        a.gdim[k] = gdim ? gdim[k] : 1; a.ph[k] = d->ph[k]; a.pw[k] = d->pw[k];
    }
    for (int n = 0; n < 2; ++n) {
        for (int p = 0; p < 3; ++p) a.feat[n][p] = d->featp[n][p];
        for (int l = 0; l < 6; ++l) { a.mlp[n].w[l] = d->dev(d->net[n].mw[l]); a.mlp[n].b[l] = d->dev(d->net[n].mb[l]); }
    }
    a.nout[0] = 1; a.nout[1] = d->cfg.tex_channels;
    const int upt = d->up_p / 32, hidt = d->hid_p / 32;
    if (upt == 2 && hidt == 8) return launch_decode<2, 8>(a, st);
    if (upt == 1 && hidt == 1) return launch_decode<1, 1>(a, st);
    if (upt == 1 && hidt == 8) return launch_decode<1, 8>(a, st);
    set_error("decoder: feat_channel_up=%d / mlp_hidden_channels=%d has no compiled kernel (supported after padding to 32: "
              "up<=64 with hidden 256, or up<=32 with hidden 32)", d->cfg.feat_channel_up, d->cfg.mlp_hidden_channels);
    return S3D_ERR_UNSUPPORTED;
}

}  // namespace s3d

extern "C" {

int s3d_decoder_create(const s3d_decoder_cfg* cfg, s3d_decoder** out) {
    S3D_CHECK(cfg && out, S3D_ERR_INVALID, "decoder_create: null argument");
    S3D_CHECK(cfg->mlp_hidden_layers == 4, S3D_ERR_UNSUPPORTED, "decoder: mlp_hidden_layers=%d (only the default 4 is built)", cfg->mlp_hidden_layers);
    S3D_CHECK(cfg->geo_feat_channels >= 1 && cfg->geo_feat_channels <= 32 && cfg->tex_feat_channels >= 1 && cfg->tex_feat_channels <= 32,
              S3D_ERR_UNSUPPORTED, "decoder: feature groups must have 1..32 channels");
    S3D_CHECK(cfg->tex_channels >= 1 && cfg->tex_channels <= 3, S3D_ERR_UNSUPPORTED, "decoder: tex_channels must be 1..3");
    S3D_CHECK(cfg->feat_channel_up >= 1 && cfg->mlp_hidden_channels >= 1, S3D_ERR_INVALID, "decoder: bad widths");
    std::unique_ptr<s3d_decoder> d(new s3d_decoder());
    d->cfg = *cfg;
    d->up_p = rup32(cfg->feat_channel_up);
    d->hid_p = rup32(cfg->mlp_hidden_channels);
    dec_specs(d.get());
    *out = d.release();
    return 0;
}
void s3d_decoder_destroy(s3d_decoder* d) { delete d; }
int s3d_decoder_num_params(const s3d_decoder* d) { return d ? int(d->specs.size()) : S3D_ERR_INVALID; }
int s3d_decoder_param_info(const s3d_decoder* d, int i, const char** name, int64_t shape[4], int* ndim) {
    S3D_CHECK(d && i >= 0 && i < int(d->specs.size()), S3D_ERR_INVALID, "decoder param_info: index %d out of range", i);
    if (name) *name = d->specs[i].name.c_str();
    if (ndim) *ndim = int(d->specs[i].shape.size());
    if (shape) for (size_t k = 0; k < d->specs[i].shape.size(); ++k) shape[k] = d->specs[i].shape[k];
    return 0;
}
int s3d_decoder_set_param(s3d_decoder* d, const char* name, const float* data, const int64_t* shape, int ndim) {
    S3D_CHECK(d && name && data && shape, S3D_ERR_INVALID, "decoder set_param: null argument");
    for (const auto& sp : d->specs) {
        if (sp.name != name) continue;
        bool ok = int(sp.shape.size()) == ndim;
        size_t n = 1;
        for (int k = 0; ok && k < ndim; ++k) { ok = sp.shape[k] == shape[k]; n *= size_t(shape[k]); }
        S3D_CHECK(ok, S3D_ERR_INVALID, "size mismatch for %s", name);
        d->host[sp.name].assign(data, data + n);
        d->packed = false;
        return 0;
    }
    set_error("unexpected key '%s' in decoder state_dict", name);
    return S3D_ERR_INVALID;
}

int s3d_decoder_prepare_triplane(s3d_decoder* d, const float* xy, const float* xz, const float* yz, int H, int W, int D,
                                 void* stream) {
    S3D_CHECK(d && xy && xz && yz && H >= 1 && W >= 1 && D >= 1, S3D_ERR_INVALID, "decoder prepare: bad argument");
    if (!d->packed) S3D_TRY(dec_pack(d));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Geo g = Geo::from_hwd(H, W, D);
    const float* in[3] = {xy, xz, yz};
    const int upp = d->up_p;
    // persistent outputs: 2 nets x 3 planes x [h][w][upp]
    size_t tot = 0;
    for (int p = 0; p < 3; ++p) tot += size_t(g.h[p]) * g.w[p] * upp;
    S3D_HIP(hipStreamSynchronize(st));               // the feature buffer may still be read by an earlier decode
    S3D_TRY(d->feat.reserve(2 * tot * sizeof(float)));
    {
        float* base = static_cast<float*>(d->feat.p);
        for (int n = 0; n < 2; ++n)
            for (int p = 0; p < 3; ++p) { d->featp[n][p] = base; base += size_t(g.h[p]) * g.w[p] * upp; }
    }
    for (int p = 0; p < 3; ++p) { d->ph[p] = g.h[p]; d->pw[p] = g.w[p]; }
    // temporaries (two passes: measure, then run)
    for (int pass = 0; pass < 2; ++pass) {
        Arena& ar = d->arena;
        ar.measuring = pass == 0;
        if (pass == 0) ar.high = 0;
        else if (ar.high > ar.buf.cap) S3D_TRY(ar.buf.reserve(ar.high));
        ar.reset();
        for (int n = 0; n < 2; ++n) {
            auto& N = d->net[n];
            const int c0 = n == 0 ? 0 : d->cfg.geo_feat_channels;
            Tri x, a, y, s; x.C = 32; a.C = y.C = s.C = upp; x.g = a.g = y.g = s.g = g;
            double* part[3];
            for (int p = 0; p < 3; ++p) {
                const size_t hw = size_t(g.h[p]) * g.w[p];
                x.p[p] = ar.alloc<float>(hw * 32); a.p[p] = ar.alloc<float>(hw * upp);
                y.p[p] = ar.alloc<float>(hw * upp); s.p[p] = ar.alloc<float>(hw * upp);
                part[p] = ar.alloc<double>(size_t(kInChunks) * upp * 2);
            }
            if (pass == 0) continue;
            for (int p = 0; p < 3; ++p) {
                const int hw = g.h[p] * g.w[p];
                hipLaunchKernelGGL(k_slice_pad, dim3((unsigned)((size_t(hw) * 32 + 255) / 256)), dim3(256), 0, st, in[p], x.p[p], hw, c0, N.cin_ch);
            }
            S3D_HIP(hipGetLastError());
            auto conv = [&](ConvKind kind, const ConvW& cw, const Tri& src, const Tri* res, float* const dst[3]) {
                ConvArgs ca; memset(&ca, 0, sizeof ca);
                ca.B = 1; ca.cin = cw.cin; ca.cout = cw.cout; ca.njobs = 3;
                for (int p = 0; p < 3; ++p) {
                    ConvJob& J = ca.job[p];
                    J.in = src.p[p]; J.wgt = d->dev(cw.dense[p]); J.bias = d->dev(cw.bias[p]);
                    J.res = res ? res->p[p] : nullptr; J.out = dst[p]; J.h = g.h[p]; J.w = g.w[p];
                }
                return launch_conv(kind, ca, st);
            };
            S3D_TRY(conv(CONV_5x5, N.cin, x, nullptr, a.p));                         // in_layers: conv5x5 (no norm/act on the input)
            for (int p = 0; p < 3; ++p) {                                            // norm_{p} -> SiLU
                const int hw = g.h[p] * g.w[p], cq = upp / 4, pl = std::max(1, 256 / cq);
                hipLaunchKernelGGL(k_chan_partials, dim3(kInChunks), dim3(cq * pl), size_t(pl) * upp * 2 * sizeof(double), st, a.p[p], part[p], hw, upp);
                hipLaunchKernelGGL(k_inorm_silu, dim3(std::min(1024, (hw * cq + 255) / 256)), dim3(256), size_t(2) * upp * sizeof(float), st,
                                   a.p[p], part[p], d->dev(N.gamma[p]), d->dev(N.beta[p]), y.p[p], hw, upp, 1e-6f);
            }
            S3D_HIP(hipGetLastError());
            S3D_TRY(conv(CONV_1x1, N.sc, x, nullptr, s.p));                          // shortcut(x)
            S3D_TRY(conv(CONV_5x5, N.cout_, y, &s, d->featp[n]));                    // out_layers conv5x5 + shortcut
        }
    }
    d->prepared = true;
    return 0;
}

int s3d_decoder_decode_points(s3d_decoder* d, const float* pts, int64_t N, const float aabb[6], int clamp_color, float* out,
                              void* stream) {
    S3D_CHECK(d && aabb && N >= 0 && (N == 0 || (pts && out)), S3D_ERR_INVALID, "decode_points: bad argument");
    if (N == 0) return 0;                                   // decoding no points is a no-op (empty tensors have null pointers)
    return run_decode(d, pts, N, aabb, nullptr, clamp_color, out, static_cast<hipStream_t>(stream));
}

int s3d_decoder_grid_dims(const float aabb[6], int reso, int dims[3]) {
    S3D_CHECK(aabb && dims && reso >= 1, S3D_ERR_INVALID, "grid_dims: bad argument");
    // resolutions = (resolution * aabb_size / aabb_size.max()).long()  in fp32 (src/encoding/utils3d.py:17-19)
    float size[3], mx = 0.f;
    for (int k = 0; k < 3; ++k) { size[k] = aabb[3 + k] - aabb[k]; mx = std::max(mx, size[k]); }
    S3D_CHECK(mx > 0.f, S3D_ERR_INVALID, "grid_dims: empty aabb");
    for (int k = 0; k < 3; ++k) dims[k] = int(float(reso) * size[k] / mx);
    return 0;
}

int s3d_decoder_decode_grid(s3d_decoder* d, int reso, const float aabb[6], float* out, void* stream) {
    S3D_CHECK(d && aabb && out, S3D_ERR_INVALID, "decode_grid: bad argument");
    int dims[3];
    S3D_TRY(s3d_decoder_grid_dims(aabb, reso, dims));
    const long long N = (long long)dims[0] * dims[1] * dims[2];
    return run_decode(d, nullptr, N, aabb, dims, /*clamp=*/1, out, static_cast<hipStream_t>(stream));
}

}  // extern "C"
