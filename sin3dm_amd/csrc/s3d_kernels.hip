// s3d_kernels.hip — the memory-bound kernels of the triplane UNet step (everything except the
// MFMA convolutions): layout changes, GroupNorm statistics, GN-apply+FiLM+SiLU with rollout means,
// resampling, the output head, the timestep MLP and the sampler update.
//
// Layout: every activation plane is NHWC fp32 [B][h][w][C]; a thread always owns one float4 of
// channels so that a wave reads/writes whole 128-byte (or longer) contiguous runs per pixel.
// All reductions are two-stage with a fixed summation order (no float atomics): results are
// bit-repeatable run to run.
#include "s3d_common.h"
#include "s3d_rank1.h"
#include "s3d_sampler.h"

namespace s3d {

// x * sigmoid(x) on the hardware exp/rcp units (v_exp_f32 / v_rcp_f32, ~1-2 ulp each): the activation kernels
// are otherwise VALU-bound on the IEEE expf + division sequence.  Limits: exp(-v) -> inf gives v * 0, -> 0 gives v.
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ layout repacks (tests only)
__global__ void k_nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out, int B, int C, int hw) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t n = size_t(B) * C * hw;
    if (i >= n) return;
    int c = int(i % C);
    size_t r = i / C;
    int pix = int(r % hw);
    int b = int(r / hw);
    out[i] = in[(size_t(b) * C + c) * hw + pix];
}
__global__ void k_nhwc_to_nchw(const float* __restrict__ in, float* __restrict__ out, int B, int C, int hw) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t n = size_t(B) * C * hw;
    if (i >= n) return;
    int pix = int(i % hw);
    size_t r = i / hw;
    int c = int(r % C);
    int b = int(r / C);
    out[i] = in[(size_t(b) * hw + pix) * C + c];
}
int launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int h, int w, hipStream_t st) {
    size_t n = size_t(B) * C * h * w;
    if (!n) return 0;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, B, C, h * w);
    S3D_HIP(hipGetLastError());
    return 0;
}
int launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int h, int w, hipStream_t st) {
    size_t n = size_t(B) * C * h * w;
    if (!n) return 0;
    hipLaunchKernelGGL(k_nhwc_to_nchw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, B, C, h * w);
    S3D_HIP(hipGetLastError());
    return 0;
}

// Block tail shared by the kernels that emit GroupNorm partials for a pixel chunk: thread (q, l) holds {sum, sumsq} of
// channel quad q over its pixels; the block adds them per group in a fixed order and writes group g at out[g * gstride][2].
__device__ __forceinline__ void gn_chunk_reduce(double* sm, const double s[4], const double ss[4], int q, int l, int C, int pl,
                                                double* out, int gstride) {
    for (int k = 0; k < 4; ++k) {
        sm[(size_t(l) * C + 4 * q + k) * 2 + 0] = s[k];
        sm[(size_t(l) * C + 4 * q + k) * 2 + 1] = ss[k];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int g = threadIdx.x, cg = C / 32;
        double S = 0, SS = 0;
        for (int ll = 0; ll < pl; ++ll)
            for (int c = g * cg; c < (g + 1) * cg; ++c) {
                S += sm[(size_t(ll) * C + c) * 2 + 0];
                SS += sm[(size_t(ll) * C + c) * 2 + 1];
            }
        out[size_t(g) * gstride * 2] = S; out[size_t(g) * gstride * 2 + 1] = SS;
    }
}
__device__ __forceinline__ void gn_acc(double s[4], double ss[4], const float4& v) {
    s[0] += v.x; ss[0] += double(v.x) * v.x;
    s[1] += v.y; ss[1] += double(v.y) * v.y;
    s[2] += v.z; ss[2] += double(v.z) * v.z;
    s[3] += v.w; ss[3] += double(v.w) * v.w;
}
static void thread_shape(int C, int& cq, int& pl) {
    cq = C / 4;
    pl = cq >= 256 ? 1 : 256 / cq;
}

// ------------------------------------------------------------------ in_conv
// decompose_featmaps (src/utils/triplane_util.py:20-25) + TriplaneConv(in, ch, 1x1, no rollout)
// (src/diffusion/unet_triplane.py:378, 482).  One thread = one pixel x 4 output channels.
struct InConvArgs {
    const float* x; const float* wT; const float* bias;
    float* out[3];
    int B, Cin, Cout, H, W, D;
    int h[3], w[3];
    long long pix_begin[4];   // prefix of per-plane pixel counts (per sample)
};
__global__ void k_in_conv(InConvArgs a) {
    const int cq = a.Cout >> 2;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_b = a.pix_begin[3] * cq;
    if (i >= per_b * a.B) return;
    const int b = int(i / per_b);
    long long r = i % per_b;
    const int q = int(r % cq);
    long long pix = r / cq;
    int p = pix >= a.pix_begin[2] ? 2 : (pix >= a.pix_begin[1] ? 1 : 0);
    pix -= a.pix_begin[p];
    const int y = int(pix / a.w[p]), xx = int(pix % a.w[p]);
    const int Hc = a.H + a.D, Wc = a.W + a.D;
    int sy, sx;
    if (p == 0) { sy = y; sx = xx; }
    else if (p == 1) { sy = y; sx = a.W + xx; }
    else { sy = a.H + xx; sx = y; }                 // yz[w][d] = composed[H + d][w]
    const float* src = a.x + (size_t(b) * a.Cin * Hc + sy) * Wc + sx;
    const float4* wq = reinterpret_cast<const float4*>(a.wT + size_t(p) * a.Cin * a.Cout) + q;
    float4 acc = reinterpret_cast<const float4*>(a.bias + size_t(p) * a.Cout)[q];
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float v = src[size_t(ci) * Hc * Wc];
        const float4 wv = wq[size_t(ci) * cq];
        acc.x = fmaf(v, wv.x, acc.x); acc.y = fmaf(v, wv.y, acc.y);
        acc.z = fmaf(v, wv.z, acc.z); acc.w = fmaf(v, wv.w, acc.w);
    }
    reinterpret_cast<float4*>(a.out[p] + ((size_t(b) * a.h[p] + y) * a.w[p] + xx) * a.Cout)[q] = acc;
}
// LDS-staged form for Cout = 64 / 128 / 256 and Cin <= 16: a block owns PX pixels that are
// consecutive in the composed input map (a row segment of xy / xz, a column segment of yz), stages their Cin x PX input
// values with coalesced loads and lets every thread (quad, pixel lane) form PX/8 outputs from LDS broadcasts.
template <int PX, int CQ>                                  // CQ = channel quads per pixel (Cout / 4): 16, 32 or 64
__global__ __launch_bounds__(256) void k_in_conv_lds(InConvArgs a, int segs0, int segs1, int segs2, double* part, int maxparts) {
    constexpr int LANES = 256 / CQ, C = 4 * CQ;             // pixel lanes of the block
    __shared__ __attribute__((aligned(16))) float sx[PX][16];
    __shared__ float sred[2][LANES][C];                      // GroupNorm partials of the block: [sum | sumsq][pixel lane][channel]
    const int b = blockIdx.y;
    int blk = blockIdx.x, p = 0;
    if (blk >= segs0) { blk -= segs0; p = 1; if (blk >= segs1) { blk -= segs1; p = 2; } }
    const int h = a.h[p], w = a.w[p];
    // run length along the contiguous input direction: w for xy / xz (pixels of a row), h for yz (pixels of a column)
    const int len = p == 2 ? h : w, nseg = (len + PX - 1) / PX;
    const int line = blk / nseg, s0 = (blk % nseg) * PX;
    const int Hc = a.H + a.D, Wc = a.W + a.D;
    const float* xb = a.x + size_t(b) * a.Cin * Hc * Wc;
    // composed coordinates of segment element e: xy (line, s0+e); xz (line, W + s0+e); yz: plane pixel (y = s0+e, xx = line) -> (H + line, s0+e)
    const int sy = p == 2 ? a.H + line : line, sx0 = p == 1 ? a.W + s0 : s0;
    for (int it = threadIdx.x; it < 16 * PX; it += 256) {            // channels Cin..15 are zero (0 x stale LDS could be NaN)
        const int ci = it / PX, e = it - ci * PX;
        sx[e][ci] = ci < a.Cin && s0 + e < len ? xb[(size_t(ci) * Hc + sy) * Wc + sx0 + e] : 0.f;
    }
    const int q = threadIdx.x % CQ, lp = threadIdx.x / CQ;
    const float4* wq = reinterpret_cast<const float4*>(a.wT + size_t(p) * a.Cin * a.Cout) + q;
    float4 wv[16];
#pragma unroll
    for (int ci = 0; ci < 16; ++ci) wv[ci] = ci < a.Cin ? wq[size_t(ci) * CQ] : make_float4(0, 0, 0, 0);
    const float4 bias = reinterpret_cast<const float4*>(a.bias + size_t(p) * a.Cout)[q];
    __syncthreads();
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PX / LANES; ++k) {
        const int e = k * LANES + lp;
        if (s0 + e >= len) continue;
        float4 acc = bias;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const float4 v = *reinterpret_cast<const float4*>(&sx[e][c4 * 4]);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 ww = wv[c4 * 4 + j];
                acc.x = fmaf(vv[j], ww.x, acc.x); acc.y = fmaf(vv[j], ww.y, acc.y);
                acc.z = fmaf(vv[j], ww.z, acc.z); acc.w = fmaf(vv[j], ww.w, acc.w);
            }
        }
        const int y = p == 2 ? s0 + e : line, xx = p == 2 ? line : s0 + e;
        reinterpret_cast<float4*>(a.out[p] + ((size_t(b) * h + y) * w + xx) * a.Cout)[q] = acc;
        gs[0] += acc.x; gs[1] += acc.y; gs[2] += acc.z; gs[3] += acc.w;
        gss[0] = fmaf(acc.x, acc.x, gss[0]); gss[1] = fmaf(acc.y, acc.y, gss[1]); gss[2] = fmaf(acc.z, acc.z, gss[2]); gss[3] = fmaf(acc.w, acc.w, gss[3]);
    }
    if (part) {                                                // one GroupNorm part per block
#pragma unroll
        for (int k = 0; k < 4; ++k) { sred[0][lp][4 * q + k] = gs[k]; sred[1][lp][4 * q + k] = gss[k]; }
        __syncthreads();
        if (threadIdx.x < 32) {
            constexpr int cg = C / 32;
            const int g = threadIdx.x;
            double S = 0, SS = 0;
            for (int l = 0; l < LANES; ++l)
#pragma unroll
                for (int k = 0; k < cg; ++k) { S += sred[0][l][cg * g + k]; SS += sred[1][l][cg * g + k]; }
            double* o = part + (((size_t(b) * 3 + p) * 32 + g) * maxparts + blk) * 2;
            o[0] = S; o[1] = SS;
        }
    }
}
constexpr int kInConvPx = 32;
bool in_conv_gn_parts(const Geo& g, int Cin, int Cout, int nparts[3]) {
    if (!((Cout == 64 || Cout == 128 || Cout == 256) && Cin <= 16)) return false;
    for (int p = 0; p < 3; ++p) { const int len = p == 2 ? g.h[p] : g.w[p], lines = p == 2 ? g.w[p] : g.h[p]; nparts[p] = lines * ((len + kInConvPx - 1) / kInConvPx); }
    return true;
}
int launch_in_conv(const float* x, int B, int Cin, int H, int W, int D, const float* wT, const float* bias,
                   int Cout, Tri& out, hipStream_t st, const GnPartials* part) {
    InConvArgs a;
    a.x = x; a.wT = wT; a.bias = bias; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.D = D;
    a.pix_begin[0] = 0;
    for (int p = 0; p < 3; ++p) {
        a.out[p] = out.p[p]; a.h[p] = out.g.h[p]; a.w[p] = out.g.w[p];
        a.pix_begin[p + 1] = a.pix_begin[p] + (long long)a.h[p] * a.w[p];
    }
    long long n = a.pix_begin[3] * (Cout / 4) * B;
    if (!n) return 0;
    int segs[3];
    if (in_conv_gn_parts(out.g, Cin, Cout, segs)) {                   // the LDS-staged form (22 -> 10 us at 128^3; 16 / 32 / 64 pixels per block measured equal)
        S3D_CHECK(!part || (part->nsub == 32 && part->nparts[0] == segs[0] && part->nparts[1] == segs[1] && part->nparts[2] == segs[2]), S3D_ERR_INVALID, "in_conv: GroupNorm partial layout");
        const dim3 grid(segs[0] + segs[1] + segs[2], B);
        double* pp = part ? part->p : nullptr;
        const int mp = part ? part->maxparts : 0;
        if (Cout == 64) hipLaunchKernelGGL((k_in_conv_lds<kInConvPx, 16>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], pp, mp);
        else if (Cout == 128) hipLaunchKernelGGL((k_in_conv_lds<kInConvPx, 32>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], pp, mp);
        else hipLaunchKernelGGL((k_in_conv_lds<kInConvPx, 64>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], pp, mp);
        S3D_HIP(hipGetLastError());
        return 0;
    }
    S3D_CHECK(!part, S3D_ERR_INVALID, "in_conv: this shape does not emit GroupNorm partials");
    hipLaunchKernelGGL(k_in_conv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ GroupNorm statistics
// GroupNorm32(32, C): per (sample, group) mean and biased variance over (C/32)*h*w (src/diffusion/nn.py:17-19,
// 93-100).  Stage 1 here: each block reduces one pixel chunk of one plane to {sum, sumsq} per group, in double.
// Stage 2 (gn_finalize, in the consumers) adds the kGnChunks partials in index order.
struct GnPartArgs {
    const float* x[3];
    int h[3], w[3];
    int C, cq, pl;
    double* part;   // [B][3][32 groups][kGnChunks][2]
};
__global__ void k_gn_partials(GnPartArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sm = reinterpret_cast<double*>(smem_raw);        // [pl][C][2]
    const int chunk = blockIdx.x, p = blockIdx.y, b = blockIdx.z;
    const int npix = a.h[p] * a.w[p];
    const int per = (npix + kGnChunks - 1) / kGnChunks;
    const int p0 = chunk * per, p1 = min(npix, p0 + per);
    const int q = threadIdx.x % a.cq, l = threadIdx.x / a.cq;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    const float4* src = reinterpret_cast<const float4*>(a.x[p] + size_t(b) * npix * a.C) + q;
#pragma unroll 8
    for (int pix = p0 + l; pix < p1; pix += a.pl) {
        const float4 v = src[size_t(pix) * a.cq];
        s[0] += v.x; ss[0] += double(v.x) * v.x;
        s[1] += v.y; ss[1] += double(v.y) * v.y;
        s[2] += v.z; ss[2] += double(v.z) * v.z;
        s[3] += v.w; ss[3] += double(v.w) * v.w;
    }
    gn_chunk_reduce(sm, s, ss, q, l, a.C, a.pl, a.part + ((size_t(b) * 3 + p) * 32 * kGnChunks + chunk) * 2, kGnChunks);
}
int launch_gn_partials(const Tri& x, int B, GnPartials out, hipStream_t st) {
    GnPartArgs a;
    for (int p = 0; p < 3; ++p) { a.x[p] = x.p[p]; a.h[p] = x.g.h[p]; a.w[p] = x.g.w[p]; }
    a.C = x.C; thread_shape(x.C, a.cq, a.pl); a.part = out.p;
    S3D_CHECK(x.C % 32 == 0 && a.cq <= 1024, S3D_ERR_INVALID, "GroupNorm(32, C): C=%d must be a multiple of 32 and <= 4096", x.C);
    size_t shm = size_t(a.pl) * a.C * 2 * sizeof(double);
    hipLaunchKernelGGL(k_gn_partials, dim3(kGnChunks, 3, B), dim3(a.cq * a.pl), shm, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// Stage 2: {mean, rstd} of every (b, plane, group) from the partials, added in part order, in double.
struct GnFinArgs {
    const double* part; float* mr;
    int maxparts, nparts[3], nsub, subs_per_group;
    double count[3];      // elements per group = (C/32)*h*w
};
// fixed-order reduction of one double per thread over a 256-thread block: butterfly inside each wave (the same pairing for
// every launch), then the four wave sums in wave order: bit-repeatable
__device__ __forceinline__ double block_sum256(double v, double* sm4) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) sm4[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = (sm4[0] + sm4[1]) + (sm4[2] + sm4[3]);
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(256) void k_gn_finalize(GnFinArgs a) {                   // one block per (group, plane, sample): 96 blocks at batch 1, not 3
    __shared__ double sm4[4];
    const int g = blockIdx.x, p = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const double* base = a.part + (size_t(b) * 3 + p) * a.maxparts * a.nsub * 2;
    double S = 0, SS = 0;
    for (int k = 0; k < a.subs_per_group; ++k) {               // [sub][part]: a group's parts are contiguous
        const double2* row = reinterpret_cast<const double2*>(base + (size_t(g) * a.subs_per_group + k) * a.maxparts * 2);
        for (int part = tid; part < a.nparts[p]; part += 256) { const double2 v = row[part]; S += v.x; SS += v.y; }
    }
    S = block_sum256(S, sm4);
    SS = block_sum256(SS, sm4);
    if (tid == 0) {
        const double m = S / a.count[p];
        double var = SS / a.count[p] - m * m;
        if (var < 0) var = 0;
        float* o = a.mr + ((size_t(b) * 3 + p) * 32 + g) * 2;
        o[0] = float(m);
        o[1] = float(1.0 / sqrt(var + 1e-5));
    }
}
int launch_gn_finalize(const GnPartials& part, const Geo& g, int C, int B, GnStats out, hipStream_t st) {
    GnFinArgs a;
    a.part = part.p; a.mr = out.mr; a.maxparts = part.maxparts; a.nsub = part.nsub;
    S3D_CHECK(part.nsub % 32 == 0, S3D_ERR_INVALID, "gn_finalize: nsub=%d", part.nsub);
    a.subs_per_group = part.nsub / 32;
    for (int p = 0; p < 3; ++p) { a.nparts[p] = part.nparts[p]; a.count[p] = double(C / 32) * g.h[p] * g.w[p]; }
    if (!B) return 0;
    hipLaunchKernelGGL(k_gn_finalize, dim3(32, 3, B), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// exact-2x bilinear sample (align_corners=False) of a low-resolution NHWC plane at output pixel (yo, xo), channel quad q:
// the arithmetic of k_upcat / F.interpolate (src = 0.5 * (dst + 0.5) - 0.5 clamped at 0, neighbour clamped to the edge)
__device__ __forceinline__ float4 up2x_sample(const float4* __restrict__ src, int hi, int wi, int cuq, int yo, int xo) {
    float fy = 0.5f * (float(yo) + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
    float fx = 0.5f * (float(xo) + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
    int y0 = int(fy); y0 = y0 > hi - 1 ? hi - 1 : y0;
    int x0 = int(fx); x0 = x0 > wi - 1 ? wi - 1 : x0;
    const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
    const float ly1 = fy - float(y0), ly0 = 1.f - ly1, lx1 = fx - float(x0), lx0 = 1.f - lx1;
    const float4 v00 = src[(size_t(y0) * wi + x0) * cuq], v01 = src[(size_t(y0) * wi + x1) * cuq];
    const float4 v10 = src[(size_t(y1) * wi + x0) * cuq], v11 = src[(size_t(y1) * wi + x1) * cuq];
    float4 o;
    o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
    o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
    o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
    o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    return o;
}

// Eight consecutive output rows i0..i0+7 (i0 a multiple of 8) of output column j: the six low-resolution rows they touch
// are interpolated horizontally once (12 loads instead of 32) and combined with the per-row weights of the exact formula;
// the values equal up2x_sample's bit for bit (same products, same order).
struct Up8Rows { int rk[6]; float ly0[8], ly1[8]; };
__device__ __forceinline__ Up8Rows up8_rows(int i0, int hi) {
    Up8Rows R;
    const int base = (i0 >> 1) - 1;
#pragma unroll
    for (int k = 0; k < 6; ++k) { const int r = base + k; R.rk[k] = r < 0 ? 0 : (r > hi - 1 ? hi - 1 : r); }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float fy = 0.5f * (float(i0 + r) + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
        int y0 = int(fy); y0 = y0 > hi - 1 ? hi - 1 : y0;
        R.ly1[r] = fy - float(y0); R.ly0[r] = 1.f - R.ly1[r];
    }
    return R;
}
__device__ __forceinline__ void up8_column(const float4* __restrict__ src, int wi, int cuq, const Up8Rows& R, int j, float4 out[8]) {
    float fx = 0.5f * (float(j) + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
    int x0 = int(fx); x0 = x0 > wi - 1 ? wi - 1 : x0;
    const int x1 = x0 + (x0 < wi - 1 ? 1 : 0);
    const float lx1 = fx - float(x0), lx0 = 1.f - lx1;
    float4 hl[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float4 a = src[(size_t(R.rk[k]) * wi + x0) * cuq], b = src[(size_t(R.rk[k]) * wi + x1) * cuq];
        hl[k].x = lx0 * a.x + lx1 * b.x; hl[k].y = lx0 * a.y + lx1 * b.y; hl[k].z = lx0 * a.z + lx1 * b.z; hl[k].w = lx0 * a.w + lx1 * b.w;
    }
    // output row r uses low-resolution rows (base + ka, base + ka + 1): ka = (r + 1) >> 1
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float4 p = hl[(r + 1) >> 1], q = hl[((r + 1) >> 1) + 1];
        out[r].x = R.ly0[r] * p.x + R.ly1[r] * q.x; out[r].y = R.ly0[r] * p.y + R.ly1[r] * q.y;
        out[r].z = R.ly0[r] * p.z + R.ly1[r] * q.z; out[r].w = R.ly0[r] * p.w + R.ly1[r] * q.w;
    }
}

// GroupNorm partials of bilinear2x(u) per subgroup of sg channels; one part per 8x8 tile of the OUTPUT.
struct GnPartUpArgs {
    const float* u[3];
    int hi[3], wi[3];
    int C, cq, pl, sg, nsub, maxparts;
    double* part;   // [B][3][nsub][maxparts][2]
};
__global__ void k_gn_partials_up(GnPartUpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sm = reinterpret_cast<double*>(smem_raw);        // [pl][C][2]
    const int p = blockIdx.y, b = blockIdx.z;
    const int hi = a.hi[p], wi = a.wi[p], h = 2 * hi, w = 2 * wi;
    const int ntc = (w + kActCols - 1) / kActCols, ntr = (h + kActRows - 1) / kActRows;
    if (int(blockIdx.x) >= ntc * ntr) return;
    const int tr = blockIdx.x / ntc, tc = blockIdx.x % ntc;
    const int i0 = tr * kActRows, j0 = tc * kActCols, i1 = min(h, i0 + kActRows), j1 = min(w, j0 + kActCols);
    const int q = threadIdx.x % a.cq, l = threadIdx.x / a.cq;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    const float4* src = reinterpret_cast<const float4*>(a.u[p] + size_t(b) * hi * wi * a.C) + q;
    const Up8Rows R = up8_rows(i0, hi);
    for (int j = j0 + l; j < j1; j += a.pl) {
        float4 v[8];
        up8_column(src, wi, a.cq, R, j, v);
#pragma unroll
        for (int r = 0; r < 8; ++r) if (i0 + r < i1) gn_acc(s, ss, v[r]);
    }
    for (int k = 0; k < 4; ++k) {
        sm[(size_t(l) * a.C + 4 * q + k) * 2 + 0] = s[k];
        sm[(size_t(l) * a.C + 4 * q + k) * 2 + 1] = ss[k];
    }
    __syncthreads();
    for (int sub = threadIdx.x; sub < a.nsub; sub += blockDim.x) {
        double S = 0, SS = 0;
        for (int ll = 0; ll < a.pl; ++ll)
            for (int c = sub * a.sg; c < (sub + 1) * a.sg; ++c) { S += sm[(size_t(ll) * a.C + c) * 2]; SS += sm[(size_t(ll) * a.C + c) * 2 + 1]; }
        double* o = a.part + (((size_t(b) * 3 + p) * a.nsub + sub) * a.maxparts + blockIdx.x) * 2;
        o[0] = S; o[1] = SS;
    }
}
void gn_up_parts(const Geo& out_g, int nparts[3]) {
    for (int p = 0; p < 3; ++p) nparts[p] = cdiv(out_g.h[p], kActRows) * cdiv(out_g.w[p], kActCols);
}
int launch_gn_partials_up(const Tri& u, int B, int sg, GnPartials out, hipStream_t st) {
    GnPartUpArgs a;
    int maxtiles = 0;
    for (int p = 0; p < 3; ++p) {
        a.u[p] = u.p[p]; a.hi[p] = u.g.h[p]; a.wi[p] = u.g.w[p];
        maxtiles = std::max(maxtiles, cdiv(2 * u.g.h[p], kActRows) * cdiv(2 * u.g.w[p], kActCols));
    }
    a.C = u.C; thread_shape(u.C, a.cq, a.pl); a.sg = sg; a.nsub = u.C / sg; a.part = out.p; a.maxparts = out.maxparts;
    S3D_CHECK(u.C % sg == 0 && a.cq <= 1024 && out.nsub == a.nsub && out.maxparts >= maxtiles, S3D_ERR_INVALID, "gn_partials_up: layout");
    if (!maxtiles || !B) return 0;
    hipLaunchKernelGGL(k_gn_partials_up, dim3(maxtiles, 3, B), dim3(a.cq * a.pl), size_t(a.pl) * a.C * 2 * sizeof(double), st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}
// {mean, rstd} per (b, plane, group) of the concat [up | skip]: the group's subgroups come first from the up partials,
// then from the skip partials; parts added in index order, in double.
struct GnFinCatArgs {
    const double* pu; const double* ps; float* mr;
    int maxparts_u, maxparts_s, nparts_u[3], nparts_s[3], nsub_u, nsub_s, subs_per_group;
    double count[3];
};
__global__ __launch_bounds__(256) void k_gn_finalize_cat(GnFinCatArgs a) {
    __shared__ double sm4[4];
    const int g = blockIdx.x, p = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    double S = 0, SS = 0;
    for (int k = 0; k < a.subs_per_group; ++k) {
        const int sub = g * a.subs_per_group + k;
        const bool up = sub < a.nsub_u;
        const double2* row = reinterpret_cast<const double2*>(
            up ? a.pu + (((size_t(b) * 3 + p) * a.nsub_u + sub) * a.maxparts_u) * 2
               : a.ps + (((size_t(b) * 3 + p) * a.nsub_s + (sub - a.nsub_u)) * a.maxparts_s) * 2);
        const int n = up ? a.nparts_u[p] : a.nparts_s[p];
        for (int part = tid; part < n; part += 256) { const double2 v = row[part]; S += v.x; SS += v.y; }
    }
    S = block_sum256(S, sm4);
    SS = block_sum256(SS, sm4);
    if (tid == 0) {
        const double m = S / a.count[p];
        double var = SS / a.count[p] - m * m;
        if (var < 0) var = 0;
        float* o = a.mr + ((size_t(b) * 3 + p) * 32 + g) * 2;
        o[0] = float(m);
        o[1] = float(1.0 / sqrt(var + 1e-5));
    }
}
int launch_gn_finalize_cat(const GnPartials& pu, const GnPartials& ps, const Geo& g, int C, int B, GnStats out, hipStream_t st) {
    GnFinCatArgs a;
    a.pu = pu.p; a.ps = ps.p; a.mr = out.mr; a.maxparts_u = pu.maxparts; a.maxparts_s = ps.maxparts; a.nsub_u = pu.nsub; a.nsub_s = ps.nsub;
    S3D_CHECK((pu.nsub + ps.nsub) % 32 == 0, S3D_ERR_INVALID, "gn_finalize_cat: nsub=%d+%d", pu.nsub, ps.nsub);
    a.subs_per_group = (pu.nsub + ps.nsub) / 32;
    for (int p = 0; p < 3; ++p) { a.nparts_u[p] = pu.nparts[p]; a.nparts_s[p] = ps.nparts[p]; a.count[p] = double(C / 32) * g.h[p] * g.w[p]; }
    if (!B) return 0;
    hipLaunchKernelGGL(k_gn_finalize_cat, dim3(32, 3, B), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ GN-apply (+FiLM) + SiLU (+ rollout partial means)
// TriplaneNorm + TriplaneSiLU (src/diffusion/unet_triplane.py:63-95), with the FiLM modulation
// h*(1+scale)+shift of TriplaneResBlock._forward (:285-297) applied between them, and the axis sums the
// next TriplaneConv's rollout needs (:37-46) accumulated on the way out.
// y = x*(rstd*gamma) + (beta - mean*rstd*gamma) is the form ATen's CPU group_norm kernel evaluates.
// GroupNorm statistics from a producer's partial sums, added by the CONSUMING block itself (k_gn_act, k_out_head_px): L =
// blockDim / 32 lanes per group, lane j adds entries j, j + L, ... of the group's [sub][part] list (16 loads in flight per
// trip, added in index order), the lanes meet in lane order (double) — the formula of k_gn_finalize.  sd: 2 * blockDim doubles
// of LDS; returns the block's {mean[32], rstd[32]} as floats right behind them (valid after the caller's next barrier).
struct GnPartSrc { const double* part; int nparts[3], maxparts, nsub, spg; double count[3]; };
__device__ __forceinline__ float* gn_stats_from_parts(const GnPartSrc& a, int b, int p, double* sd, float* mr_out) {
    const int L = int(blockDim.x) >> 5, tid = threadIdx.x;
    if (tid < 32 * L) {
        const int g = tid / L, j = tid - g * L;
        const int n = a.nparts[p], tot = a.spg * n;
        const double2* base = reinterpret_cast<const double2*>(a.part) + ((size_t(b) * 3 + p) * a.nsub + size_t(g) * a.spg) * a.maxparts;
        double S = 0, SS = 0;
        constexpr int NB = 16;                          // loads in flight per trip (added in index order afterwards)
        for (int k0 = j; k0 < tot; k0 += NB * L) {
            double2 v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int k = k0 + u * L, kc = k < tot ? k : j;
                const int sub = kc / n, part = kc - sub * n;
                v[u] = base[size_t(sub) * a.maxparts + part];
                if (k >= tot) v[u] = make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) { S += v[u].x; SS += v[u].y; }
        }
        sd[tid * 2] = S; sd[tid * 2 + 1] = SS;
    }
    __syncthreads();
    float* stf = reinterpret_cast<float*>(sd + 2 * blockDim.x);     // behind the lanes' sums: no second barrier
    if (tid < 32) {
        double S = 0, SS = 0;
        for (int j = 0; j < L; ++j) { S += sd[(tid * L + j) * 2]; SS += sd[(tid * L + j) * 2 + 1]; }
        const double m = S / a.count[p];
        double var = SS / a.count[p] - m * m;
        if (var < 0) var = 0;
        stf[tid] = float(m); stf[32 + tid] = float(1.0 / sqrt(var + 1e-5));
        if (mr_out) {
            float* o = mr_out + ((size_t(b) * 3 + p) * 32 + tid) * 2;
            o[0] = stf[tid]; o[1] = stf[32 + tid];
        }
    }
    return stf;
}

static GnPartSrc gn_part_src(const GnPartials& part, const Geo& g, int C) {
    GnPartSrc s;
    s.part = part.p; s.maxparts = part.maxparts; s.nsub = part.nsub; s.spg = part.nsub / 32;
    for (int p = 0; p < 3; ++p) { s.nparts[p] = part.nparts[p]; s.count[p] = double(C / 32) * g.h[p] * g.w[p]; }
    return s;
}

// The same sums as a consumer that adds the partials itself, as a launch of their own: one block of the CONSUMER's size per
// (plane, sample) runs gn_stats_from_parts — lane count per group, trip order and meeting order are the consumer's, so the
// statistics are bit-identical to the in-consumer form.  It pays when the consumer has many rounds of blocks (batch >= 2, the
// (256,256,128) planes): every one of them repeats the ~3-us addition as its prologue (config 3: 5.42 -> 5.35 ms/step).
__global__ void k_gn_finalize_as(GnPartSrc ps, float* mr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    gn_stats_from_parts(ps, blockIdx.y, blockIdx.x, reinterpret_cast<double*>(smem_raw), mr);
}
int launch_gn_finalize_as(const GnPartials& part, const Geo& g, int C, int B, int threads, GnStats out, hipStream_t st) {
    S3D_CHECK(gn_act_can_add_parts(part, C) && threads >= 32 && threads % 32 == 0 && out.mr, S3D_ERR_INVALID, "gn_finalize_as: layout");
    if (!B) return 0;
    const GnPartSrc ps = gn_part_src(part, g, C);
    hipLaunchKernelGGL(k_gn_finalize_as, dim3(3, B), dim3(threads), size_t(threads) * 2 * sizeof(double) + 64 * sizeof(float), st, ps, out.mr);
    S3D_HIP(hipGetLastError());
    return 0;
}

struct GnActArgs {
    const float* x[3]; float* y[3];
    const float* gamma[3]; const float* beta[3];
    float* rowpart[3]; float* colpart[3];
    const float* mr;
    const float* film; int film_stride;
    int h[3], w[3];
    int C, cq, pl, with_means;
    // mr == null && part != null: the statistics come from the producer's partial sums, added here (every block adds the
    // parts of its plane's 32 groups itself: 64 KB from L2 at 128^2 x 128 channels, instead of a k_gn_finalize launch)
    GnPartSrc ps;
    float* mr_out;            // ... and block 0 of each (plane, sample) also leaves {mean, rstd} [B][3][32][2] (the training tape)
};
__global__ __launch_bounds__(256) void k_gn_act(GnActArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);           // [64] stats, then [pl][kActRows][C] row sums
    const int p = blockIdx.y, b = blockIdx.z;
    const int h = a.h[p], w = a.w[p];
    const int ntc = (w + kActCols - 1) / kActCols, ntr = (h + kActRows - 1) / kActRows;
    if (int(blockIdx.x) >= ntc * ntr) return;
    const int tr = blockIdx.x / ntc, tc = blockIdx.x % ntc;
    const bool ident = a.mr == nullptr && a.ps.part == nullptr;
    const int q = threadIdx.x % a.cq, l = threadIdx.x / a.cq;
    const int i0 = tr * kActRows, j0 = tc * kActCols;
    const int i1 = min(h, i0 + kActRows), j1 = min(w, j0 + kActCols);
    const float4* xs = reinterpret_cast<const float4*>(a.x[p] + size_t(b) * h * w * a.C) + q;
    float4* ys = reinterpret_cast<float4*>(a.y[p] + size_t(b) * h * w * a.C) + q;
    // the thread's first column of pixels and its channel quad's parameters are requested NOW: their round trip (the tensor
    // was just written by another kernel: a miss all the way to memory) runs beside the statistics' (the same kind of miss)
    // instead of behind them.  (Round 2 measured this slower — under the 128-VGPR budget the kernel then had, see __launch_bounds__.)
    float4 xfirst[kActRows];
    {
        const int j = min(j0 + l, j1 - 1);
#pragma unroll
        for (int r = 0; r < kActRows; ++r) xfirst[r] = xs[(size_t(min(i0 + r, i1 - 1)) * w + j) * a.cq];
    }
    const bool film = a.film != nullptr;
    float4 gam4 = make_float4(1, 1, 1, 1), bet4 = make_float4(0, 0, 0, 0), fsc4 = make_float4(0, 0, 0, 0), fsh4 = make_float4(0, 0, 0, 0);
    if (!ident) { gam4 = reinterpret_cast<const float4*>(a.gamma[p])[q]; bet4 = reinterpret_cast<const float4*>(a.beta[p])[q]; }
    if (film) {
        fsc4 = reinterpret_cast<const float4*>(a.film + size_t(b) * a.film_stride)[q];
        fsh4 = reinterpret_cast<const float4*>(a.film + size_t(b) * a.film_stride + a.C)[q];
    }
    if (a.mr) {
        if (threadIdx.x < 32) {
            const float* mr = a.mr + ((size_t(b) * 3 + p) * 32 + threadIdx.x) * 2;
            sm[threadIdx.x] = mr[0]; sm[32 + threadIdx.x] = mr[1];
        }
    } else if (a.ps.part) {
        sm = gn_stats_from_parts(a.ps, b, p, reinterpret_cast<double*>(smem_raw), a.mr_out && blockIdx.x == 0 ? a.mr_out : nullptr);
    }
    __syncthreads();
    const int cg = a.C / 32;
    float A[4], Bc[4], sc[4], sh[4];
    {
        const float gv[4] = {gam4.x, gam4.y, gam4.z, gam4.w}, bv[4] = {bet4.x, bet4.y, bet4.z, bet4.w};
        const float s1[4] = {fsc4.x, fsc4.y, fsc4.z, fsc4.w}, s2[4] = {fsh4.x, fsh4.y, fsh4.z, fsh4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * q + k, g = c / cg;
            const float scale = ident ? 1.0f : sm[32 + g] * gv[k];
            A[k] = scale;
            Bc[k] = ident ? 0.0f : bv[k] - scale * sm[g];
            sc[k] = film ? 1.0f + s1[k] : 1.0f;
            sh[k] = film ? s2[k] : 0.0f;
        }
    }
    __syncthreads();                                            // stats region of sm is reused below
    float4 rowacc[kActRows];
#pragma unroll
    for (int r = 0; r < kActRows; ++r) rowacc[r] = make_float4(0, 0, 0, 0);
    for (int j = j0 + l; j < j1; j += a.pl) {
        float4 colacc = make_float4(0, 0, 0, 0);
        float4 xin[kActRows];
        if (j == j0 + l) {
#pragma unroll
            for (int r = 0; r < kActRows; ++r) xin[r] = xfirst[r];
        } else {
#pragma unroll
            for (int r = 0; r < kActRows; ++r) xin[r] = xs[(size_t(min(i0 + r, i1 - 1)) * w + j) * a.cq];   // all loads first
        }
#pragma unroll
        for (int r = 0; r < kActRows; ++r) {
            const int i = i0 + r;
            if (i < i1) {
                const size_t off = (size_t(i) * w + j) * a.cq;
                const float4 v = xin[r];
                float4 o;
                o.x = fmaf(v.x, A[0], Bc[0]); o.y = fmaf(v.y, A[1], Bc[1]);
                o.z = fmaf(v.z, A[2], Bc[2]); o.w = fmaf(v.w, A[3], Bc[3]);
                if (film) {
                    o.x = o.x * sc[0] + sh[0]; o.y = o.y * sc[1] + sh[1];
                    o.z = o.z * sc[2] + sh[2]; o.w = o.w * sc[3] + sh[3];
                }
                if (!ident) { o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w); }
                ys[off] = o;
                colacc.x += o.x; colacc.y += o.y; colacc.z += o.z; colacc.w += o.w;
                rowacc[r].x += o.x; rowacc[r].y += o.y; rowacc[r].z += o.z; rowacc[r].w += o.w;
            }
        }
        if (a.with_means)
            reinterpret_cast<float4*>(a.colpart[p] + ((size_t(b) * ntr + tr) * w + j) * a.C)[q] = colacc;
    }
    if (!a.with_means) return;
    float4* smr = reinterpret_cast<float4*>(smem_raw);         // [pl][kActRows][cq] (from the start of the block's LDS)
#pragma unroll
    for (int r = 0; r < kActRows; ++r) smr[(size_t(l) * kActRows + r) * a.cq + q] = rowacc[r];
    __syncthreads();
    const int nthr = a.cq * a.pl;
    for (int it = threadIdx.x; it < kActRows * a.cq; it += nthr) {
        const int r = it / a.cq, qq = it % a.cq;
        const int i = i0 + r;
        if (i >= i1) continue;
        float4 s = make_float4(0, 0, 0, 0);
        for (int ll = 0; ll < a.pl; ++ll) {
            const float4 v = smr[(size_t(ll) * kActRows + r) * a.cq + qq];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(a.rowpart[p] + ((size_t(b) * ntc + tc) * h + i) * a.C)[qq] = s;
    }
}
bool gn_act_can_add_parts(const GnPartials& part, int C) {
    if (!part.p || part.nsub % 32 != 0 || C % 32 != 0) return false;
    return (part.nsub / 32) * part.maxparts <= 160;      // entries per group a block adds (x 32 groups x 16 bytes; the 256 of an avgpool producer: +4.1 us in the act kernel for a 4.6-us launch, not taken)
}
// Who adds them: the consumer's own blocks (S3D_GN_FUSED=1: always) when the launch is at most two rounds of blocks — one
// dependent launch less on a latency-bound step —, k_gn_finalize_as ahead of it (S3D_GN_FUSED=0: always) otherwise.  Same bits.
bool gn_parts_in_consumer(long long consumer_blocks) {
    const int mode = opt(OPT_GN_FUSED);
    const int cus = device_cus();
    return mode < 0 ? consumer_blocks <= 6LL * cus : mode != 0;          // two rounds of three blocks per CU (1536 on an MI355X)
}
int gn_act_threads(int C) { int cq, pl; thread_shape(C, cq, pl); return cq * std::min(pl, kActCols); }
long long gn_act_blocks(const Geo& g, int B) {
    int maxtiles = 0;
    for (int p = 0; p < 3; ++p) maxtiles = std::max(maxtiles, cdiv(g.h[p], kActRows) * cdiv(g.w[p], kActCols));
    return (long long)maxtiles * 3 * B;
}
int launch_gn_act(const Tri& x, int B, GnStats stats, const ActArgs& aa, Tri& y, const MeanPartials* mp,
                  hipStream_t st, const GnPartials* stats_part) {
    GnActArgs a;
    a.ps.part = nullptr; a.mr_out = nullptr;
    if (stats_part) {                                     // stats.mr (optional) then receives the statistics instead of providing them
        S3D_CHECK(gn_act_can_add_parts(*stats_part, x.C), S3D_ERR_INVALID, "gn_act: partial-sum statistics layout");
        a.mr_out = stats.mr; stats.mr = nullptr;
        a.ps = gn_part_src(*stats_part, x.g, x.C);
    }
    int maxtiles = 0;
    for (int p = 0; p < 3; ++p) {
        a.x[p] = x.p[p]; a.y[p] = y.p[p]; a.h[p] = x.g.h[p]; a.w[p] = x.g.w[p];
        a.gamma[p] = aa.gamma[p]; a.beta[p] = aa.beta[p];
        a.rowpart[p] = mp ? mp->rowpart[p] : nullptr;
        a.colpart[p] = mp ? mp->colpart[p] : nullptr;
        maxtiles = std::max(maxtiles, cdiv(a.h[p], kActRows) * cdiv(a.w[p], kActCols));
    }
    a.mr = stats.mr; a.film = aa.film; a.film_stride = aa.film_stride;
    a.C = x.C; thread_shape(x.C, a.cq, a.pl); a.with_means = mp ? 1 : 0;
    a.pl = std::min(a.pl, kActCols);          // a pixel lane per tile column at most (64 channels: 16 lanes left half the block idle)
    S3D_CHECK(x.C % 32 == 0 && a.cq * a.pl <= 256, S3D_ERR_INVALID, "GroupNorm(32, C): C=%d unsupported", x.C);
    if (!maxtiles || !B) return 0;
    size_t shm = std::max(size_t(64) * sizeof(float), size_t(a.pl) * kActRows * a.C * sizeof(float));
    if (a.ps.part) shm = std::max(shm, size_t(a.cq) * a.pl * 2 * sizeof(double) + 64 * sizeof(float));
    hipLaunchKernelGGL(k_gn_act, dim3(maxtiles, 3, B), dim3(a.cq * a.pl), shm, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// k_gn_act on the virtual tensor [bilinear2x(u) | sk]: the concat of TriplaneUNetModelSmall.forward (:494-503) is never
// written; its upsampled half is sampled from the low-resolution tensor on the way in.
struct GnActCatArgs {
    GnActArgs base;                  // x unused; C / cq / pl describe the concat
    const float* u[3]; const float* sk[3];
    int cuq, csq;
};
// (__launch_bounds__: without one the compiler budgets for 1024-thread blocks — 128 VGPRs — and this kernel spilled 84 bytes per
// thread into scratch inside its column loop: 38.5 us for 57 MB at 128^3.  Its blocks are 192-512 threads.)
__global__ __launch_bounds__(512) void k_gn_act_cat(GnActCatArgs ca) {
    const GnActArgs& a = ca.base;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    const int p = blockIdx.y, b = blockIdx.z;
    const int h = a.h[p], w = a.w[p];
    const int ntc = (w + kActCols - 1) / kActCols, ntr = (h + kActRows - 1) / kActRows;
    if (int(blockIdx.x) >= ntc * ntr) return;
    const int tr = blockIdx.x / ntc, tc = blockIdx.x % ntc;
    // threads of the upsampled half first (cuq quads x pl pixel lanes), then the skip half: with cuq a multiple of 64 every
    // wave runs one of the two load paths only
    const int nup = ca.cuq * a.pl;
    const int q = int(threadIdx.x) < nup ? int(threadIdx.x) % ca.cuq : ca.cuq + (int(threadIdx.x) - nup) % ca.csq;
    const int l = int(threadIdx.x) < nup ? int(threadIdx.x) / ca.cuq : (int(threadIdx.x) - nup) / ca.csq;
    const bool film = a.film != nullptr;
    // (the quad's parameters are requested beside the statistics, not behind them)
    const float4 gam4 = reinterpret_cast<const float4*>(a.gamma[p])[q], bet4 = reinterpret_cast<const float4*>(a.beta[p])[q];
    float4 fsc4 = make_float4(0, 0, 0, 0), fsh4 = make_float4(0, 0, 0, 0);
    if (film) {
        fsc4 = reinterpret_cast<const float4*>(a.film + size_t(b) * a.film_stride)[q];
        fsh4 = reinterpret_cast<const float4*>(a.film + size_t(b) * a.film_stride + a.C)[q];
    }
    if (threadIdx.x < 32) {
        const float* mr = a.mr + ((size_t(b) * 3 + p) * 32 + threadIdx.x) * 2;
        sm[threadIdx.x] = mr[0]; sm[32 + threadIdx.x] = mr[1];
    }
    __syncthreads();
    const int cg = a.C / 32;
    float A[4], Bc[4], sc[4], sh[4];
    {
        const float gv[4] = {gam4.x, gam4.y, gam4.z, gam4.w}, bv[4] = {bet4.x, bet4.y, bet4.z, bet4.w};
        const float s1[4] = {fsc4.x, fsc4.y, fsc4.z, fsc4.w}, s2[4] = {fsh4.x, fsh4.y, fsh4.z, fsh4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * q + k, g = c / cg;
            const float scale = sm[32 + g] * gv[k];
            A[k] = scale;
            Bc[k] = bv[k] - scale * sm[g];
            sc[k] = film ? 1.0f + s1[k] : 1.0f;
            sh[k] = film ? s2[k] : 0.0f;
        }
    }
    __syncthreads();
    const int i0 = tr * kActRows, j0 = tc * kActCols;
    const int i1 = min(h, i0 + kActRows), j1 = min(w, j0 + kActCols);
    const bool is_up = q < ca.cuq;
    const int hi = h / 2, wi = w / 2;
    const float4* us = reinterpret_cast<const float4*>(ca.u[p] + size_t(b) * hi * wi * ca.cuq * 4) + q;
    const float4* ss = reinterpret_cast<const float4*>(ca.sk[p] + size_t(b) * h * w * ca.csq * 4) + (q - ca.cuq);
    float4* ys = reinterpret_cast<float4*>(a.y[p] + size_t(b) * h * w * a.C) + q;
    float4 rowacc[kActRows];
#pragma unroll
    for (int r = 0; r < kActRows; ++r) rowacc[r] = make_float4(0, 0, 0, 0);
    static_assert(kActRows == 8, "up8_column produces eight rows");
    const Up8Rows R = up8_rows(i0, hi);
    for (int j = j0 + l; j < j1; j += a.pl) {
        float4 colacc = make_float4(0, 0, 0, 0);
        float4 xin[kActRows];
        if (is_up) up8_column(us, wi, ca.cuq, R, j, xin);
        else {
#pragma unroll
            for (int r = 0; r < kActRows; ++r) xin[r] = ss[(size_t(min(i0 + r, i1 - 1)) * w + j) * ca.csq];
        }
#pragma unroll
        for (int r = 0; r < kActRows; ++r) {
            const int i = i0 + r;
            if (i < i1) {
                const size_t off = (size_t(i) * w + j) * a.cq;
                const float4 v = xin[r];
                float4 o;
                o.x = fmaf(v.x, A[0], Bc[0]); o.y = fmaf(v.y, A[1], Bc[1]);
                o.z = fmaf(v.z, A[2], Bc[2]); o.w = fmaf(v.w, A[3], Bc[3]);
                if (film) {
                    o.x = o.x * sc[0] + sh[0]; o.y = o.y * sc[1] + sh[1];
                    o.z = o.z * sc[2] + sh[2]; o.w = o.w * sc[3] + sh[3];
                }
                o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w);
                ys[off] = o;
                colacc.x += o.x; colacc.y += o.y; colacc.z += o.z; colacc.w += o.w;
                rowacc[r].x += o.x; rowacc[r].y += o.y; rowacc[r].z += o.z; rowacc[r].w += o.w;
            }
        }
        if (a.with_means)
            reinterpret_cast<float4*>(a.colpart[p] + ((size_t(b) * ntr + tr) * w + j) * a.C)[q] = colacc;
    }
    if (!a.with_means) return;
    float4* smr = reinterpret_cast<float4*>(sm);
#pragma unroll
    for (int r = 0; r < kActRows; ++r) smr[(size_t(l) * kActRows + r) * a.cq + q] = rowacc[r];
    __syncthreads();
    const int nthr = a.cq * a.pl;
    for (int it = threadIdx.x; it < kActRows * a.cq; it += nthr) {
        const int r = it / a.cq, qq = it % a.cq;
        const int i = i0 + r;
        if (i >= i1) continue;
        float4 s = make_float4(0, 0, 0, 0);
        for (int ll = 0; ll < a.pl; ++ll) {
            const float4 v = smr[(size_t(ll) * kActRows + r) * a.cq + qq];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(a.rowpart[p] + ((size_t(b) * ntc + tc) * h + i) * a.C)[qq] = s;
    }
}
int launch_gn_act_cat(const Tri& u, const Tri& sk, int B, GnStats stats, const ActArgs& aa, Tri& y, const MeanPartials* mp,
                      hipStream_t st) {
    GnActCatArgs ca;
    GnActArgs& a = ca.base;
    int maxtiles = 0;
    for (int p = 0; p < 3; ++p) {
        a.x[p] = nullptr; a.y[p] = y.p[p]; a.h[p] = y.g.h[p]; a.w[p] = y.g.w[p];
        ca.u[p] = u.p[p]; ca.sk[p] = sk.p[p];
        a.gamma[p] = aa.gamma[p]; a.beta[p] = aa.beta[p];
        a.rowpart[p] = mp ? mp->rowpart[p] : nullptr;
        a.colpart[p] = mp ? mp->colpart[p] : nullptr;
        maxtiles = std::max(maxtiles, cdiv(a.h[p], kActRows) * cdiv(a.w[p], kActCols));
        S3D_CHECK(y.g.h[p] == 2 * u.g.h[p] && y.g.w[p] == 2 * u.g.w[p] && sk.g.h[p] == y.g.h[p] && sk.g.w[p] == y.g.w[p], S3D_ERR_INVALID, "gn_act_cat: geometry");
    }
    ca.cuq = u.C / 4; ca.csq = sk.C / 4;
    a.mr = stats.mr; a.film = aa.film; a.film_stride = aa.film_stride;
    a.C = y.C; thread_shape(y.C, a.cq, a.pl); a.with_means = mp ? 1 : 0;
    S3D_CHECK(y.C == u.C + sk.C && y.C % 32 == 0 && a.cq * a.pl <= 512 && stats.mr, S3D_ERR_INVALID, "gn_act_cat: C=%d unsupported", y.C);
    if (!maxtiles || !B) return 0;
    size_t shm = std::max(size_t(64) * sizeof(float), size_t(a.pl) * kActRows * a.C * sizeof(float));
    hipLaunchKernelGGL(k_gn_act_cat, dim3(maxtiles, 3, B), dim3(a.cq * a.pl), shm, st, ca);
    S3D_HIP(hipGetLastError());
    return 0;
}

// th.mean over one axis of the activated planes (src/diffusion/unet_triplane.py:38-46): add the tile partials
// in index order and divide by the axis length.
__global__ __launch_bounds__(256) void k_means_finalize(MeanFinArgs a) {
    means_finalize_thread(a, blockIdx.y, blockIdx.z, int(blockIdx.x * 128 + threadIdx.x));
}
MeanFinArgs means_finalize_args(const Geo& g, int C, int B, const MeanPartials& mp, const MeanVecs& mv) {
    MeanFinArgs a;
    a.C = C; a.cq = C / 4; a.B = B;
    for (int p = 0; p < 3; ++p) {
        const int h = g.h[p], w = g.w[p];
        a.src[2 * p] = mp.rowpart[p]; a.dst[2 * p] = mv.rowmean[p];
        a.len[2 * p] = h; a.nt[2 * p] = cdiv(w, kActCols); a.inv[2 * p] = 1.0f / float(w);
        a.src[2 * p + 1] = mp.colpart[p]; a.dst[2 * p + 1] = mv.colmean[p];
        a.len[2 * p + 1] = w; a.nt[2 * p + 1] = cdiv(h, kActRows); a.inv[2 * p + 1] = 1.0f / float(h);
    }
    return a;
}
int launch_means_finalize(const Geo& g, int C, int B, const MeanPartials& mp, MeanVecs mv, hipStream_t st) {
    MeanFinArgs a = means_finalize_args(g, C, B, mp, mv);
    int maxlen = 0;
    for (int v = 0; v < 6; ++v) maxlen = std::max(maxlen, a.len[v]);
    if (!maxlen || !B || !a.cq) return 0;
    S3D_CHECK((long long)maxlen * a.cq < (1ll << 31), S3D_ERR_INVALID, "means_finalize: vector too long");
    hipLaunchKernelGGL(k_means_finalize, dim3(cdiv(maxlen * a.cq, 128), 6, B), dim3(128), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ resampling
// F.avg_pool2d(k=2, s=2)  (src/diffusion/unet_triplane.py:137-139): floor(h/2) x floor(w/2)
struct PoolArgs { const float* x[3]; float* y[3]; int h[3], w[3]; int cq, B; long long begin[4]; };
__global__ void k_avgpool(PoolArgs a) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.begin[3] * a.B) return;
    const int b = int(i / a.begin[3]);
    long long r = i % a.begin[3];
    int p = r >= a.begin[2] ? 2 : (r >= a.begin[1] ? 1 : 0);
    r -= a.begin[p];
    const int q = int(r % a.cq);
    const int ho = a.h[p] / 2, wo = a.w[p] / 2;
    const int xo = int((r / a.cq) % wo), yo = int(r / a.cq / wo);
    const float4* src = reinterpret_cast<const float4*>(a.x[p]) + ((size_t(b) * a.h[p] + 2 * yo) * a.w[p] + 2 * xo) * a.cq + q;
    const float4 v00 = src[0], v01 = src[a.cq], v10 = src[size_t(a.w[p]) * a.cq], v11 = src[size_t(a.w[p] + 1) * a.cq];
    float4 o;
    o.x = (v00.x + v01.x + v10.x + v11.x) * 0.25f; o.y = (v00.y + v01.y + v10.y + v11.y) * 0.25f;
    o.z = (v00.z + v01.z + v10.z + v11.z) * 0.25f; o.w = (v00.w + v01.w + v10.w + v11.w) * 0.25f;
    reinterpret_cast<float4*>(a.y[p])[((size_t(b) * ho + yo) * wo + xo) * a.cq + q] = o;
    (void)ho;
}
__global__ void k_avgpool_gn(PoolArgs a, double* part, int pl) {        // pixel-chunk form that also emits GroupNorm partials of y
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int chunk = blockIdx.x, p = blockIdx.y, b = blockIdx.z;
    const int ho = a.h[p] / 2, wo = a.w[p] / 2, npix = ho * wo;
    const int per = (npix + kGnChunks - 1) / kGnChunks;
    const int p0 = chunk * per, p1 = min(npix, p0 + per);
    const int q = threadIdx.x % a.cq, l = threadIdx.x / a.cq;
    const float4* xb = reinterpret_cast<const float4*>(a.x[p]) + size_t(b) * a.h[p] * a.w[p] * a.cq + q;
    float4* yb = reinterpret_cast<float4*>(a.y[p]) + size_t(b) * npix * a.cq + q;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int pix = p0 + l; pix < p1; pix += pl) {
        const int yo = pix / wo, xo = pix - yo * wo;
        const float4* src = xb + (size_t(2 * yo) * a.w[p] + 2 * xo) * a.cq;
        const float4 v00 = src[0], v01 = src[a.cq], v10 = src[size_t(a.w[p]) * a.cq], v11 = src[size_t(a.w[p] + 1) * a.cq];
        float4 o;
        o.x = (v00.x + v01.x + v10.x + v11.x) * 0.25f; o.y = (v00.y + v01.y + v10.y + v11.y) * 0.25f;
        o.z = (v00.z + v01.z + v10.z + v11.z) * 0.25f; o.w = (v00.w + v01.w + v10.w + v11.w) * 0.25f;
        yb[size_t(pix) * a.cq] = o;
        gn_acc(s, ss, o);
    }
    gn_chunk_reduce(reinterpret_cast<double*>(smem_raw), s, ss, q, l, a.cq * 4, pl, part + ((size_t(b) * 3 + p) * 32 * kGnChunks + chunk) * 2, kGnChunks);
}
int launch_avgpool(const Tri& x, int B, Tri& y, hipStream_t st, const GnPartials* part) {
    PoolArgs a;
    a.cq = x.C / 4; a.B = B; a.begin[0] = 0;
    for (int p = 0; p < 3; ++p) {
        a.x[p] = x.p[p]; a.y[p] = y.p[p]; a.h[p] = x.g.h[p]; a.w[p] = x.g.w[p];
        a.begin[p + 1] = a.begin[p] + (long long)(a.h[p] / 2) * (a.w[p] / 2) * a.cq;
    }
    long long n = a.begin[3] * B;
    if (!n) return 0;
    if (part) {
        int cq, pl; thread_shape(x.C, cq, pl);
        S3D_CHECK(part->maxparts == kGnChunks && part->nsub == 32 && x.C % 32 == 0 && cq <= 1024, S3D_ERR_INVALID, "avgpool: GroupNorm partial layout");
        hipLaunchKernelGGL(k_avgpool_gn, dim3(kGnChunks, 3, B), dim3(cq * pl), size_t(pl) * x.C * 2 * sizeof(double), st, a, part->p, pl);
        S3D_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_avgpool, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// F.interpolate(mode='bilinear', align_corners=False), scale_factor=2 (:116-118) or size=... (:494-499):
// src = (in/out)*(dst+0.5)-0.5 clamped at 0; neighbour index clamped to in-1.
__global__ void k_bilinear(const float* __restrict__ in, float* __restrict__ out, int B, int cq, int hi, int wi,
                           int ho, int wo, int out_cq, int out_q0) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)B * ho * wo * cq;
    if (i >= n) return;
    const int q = int(i % cq);
    long long r = i / cq;
    const int xo = int(r % wo); r /= wo;
    const int yo = int(r % ho);
    const int b = int(r / ho);
    const float sh = float(hi) / float(ho), sw = float(wi) / float(wo);
    float fy = sh * (float(yo) + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
    float fx = sw * (float(xo) + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
    int y0 = int(fy); y0 = y0 > hi - 1 ? hi - 1 : y0;
    int x0 = int(fx); x0 = x0 > wi - 1 ? wi - 1 : x0;
    const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
    const float ly1 = fy - float(y0), ly0 = 1.f - ly1, lx1 = fx - float(x0), lx0 = 1.f - lx1;
    const float4* src = reinterpret_cast<const float4*>(in) + size_t(b) * hi * wi * cq + q;
    const float4 a = src[(size_t(y0) * wi + x0) * cq], bq = src[(size_t(y0) * wi + x1) * cq];
    const float4 c = src[(size_t(y1) * wi + x0) * cq], d = src[(size_t(y1) * wi + x1) * cq];
    float4 o;
    o.x = ly0 * (lx0 * a.x + lx1 * bq.x) + ly1 * (lx0 * c.x + lx1 * d.x);
    o.y = ly0 * (lx0 * a.y + lx1 * bq.y) + ly1 * (lx0 * c.y + lx1 * d.y);
    o.z = ly0 * (lx0 * a.z + lx1 * bq.z) + ly1 * (lx0 * c.z + lx1 * d.z);
    o.w = ly0 * (lx0 * a.w + lx1 * bq.w) + ly1 * (lx0 * c.w + lx1 * d.w);
    reinterpret_cast<float4*>(out)[((size_t(b) * ho + yo) * wo + xo) * out_cq + out_q0 + q] = o;
}
int launch_bilinear(const float* in, int B, int C, int hi, int wi, float* out, int ho, int wo, int out_cstride,
                    int out_coff, hipStream_t st) {
    long long n = (long long)B * ho * wo * (C / 4);
    if (!n) return 0;
    hipLaunchKernelGGL(k_bilinear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, B, C / 4, hi, wi, ho,
                       wo, out_cstride / 4, out_coff / 4);
    S3D_HIP(hipGetLastError());
    return 0;
}
// TriplaneUpsample2x + concat in one pass over all three planes: out[..., 0:Cu] = bilinear2x(u), out[..., Cu:] = skip
// (src/diffusion/unet_triplane.py:106-124, 501-503).  Used when 2x the low-resolution size equals the skip's size.
struct UpCatArgs {
    const float* u[3]; const float* sk[3]; float* out[3];
    int hi[3], wi[3];          // low-resolution sizes; outputs are 2x
    int cuq, csq, B;           // float4 per pixel of the upsampled / skip parts
    long long begin[4];
};
__global__ void k_upcat(UpCatArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.begin[3] * a.B) return;
    const int b = int(i / a.begin[3]);
    long long r = i % a.begin[3];
    const int p = r >= a.begin[2] ? 2 : (r >= a.begin[1] ? 1 : 0);
    r -= a.begin[p];
    const int oq = a.cuq + a.csq;
    const int q = int(r % oq);
    const long long pix = r / oq;
    const int ho = 2 * a.hi[p], wo = 2 * a.wi[p];
    const int xo = int(pix % wo), yo = int(pix / wo);
    float4 o;
    if (q >= a.cuq) {
        o = reinterpret_cast<const float4*>(a.sk[p])[((size_t(b) * ho + yo) * wo + xo) * a.csq + (q - a.cuq)];
    } else {
        const int hi = a.hi[p], wi = a.wi[p];
        float fy = 0.5f * (float(yo) + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
        float fx = 0.5f * (float(xo) + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
        int y0 = int(fy); y0 = y0 > hi - 1 ? hi - 1 : y0;
        int x0 = int(fx); x0 = x0 > wi - 1 ? wi - 1 : x0;
        const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
        const float ly1 = fy - float(y0), ly0 = 1.f - ly1, lx1 = fx - float(x0), lx0 = 1.f - lx1;
        const float4* src = reinterpret_cast<const float4*>(a.u[p]) + size_t(b) * hi * wi * a.cuq + q;
        const float4 v00 = src[(size_t(y0) * wi + x0) * a.cuq], v01 = src[(size_t(y0) * wi + x1) * a.cuq];
        const float4 v10 = src[(size_t(y1) * wi + x0) * a.cuq], v11 = src[(size_t(y1) * wi + x1) * a.cuq];
        o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
        o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
        o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
        o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    }
    reinterpret_cast<float4*>(a.out[p])[((size_t(b) * ho + yo) * wo + xo) * oq + q] = o;
}
// The same pass with 32 pixels x all channel quads per block (four pixel lanes, eight trips), which also leaves the
// block's GroupNorm partial of the concatenated tensor: a thread adds its eight pixels per channel in double, the lanes
// meet through LDS and 32 threads add their group's channels.  (Four pixels per block — 35 328 blocks and as many partial
// records at towerruins size, batch 4 — took 51 us.)
constexpr int kUpcatPx = 32, kUpcatLanes = 4;
__global__ void k_upcat_gn(UpCatArgs a, double* part, int maxparts) {
    extern __shared__ __attribute__((aligned(16))) double up_smd[];     // [2][kUpcatLanes][C]
    const int p = blockIdx.y, b = blockIdx.z;
    const int hi = a.hi[p], wi = a.wi[p], ho = 2 * hi, wo = 2 * wi, npix = ho * wo;
    const int oq = a.cuq + a.csq, C = oq * 4;
    const int q = threadIdx.x % oq, lp = threadIdx.x / oq;
    if (int(blockIdx.x) * kUpcatPx >= npix) return;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int it = 0; it < kUpcatPx / kUpcatLanes; ++it) {
        const int pix = blockIdx.x * kUpcatPx + it * kUpcatLanes + lp;
        if (pix >= npix) break;
        const int xo = pix % wo, yo = pix / wo;
        float4 o;
        if (q >= a.cuq) {
            o = reinterpret_cast<const float4*>(a.sk[p])[((size_t(b) * ho + yo) * wo + xo) * a.csq + (q - a.cuq)];
        } else {
            float fy = 0.5f * (float(yo) + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
            float fx = 0.5f * (float(xo) + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
            int y0 = int(fy); y0 = y0 > hi - 1 ? hi - 1 : y0;
            int x0 = int(fx); x0 = x0 > wi - 1 ? wi - 1 : x0;
            const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
            const float ly1 = fy - float(y0), ly0 = 1.f - ly1, lx1 = fx - float(x0), lx0 = 1.f - lx1;
            const float4* src = reinterpret_cast<const float4*>(a.u[p]) + size_t(b) * hi * wi * a.cuq + q;
            const float4 v00 = src[(size_t(y0) * wi + x0) * a.cuq], v01 = src[(size_t(y0) * wi + x1) * a.cuq];
            const float4 v10 = src[(size_t(y1) * wi + x0) * a.cuq], v11 = src[(size_t(y1) * wi + x1) * a.cuq];
            o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
            o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
            o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
            o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
        }
        reinterpret_cast<float4*>(a.out[p])[((size_t(b) * ho + yo) * wo + xo) * oq + q] = o;
        const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] += double(ov[k]); ss[k] += double(ov[k]) * double(ov[k]); }
    }
    double* sq = up_smd + kUpcatLanes * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) { up_smd[lp * C + 4 * q + k] = s[k]; sq[lp * C + 4 * q + k] = ss[k]; }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int g = threadIdx.x, cg = C / 32;
        double S = 0, SS = 0;
        for (int l = 0; l < kUpcatLanes; ++l)
            for (int c = g * cg; c < (g + 1) * cg; ++c) { S += up_smd[l * C + c]; SS += sq[l * C + c]; }
        double* op = part + (((size_t(b) * 3 + p) * 32 + g) * maxparts + blockIdx.x) * 2;
        op[0] = S; op[1] = SS;
    }
}
bool upcat_gn_parts(const Geo& out_g, int C, int nparts[3]) {
    if (C % 32 != 0 || C / 4 * kUpcatLanes > 1024) return false;
    for (int p = 0; p < 3; ++p) nparts[p] = (out_g.h[p] * out_g.w[p] + kUpcatPx - 1) / kUpcatPx;
    return true;
}
int launch_upcat(const Tri& u, const Tri& sk, int B, Tri& out, hipStream_t st, const GnPartials* part) {
    UpCatArgs a;
    a.cuq = u.C / 4; a.csq = sk.C / 4; a.B = B; a.begin[0] = 0;
    for (int p = 0; p < 3; ++p) {
        a.u[p] = u.p[p]; a.sk[p] = sk.p[p]; a.out[p] = out.p[p]; a.hi[p] = u.g.h[p]; a.wi[p] = u.g.w[p];
        a.begin[p + 1] = a.begin[p] + (long long)4 * u.g.h[p] * u.g.w[p] * (a.cuq + a.csq);
    }
    const long long n = a.begin[3] * B;
    if (!n) return 0;
    if (part) {
        int np[3];
        S3D_CHECK(upcat_gn_parts(out.g, out.C, np) && part->nsub == 32 && part->nparts[0] == np[0] && part->nparts[1] == np[1] && part->nparts[2] == np[2],
                  S3D_ERR_INVALID, "upcat: GroupNorm partial layout");
        const int mx = std::max(np[0], std::max(np[1], np[2]));
        hipLaunchKernelGGL(k_upcat_gn, dim3(mx, 3, B), dim3(out.C / 4 * kUpcatLanes), size_t(2) * kUpcatLanes * out.C * sizeof(double), st, a, part->p, part->maxparts);
        S3D_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_upcat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

__global__ void k_copy_slice(const float* __restrict__ in, float* __restrict__ out, long long npix, int cq, int out_cq,
                             int out_q0) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * cq) return;
    const int q = int(i % cq);
    const long long pix = i / cq;
    reinterpret_cast<float4*>(out)[pix * out_cq + out_q0 + q] = reinterpret_cast<const float4*>(in)[i];
}
int launch_copy_slice(const float* in, int B, int C, int h, int w, float* out, int out_cstride, int out_coff,
                      hipStream_t st) {
    long long npix = (long long)B * h * w;
    long long n = npix * (C / 4);
    if (!n) return 0;
    hipLaunchKernelGGL(k_copy_slice, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, npix, C / 4,
                       out_cstride / 4, out_coff / 4);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ output head
// out = TriplaneNorm -> TriplaneSiLU -> TriplaneConv(ch, out_channels, 1x1) then compose_featmaps
// (src/diffusion/unet_triplane.py:441-445, 507-508 ; src/utils/triplane_util.py:7-17).
// blockIdx.y: plane 0..2, 3 = the DxD corner that compose fills with zeros.
struct OutHeadArgs {
    const float* x[3]; const float* gamma[3]; const float* beta[3];
    const float* mr; const float* w; const float* bias; float* out;
    int h[3], wd[3];
    int C, cq, ppb, Cout, H, W, D;
    GnPartSrc ps;             // mr == null (k_out_head_px only): the block adds the producing convolution's partial sums itself
};
constexpr int kOutCo = 16;     // output channels handled per pass
// One thread = one pixel x one float4 of channels: the activation is evaluated once per element, each thread
// forms its 4-channel partial dot products with the Cout weight rows, and the cq partials of a pixel are added in
// LDS in index order.
__global__ void k_out_head(OutHeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);           // [64] stats, then [ppb][kOutCo][cq] partials
    const int p = blockIdx.y, b = blockIdx.z;
    const int Hc = a.H + a.D, Wc = a.W + a.D;
    if (p == 3) {                                              // zero corner
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)a.Cout * a.D * a.D;
             i += (long long)gridDim.x * blockDim.x) {
            const int dx = int(i % a.D), dy = int((i / a.D) % a.D), co = int(i / a.D / a.D);
            a.out[((size_t(b) * a.Cout + co) * Hc + a.H + dy) * Wc + a.W + dx] = 0.f;
        }
        return;
    }
    const int h = a.h[p], w = a.wd[p], npix = h * w;
    if (int(blockIdx.x) * a.ppb >= npix) return;
    if (threadIdx.x < 32) {
        const float* mr = a.mr + ((size_t(b) * 3 + p) * 32 + threadIdx.x) * 2;
        sm[threadIdx.x] = mr[0]; sm[32 + threadIdx.x] = mr[1];
    }
    __syncthreads();
    const int q = threadIdx.x % a.cq, lp = threadIdx.x / a.cq;
    const int pix = blockIdx.x * a.ppb + lp;
    const bool live = pix < npix;
    const int cg = a.C / 32;
    float act[4];
    {
        const float4 v = live ? reinterpret_cast<const float4*>(a.x[p] + (size_t(b) * npix + pix) * a.C)[q]
                              : make_float4(0, 0, 0, 0);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * q + k, g = c / cg;
            const float scale = sm[32 + g] * a.gamma[p][c];
            act[k] = silu_f(fmaf(vv[k], scale, a.beta[p][c] - scale * sm[g]));
        }
    }
    __syncthreads();                                            // stats region is reused for the partials
    for (int co0 = 0; co0 < a.Cout; co0 += kOutCo) {
        const int nco = min(kOutCo, a.Cout - co0);
        for (int k = 0; k < nco; ++k) {
            const float4 wv = reinterpret_cast<const float4*>(a.w + (size_t(p) * a.Cout + co0 + k) * a.C)[q];
            // rows of cq+1 floats: the column-wise reads of the reduction below hit distinct banks
            sm[(size_t(k) * a.ppb + lp) * (a.cq + 1) + q] = act[0] * wv.x + act[1] * wv.y + act[2] * wv.z + act[3] * wv.w;
        }
        __syncthreads();
        for (int it = threadIdx.x; it < a.ppb * nco; it += blockDim.x) {
            const int k = it / a.ppb, l2 = it % a.ppb;           // consecutive threads -> consecutive pixels
            const int px = blockIdx.x * a.ppb + l2;
            if (px >= npix) continue;
            const float* row = sm + (size_t(k) * a.ppb + l2) * (a.cq + 1);
            float acc = a.bias[p * a.Cout + co0 + k];
            for (int j = 0; j < a.cq; ++j) acc += row[j];
            const int y = px / w, xx = px % w;
            int sy, sx;
            if (p == 0) { sy = y; sx = xx; } else if (p == 1) { sy = y; sx = a.W + xx; } else { sy = a.H + xx; sx = y; }
            a.out[((size_t(b) * a.Cout + co0 + k) * Hc + sy) * Wc + sx] = acc;
        }
        __syncthreads();
    }
}

// The default form for C = 64 / 128 / 256 and Cout <= 16: a block owns 64 pixels that are consecutive in the COMPOSED output
// (a row segment of xy / xz, a column segment of yz), activates them once on the way into LDS (coalesced 16-byte loads,
// one channel quad per thread), then wave w contracts channels [w*C/4, (w+1)*C/4) of its lane's pixel with weight rows
// that are wave-uniform (scalar loads), and the four partial sums per (pixel, cout) meet in LDS for a coalesced NCHW store.
// 28.7 -> 12 us at 128 channels, 128^3 (the thread-per-quad form above spends its time in 12 LDS reduction rounds).
constexpr int kOhPx = 64;
// FUSED (SURVEY.md §2b "K8: fuse with K9"; the sampling loops): the model output of a composed position is not stored (unless
// a.out is set) — the p_sample / ddim_sample update of s3d_sampler.h is applied to it where it is formed, x_t and the noise are
// read and x_{t-1} + pred_xstart written with the same coalesced NCHW accesses; the D x D corner takes the update with a model
// output of exactly 0 (compose_featmaps' zero fill, src/utils/triplane_util.py:7-18).  Same arithmetic in the same order as
// k_out_head_px<CQ, false> followed by k_sampler: bit-identical results.
// CARRY (FUSED only): the block then runs the NEXT step's in_conv on the 64 x_{t-1} pixels it has just formed —
// k_in_conv_lds<32, CQ>'s body twice (its blocks own 32 pixels), fed from LDS instead of from memory: the same products in the same
// order, the same thread mapping and the same per-block GroupNorm partial sums, written to the tensors the next forward reads.
template <int CQ, bool FUSED, bool CARRY = false>          // channel quads per pixel: 16, 32 or 64
__global__ __launch_bounds__(256, 3) void k_out_head_px(OutHeadArgs a, int segs0, int segs1, int segs2, s3d_sampler_args sa, InConvCarry cy = InConvCarry()) {
    constexpr int C = 4 * CQ, CPW = C / 4, LANES = 256 / CQ, LD = C + 4;
    __shared__ __attribute__((aligned(16))) float sx[kOhPx * LD];
    __shared__ float sp[4][16][kOhPx];
    __shared__ float sst[64];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int Hc = a.H + a.D, Wc = a.W + a.D;
    int blk = blockIdx.x, p = 0;
    if (blk >= segs0) { blk -= segs0; p = 1; if (blk >= segs1) { blk -= segs1; p = 2; } }
    // The D x D corner of compose_featmaps (zeros; FUSED: the sampler update with a model output of 0 — a quarter of the composed
    // map at D = H = W) is spread over the pixel blocks: element i = block * 256 + thread (+ k * all threads).  Extra corner
    // blocks behind the pixel blocks were a second, nearly empty round of the chip (768 + 192 blocks on 768 slots: 25 us);
    // here x_t / eps of the thread's first two corner elements are requested with the block's other early loads.
    const long long n_corner = (long long)a.Cout * a.D * a.D;
    const long long corner_stride = 256LL * gridDim.x;
    const long long corner_i0 = (long long)blockIdx.x * 256 + tid;
    constexpr int CU_ = 2;
    size_t corner_o[CU_]; float corner_x[CU_], corner_n[CU_];
#pragma unroll
    for (int u = 0; u < CU_; ++u) {
        const long long i = corner_i0 + u * corner_stride;
        corner_o[u] = 0; corner_x[u] = corner_n[u] = 0.f;
        if (i < n_corner) {
            const int dx = int(i % a.D), dy = int((i / a.D) % a.D), co = int(i / a.D / a.D);
            corner_o[u] = ((size_t(b) * a.Cout + co) * Hc + a.H + dy) * Wc + a.W + dx;
            // (CARRY: requested behind the output loop instead — their latency then runs under the in_conv tail; kept from here they
            // were live across the whole kernel and spilled behind a vmcnt(0) at its very start)
            if (FUSED && !CARRY) { corner_x[u] = sa.x[corner_o[u]]; if (sa.noise) corner_n[u] = sa.noise[corner_o[u]]; }
        }
    }
    const int h = a.h[p], w = a.wd[p];
    const int len = p == 2 ? h : w, nseg = (len + kOhPx - 1) / kOhPx;
    const int line = blk / nseg, s0 = (blk % nseg) * kOhPx;
    // the plane's weight rows [Cout][C] are requested NOW and parked in registers: read from memory inside the contraction
    // (wave-uniform addresses, eight 16-byte loads and a full wait per pair of output channels) they were ~8 us of exposed L2
    // round trips per block (batch 8: 171 -> see profiles/r04_out_head.txt); they move into the staging area once every lane
    // holds its activations
    constexpr int kWIts = (16 * CQ + 255) / 256;             // float4 items of Cout <= 16 rows per thread
    float4 wreg[kWIts];
#pragma unroll
    for (int k = 0; k < kWIts; ++k) {
        const int it = tid + 256 * k;
        wreg[k] = it < a.Cout * CQ ? reinterpret_cast<const float4*>(a.w + size_t(p) * a.Cout * C)[it] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // FUSED: x_t and the step's eps of this thread's output positions are requested NOW — as the tail of the block (behind the
    // last barrier) their latency was the block's critical path: 31 us per launch instead of 15 + 7 for head + sampler kernel
    constexpr int kOutIts = 16 * kOhPx / 256;                 // Cout <= 16
    float pre_x[kOutIts], pre_n[kOutIts];
    SamplerCoef sc;
    if (FUSED) {
#pragma unroll
        for (int k = 0; k < kOutIts; ++k) {
            const int it = tid + 256 * k, co = it >> 6, e = it & 63;
            pre_x[k] = pre_n[k] = 0.f;
            if (co < a.Cout && s0 + e < len) {
                const int sy = p == 2 ? a.H + line : line, sx0 = p == 1 ? a.W + s0 : s0;
                const size_t o = ((size_t(b) * a.Cout + co) * Hc + sy) * Wc + sx0 + e;
                pre_x[k] = sa.x[o];
                if (sa.noise) pre_n[k] = sa.noise[o];
            }
        }
        sc = sampler_coef(sa, int(sa.t[b]));                  // (t, then the seven table entries it selects: two dependent round trips, behind the requests above)
    }
    const float* st = sst;
    if (a.mr) {
        if (tid < 32) {
            const float* mr = a.mr + ((size_t(b) * 3 + p) * 32 + tid) * 2;
            sst[tid] = mr[0]; sst[32 + tid] = mr[1];
        }
    } else {
        // the statistics of the head's input from its producer's partial sums, as k_gn_act does it: one k_gn_finalize launch less
        // on the step's dependent chain (the L2-hot reads run beside the x_t / eps requests above)
        st = gn_stats_from_parts(a.ps, b, p, reinterpret_cast<double*>(sx), nullptr);
    }
    __syncthreads();
    {
        const int q = tid % CQ, l = tid / CQ;
        constexpr int cg = C / 32;
        float A[4], Bc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * q + k, g = c / cg;
            A[k] = st[32 + g] * a.gamma[p][c];
            Bc[k] = a.beta[p][c] - A[k] * st[g];
        }
        if (!a.mr) __syncthreads();                         // the added statistics live in sx, which the staging below overwrites
        const float* xb = a.x[p] + size_t(b) * h * w * C;
#pragma unroll
        for (int k = 0; k < kOhPx / LANES; ++k) {
            const int e = k * LANES + l;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s0 + e < len) {
                const int y = p == 2 ? s0 + e : line, xx = p == 2 ? line : s0 + e;
                const float4 v = reinterpret_cast<const float4*>(xb + (size_t(y) * w + xx) * C)[q];
                o.x = silu_f(fmaf(v.x, A[0], Bc[0])); o.y = silu_f(fmaf(v.y, A[1], Bc[1]));
                o.z = silu_f(fmaf(v.z, A[2], Bc[2])); o.w = silu_f(fmaf(v.w, A[3], Bc[3]));
            }
            *reinterpret_cast<float4*>(sx + e * LD + 4 * q) = o;
        }
    }
    __syncthreads();
    {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), e = tid & 63;
        float act[CPW];
#pragma unroll
        for (int j = 0; j < CPW / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(sx + e * LD + wv * CPW + 4 * j);
            act[4 * j] = v.x; act[4 * j + 1] = v.y; act[4 * j + 2] = v.z; act[4 * j + 3] = v.w;
        }
        __syncthreads();                                    // every lane holds its activations: the staging area takes the weight rows
#pragma unroll
        for (int k = 0; k < kWIts; ++k) {
            const int it = tid + 256 * k;
            if (it < 16 * CQ) reinterpret_cast<float4*>(sx)[it] = wreg[k];
        }
        __syncthreads();
        const float* wb = sx + wv * CPW;                    // [Cout][C]; wave-uniform addresses: LDS broadcast reads
        for (int co = 0; co < a.Cout; ++co) {
            const float4* wr = reinterpret_cast<const float4*>(wb + co * C);
            float s0a = 0.f, s1a = 0.f;
#pragma unroll
            for (int j = 0; j < CPW / 4; ++j) {
                const float4 w4 = wr[j];
                s0a = fmaf(act[4 * j], w4.x, s0a); s1a = fmaf(act[4 * j + 1], w4.y, s1a);
                s0a = fmaf(act[4 * j + 2], w4.z, s0a); s1a = fmaf(act[4 * j + 3], w4.w, s1a);
            }
            sp[wv][co][e] = s0a + s1a;
        }
    }
    __syncthreads();
    float xnext[kOutIts];
#pragma unroll
    for (int k = 0; k < kOutIts; ++k) xnext[k] = 0.f;
    // CARRY: the next in_conv's weight rows for input channels 0..7 are requested now (their latency runs under the output loop), the
    // other eight once the x_{t-1} values are staged
    float4 wA[8], wB[8];
    float4 biasn = make_float4(0, 0, 0, 0);
    if (FUSED && CARRY) {
        const float4* wq0 = reinterpret_cast<const float4*>(cy.wT + size_t(p) * cy.Cin * C) + tid % CQ;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) wA[ci] = ci < cy.Cin ? wq0[size_t(ci) * CQ] : make_float4(0, 0, 0, 0);
        biasn = reinterpret_cast<const float4*>(cy.bias + size_t(p) * C)[tid % CQ];
    }
#pragma unroll
    for (int k = 0; k < kOutIts; ++k) {
        const int it = tid + 256 * k, co = it >> 6, e = it & 63;
        if (co >= a.Cout || s0 + e >= len) continue;
        const float v = ((sp[0][co][e] + sp[1][co][e]) + (sp[2][co][e] + sp[3][co][e])) + a.bias[p * a.Cout + co];
        const int sy = p == 2 ? a.H + line : line, sx0 = p == 1 ? a.W + s0 : s0;
        const size_t o = ((size_t(b) * a.Cout + co) * Hc + sy) * Wc + sx0 + e;
        if (!FUSED || a.out) a.out[o] = v;
        if (FUSED) {
            const float xprev = sampler_element(sa, sc, (long long)o, v, pre_x[k], pre_n[k]);
            if (CARRY) xnext[k] = xprev;
        }
    }
    if (FUSED && CARRY) {
#pragma unroll
        for (int u = 0; u < CU_; ++u) {
            const long long i = (long long)blockIdx.x * 256 + tid + u * corner_stride;
            if (i < n_corner) {
                const int dx = int(i % a.D), dy = int((i / a.D) % a.D), co = int(i / a.D / a.D);
                corner_o[u] = ((size_t(b) * a.Cout + co) * Hc + a.H + dy) * Wc + a.W + dx;
                corner_x[u] = sa.x[corner_o[u]]; if (sa.noise) corner_n[u] = sa.noise[corner_o[u]];
            }
        }
        // ---- the next step's in_conv on this block's x_{t-1} (k_in_conv_lds<32, CQ>, twice).  sx is dead since the barrier above.
        constexpr int LANES_I = 256 / CQ, PXI = 32;
        float (*sxn)[16] = reinterpret_cast<float (*)[16]>(sx);                        // [64 pixels][16 input channels]
        float (*sred)[LANES_I][C] = reinterpret_cast<float (*)[LANES_I][C]>(sx + kOhPx * 16);      // [sum | sumsq][pixel lane][channel]
        static_assert(kOhPx * 16 + 2 * LANES_I * C <= kOhPx * LD, "carry scratch inside the staging area");
#pragma unroll
        for (int k = 0; k < kOutIts; ++k) {
            const int it = tid + 256 * k, co = it >> 6, e = it & 63;
            sxn[e][co] = (co < cy.Cin && s0 + e < len) ? xnext[k] : 0.f;               // (channels Cin..15 and pixels past the line: zeros, as k_in_conv_lds stages them)
        }
        const int q = tid % CQ, lp = tid / CQ;
        const float4* wq = reinterpret_cast<const float4*>(cy.wT + size_t(p) * cy.Cin * C) + q;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) wB[ci] = 8 + ci < cy.Cin ? wq[size_t(8 + ci) * CQ] : make_float4(0, 0, 0, 0);
        const int nseg_in = (len + PXI - 1) / PXI;
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < kOhPx / PXI; ++hh) {
            const int s0i = s0 + hh * PXI;
            if (s0i >= len) break;                                                     // (block-uniform: no such in_conv block)
            // every output's chain is bias, then input channels 0..15 in order (k_in_conv_lds's)
            constexpr int NK = PXI / LANES_I;
            float4 acc[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) acc[k] = biasn;
#pragma unroll
            for (int half8 = 0; half8 < 2; ++half8) {
                const float4* wvn = half8 ? wB : wA;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int e = k * LANES_I + lp;
#pragma unroll
                    for (int c4 = 0; c4 < 2; ++c4) {
                        const float4 v = *reinterpret_cast<const float4*>(&sxn[hh * PXI + e][(half8 * 2 + c4) * 4]);
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float4 ww = wvn[c4 * 4 + j];
                            acc[k].x = fmaf(vv[j], ww.x, acc[k].x); acc[k].y = fmaf(vv[j], ww.y, acc[k].y);
                            acc[k].z = fmaf(vv[j], ww.z, acc[k].z); acc[k].w = fmaf(vv[j], ww.w, acc[k].w);
                        }
                    }
                }
            }
            float gs[4] = {0.f, 0.f, 0.f, 0.f}, gss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int e = k * LANES_I + lp;
                if (s0i + e >= len) continue;
                const int y = p == 2 ? s0i + e : line, xx = p == 2 ? line : s0i + e;
                reinterpret_cast<float4*>(cy.out[p] + ((size_t(b) * h + y) * w + xx) * C)[q] = acc[k];
                gs[0] += acc[k].x; gs[1] += acc[k].y; gs[2] += acc[k].z; gs[3] += acc[k].w;
                gss[0] = fmaf(acc[k].x, acc[k].x, gss[0]); gss[1] = fmaf(acc[k].y, acc[k].y, gss[1]); gss[2] = fmaf(acc[k].z, acc[k].z, gss[2]); gss[3] = fmaf(acc[k].w, acc[k].w, gss[3]);
            }
            if (cy.part) {
                if (hh) __syncthreads();                                               // the previous half's partial sums have been read
#pragma unroll
                for (int k = 0; k < 4; ++k) { sred[0][lp][4 * q + k] = gs[k]; sred[1][lp][4 * q + k] = gss[k]; }
                __syncthreads();
                if (tid < 32) {
                    constexpr int cg = C / 32;
                    const int g = tid;
                    double S = 0, SS = 0;
                    for (int l = 0; l < LANES_I; ++l)
#pragma unroll
                        for (int k = 0; k < cg; ++k) { S += sred[0][l][cg * g + k]; SS += sred[1][l][cg * g + k]; }
                    const int blk_in = line * nseg_in + s0i / PXI;
                    double* o = cy.part + (((size_t(b) * 3 + p) * 32 + g) * cy.maxparts + blk_in) * 2;
                    o[0] = S; o[1] = SS;
                }
            }
        }
    }
    // (the corner index again from an opaque thread id: kept from the top of the kernel it was a 64-bit value live across the CARRY tail — spilled)
    int tid_c = threadIdx.x;
    if (CARRY) asm volatile("" : "+v"(tid_c));
    const long long corner_i1 = CARRY ? (long long)blockIdx.x * 256 + tid_c : corner_i0;
#pragma unroll
    for (int u = 0; u < CU_; ++u) {
        if (corner_i1 + u * corner_stride >= n_corner) continue;
        if (!FUSED || a.out) a.out[corner_o[u]] = 0.f;
        if (FUSED) sampler_element(sa, sc, (long long)corner_o[u], 0.f, corner_x[u], corner_n[u]);
    }
    for (long long i = corner_i1 + CU_ * corner_stride; i < n_corner; i += corner_stride) {      // (shapes with D*D > 2 x the plane pixels)
        const int dx = int(i % a.D), dy = int((i / a.D) % a.D), co = int(i / a.D / a.D);
        const size_t o = ((size_t(b) * a.Cout + co) * Hc + a.H + dy) * Wc + a.W + dx;
        if (!FUSED || a.out) a.out[o] = 0.f;
        if (FUSED) sampler_element(sa, sc, (long long)o, 0.f);
    }
}
static bool out_head_px_form(int C, int Cout) { return (C == 64 || C == 128 || C == 256) && Cout <= 16; }      // (other widths: k_out_head)
// The in-head sampler update: head + sampler kernel + their launch gap 23 -> 20 us at 128^3, batch 1.  Until the head's weight
// rows stopped being fetched inside the contraction (round 4) it lost from batch 2 on (batch 8: 172 us fused against 82 + 35);
// now it wins there too (config 3: 5.173 -> 5.160 ms/step) and is taken at every batch.
bool out_head_fuses_sampler(int C, int Cout, int B) { (void)B; return out_head_px_form(C, Cout); }
bool out_head_px_takes(int C, int Cout) { return out_head_px_form(C, Cout); }
// fuse != null: the sampler update of one denoising step is applied to the model output (fuse->model_out is ignored).  When
// the pixel-chunk form takes the launch it happens in the same kernel and `out` may be null (the model output is then never
// stored); otherwise `out` is required and the stand-alone k_sampler follows.
// part != null (stats.mr == null; only when out_head_adds_parts() says so): the kernel adds the producer's GroupNorm partials itself.
long long out_head_px_blocks(const Geo& g, int B) {
    long long n = 0;
    for (int p = 0; p < 3; ++p) { const int len = p == 2 ? g.h[p] : g.w[p], lines = p == 2 ? g.w[p] : g.h[p]; n += (long long)lines * cdiv(len, kOhPx); }
    return n * B;
}
bool out_head_adds_parts(const GnPartials& part, int C, int Cout) { return out_head_px_form(C, Cout) && gn_act_can_add_parts(part, C); }
bool out_head_can_carry(int C, int Cin, int Cout_head) { return out_head_px_form(C, Cout_head) && Cin <= 16 && Cin == Cout_head; }
int launch_out_head(const Tri& x, int B, GnStats stats, const ActArgs& aa, const float* w, const float* bias,
                    int Cout, int H, int W, int D, float* out, hipStream_t st, const s3d_sampler_args* fuse, const GnPartials* part,
                    const InConvCarry* carry) {
    OutHeadArgs a;
    int maxpix = D * D ? 1 : 0;
    for (int p = 0; p < 3; ++p) {
        a.x[p] = x.p[p]; a.gamma[p] = aa.gamma[p]; a.beta[p] = aa.beta[p];
        a.h[p] = x.g.h[p]; a.wd[p] = x.g.w[p];
        maxpix = std::max(maxpix, a.h[p] * a.wd[p]);
    }
    a.mr = stats.mr; a.w = w; a.bias = bias; a.out = out; a.C = x.C; a.Cout = Cout; a.H = H; a.W = W; a.D = D;
    memset(&a.ps, 0, sizeof a.ps);
    if (part) {
        S3D_CHECK(!stats.mr && out_head_adds_parts(*part, x.C, Cout), S3D_ERR_INVALID, "out head: partial-sum statistics layout");
        a.ps = gn_part_src(*part, x.g, x.C);
    } else S3D_CHECK(stats.mr, S3D_ERR_INVALID, "out head: no GroupNorm statistics");
    thread_shape(x.C, a.cq, a.ppb);
    S3D_CHECK(x.C % 32 == 0 && a.cq <= 1024, S3D_ERR_INVALID, "out head: C=%d unsupported", x.C);
    if (!maxpix || !B) return 0;
    if (fuse) S3D_CHECK(fuse->batch == B && fuse->per_sample == (long long)Cout * (H + D) * (W + D), S3D_ERR_INVALID, "out head: the sampler step does not match the model output's shape");
    const bool fuse_here = fuse && out_head_fuses_sampler(x.C, Cout, B);
    if (fuse && !fuse_here) S3D_CHECK(out, S3D_ERR_INVALID, "out head: this step needs a model-output buffer");
    if (out_head_px_form(x.C, Cout)) {
        int segs[3];
        for (int p = 0; p < 3; ++p) { const int len = p == 2 ? a.h[p] : a.wd[p], lines = p == 2 ? a.wd[p] : a.h[p]; segs[p] = lines * cdiv(len, kOhPx); }
        const dim3 grid(segs[0] + segs[1] + segs[2], B);
        s3d_sampler_args sa;
        memset(&sa, 0, sizeof sa);
        if (carry) S3D_CHECK(fuse_here && out_head_can_carry(x.C, carry->Cin, Cout) && fuse->mode != S3D_STEP_MEAN_ONLY, S3D_ERR_INVALID, "out head: this step cannot carry the next in_conv");
        if (fuse_here && carry) {
            sa = *fuse;
            if (x.C == 64) hipLaunchKernelGGL((k_out_head_px<16, true, true>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], sa, *carry);
            else if (x.C == 128) hipLaunchKernelGGL((k_out_head_px<32, true, true>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], sa, *carry);
            else hipLaunchKernelGGL((k_out_head_px<64, true, true>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], sa, *carry);
        } else if (fuse_here) {
            sa = *fuse;
            if (x.C == 64) hipLaunchKernelGGL((k_out_head_px<16, true>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], sa);
            else if (x.C == 128) hipLaunchKernelGGL((k_out_head_px<32, true>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], sa);
            else hipLaunchKernelGGL((k_out_head_px<64, true>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], sa);
        } else {
            if (x.C == 64) hipLaunchKernelGGL((k_out_head_px<16, false>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], sa);
            else if (x.C == 128) hipLaunchKernelGGL((k_out_head_px<32, false>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], sa);
            else hipLaunchKernelGGL((k_out_head_px<64, false>), grid, dim3(256), 0, st, a, segs[0], segs[1], segs[2], sa);
        }
        S3D_HIP(hipGetLastError());
        if (fuse && !fuse_here) { s3d_sampler_args sb = *fuse; sb.model_out = out; return launch_sampler(sb, st); }
        return 0;
    }
    S3D_CHECK(out, S3D_ERR_INVALID, "out head: this width needs a model-output buffer");
    size_t shm = std::max(size_t(64), size_t(a.ppb) * kOutCo * (a.cq + 1)) * sizeof(float);
    hipLaunchKernelGGL(k_out_head, dim3(cdiv(maxpix, a.ppb), 4, B), dim3(a.cq * a.ppb), shm, st, a);
    S3D_HIP(hipGetLastError());
    if (fuse) { s3d_sampler_args sa = *fuse; sa.model_out = out; return launch_sampler(sa, st); }
    return 0;
}

// ------------------------------------------------------------------ timestep path
// timestep_embedding (src/diffusion/nn.py:103-121) -> time_embed (unet_triplane.py:371-375, 477) ->
// per-block emb_layers SiLU+Linear (:232-238, 281).  One wave per output feature.
__global__ void k_linear(const float* __restrict__ in, int I, const float* __restrict__ W, const float* __restrict__ bias,
                         int O, float* __restrict__ out, int B, int in_mode, int out_silu) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * O) return;
    const int b = wave / O, o = wave % O;
    const float* wr = W + size_t(o) * I;
    float acc = 0.f;
    if (in_mode == 2) {
        const float t = in[b];
        const int half = I / 2;
        for (int i = lane; i < I; i += 64) {
            float v;
            if (i < 2 * half) {
                const int k = i < half ? i : i - half;
                const float f = expf(-logf(10000.f) * float(k) / float(half));
                const float arg = t * f;
                v = i < half ? cosf(arg) : sinf(arg);
            } else v = 0.f;                                  // odd dim: trailing zero column (nn.py:119-120)
            acc = fmaf(v, wr[i], acc);
        }
    } else {
        const float* xr = in + size_t(b) * I;
        for (int i = lane; i < I; i += 64) {
            float v = xr[i];
            if (in_mode == 1) v = silu_f(v);
            acc = fmaf(v, wr[i], acc);
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) {
        float r = acc + (bias ? bias[o] : 0.f);
        out[size_t(b) * O + o] = out_silu ? silu_f(r) : r;
    }
}
int launch_linear(const float* in, int B, int I, const float* W, const float* bias, int O, float* out, int in_mode,
                  int out_silu, hipStream_t st) {
    long long waves = (long long)B * O;
    if (!waves) return 0;
    hipLaunchKernelGGL(k_linear, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, in, I, W, bias, O, out, B, in_mode,
                       out_silu);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ sampler update
// the stand-alone form of s3d_sampler.h (s3d_sampler_step; the sampling loops use the form fused into the output head)
__global__ void k_sampler(s3d_sampler_args a) {
    const long long n = a.batch * a.per_sample;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = int(i / a.per_sample);
        const SamplerCoef c = sampler_coef(a, int(a.t[b]));
        sampler_element(a, c, i, a.model_out[i]);
    }
}
int launch_sampler(const s3d_sampler_args& a, hipStream_t st) {
    const long long n = a.batch * a.per_sample;
    if (n <= 0) return 0;
    unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_sampler, dim3(blocks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

}  // namespace s3d
