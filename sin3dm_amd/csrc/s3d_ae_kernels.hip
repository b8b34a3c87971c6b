// s3d_ae_kernels.hip — kernels of the auto-encoder training tier that are not shared with the
// denoiser: the projected encoder, the point gather / scatter, the thin first/last MLP layers and the losses.
//
// Encoder (AutoEncoderGroupSkip.encode, src/encoding/networks.py:164-180): Conv3d(k4, s2, p1) over the whole input
// volume followed by a mean over one axis.  The mean commutes with the convolution, so the 3-D feature volume is never
// formed: for the xy plane
//     mean_z conv3d(vol)[co,x,y] = b + 1/D * sum_{ci,kx,ky,kz} W[co,ci,kx,ky,kz] * Pz[X=2x+kx-1][Y=2y+ky-1][kz][ci]
// with Pz[X][Y][kz][ci] = sum_z vol[ci,X,Y,2z+kz-1] — a 4x4 stride-2 2-D convolution over a 4*Cin-channel projection
// that is computed ONCE per training volume (k_project).  The same holds for xz (Py) and yz (Px); the weight gradient
// is the matching 2-D correlation (k_enc_wgrad).  Per step this replaces 9.7 GFLOP of Conv3d by 0.15 GFLOP.
#include <hipcub/hipcub.hpp>

#include "s3d_ae.h"

namespace s3d {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline unsigned cdivll(long long a, int b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------ volume projections
// vol [C][X2][Y2][Z2] -> P[axis] laid out [A][B][4 taps][C] where (A, B) are the two kept axes (xy: X,Y ; xz: X,Z ;
// yz: Y,Z) and tap k sums the removed axis over indices 2j+k-1, j = 0..n/2-1, that fall inside [0, n).
__global__ void k_project(const float* __restrict__ vol, float* __restrict__ P, int C, int X2, int Y2, int Z2, int axis) {
    // one thread per (a, b, c); loops the removed axis once and feeds the four tap sums
    const int n0 = axis == 2 ? X2 : (axis == 1 ? X2 : Y2), n1 = axis == 2 ? Y2 : Z2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n0 * n1 * C) return;
    const int c = int(idx % C);
    const int b = int((idx / C) % n1), a = int(idx / ((long long)C * n1));
    const int n = axis == 2 ? Z2 : (axis == 1 ? Y2 : X2);
    double s[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        size_t off;
        if (axis == 2) off = ((size_t(c) * X2 + a) * Y2 + b) * Z2 + i;
        else if (axis == 1) off = ((size_t(c) * X2 + a) * Y2 + i) * Z2 + b;
        else off = ((size_t(c) * X2 + i) * Y2 + a) * Z2 + b;
        const double v = vol[off];
        // tap k sums i = 2j+k-1, j = 0..n/2-1, inside [0, n):
        //   k=0: odd i <= n-3 ; k=1: every even i ; k=2: every odd i ; k=3: even i >= 2
        if (i & 1) { s[2] += v; if (i <= n - 3) s[0] += v; }
        else { s[1] += v; if (i >= 2) s[3] += v; }
    }
    float* o = P + ((size_t(a) * n1 + b) * 4) * C + c;
    for (int k = 0; k < 4; ++k) o[size_t(k) * C] = float(s[k]);
}
int launch_project(const float* vol, int C, int X2, int Y2, int Z2, float* const P[3], hipStream_t st) {
    // P[0] = xy plane (sum over z, axis 2), P[1] = xz (sum over y), P[2] = yz (sum over x)
    const long long n[3] = {(long long)X2 * Y2 * C, (long long)X2 * Z2 * C, (long long)Y2 * Z2 * C};
    const int axis[3] = {2, 1, 0};
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(k_project, dim3(cdivll(n[p], 256)), dim3(256), 0, st, vol, P[p], C, X2, Y2, Z2, axis[p]);
        S3D_HIP(hipGetLastError());
    }
    return 0;
}

// ------------------------------------------------------------------ encoder forward on the projections
// pre[p][a][b][co] = bias[co] + inv_len * sum_{ka,kb,ks,ci} Wp[p][co][(ka*4+kb)*4*C + ks*C + ci] * P[p][2a+ka-1][2b+kb-1][ks][ci]
// Wp is the Conv3d weight permuted per plane by k_enc_pack (taps of the two kept axes outermost).
struct EncArgs {
    const float* P[3]; const float* Wp; const float* bias;     // Wp [3][CO][16*4*C]
    float* pre[3];
    int h[3], w[3];                                              // plane sizes (feature map); projections are 2h x 2w
    float inv_len[3];
    int C, CO;
    long long begin[4];
};
// One thread per output PIXEL, one wave per block (round 5; it was one thread per (pixel, output channel): 256 scalar
// multiply-adds on 512 scalar loads each, 300 us for 0.3 GFLOP): the wave first copies its plane's weights into LDS as
// [tap][k/4][co][4] (12 KB for CO = 12, C = 4), then every pixel's 4x4 taps x 4C projection values are read once as float4 and
// feed all CO accumulators from broadcast LDS reads.  Same order of the multiply-adds as before.
constexpr int kEncFwdMaxCO = 12, kEncFwdMaxK4 = 4;
__global__ __launch_bounds__(64) void k_enc_fwd(EncArgs a) {
    __shared__ float4 sw[16 * kEncFwdMaxK4 * kEncFwdMaxCO];
    const int p = blockIdx.y;
    const int h = a.h[p], w = a.w[p];
    const int K = 4 * a.C, h2 = 2 * h, w2 = 2 * w, CO = a.CO, K4 = K / 4;
    {
        const float4* src = reinterpret_cast<const float4*>(a.Wp + size_t(p) * CO * 16 * K);      // [co][tap][k4]
        if (CO < kEncFwdMaxCO || K4 < kEncFwdMaxK4) {                                                 // [tap][k4][co], zero past K4 / CO
            for (int e = threadIdx.x; e < 16 * kEncFwdMaxK4 * kEncFwdMaxCO; e += 64) sw[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
        }
        for (int e = threadIdx.x; e < CO * 16 * K4; e += 64) {                                      // consecutive lanes read consecutive float4
            const int k4 = e % K4, tap = (e / K4) % 16, co = e / (16 * K4);
            sw[(tap * kEncFwdMaxK4 + k4) * kEncFwdMaxCO + co] = src[e];
        }
    }
    __syncthreads();
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= h * w) return;
    const int x = idx / w, y = idx - x * w;
    float acc[kEncFwdMaxCO];
#pragma unroll
    for (int c = 0; c < kEncFwdMaxCO; ++c) acc[c] = 0.f;
    for (int ka = 0; ka < 4; ++ka) {
        const int X = 2 * x + ka - 1;
        float4 v[4][kEncFwdMaxK4];                           // the row's 4 taps x K values, every load issued before the first use
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int Y = 2 * y + kb - 1;
            const bool in = X >= 0 && X < h2 && Y >= 0 && Y < w2;
            const float4* pr = reinterpret_cast<const float4*>(a.P[p] + (size_t(in ? X : 0) * w2 + (in ? Y : 0)) * K);
#pragma unroll
            for (int k4 = 0; k4 < kEncFwdMaxK4; ++k4) {
                const float4 ld = pr[min(k4, K4 - 1)];       // (weights past K4 are zero)
                v[kb][k4] = in ? ld : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int k4 = 0; k4 < kEncFwdMaxK4; ++k4) {
                const float4* wt = sw + ((ka * 4 + kb) * kEncFwdMaxK4 + k4) * kEncFwdMaxCO;
#pragma unroll
                for (int c = 0; c < kEncFwdMaxCO; ++c) {
                    const float4 wc = wt[c], u = v[kb][k4];
                    acc[c] = fmaf(wc.w, u.w, fmaf(wc.z, u.z, fmaf(wc.y, u.y, fmaf(wc.x, u.x, acc[c]))));
                }
            }
    }
    float* o = a.pre[p] + size_t(idx) * CO;
#pragma unroll
    for (int c = 0; c < kEncFwdMaxCO; ++c) if (c < CO) o[c] = a.bias[c] + a.inv_len[p] * acc[c];
}
// Conv3d weights [CO][C][4][4][4] (geo rows use input channel 0 only) -> Wp[p][co][(ka*4+kb)*4*C + ks*C + ci]
__global__ void k_enc_pack(const float* __restrict__ wgeo, const float* __restrict__ wtex, int geo, int tex, int C, float* __restrict__ Wp) {
    const int CO = geo + tex, K = 4 * C;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * CO * 16 * K) return;
    const int ci = idx % C, ks = (idx / C) % 4, kb = (idx / K) % 4, ka = (idx / (4 * K)) % 4;
    const int co = (idx / (16 * K)) % CO, p = idx / (16 * K * CO);
    // plane 0 (xy): (ka,kb,ks) = (kx,ky,kz); plane 1 (xz): (kx,kz,ky); plane 2 (yz): (ky,kz,kx)
    const int kx = p == 2 ? ks : ka, ky = p == 0 ? kb : (p == 1 ? ks : ka), kz = p == 0 ? ks : kb;
    float v = 0.f;
    if (co < geo) { if (ci == 0) v = wgeo[(size_t(co) * 64) + (kx * 4 + ky) * 4 + kz]; }
    else v = wtex[((size_t(co - geo) * C + ci) * 64) + (kx * 4 + ky) * 4 + kz];
    Wp[idx] = v;
}
int launch_enc_pack(const float* wgeo, const float* wtex, int geo, int tex, int C, float* Wp, hipStream_t st) {
    const int n = 3 * (geo + tex) * 16 * 4 * C;
    hipLaunchKernelGGL(k_enc_pack, dim3(cdiv(n, 256)), dim3(256), 0, st, wgeo, wtex, geo, tex, C, Wp);
    S3D_HIP(hipGetLastError());
    return 0;
}
int launch_enc_fwd(const EncDesc& e, const float* Wp, const float* bias, float* const pre[3], hipStream_t st) {
    EncArgs a;
    a.Wp = Wp; a.bias = bias; a.C = e.C; a.CO = e.CO; a.begin[0] = 0;
    for (int p = 0; p < 3; ++p) {
        a.P[p] = e.P[p]; a.pre[p] = pre[p]; a.h[p] = e.g.h[p]; a.w[p] = e.g.w[p]; a.inv_len[p] = e.inv_len[p];
        a.begin[p + 1] = a.begin[p] + (long long)e.g.h[p] * e.g.w[p] * e.CO;
    }
    S3D_CHECK(e.CO <= kEncFwdMaxCO && e.C <= kEncFwdMaxK4, S3D_ERR_UNSUPPORTED, "encoder: %d feature channels (max %d), %d input channels (max %d)",
              e.CO, kEncFwdMaxCO, e.C, kEncFwdMaxK4);
    int maxpix = 0;
    for (int p = 0; p < 3; ++p) maxpix = std::max(maxpix, e.g.h[p] * e.g.w[p]);
    hipLaunchKernelGGL(k_enc_fwd, dim3(cdiv(maxpix, 64), 3), dim3(64), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ InstanceNorm2d (no affine, eps 1e-5) + tanh(0.5 x)
// forward: y = tanh(0.5 * (x - mean_c) * rstd_c) per (plane, channel) over the plane's pixels; one block per (plane, channel)
struct EncNormArgs { const float* x[3]; float* y[3]; float* mr; const float* dy[3]; float* dx[3]; int hw[3]; int CO; };
// (1024 threads per (plane, channel): the 36 blocks walk their 16 384 pixels in 16 trips instead of 64)
constexpr int kEncNormThreads = 1024;
__global__ __launch_bounds__(kEncNormThreads) void k_enc_norm_fwd(EncNormArgs a) {
    __shared__ double red[2][kEncNormThreads];
    const int c = blockIdx.x, p = blockIdx.y, n = a.hw[p], CO = a.CO;
    double s = 0, ss = 0;
    for (int i = threadIdx.x; i < n; i += kEncNormThreads) { const double v = a.x[p][size_t(i) * CO + c]; s += v; ss += v * v; }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = ss;
    __syncthreads();
    for (int o = kEncNormThreads / 2; o > 0; o >>= 1) {
        if (int(threadIdx.x) < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    const double mean = red[0][0] / n;
    double var = red[1][0] / n - mean * mean; if (var < 0) var = 0;
    const float m = float(mean), rstd = float(1.0 / sqrt(var + 1e-5));
    if (threadIdx.x == 0) { a.mr[(p * CO + c) * 2] = m; a.mr[(p * CO + c) * 2 + 1] = rstd; }
    for (int i = threadIdx.x; i < n; i += kEncNormThreads) a.y[p][size_t(i) * CO + c] = tanhf(0.5f * (a.x[p][size_t(i) * CO + c] - m) * rstd);
}
// backward: dn = dy * 0.5 * (1 - y^2) ; dx = rstd * (dn - mean(dn) - n_hat * mean(dn * n_hat))
// Two launches over (pixel chunk, plane) blocks of 16 pixels x CO channels (round 5; it was one block per (plane, channel)
// reading every twelfth float, 71 us): a thread keeps its channel, consecutive threads read consecutive floats; the chunk sums
// go out as doubles and every block of the second launch adds the chunks of its plane in the same order.
constexpr int kEncNormChunks = 64, kEncNormLanes = 16;
struct EncNormBwdArgs { EncNormArgs n; double* part; };          // part [3][kEncNormChunks][CO][2]
__device__ __forceinline__ float enc_norm_dn(const EncNormArgs& a, int p, size_t e, float m, float rstd, float& nh) {
    const float y = a.y[p][e];
    nh = (a.x[p][e] - m) * rstd;
    return a.dy[p][e] * 0.5f * (1.f - y * y);
}
__global__ __launch_bounds__(kEncNormLanes * 12) void k_enc_norm_bwd_part(EncNormBwdArgs b) {
    __shared__ double red[2][kEncNormLanes * 12];
    const EncNormArgs& a = b.n;
    const int chunk = blockIdx.x, p = blockIdx.y, n = a.hw[p], CO = a.CO;
    const int c = threadIdx.x % CO, lane = threadIdx.x / CO;
    const int p0 = int((long long)n * chunk / kEncNormChunks), p1 = int((long long)n * (chunk + 1) / kEncNormChunks);
    const float m = a.mr[(p * CO + c) * 2], rstd = a.mr[(p * CO + c) * 2 + 1];
    double s1 = 0, s2 = 0;
    for (int i = p0 + lane; i < p1; i += kEncNormLanes) {
        float nh;
        const float dn = enc_norm_dn(a, p, size_t(i) * CO + c, m, rstd, nh);
        s1 += dn; s2 += double(dn) * nh;
    }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (lane == 0) {
        for (int l = 1; l < kEncNormLanes; ++l) { s1 += red[0][l * CO + c]; s2 += red[1][l * CO + c]; }
        double* o = b.part + ((size_t(p) * kEncNormChunks + chunk) * CO + c) * 2;
        o[0] = s1; o[1] = s2;
    }
}
__global__ __launch_bounds__(kEncNormLanes * 12) void k_enc_norm_bwd_apply(EncNormBwdArgs b) {
    __shared__ float kk[2][12];
    const EncNormArgs& a = b.n;
    const int chunk = blockIdx.x, p = blockIdx.y, n = a.hw[p], CO = a.CO;
    const int c = threadIdx.x % CO, lane = threadIdx.x / CO;
    if (int(threadIdx.x) < 2 * CO) {
        const int cc = threadIdx.x >> 1, which = threadIdx.x & 1;
        double s = 0;
        for (int k = 0; k < kEncNormChunks; ++k) s += b.part[((size_t(p) * kEncNormChunks + k) * CO + cc) * 2 + which];
        kk[which][cc] = float(s / n);
    }
    __syncthreads();
    const int p0 = int((long long)n * chunk / kEncNormChunks), p1 = int((long long)n * (chunk + 1) / kEncNormChunks);
    const float m = a.mr[(p * CO + c) * 2], rstd = a.mr[(p * CO + c) * 2 + 1], k1 = kk[0][c], k2 = kk[1][c];
    for (int i = p0 + lane; i < p1; i += kEncNormLanes) {
        float nh;
        const float dn = enc_norm_dn(a, p, size_t(i) * CO + c, m, rstd, nh);
        a.dx[p][size_t(i) * CO + c] = rstd * (dn - k1 - nh * k2);
    }
}
size_t enc_norm_bwd_ws_bytes(int CO) { return size_t(3) * kEncNormChunks * CO * 2 * sizeof(double); }
int launch_enc_norm(bool backward, float* const x[3], float* const y[3], float* mr, float* const dy[3], float* const dx[3],
                    const Geo& g, int CO, void* ws, hipStream_t st) {
    EncNormArgs a;
    for (int p = 0; p < 3; ++p) {
        a.x[p] = x[p]; a.y[p] = y[p]; a.dy[p] = dy ? dy[p] : nullptr; a.dx[p] = dx ? dx[p] : nullptr; a.hw[p] = g.h[p] * g.w[p];
    }
    a.mr = mr; a.CO = CO;
    if (backward) {
        S3D_CHECK(CO <= 12 && ws, S3D_ERR_UNSUPPORTED, "encoder norm backward: %d channels (max 12) / no workspace", CO);
        EncNormBwdArgs b{a, static_cast<double*>(ws)};
        hipLaunchKernelGGL(k_enc_norm_bwd_part, dim3(kEncNormChunks, 3), dim3(kEncNormLanes * CO), 0, st, b);
        hipLaunchKernelGGL(k_enc_norm_bwd_apply, dim3(kEncNormChunks, 3), dim3(kEncNormLanes * CO), 0, st, b);
    } else hipLaunchKernelGGL(k_enc_norm_fwd, dim3(CO, 3), dim3(kEncNormThreads), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ encoder weight gradient
// dWp[p][co][j] = inv_len[p] * sum_{a,b} dpre[p][a][b][co] * P[p][2a+ka-1][2b+kb-1][ks][ci],  j = (ka*4+kb)*4C + ks*C + ci
// Stage 1: block = (pixel chunk, plane), thread = j (16*4*C = 256 for C = 4), CO accumulators per thread.
// Stage 2 (k_enc_wgrad_reduce): add chunks in order, un-permute the taps, add the three planes, write Conv3d layouts.
constexpr int kEncChunks = 256, kEncMaxCO = 12;
struct EncWgArgs { const float* P[3]; const float* dpre[3]; float* part; float* bpart; int h[3], w[3]; int C, CO, J; };
// (round 5: 256 chunks instead of 64 — three blocks per CU instead of 0.75 — and 64 pixels of dpre per barrier pair with 32
// projection loads issued before their first use; the launch was a chain of 256 x {barrier, 12 floats into LDS, barrier, one
// load, 12 multiply-adds} per block, 190-250 us.  The sums are formed in the same order as before within a chunk.)
constexpr int kEncStage = 64, kEncHalf = 32;
__global__ __launch_bounds__(256) void k_enc_wgrad(EncWgArgs a) {
    __shared__ float sd[kEncStage][kEncMaxCO];
    const int chunk = blockIdx.x, p = blockIdx.y;
    const int h = a.h[p], w = a.w[p], K = 4 * a.C, CO = a.CO;
    const int npix = h * w, p0 = int((long long)npix * chunk / kEncChunks), p1 = int((long long)npix * (chunk + 1) / kEncChunks);
    const int j = blockIdx.z * 256 + threadIdx.x;
    const bool active = j < a.J;
    const int k = j % K, kb = (j / K) % 4, ka = j / (4 * K);
    float acc[kEncMaxCO];
#pragma unroll
    for (int c = 0; c < kEncMaxCO; ++c) acc[c] = 0.f;
    float bsum = 0.f;                                        // threads < CO: sum of dpre over the chunk (bias gradient), in pixel order
    for (int i0 = p0; i0 < p1; i0 += kEncStage) {
        const int ns = min(kEncStage, p1 - i0);
        __syncthreads();
        for (int e = threadIdx.x; e < kEncStage * kEncMaxCO; e += 256) {      // rows past ns and columns past CO are zero
            const int t = e / kEncMaxCO, c = e - t * kEncMaxCO;
            sd[t][c] = t < ns && c < CO ? a.dpre[p][size_t(i0 + t) * CO + c] : 0.f;
        }
        __syncthreads();
        if (int(threadIdx.x) < CO) {
#pragma unroll
            for (int t = 0; t < kEncStage; ++t) bsum += sd[t][threadIdx.x];
        }
        if (!active) continue;
        int x = i0 / w, y = i0 - x * w;
        for (int t0 = 0; t0 < ns; t0 += kEncHalf) {
            float v[kEncHalf];
#pragma unroll
            for (int t = 0; t < kEncHalf; ++t) {
                const int X = 2 * x + ka - 1, Y = 2 * y + kb - 1;
                const bool in = t0 + t < ns && X >= 0 && X < 2 * h && Y >= 0 && Y < 2 * w;
                const float ld = a.P[p][(size_t(min(max(X, 0), 2 * h - 1)) * 2 * w + min(max(Y, 0), 2 * w - 1)) * K + k];
                v[t] = in ? ld : 0.f;
                if (++y == w) { y = 0; ++x; }
            }
#pragma unroll
            for (int t = 0; t < kEncHalf; ++t) {
#pragma unroll
                for (int c = 0; c < kEncMaxCO; ++c) acc[c] = fmaf(sd[t0 + t][c], v[t], acc[c]);
            }
        }
    }
    if (blockIdx.z == 0 && int(threadIdx.x) < CO) a.bpart[(size_t(p) * kEncChunks + chunk) * CO + threadIdx.x] = bsum;
    if (!active) return;
    float* o = a.part + ((size_t(p) * kEncChunks + chunk) * CO) * a.J + j;
#pragma unroll
    for (int c = 0; c < kEncMaxCO; ++c) if (c < CO) o[size_t(c) * a.J] = acc[c];
}
struct EncWgRedArgs { const double* psum; float* dwgeo; float* dwtex; float* dbgeo; float* dbtex; float inv_len[3]; int hw[3]; int geo, tex, C, J; };
// Stage 2a: psum[p][co][j] = sum over the chunks, eight lanes per element (32 chunks each, in order; the eight sums in lane order);
// consecutive elements are consecutive in memory.  The biases ride along as j == J.
constexpr int kEncRedLanes = 8;
struct EncWgSumArgs { const float* part; const float* bpart; double* psum; int CO, J; };
__global__ __launch_bounds__(256) void k_enc_wgrad_sum(EncWgSumArgs a) {
    const int gt = blockIdx.x * blockDim.x + threadIdx.x, lane = gt & (kEncRedLanes - 1);
    const int e = gt / kEncRedLanes, n = 3 * a.CO * (a.J + 1);
    double s = 0;
    if (e < n) {
        const int j = e % (a.J + 1), co = (e / (a.J + 1)) % a.CO, p = e / ((a.J + 1) * a.CO);
        const int k0 = lane * (kEncChunks / kEncRedLanes);
        if (j < a.J) for (int k = k0; k < k0 + kEncChunks / kEncRedLanes; ++k) s += a.part[((size_t(p) * kEncChunks + k) * a.CO + co) * a.J + j];
        else for (int k = k0; k < k0 + kEncChunks / kEncRedLanes; ++k) s += a.bpart[(size_t(p) * kEncChunks + k) * a.CO + co];
    }
    const int base = threadIdx.x & 63 & ~(kEncRedLanes - 1);
    double sum = __shfl(s, base);
    for (int l = 1; l < kEncRedLanes; ++l) sum += __shfl(s, base + l);
    if (lane == 0 && e < n) a.psum[e] = sum;
}
// Stage 2b: un-permute the taps, add the three planes, write the Conv3d layouts
__global__ void k_enc_wgrad_reduce(EncWgRedArgs a) {
    const int CO = a.geo + a.tex, C = a.C;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // over CO * C * 64 Conv3d weight elements, then CO biases
    if (idx < CO * C * 64) {
        const int kz = idx % 4, ky = (idx / 4) % 4, kx = (idx / 16) % 4, ci = (idx / 64) % C, co = idx / (64 * C);
        double tot = 0;
        for (int p = 0; p < 3; ++p) {
            const int ka = p == 2 ? ky : kx, kb = p == 0 ? ky : kz, ks = p == 0 ? kz : (p == 1 ? ky : kx);
            const int j = (ka * 4 + kb) * 4 * C + ks * C + ci;
            tot += a.psum[(size_t(p) * CO + co) * (a.J + 1) + j] * a.inv_len[p];
        }
        if (co < a.geo) { if (ci == 0) a.dwgeo[size_t(co) * 64 + (kx * 4 + ky) * 4 + kz] = float(tot); }
        else a.dwtex[(size_t(co - a.geo) * C + ci) * 64 + (kx * 4 + ky) * 4 + kz] = float(tot);
    } else if (idx < CO * C * 64 + CO) {
        const int co = idx - CO * C * 64;
        double tot = 0;
        for (int p = 0; p < 3; ++p) tot += a.psum[(size_t(p) * CO + co) * (a.J + 1) + a.J];
        if (co < a.geo) a.dbgeo[co] = float(tot); else a.dbtex[co - a.geo] = float(tot);
    }
}
// partial sums [3][chunks][CO][J], bias partials [3][chunks][CO], then (8-byte aligned) the chunk totals [3][CO][J + 1] as doubles
size_t enc_wgrad_ws_floats(int C, int CO) { return size_t(3) * kEncChunks * CO * (16 * 4 * C + 1) + 2 + size_t(2) * 3 * CO * (16 * 4 * C + 1); }
int launch_enc_wgrad(const EncDesc& e, float* const dpre[3], int geo, int tex, float* ws, float* dwgeo, float* dbgeo, float* dwtex,
                     float* dbtex, hipStream_t st) {
    S3D_CHECK(e.CO <= kEncMaxCO, S3D_ERR_UNSUPPORTED, "encoder: %d feature channels (max %d)", e.CO, kEncMaxCO);
    EncWgArgs a;
    a.part = ws; a.C = e.C; a.CO = e.CO; a.J = 16 * 4 * e.C;
    a.bpart = ws + size_t(3) * kEncChunks * e.CO * a.J;
    for (int p = 0; p < 3; ++p) { a.P[p] = e.P[p]; a.dpre[p] = dpre[p]; a.h[p] = e.g.h[p]; a.w[p] = e.g.w[p]; }
    hipLaunchKernelGGL(k_enc_wgrad, dim3(kEncChunks, 3, cdiv(a.J, 256)), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    EncWgSumArgs q;
    q.part = ws; q.bpart = a.bpart; q.CO = e.CO; q.J = a.J;
    const size_t pfl = size_t(3) * kEncChunks * e.CO * (a.J + 1);
    q.psum = reinterpret_cast<double*>(ws + pfl + (pfl & 1));
    hipLaunchKernelGGL(k_enc_wgrad_sum, dim3(cdiv(3 * e.CO * (a.J + 1) * kEncRedLanes, 256)), dim3(256), 0, st, q);
    S3D_HIP(hipGetLastError());
    EncWgRedArgs r;
    r.psum = q.psum; r.dwgeo = dwgeo; r.dwtex = dwtex; r.dbgeo = dbgeo; r.dbtex = dbtex; r.geo = geo; r.tex = tex; r.C = e.C; r.J = a.J;
    for (int p = 0; p < 3; ++p) { r.inv_len[p] = e.inv_len[p]; r.hw[p] = e.g.h[p] * e.g.w[p]; }
    hipLaunchKernelGGL(k_enc_wgrad_reduce, dim3(cdiv(e.CO * e.C * 64 + e.CO, 256)), dim3(256), 0, st, r);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ channel slices between the 12-channel latent and the 32-padded conv input
// feat NHWC [hw][CT] -> x [hw][32] = channels [c0, c0+cin) zero-padded (forward) ; and the reverse scatter (backward)
// (three planes per launch — blockIdx.y — since round 5: each of these was three 5-us launches on a plane-block chain)
struct Slice3Args { const float* in[3]; float* out[3]; long long hw[3]; int CT, c0, cin; };
__global__ void k_slice_pad_nhwc(Slice3Args a) {
    const int p = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.hw[p] * 32) return;
    const int c = int(i & 31);
    a.out[p][i] = c < a.cin ? a.in[p][(i >> 5) * a.CT + a.c0 + c] : 0.f;
}
__global__ void k_unslice_nhwc(Slice3Args a) {              // in: dx [hw][32], out: dfeat [hw][CT]
    const int p = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.hw[p] * a.cin) return;
    const int c = int(i % a.cin);
    const long long pix = i / a.cin;
    a.out[p][pix * a.CT + a.c0 + c] = a.in[p][pix * 32 + c];
}
int launch_slice_pad_nhwc3(const float* const in[3], float* const out[3], const size_t hw[3], int CT, int c0, int cin, hipStream_t st) {
    Slice3Args a; a.CT = CT; a.c0 = c0; a.cin = cin;
    long long mx = 0;
    for (int p = 0; p < 3; ++p) { a.in[p] = in[p]; a.out[p] = out[p]; a.hw[p] = (long long)hw[p]; mx = std::max(mx, a.hw[p]); }
    hipLaunchKernelGGL(k_slice_pad_nhwc, dim3(cdivll(mx * 32, 256), 3), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}
int launch_unslice_nhwc3(const float* const dx[3], float* const dfeat[3], const size_t hw[3], int CT, int c0, int cin, hipStream_t st) {
    Slice3Args a; a.CT = CT; a.c0 = c0; a.cin = cin;
    long long mx = 0;
    for (int p = 0; p < 3; ++p) { a.in[p] = dx[p]; a.out[p] = dfeat[p]; a.hw[p] = (long long)hw[p]; mx = std::max(mx, a.hw[p]); }
    hipLaunchKernelGGL(k_unslice_nhwc, dim3(cdivll(mx * cin, 256), 3), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// InstanceNorm statistics {mean, rstd} per channel from the per-chunk double partials of k_chan_partials (s3d_decoder.hip)
// eight lanes per channel: lane l adds chunks l, l + 8, ... in order, the eight sums meet in a fixed xor tree (one thread per channel
// walked 64 dependent loads: 12 us, six times per iteration on the plane blocks' chains)
struct Mr3Args { const double* part[3]; double count[3]; float* mr; int nchunks, C; float eps; };      // mr [3][C][2]
__global__ __launch_bounds__(256) void k_mr_from_partials(Mr3Args a) {
    const int p = blockIdx.y, C = a.C;
    const int gt = blockIdx.x * blockDim.x + threadIdx.x, c = gt >> 3, l = gt & 7;
    const double* part = a.part[p];
    double S = 0, SS = 0;
    if (c < C)
#pragma unroll 4
        for (int k = l; k < a.nchunks; k += 8) { S += part[(size_t(k) * C + c) * 2]; SS += part[(size_t(k) * C + c) * 2 + 1]; }
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) { S += __shfl_xor(S, d, 8); SS += __shfl_xor(SS, d, 8); }
    if (c >= C || l != 0) return;
    const double m = S / a.count[p];
    double var = SS / a.count[p] - m * m; if (var < 0) var = 0;
    float* mr = a.mr + size_t(p) * C * 2;
    mr[c * 2] = float(m); mr[c * 2 + 1] = float(1.0 / sqrt(var + double(a.eps)));
}
int launch_mr_from_partials3(double* const part[3], int nchunks, int C, const size_t hw[3], float eps, float* mr, hipStream_t st) {
    Mr3Args a; a.mr = mr; a.nchunks = nchunks; a.C = C; a.eps = eps;
    for (int p = 0; p < 3; ++p) { a.part[p] = part[p]; a.count[p] = double(hw[p]); }
    hipLaunchKernelGGL(k_mr_from_partials, dim3(cdiv(C * 8, 256), 3), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ point gather and its backward (F.grid_sample, bilinear, border, align_corners=False)
// sample_feature_plane2D (networks.py:182-190): plane [h][w][C] indexed by (u -> rows, v -> columns) of the
// aabb-normalised point; h_net = sum over the three planes.
__device__ __forceinline__ void gs_coord(float xn, int size, int& i0, int& i1, float& w0, float& w1) {
    float f = ((xn + 1.f) * float(size) - 1.f) * 0.5f;         // unnormalize, align_corners=False
    f = fminf(fmaxf(f, 0.f), float(size - 1));                 // border padding: clip the coordinate
    const float fl = floorf(f);
    i0 = int(fl); i1 = i0 + 1;
    w1 = f - fl; w0 = 1.f - w1;
    if (i1 > size - 1) { i1 = size - 1; w1 = 0.f; }            // out-of-range corner contributes nothing
}
__global__ void k_gather(GatherArgs a) {
    const int cq = a.C / 4;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.Np * cq * a.nnets) return;
    const int q = int(idx % cq);
    const long long r = idx / cq;
    const long long n = r % a.Np;
    const int net = int(r / a.Np);
    if (n >= a.N) { reinterpret_cast<float4*>(a.X[net])[n * cq + q] = make_float4(0, 0, 0, 0); return; }
    float xn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) xn[k] = 2.f * (a.pts[n * 3 + k] - a.amin[k]) * a.ainv[k] - 1.f;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int ku = p == 2 ? 1 : 0, kv = p == 0 ? 1 : 2;      // xy: (x,y) ; xz: (x,z) ; yz: (y,z)
        int r0, r1, c0, c1; float wr0, wr1, wc0, wc1;
        gs_coord(xn[ku], a.ph[p], r0, r1, wr0, wr1);
        gs_coord(xn[kv], a.pw[p], c0, c1, wc0, wc1);
        const int w = a.pw[p];
        const float4* f = reinterpret_cast<const float4*>(a.feat[net][p]);
        const float4 v00 = f[(size_t(r0) * w + c0) * cq + q], v01 = f[(size_t(r0) * w + c1) * cq + q];
        const float4 v10 = f[(size_t(r1) * w + c0) * cq + q], v11 = f[(size_t(r1) * w + c1) * cq + q];
        const float w00 = wr0 * wc0, w01 = wr0 * wc1, w10 = wr1 * wc0, w11 = wr1 * wc1;
        acc.x += w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
        acc.y += w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
        acc.z += w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
        acc.w += w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
    }
    reinterpret_cast<float4*>(a.X[net])[n * cq + q] = acc;
}
int launch_gather(const PointSet& ps, const float* const feat[2][3], const int ph[3], const int pw[3], int C, int nnets,
                  float* const X[2], hipStream_t st) {
    GatherArgs a; memset(&a, 0, sizeof a);
    a.pts = ps.pts; a.N = ps.N; a.Np = ps.Np; a.C = C; a.nnets = nnets;
    for (int k = 0; k < 3; ++k) { a.amin[k] = ps.aabb[k]; a.ainv[k] = 1.f / (ps.aabb[3 + k] - ps.aabb[k]); a.ph[k] = ph[k]; a.pw[k] = pw[k]; }
    for (int n = 0; n < nnets; ++n) { a.X[n] = X[n]; for (int p = 0; p < 3; ++p) a.feat[n][p] = feat[n][p]; }
    const long long tot = ps.Np * (C / 4) * nnets;
    if (!tot) return 0;
    hipLaunchKernelGGL(k_gather, dim3(cdivll(tot, 256)), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}
// Backward of the gather WITHOUT float atomics: the points are sorted by the plane cell of their top-left bilinear
// corner (stable radix sort: equal cells keep point order), every plane pixel then gathers, in that fixed order, from the
// four cells whose corner set contains it.  Bit-repeatable, and several times faster than 100 M fp32 atomics per step.
__global__ void k_cell_keys(GatherArgs a, unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * a.Np) return;
    const int p = int(idx / a.Np);
    const long long n = idx % a.Np;
    unsigned key = unsigned(a.ph[p]) * unsigned(a.pw[p]);          // padding rows sort behind every real cell of their plane
    if (n < a.N) {
        const int ku = p == 2 ? 1 : 0, kv = p == 0 ? 1 : 2;
        const float xu = 2.f * (a.pts[n * 3 + ku] - a.amin[ku]) * a.ainv[ku] - 1.f;
        const float xv = 2.f * (a.pts[n * 3 + kv] - a.amin[kv]) * a.ainv[kv] - 1.f;
        int r0, r1, c0, c1; float w0, w1, v0, v1;
        gs_coord(xu, a.ph[p], r0, r1, w0, w1);
        gs_coord(xv, a.pw[p], c0, c1, v0, v1);
        key = unsigned(r0) * unsigned(a.pw[p]) + unsigned(c0);
    }
    // one key space for the three planes (round 5: ONE stable radix sort instead of three — 21 small dependent launches were 0.33 ms
    // between the MLPs' backward and the plane blocks'): plane p's cells and its padding key follow plane p-1's
    unsigned base = 0;
    for (int k = 0; k < p; ++k) base += unsigned(a.ph[k]) * unsigned(a.pw[k]) + 1u;
    keys[idx] = base + key; vals[idx] = unsigned(n);
}
__global__ void k_bin_starts(const unsigned* __restrict__ keys_sorted, long long Np, int ncell, unsigned* __restrict__ start) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;          // start[cell] = first sorted position with key >= cell
    if (cell > ncell) return;
    long long lo = 0, hi = Np;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (keys_sorted[mid] < unsigned(cell)) lo = mid + 1; else hi = mid; }
    start[cell] = unsigned(lo);
}
__global__ void k_scatter_sorted(ScatterSortedArgs s) {
    const GatherArgs& a = s.g;
    const int cq = a.C / 4;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= s.begin[3] * a.nnets) return;
    const int net = int(idx / s.begin[3]);
    long long r = idx % s.begin[3];
    const int p = r >= s.begin[2] ? 2 : (r >= s.begin[1] ? 1 : 0);
    r -= s.begin[p];
    const int q = int(r % cq);
    const int col = int((r / cq) % a.pw[p]), row = int(r / ((long long)cq * a.pw[p]));
    const int ku = p == 2 ? 1 : 0, kv = p == 0 ? 1 : 2;
    const int h = a.ph[p], w = a.pw[p];
    float4 acc = make_float4(0, 0, 0, 0);
    const float4* dX = reinterpret_cast<const float4*>(a.dX[net]);
    // cells whose corners can be (row, col): top-left at (row-1|row, col-1|col), visited in a fixed order
#pragma unroll
    for (int dr = 1; dr >= 0; --dr)
#pragma unroll
        for (int dc = 1; dc >= 0; --dc) {
            const int cr = row - dr, cc = col - dc;
            if (cr < 0 || cc < 0) continue;
            const unsigned cell = unsigned(cr) * unsigned(w) + unsigned(cc);
            for (unsigned k = s.start[p][cell]; k < s.start[p][cell + 1]; ++k) {
                const unsigned n = s.order[p][k];
                const float xu = 2.f * (a.pts[size_t(n) * 3 + ku] - a.amin[ku]) * a.ainv[ku] - 1.f;
                const float xv = 2.f * (a.pts[size_t(n) * 3 + kv] - a.amin[kv]) * a.ainv[kv] - 1.f;
                int r0, r1, c0, c1; float wr0, wr1, wc0, wc1;
                gs_coord(xu, h, r0, r1, wr0, wr1);
                gs_coord(xv, w, c0, c1, wc0, wc1);
                // weight of corner (row, col) for this point (a clipped corner coincides with its neighbour and has weight 0)
                const float wr = (r0 == row ? wr0 : 0.f) + (r1 == row && r1 != r0 ? wr1 : 0.f);
                const float wc = (c0 == col ? wc0 : 0.f) + (c1 == col && c1 != c0 ? wc1 : 0.f);
                const float wgt = wr * wc;
                if (wgt == 0.f) continue;
                const float4 g = dX[size_t(n) * cq + q];
                acc.x = fmaf(wgt, g.x, acc.x); acc.y = fmaf(wgt, g.y, acc.y); acc.z = fmaf(wgt, g.z, acc.z); acc.w = fmaf(wgt, g.w, acc.w);
            }
        }
    reinterpret_cast<float4*>(a.dfeat[net][p])[(size_t(row) * w + col) * cq + q] = acc;
}
size_t scatter_ws_bytes(long long Np, const int ph[3], const int pw[3]) {
    size_t cub = 0;
    unsigned* d = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cub, d, d, d, d, int(3 * Np), 0, 32, nullptr);
    size_t cells = 8;
    for (int p = 0; p < 3; ++p) cells += size_t(ph[p]) * pw[p] + 2;
    return ((cub + 255) & ~size_t(255)) + (size_t(12) * Np + cells) * sizeof(unsigned) + 1024;
}
// Two halves: the point ORDER (cell keys, one radix sort over the three planes' keys, bin starts) depends on the points alone;
// the scatter proper follows when dX is final.  (Enqueued early — on the second chain's stream beside the MLPs' forward pass, or
// on the idle weight-gradient stream from the first moment of the iteration — the order costs more than it returns: DESIGN.md §9.)
int launch_scatter_prepare(const PointSet& ps, const int ph[3], const int pw[3], int C, int nnets, void* ws, ScatterPlan& plan, hipStream_t st) {
    ScatterSortedArgs& s = plan.args; memset(&s, 0, sizeof s);
    GatherArgs& a = s.g;
    a.pts = ps.pts; a.N = ps.N; a.Np = ps.Np; a.C = C; a.nnets = nnets;
    for (int k = 0; k < 3; ++k) { a.amin[k] = ps.aabb[k]; a.ainv[k] = 1.f / (ps.aabb[3 + k] - ps.aabb[k]); a.ph[k] = ph[k]; a.pw[k] = pw[k]; }
    const long long Np = ps.Np;
    plan.ready = false;
    if (!Np) return 0;
    size_t cub = 0;
    unsigned* d0 = nullptr;
    S3D_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, cub, d0, d0, d0, d0, int(3 * Np), 0, 32, st));
    char* base = static_cast<char*>(ws);
    void* cub_tmp = base; base += (cub + 255) & ~size_t(255);
    unsigned* keys = reinterpret_cast<unsigned*>(base); unsigned* vals = keys + 3 * Np;
    unsigned* keys_s = vals + 3 * Np; unsigned* vals_s = keys_s + 3 * Np;
    unsigned* start = vals_s + 3 * Np;
    hipLaunchKernelGGL(k_cell_keys, dim3(cdivll(3 * Np, 256)), dim3(256), 0, st, a, keys, vals);
    S3D_HIP(hipGetLastError());
    long long nbins = 0;
    for (int p = 0; p < 3; ++p) nbins += (long long)ph[p] * pw[p] + 1;      // every plane's cells + its padding key
    S3D_CHECK(nbins < (1ll << 31) && 3 * Np < (1ll << 31), S3D_ERR_UNSUPPORTED, "scatter: %lld cells / %lld points", nbins, Np);
    int bits = 1; while ((1ll << bits) < nbins) ++bits;
    size_t tb = cub;
    S3D_HIP(hipcub::DeviceRadixSort::SortPairs(cub_tmp, tb, keys, keys_s, vals, vals_s, int(3 * Np), 0, bits, st));
    hipLaunchKernelGGL(k_bin_starts, dim3(cdivll(nbins + 1, 256)), dim3(256), 0, st, keys_s, 3 * Np, int(nbins), start);
    S3D_HIP(hipGetLastError());
    s.begin[0] = 0;
    long long b0 = 0;
    for (int p = 0; p < 3; ++p) {
        const int ncell = ph[p] * pw[p];
        s.order[p] = vals_s; s.start[p] = start + b0;          // (positions in the one sorted array)
        b0 += ncell + 1;
        s.begin[p + 1] = s.begin[p] + (long long)ncell * (C / 4);
    }
    plan.ready = true;
    return 0;
}
int launch_scatter_apply(ScatterPlan& plan, float* const dfeat[2][3], const float* const dX[2], hipStream_t st) {
    if (!plan.ready) return 0;
    ScatterSortedArgs& s = plan.args;
    for (int n = 0; n < s.g.nnets; ++n) { s.g.dX[n] = dX[n]; for (int p = 0; p < 3; ++p) s.g.dfeat[n][p] = dfeat[n][p]; }
    hipLaunchKernelGGL(k_scatter_sorted, dim3(cdivll(s.begin[3] * s.g.nnets, 256)), dim3(256), 0, st, s);
    S3D_HIP(hipGetLastError());
    return 0;
}
int launch_scatter(const PointSet& ps, float* const dfeat[2][3], const int ph[3], const int pw[3], int C, int nnets,
                   const float* const dX[2], void* ws, hipStream_t st) {
    ScatterPlan plan;
    S3D_TRY(launch_scatter_prepare(ps, ph, pw, C, nnets, ws, plan, st));
    return launch_scatter_apply(plan, dfeat, dX, st);
}

// ------------------------------------------------------------------ small element-wise pieces of the MLP
// dpre[n][c] = dact[n][coff + c] * (act[n][c] > 0)   (dact row stride dstride), and the column sums of dpre (the bias
// gradient) in the same pass: per-chunk partials here, added in chunk order by k_colsum_fin
// (round 5: 1024 threads per chunk with four rows of loads in flight per thread — 16 waves per CU instead of 4 walking 64
// dependent trips: the launch moved 200 MB in 115 us —; the column-sum pass below is the same kernel without the mask)
constexpr int kColChunks = 256, kColThreads = 1024;
template <bool MASK>
__global__ __launch_bounds__(kColThreads) void k_relu_bwd(const float* __restrict__ dact, int dstride, int coff, const float* __restrict__ act,
                                                          float* __restrict__ dpre, long long rows, int C, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm_rb[];      // [pl][C]
    const int cq = C / 4, pl = blockDim.x / cq;
    const int q = threadIdx.x % cq, l = threadIdx.x / cq;
    const long long r0 = rows * blockIdx.x / kColChunks, r1 = rows * (blockIdx.x + 1) / kColChunks;
    float4 s = make_float4(0, 0, 0, 0);
    for (long long n0 = r0 + l; n0 < r1; n0 += 4 * pl) {
        float4 a[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long n = min(n0 + (long long)u * pl, r1 - 1);
            if (MASK) a[u] = reinterpret_cast<const float4*>(act)[n * cq + q];
            d[u] = *reinterpret_cast<const float4*>(dact + n * dstride + coff + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long n = n0 + (long long)u * pl;
            if (n >= r1) break;
            float4 o = d[u];
            if (MASK) { o.x = a[u].x > 0.f ? o.x : 0.f; o.y = a[u].y > 0.f ? o.y : 0.f; o.z = a[u].z > 0.f ? o.z : 0.f; o.w = a[u].w > 0.f ? o.w : 0.f; }
            if (MASK) reinterpret_cast<float4*>(dpre)[n * cq + q] = o;
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
    }
    reinterpret_cast<float4*>(sm_rb)[l * cq + q] = s;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float t = 0.f;
        for (int ll = 0; ll < pl; ++ll) t += sm_rb[ll * C + c];
        part[size_t(blockIdx.x) * C + c] = t;
    }
}
__global__ __launch_bounds__(1024) void k_colsum_fin(const float* __restrict__ part, float* __restrict__ out, int C, float* __restrict__ out2 = nullptr) {
    // block = 64 columns x 16 chunk lanes (lane j adds chunks j, j+16, ... in double; the sixteen sums meet through LDS in lane
    // order): one thread per column walked 256 dependent loads in four blocks, 11 us, 28 times per iteration
    __shared__ double red[16][64];
    const int l = threadIdx.x & 63, kl = threadIdx.x >> 6, c = blockIdx.x * 64 + l;
    double s = 0;
    if (c < C)
#pragma unroll
        for (int k = 0; k < kColChunks / 16; ++k) s += part[size_t(kl + 16 * k) * C + c];
    red[kl][l] = s;
    __syncthreads();
    if (kl == 0 && c < C) {
        double t = red[0][l];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][l];
        const float v = float(t);
        out[c] = v;
        if (out2) out2[c] = v;                            // (two parameters with the same gradient: the out convolution's and the shortcut's bias)
    }
}
int launch_relu_bwd(const float* dact, int dstride, int coff, const float* act, float* dpre, long long rows, int C, float* ws,
                    float* colsum, hipStream_t st) {
    if (!rows) return 0;
    S3D_CHECK(C % 4 == 0 && C <= 1024 && dstride % 4 == 0 && coff % 4 == 0, S3D_ERR_INVALID, "relu_bwd: C=%d stride=%d off=%d", C, dstride, coff);
    const int cq = C / 4, pl = std::max(1, kColThreads / cq);
    hipLaunchKernelGGL(k_relu_bwd<true>, dim3(kColChunks), dim3(cq * pl), size_t(pl) * C * sizeof(float), st, dact, dstride, coff, act, dpre, rows, C, ws);
    S3D_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_colsum_fin, dim3(cdiv(C, 64)), dim3(1024), 0, st, ws, colsum, C, (float*)nullptr);
    S3D_HIP(hipGetLastError());
    return 0;
}
// out[n][c] = a[n][c] + b[n][coff + c]   (b row stride bstride)
__global__ void k_add_slice(const float* __restrict__ a, const float* __restrict__ b, int bstride, int coff, float* __restrict__ out,
                            long long rows, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    out[i] = a[i] + b[(i / C) * bstride + coff + int(i % C)];
}
int launch_add_slice(const float* a, const float* b, int bstride, int coff, float* out, long long rows, int C, hipStream_t st) {
    if (!rows) return 0;
    hipLaunchKernelGGL(k_add_slice, dim3(cdivll(rows * C, 256)), dim3(256), 0, st, a, b, bstride, coff, out, rows, C);
    S3D_HIP(hipGetLastError());
    return 0;
}
// column sums of a [rows][C] matrix (bias gradients): two-stage, fixed order
size_t colsum_ws_floats(int C) { return size_t(kColChunks) * C; }
int launch_colsum(const float* x, long long rows, int C, float* ws, float* out, hipStream_t st, float* out2) {
    S3D_CHECK(C % 4 == 0 && C <= 1024, S3D_ERR_INVALID, "colsum: C=%d", C);
    const int cq = C / 4, pl = std::max(1, kColThreads / cq);
    hipLaunchKernelGGL(k_relu_bwd<false>, dim3(kColChunks), dim3(cq * pl), size_t(pl) * C * sizeof(float), st, x, C, 0, (const float*)nullptr,
                       (float*)nullptr, rows, C, ws);
    S3D_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_colsum_fin, dim3(cdiv(C, 64)), dim3(1024), 0, st, ws, out, C, out2);
    S3D_HIP(hipGetLastError());
    return 0;
}

// the same for three planes in one launch pair (blockIdx.y; ws: 3 * colsum_ws_floats(C)) — the plane blocks' bias gradients were
// twelve launch pairs per iteration on the weight-gradient stream, which is what the iteration waits for
struct ColSum3Args { const float* x[3]; long long rows[3]; float* out[3]; float* out2[3]; float* part; int C; };
__global__ __launch_bounds__(kColThreads) void k_colsum3_part(ColSum3Args a) {
    extern __shared__ __attribute__((aligned(16))) float sm_rb[];      // [pl][C]
    const int p = blockIdx.y, C = a.C, cq = C / 4, pl = blockDim.x / cq;
    const int q = threadIdx.x % cq, l = threadIdx.x / cq;
    const long long rows = a.rows[p], r0 = rows * blockIdx.x / kColChunks, r1 = rows * (blockIdx.x + 1) / kColChunks;
    const float* x = a.x[p];
    float4 s = make_float4(0, 0, 0, 0);
    for (long long n0 = r0 + l; n0 < r1; n0 += 4 * pl) {
        float4 d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = *reinterpret_cast<const float4*>(x + min(n0 + (long long)u * pl, r1 - 1) * C + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (n0 + (long long)u * pl >= r1) break;
            s.x += d[u].x; s.y += d[u].y; s.z += d[u].z; s.w += d[u].w;
        }
    }
    reinterpret_cast<float4*>(sm_rb)[l * cq + q] = s;
    __syncthreads();
    float* part = a.part + size_t(p) * kColChunks * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float t = 0.f;
        for (int ll = 0; ll < pl; ++ll) t += sm_rb[ll * C + c];
        part[size_t(blockIdx.x) * C + c] = t;
    }
}
__global__ __launch_bounds__(1024) void k_colsum3_fin(ColSum3Args a) {
    __shared__ double red[16][64];
    const int p = blockIdx.y, C = a.C;
    const int l = threadIdx.x & 63, kl = threadIdx.x >> 6, c = blockIdx.x * 64 + l;
    const float* part = a.part + size_t(p) * kColChunks * C;
    double s = 0;
    if (c < C)
#pragma unroll
        for (int k = 0; k < kColChunks / 16; ++k) s += part[size_t(kl + 16 * k) * C + c];
    red[kl][l] = s;
    __syncthreads();
    if (kl == 0 && c < C) {
        double t = red[0][l];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][l];
        const float v = float(t);
        a.out[p][c] = v;
        if (a.out2[p]) a.out2[p][c] = v;
    }
}
int launch_colsum3(const float* const x[3], const size_t rows[3], int C, float* ws, float* const out[3], float* const out2[3], hipStream_t st) {
    S3D_CHECK(C % 4 == 0 && C <= 1024, S3D_ERR_INVALID, "colsum: C=%d", C);
    ColSum3Args a; a.part = ws; a.C = C;
    for (int p = 0; p < 3; ++p) { a.x[p] = x[p]; a.rows[p] = (long long)rows[p]; a.out[p] = out[p]; a.out2[p] = out2 ? out2[p] : nullptr; }
    const int cq = C / 4, pl = std::max(1, kColThreads / cq);
    hipLaunchKernelGGL(k_colsum3_part, dim3(kColChunks, 3), dim3(cq * pl), size_t(pl) * C * sizeof(float), st, a);
    S3D_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_colsum3_fin, dim3(cdiv(C, 64), 3), dim3(1024), 0, st, a);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ the last MLP layer (hid -> 1 or 3 outputs)
// forward: out[n][o] = b[o] + sum_i W[o][i] * h[n][i]  (+ sigmoid), written into pred[n][ooff + o] (row stride 4-ish)
// one wave per point: lanes stride the hidden units, shuffle-reduce
__global__ __launch_bounds__(256) void k_last_fwd(const float* __restrict__ h, const float* __restrict__ W, const float* __restrict__ b,
                                                  int I, int O, int sigm, float* __restrict__ pred, int pstride, int ooff, long long N) {
    const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    for (int o = 0; o < O; ++o) {
        float s = 0.f;
        for (int i = lane; i < I; i += 64) s = fmaf(W[size_t(o) * I + i], h[n * I + i], s);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) {
            s += b[o];
            pred[n * pstride + ooff + o] = sigm ? 1.f / (1.f + expf(-s)) : s;
        }
    }
}
int launch_last_fwd(const float* h, const float* W, const float* b, int I, int O, int sigm, float* pred, int pstride, int ooff,
                    long long N, hipStream_t st) {
    if (!N) return 0;
    hipLaunchKernelGGL(k_last_fwd, dim3(cdivll(N, 4)), dim3(256), 0, st, h, W, b, I, O, sigm, pred, pstride, ooff, N);
    S3D_HIP(hipGetLastError());
    return 0;
}
// backward: dh[n][i] = sum_o dout[n][o] * W[o][i] ; dW[o][i] = sum_n dout[n][o] * h[n][i] ; db[o] = sum_n dout[n][o]
// dout [Np][4-wide rows: column ooff + o]; rows >= N carry zero.  dW/db: per-chunk partials then ordered sum.
constexpr int kLastChunks = 512;
__global__ __launch_bounds__(256) void k_last_bwd(const float* __restrict__ dout, int dstride, int ooff, const float* __restrict__ W,
                                                  const float* __restrict__ h, int I, int O, long long Np, float* __restrict__ dh,
                                                  float* __restrict__ part) {
    const long long r0 = Np * blockIdx.x / kLastChunks, r1 = Np * (blockIdx.x + 1) / kLastChunks;
    for (int i = threadIdx.x; i < I; i += 256) {
        float w[3] = {0, 0, 0}, acc[3] = {0, 0, 0};
        for (int o = 0; o < O; ++o) w[o] = W[size_t(o) * I + i];
#pragma unroll 8
        for (long long n = r0; n < r1; ++n) {                // (unrolled: eight rows of loads in flight per thread)
            const float hv = h[n * I + i];
            float d = 0.f;
            for (int o = 0; o < O; ++o) { const float g = dout[n * dstride + ooff + o]; d = fmaf(g, w[o], d); acc[o] = fmaf(g, hv, acc[o]); }
            dh[n * I + i] = d;
        }
        for (int o = 0; o < O; ++o) part[(size_t(blockIdx.x) * O + o) * (I + 1) + i] = acc[o];
    }
    if (int(threadIdx.x) < O) {
        float s = 0.f;
        for (long long n = r0; n < r1; ++n) s += dout[n * dstride + ooff + threadIdx.x];
        part[(size_t(blockIdx.x) * O + threadIdx.x) * (I + 1) + I] = s;
    }
}
// 32 lanes per output: lane l adds chunks l, l + 32, ... (double), the 32 sums meet in a fixed xor tree.  (One thread per output
// walked 512 dependent loads: 38 us for the geometry net's 257 outputs — one block and one wave of it — at the head of each net's
// backward chain.)
__global__ __launch_bounds__(256) void k_last_bwd_fin(const float* __restrict__ part, int I, int O, float* __restrict__ dW, float* __restrict__ db) {
    const int idx = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    const bool ok = idx < O * (I + 1);
    const int i = ok ? idx % (I + 1) : 0, o = ok ? idx / (I + 1) : 0;
    double s = 0;
    if (ok)
#pragma unroll 4
        for (int k = l; k < kLastChunks; k += 32) s += part[(size_t(k) * O + o) * (I + 1) + i];
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) s += __shfl_xor(s, d, 32);
    if (ok && l == 0) { if (i < I) dW[size_t(o) * I + i] = float(s); else db[o] = float(s); }
}
size_t last_bwd_ws_floats(int I, int O) { return size_t(kLastChunks) * O * (I + 1); }
int launch_last_bwd(const float* dout, int dstride, int ooff, const float* W, const float* h, int I, int O, long long Np, float* dh,
                    float* ws, float* dW, float* db, hipStream_t st) {
    S3D_CHECK(O >= 1 && O <= 3, S3D_ERR_UNSUPPORTED, "last layer: %d outputs", O);
    hipLaunchKernelGGL(k_last_bwd, dim3(kLastChunks), dim3(256), 0, st, dout, dstride, ooff, W, h, I, O, Np, dh, ws);
    S3D_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_last_bwd_fin, dim3(cdiv(O * (I + 1), 8)), dim3(256), 0, st, ws, I, O, dW, db);
    S3D_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ losses (src/encoding/model.py:186-237) and d loss / d pre-activation
// sdf: weightedl1  mean(|p - g| * (1 + 0.5 sign(g) sign(g - p))) or l1 ; tex (on rows with |g_sdf| < band): l1 | l2 | huber(0.1), * tex_weight
// pred [N][1+TC] holds sdf and SIGMOID-ed texture; dout gets d(sdf_loss + tex_loss)/d(pre-activation) (through the sigmoid)
struct LossArgs {
    const float* pred; const float* sdf; const float* tex; float* dout; float* part; float* losses;
    long long N, Np; int TC, sdf_mode, tex_mode; float band, tex_weight;
};
constexpr int kLossChunks = 64;
__device__ __forceinline__ float tex_term(int mode, float d) {
    if (mode == 0) return fabsf(d);
    if (mode == 1) return d * d;
    return fabsf(d) < 0.1f ? 0.5f * d * d : 0.1f * (fabsf(d) - 0.05f);
}
__device__ __forceinline__ float tex_dterm(int mode, float d) {
    if (mode == 0) return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (mode == 1) return 2.f * d;
    return fabsf(d) < 0.1f ? d : (d > 0.f ? 0.1f : -0.1f);
}
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }
__global__ __launch_bounds__(256) void k_loss_part(LossArgs a) {
    __shared__ double red[3][256];
    const long long r0 = a.N * blockIdx.x / kLossChunks, r1 = a.N * (blockIdx.x + 1) / kLossChunks;
    double s_sdf = 0, s_tex = 0, cnt = 0;
    const int S = 1 + a.TC;
    for (long long n = r0 + threadIdx.x; n < r1; n += 256) {
        const float g = a.sdf[n], p = a.pred[n * S];
        const float wgt = a.sdf_mode == 1 ? 1.f + 0.5f * sgn(g) * sgn(g - p) : 1.f;
        s_sdf += fabsf(p - g) * wgt;
        if (a.TC && fabsf(g) < a.band) {
            cnt += 1;
            for (int k = 0; k < a.TC; ++k) s_tex += tex_term(a.tex_mode, a.pred[n * S + 1 + k] - a.tex[n * a.TC + k]);
        }
    }
    red[0][threadIdx.x] = s_sdf; red[1][threadIdx.x] = s_tex; red[2][threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (int(threadIdx.x) < o) for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) a.part[blockIdx.x * 3 + threadIdx.x] = float(red[threadIdx.x][0]);
}
__global__ void k_loss_fin(LossArgs a) {
    double s_sdf = 0, s_tex = 0, cnt = 0;
    for (int k = 0; k < kLossChunks; ++k) { s_sdf += a.part[k * 3]; s_tex += a.part[k * 3 + 1]; cnt += a.part[k * 3 + 2]; }
    a.losses[0] = float(s_sdf / double(a.N));
    a.losses[1] = a.TC ? float(s_tex / (cnt * a.TC) * a.tex_weight) : 0.f;       // 0/0 = nan like F.l1_loss on an empty selection
    a.losses[2] = float(cnt);
}
__global__ void k_loss_grad(LossArgs a) {
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.Np) return;
    const int S = 1 + a.TC;
    if (n >= a.N) { for (int k = 0; k < S; ++k) a.dout[n * S + k] = 0.f; return; }
    const float g = a.sdf[n], p = a.pred[n * S];
    const float wgt = a.sdf_mode == 1 ? 1.f + 0.5f * sgn(g) * sgn(g - p) : 1.f;
    a.dout[n * S] = sgn(p - g) * wgt / float(a.N);
    const bool in = a.TC && fabsf(g) < a.band;
    const float scale = in ? a.tex_weight / (a.losses[2] * float(a.TC)) : 0.f;
    for (int k = 0; k < a.TC; ++k) {
        const float s = a.pred[n * S + 1 + k];
        a.dout[n * S + 1 + k] = in ? tex_dterm(a.tex_mode, s - a.tex[n * a.TC + k]) * scale * s * (1.f - s) : 0.f;
    }
}
int launch_ae_loss(const float* pred, const float* sdf, const float* tex, long long N, long long Np, int TC, int sdf_mode, int tex_mode,
                   float band, float tex_weight, float* ws, float* losses, float* dout, hipStream_t st) {
    LossArgs a{pred, sdf, tex, dout, ws, losses, N, Np, TC, sdf_mode, tex_mode, band, tex_weight};
    hipLaunchKernelGGL(k_loss_part, dim3(kLossChunks), dim3(256), 0, st, a);
    S3D_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_loss_fin, dim3(1), dim3(1), 0, st, a);
    S3D_HIP(hipGetLastError());
    if (dout) {
        hipLaunchKernelGGL(k_loss_grad, dim3(cdivll(Np, 256)), dim3(256), 0, st, a);
        S3D_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace s3d
