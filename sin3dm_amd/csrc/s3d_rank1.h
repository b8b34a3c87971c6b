// s3d_rank1.h — device bodies of the rollout's small dependent stages (k_means_finalize in s3d_kernels.hip, k_rank1 in
// s3d_conv.hip):
//
//   gn_act --(tile partial sums)--> means  --(six mean vectors)--> rank-1 tables --> convolution EPILOGUE
//
// Each stage is its own launch; stream order is the hand-off.  (Round 3 also ran the two stages as producer blocks inside the
// 3x3 convolution launch and as one fused launch — bit-identical, measured slower, removed in round 4: DESIGN.md §12.1,
// profiles/r03_rank1_inline.txt, commits c0b16ff / 6acaefc hold the code.)
#pragma once
#include "s3d_common.h"

namespace s3d {

typedef float r1_f32x16 __attribute__((ext_vector_type(16)));
typedef float r1_f32x4 __attribute__((ext_vector_type(4)));
typedef const r1_f32x4 __attribute__((address_space(1)))* r1_gf4ptr;

constexpr int kR1Chunk = 128, kR1Ld = kR1Chunk + 4;
constexpr int kR1LdsFloats = (34 + 2 * 32) * kR1Ld;       // A tile (34 rows) + two B tiles: 51 744 bytes

// ------------------------------------------------------------------ stage A: th.mean over one axis of the activated planes
// (src/diffusion/unet_triplane.py:38-46): add the tile partials in index order and divide by the axis length.
// item = (position, channel quad) of vector v = blockIdx.y of sample blockIdx.z, one thread each.  At batch 1 the launch is a
// pure latency chain (dispatch + one cold round of loads + a store: 4.9 us whatever the index arithmetic costs — a 64-bit
// version with a search over the six vectors measured the same); at batch 8 it is bound by the 19 MB of partials per sample it
// reads (14.3 -> 13.4 us per call with all sixteen loads of an item in flight in one thread instead of four lanes of four).
__device__ __forceinline__ void means_finalize_thread(const MeanFinArgs& a, int v, int b, int gthread) {
    // one thread per item, all of its partials requested together; the sums are formed as the four-lane form formed them —
    // s_k = p_k + p_{k+4} + p_{k+8} + ... in order, then (s0 + s1) + (s2 + s3) — so the bits are those of every earlier round
    const int i = gthread;
    const int len = a.len[v], nt = a.nt[v], cq = a.cq;
    if (i >= len * cq) return;
    const int pos = i / cq, q = i - pos * cq;
    const float4* src = reinterpret_cast<const float4*>(a.src[v]) + (size_t(b) * nt * len + pos) * cq + q;
    const size_t tstride = size_t(len) * cq;
    float4 s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = make_float4(0, 0, 0, 0);
    for (int t0 = 0; t0 < nt; t0 += 16) {
        float4 u[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int t = t0 + k;
            u[k] = src[size_t(t < nt ? t : 0) * tstride];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (t0 + k < nt) { float4& d = s[k & 3]; d.x += u[k].x; d.y += u[k].y; d.z += u[k].z; d.w += u[k].w; }
    }
    float4 r;
    r.x = (s[0].x + s[1].x) + (s[2].x + s[3].x); r.y = (s[0].y + s[1].y) + (s[2].y + s[3].y);
    r.z = (s[0].z + s[1].z) + (s[2].z + s[3].z); r.w = (s[0].w + s[1].w) + (s[2].w + s[3].w);
    const float inv = a.inv[v];
    r.x *= inv; r.y *= inv; r.z *= inv; r.w *= inv;
    reinterpret_cast<float4*>(a.dst[v])[(size_t(b) * len + pos) * cq + q] = r;
}

// ------------------------------------------------------------------ stage B: rank-1 rollout tables (skinny GEMMs)
// out[b][pos][n] = sum_{tap, c} W[tap][n][c] * v[b][pos + tap - 1][c]   (zero outside [0, L)), n = variant*cout + co.
// M = L is only ~128 rows, so the work is split along K instead.  One block = 32 positions x 32 columns.  K = 3 taps x C
// is walked in stages of (tap, <=128-channel chunk): whole 512-byte rows of the vector (with its +-1 halo, loaded once per
// chunk) and of the weights are staged in LDS with coalesced loads, register-prefetched one stage ahead; inside a stage the
// four waves each contract a quarter of the chunk, and their partial accumulators are added through LDS in wave order.
// ROLL3 (CONV_1x3_ROLL, the forward rollout tables): the four edge variants of a table entry are sums over subsets of the
// three taps o of the SUMMED-OUT axis (interior o0+o1+o2, first o1+o2, last o0+o1, single o1), so only the three per-tap
// products U_o are contracted — weights [tap][n][cin] with n = (co / 8) * 24 + o * 8 + co % 8, a block owns 32 positions x
// 8 output channels = 24 weight rows — and the variants are formed while the four waves' partials are added.
// lds: kR1LdsFloats floats.
// chunk0 / nch: the 128-channel K chunks this block contracts (0 / 0 = all).  A table may be cut into K SLICES, one per
// chunk, written by different blocks to out + slice * slice_stride and added by the consuming convolution's epilogue in slice
// order: the stage chain of a block is then 3 stages whatever the channel count (256-channel layers: 14 -> 9.5 us).
// NS (1 / 2) samples per block, b .. b + bcount - 1: the weight tile of a stage is staged once and contracted with each
// sample's vector tile (batch > 1: a stage's bytes per sample drop from 23 to 14.5 KB); every sample's sums are formed in the
// same order as with NS = 1.  lds: kR1LdsFloats + (NS - 1) * 34 * kR1Ld floats.
struct R1Block { const float* vin; const float* wgt; float* out; int L, cin, cout4, n_tiles_n, b, mtile, ntile; int chunk0 = 0, nch = 0; int bcount = 1; };

template <bool ROLL3, int NS = 1>
__device__ __forceinline__ void rank1_block(const R1Block& J, float* lds) {
    constexpr int kATile = 34 * kR1Ld;
    float* sA = lds;
    float* sB0 = lds + NS * kATile;
    const int L = J.L, cin = J.cin, cout4 = J.cout4, mtile = J.mtile, ntile = J.ntile, b = J.b;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 31, half = lane >> 5;
    const int nchunks = (cin + kR1Chunk - 1) / kR1Chunk;   // the last chunk is narrower when cin % 128 != 0
    constexpr int q4 = kR1Chunk / 4;                        // float4 slots per staged row
    const float* vb = J.vin + size_t(b) * L * cin;
    const size_t tapStride = ROLL3 ? size_t(J.n_tiles_n) * 24 * cin : size_t(cout4) * cin;
    const r1_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    r1_f32x16 acc[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    // staging items: A has 34 rows (positions mtile*32-1 .. +32), B 32 rows (columns ntile*32 ..); <= 5 + 4 float4 each
    constexpr int NA = (34 * (kR1Chunk / 4) + 255) / 256, NB = (32 * (kR1Chunk / 4) + 255) / 256;
    r1_f32x4 ra[NS][NA], rb[NB];
    auto loadA = [&](int chunk) {
        const int c0 = chunk * kR1Chunk, wq = (min(cin - c0, kR1Chunk)) / 4;
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int idx = it * 256 + tid, row = idx / q4, q = idx - row * q4;
            const int pos = mtile * 32 - 1 + row;
            const bool ok = row < 34 && pos >= 0 && pos < L && q < wq && n < J.bcount;
            ra[n][it] = ((r1_gf4ptr)(uintptr_t)(vb + (size_t(ok ? n : 0) * L + (ok ? pos : 0)) * cin + c0 + (ok ? q : 0) * 4))[0];
            if (!ok) ra[n][it] = zero4;
        }
    };
    auto loadB = [&](int tap, int chunk) {
        const int c0 = chunk * kR1Chunk, wq = (min(cin - c0, kR1Chunk)) / 4;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = it * 256 + tid, row = idx / q4, q = idx - row * q4;
            const int n = ROLL3 ? ntile * 24 + row : ntile * 32 + row;
            const bool ok = (ROLL3 ? row < 24 && ntile * 8 + (row & 7) < cout4 : row < 32 && n < cout4) && q < wq;
            rb[it] = ((r1_gf4ptr)(uintptr_t)(J.wgt + tap * tapStride + size_t(ok ? n : 0) * cin + c0 + (ok ? q : 0) * 4))[0];
            if (!ok) rb[it] = zero4;
        }
    };
    auto storeA = [&]() {
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int idx = it * 256 + tid, row = idx / q4, q = idx - row * q4;
            if (row < 34) *reinterpret_cast<r1_f32x4*>(sA + n * kATile + row * kR1Ld + q * 4) = ra[n][it];
        }
    };
    auto storeB = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = it * 256 + tid, row = idx / q4, q = idx - row * q4;
            if (row < 32) *reinterpret_cast<r1_f32x4*>(sB0 + buf * (32 * kR1Ld) + row * kR1Ld + q * 4) = rb[it];
        }
    };

    const int ch0 = J.nch > 0 ? J.chunk0 : 0;
    const int nstages = (J.nch > 0 ? J.nch : nchunks) * 3; // stage s -> chunk = ch0 + s / 3, tap = s % 3
    loadB(0, ch0);
    loadA(ch0);
    storeA(); storeB(0);
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        const int chunk = ch0 + s / 3, tap = s - (s / 3) * 3;
        const int ns = s + 1 < nstages ? s + 1 : s;
        const int nchunk = ch0 + ns / 3, ntap = ns - (ns / 3) * 3;
        loadB(ntap, nchunk);
        const bool newA = nchunk != chunk;
        if (newA) loadA(nchunk);
        __builtin_amdgcn_sched_barrier(0);
        // this wave's quarter of the chunk: k8 steps [wid*cw/32, (wid+1)*cw/32)
        const float* Ar = sA + (i + tap) * kR1Ld + half * 4;
        const float* Br = sB0 + (s & 1) * (32 * kR1Ld) + i * kR1Ld + half * 4;
        const int k8n = min(cin - chunk * kR1Chunk, kR1Chunk) / 32;
        for (int k8 = 0; k8 < k8n; ++k8) {
            const int c = (wid * k8n + k8) * 8;
            const r1_f32x4 b4 = *reinterpret_cast<const r1_f32x4*>(Br + c);
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                const r1_f32x4 a4 = *reinterpret_cast<const r1_f32x4*>(Ar + n * kATile + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[n], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (newA) __syncthreads();                          // everyone is done reading the old A tile
        storeB((s + 1) & 1);
        if (newA) storeA();
        __syncthreads();
    }
    // add the four waves' partials (reuse the B tiles as [NS][4][16][64] floats = 16 KB per sample)
    float* red0 = sB0;
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) red0[n * 4096 + (wid * 16 + r) * 64 + lane] = acc[n][r];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        if (n >= J.bcount) break;
        const float* red = red0 + n * 4096;
        const int bs = b + n;
        if (ROLL3) {
            // MFMA row p sits in register (p&3) + 4*(p>>3) of lane half (p>>2)&1
            auto usum = [&](int p, int c8, int o) {
                const int r = (p & 3) + 4 * (p >> 3), l = ((p >> 2) & 1) * 32 + o * 8 + c8;
                return red[(0 * 16 + r) * 64 + l] + red[(1 * 16 + r) * 64 + l] + red[(2 * 16 + r) * 64 + l] + red[(3 * 16 + r) * 64 + l];
            };
            const int p = tid >> 3, c8 = tid & 7;
            const float u0 = usum(p, c8, 0), u1 = usum(p, c8, 1), u2 = usum(p, c8, 2);
            const int row = mtile * 32 + p, co = ntile * 8 + c8;
            if (row < L && co < cout4) {
                float* o4 = J.out + (size_t(bs) * L + row) * 4 * cout4 + co;      // [pos][variant][cout]
                o4[0] = (u0 + u1) + u2; o4[cout4] = u1 + u2; o4[2 * cout4] = u0 + u1; o4[3 * cout4] = u1;
            }
            continue;
        }
        for (int it = tid; it < 1024; it += 256) {
            const int r = it >> 6, l = it & 63;
            const float v = red[(0 * 16 + r) * 64 + l] + red[(1 * 16 + r) * 64 + l] + red[(2 * 16 + r) * 64 + l] + red[(3 * 16 + r) * 64 + l];
            const int row = mtile * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = ntile * 32 + (l & 31);
            if (row < L && col < cout4) J.out[(size_t(bs) * L + row) * cout4 + col] = v;
        }
    }
}

}  // namespace s3d
