// s3d_common.h — internal declarations shared by the HIP translation units of libsin3dm_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/sin3dm_hip.h"

namespace s3d {

// ------------------------------------------------------------------ errors
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* get_error();

#define S3D_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            s3d::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return S3D_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

#define S3D_CHECK(cond, code, ...)                                                                 \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            s3d::set_error(__VA_ARGS__);                                                           \
            return (code);                                                                         \
        }                                                                                          \
    } while (0)

#define S3D_TRY(expr)                                                                              \
    do {                                                                                           \
        int rc__ = (expr);                                                                         \
        if (rc__ != 0) return rc__;                                                                \
    } while (0)

// Opt a kernel in to more than 64 KB of dynamic LDS.  The attribute belongs to the (kernel, device) pair: it is set once per
// device the calling code ever launches on (`opted` = the call site's bit mask of devices done); a failure is reported and not
// remembered, so a transient error does not poison later launches.
inline int ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& opted) {
    int dev = 0;
    S3D_HIP(hipGetDevice(&dev));
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (bit && (opted.load(std::memory_order_acquire) & bit)) return 0;
    S3D_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (bit) opted.fetch_or(bit, std::memory_order_release);
    return 0;
}

// ------------------------------------------------------------------ device memory helpers
struct DevBuf {                       // owning device allocation (grow-only)
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

struct Arena {                        // bump allocator over one DevBuf; two-pass (measure, then run)
    DevBuf buf;
    size_t off = 0, high = 0;
    bool measuring = false;
    void reset() { off = 0; }
    template <class T>
    T* alloc(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        size_t o = off;
        off += bytes;
        if (off > high) high = off;
        if (measuring) return reinterpret_cast<T*>(uintptr_t(256));   // never dereferenced
        return reinterpret_cast<T*>(static_cast<char*>(buf.p) + o);
    }
};

int upload(DevBuf& dst, const void* host, size_t bytes);

// ------------------------------------------------------------------ process-wide options (s3d_set_option / environment)
// Every kernel-form switch of the library.  Value = what s3d_set_option last stored; before that the environment variable
// S3D_<NAME> (read once, at the first query of that option); before that the default (kOptUnset for the choices the library
// makes by launch size).  Queries are a table read: cheap enough for every launch.
enum Opt { OPT_WINO = 0, OPT_WINO24W, OPT_VCAT, OPT_WGRAD_WINO, OPT_RANK1_SLICES, OPT_RANK1_BATCH, OPT_CONV_IMPL, OPT_CONV1X1_T,
           OPT_GN_FUSED, OPT_BWD_SIDE, OPT_GNB_FUSED, OPT_WINO24G, OPT_EDGE_SIGNAL, OPT_COUNT };
constexpr int kOptUnset = -1;
int opt(Opt o);                       // kOptUnset when neither set nor in the environment
inline bool opt_on(Opt o) { return opt(o) != 0; }      // switches that default to on: anything but an explicit 0
// compute units of the CURRENT device (cached per device index: one process may drive several)
int device_cus();

// ------------------------------------------------------------------ LDS-DMA (device code only)
#if defined(__HIPCC__)
typedef int i32x4 __attribute__((ext_vector_type(4)));
// One 1-KB LDS-DMA piece: lane l's 16 bytes at buffer offset voff + soff land at LDS byte lds_addr + 16 l (lds_addr wave-uniform);
// an out-of-range offset (bit 31) lands as ZEROS, exec-masked lanes write nothing (tools/glds_probe.hip).
// M0 (the destination base) belongs to the compiler: saved and restored inside the statement.  Nothing here is visible to hipcc's
// s_waitcnt bookkeeping: the caller retires the piece with its own s_waitcnt vmcnt + a barrier before any lane reads it.
__device__ __forceinline__ void lds_dma16(unsigned lds_addr, unsigned voff, i32x4 rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// buffer descriptor of `bytes` bytes at p (raw dword addressing, out-of-range reads return 0)
__device__ __forceinline__ i32x4 lds_dma_desc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    return i32x4{int(unsigned(a)), int(unsigned(a >> 32) & 0xFFFF), int(bytes), 0x00020000};
}
#endif

// ------------------------------------------------------------------ plane geometry
// The three planes of a triplane: xy[H,W], xz[H,D], yz[W,D]  (src/utils/triplane_util.py:20-25)
struct Geo {
    int h[3], w[3];
    static Geo from_hwd(int H, int W, int D) { return Geo{{H, H, W}, {W, D, D}}; }
    Geo half() const { return Geo{{h[0] / 2, h[1] / 2, h[2] / 2}, {w[0] / 2, w[1] / 2, w[2] / 2}}; }
    Geo twice() const { return Geo{{h[0] * 2, h[1] * 2, h[2] * 2}, {w[0] * 2, w[1] * 2, w[2] * 2}}; }
    bool operator==(const Geo& o) const {
        for (int i = 0; i < 3; ++i) if (h[i] != o.h[i] || w[i] != o.w[i]) return false;
        return true;
    }
    size_t pixels() const { return size_t(h[0]) * w[0] + size_t(h[1]) * w[1] + size_t(h[2]) * w[2]; }
};

// GroupNorm partial sums left by a producer (layout: see "GroupNorm statistics" below)
struct GnPartials { double* p; int maxparts; int nparts[3]; int nsub; };
// A triplane activation in the library's layout: per plane NHWC fp32 [B][h][w][C].
struct Tri {
    float* p[3] = {nullptr, nullptr, nullptr};
    int C = 0;
    Geo g{};
    float* gn = nullptr;      // GroupNorm {mean, rstd} [B][3][32][2] of this tensor when its producer already reduced them
    GnPartials part{nullptr, 0, {0, 0, 0}, 0};   // ... or only the partial sums (a tensor that is normalised later as part of a concat)
};

// ------------------------------------------------------------------ MFMA conv (s3d_conv.hip)
// One implicit-GEMM convolution problem: M = B*h*w pixels, N = cout, K = taps*cin.
struct ConvJob {
    const float* in;      // [B][h][w][cin] NHWC
    const float* wgt;     // packed [taps][cout][cin]
    const float* wgt_wino;// 3x3 only: Winograd-transformed weights in MFMA fragment order (s3d_wino.hip) or null
    const float* wgt_wino24s;// 3x3 only: the mixed F(2x4,3x3) image of k_conv_wino24s / k_conv_wino24w (s3d_wino24.hip) or null (null: the F(2x2) kernels are used)
    const float* wgt_r1f; // CONV_1x3_ROLL only: the rank-1 weights in MFMA fragment order (k_rank1b, pack_rank1_frag) or null (null: k_rank1 on `wgt`)
    const float* bias;    // [cout] or null
    const float* bbias;   // [B][bbias_stride] per-sample bias (h + emb path) or null
    const float* rrow;    // [B][h][4][cout] rank-1 rollout term indexed by pixel row, variant by column; or null
    const float* rcol;    // [B][w][4][cout] rank-1 rollout term indexed by pixel column, variant by row; or null
    const float* res;     // [B][h][w][cout] residual or null
    int res_up;           // direct kernels only: res is [B][h/2][w/2][cout] and is upsampled 2x bilinearly (align_corners=False) on the way in
    float* out;           // [B][h][w][cout]
    int h, w;
    int bbias_stride;
    double* gn_part;      // GroupNorm partial sums of `out` for this plane (see GnPartials) or null
    int tiles_x, tiles_per_img, n_tiles_n;   // filled by the launcher
    int block_begin;                          // first block id of this job
};
constexpr int kMaxConvJobs = 8;
// Training: what the input-gradient convolution in front of a GroupNorm(+FiLM)+SiLU backward needs to form that backward's two
// per-channel sums (sum dz, sum dz * xh) in its epilogue (k_conv_wino24s_gnb).  Index = plane = job of the launch.
struct GnbArgs {
    const float* x[3];                       // the norm's input [B][h][w][C]
    const float* rowadd[3]; const float* coladd[3];   // broadcast gradients of the rollout means, [B][h][C] / [B][w][C], or null
    float rowscale[3], colscale[3];
    const float* gamma[3]; const float* beta[3];
    const float* mr;                         // {mean, rstd} [B][3][groups]
    const float* film; int film_stride;      // FiLM row(s): scale at [0, C), shift at [C, 2C); or null
    int groups;
};
struct ConvArgs {
    ConvJob job[kMaxConvJobs];
    int njobs;
    int B, cin, cout;
    int gn_sg, gn_nsub, gn_maxparts;          // GroupNorm partial layout when job[].gn_part is set
    int relu;                                 // direct kernels only: ReLU in the epilogue (the decoder MLPs as 1x1 convs)
    int xcd_swizzle;                          // Winograd kernel: logical block order contiguous per XCD (set by the launcher)
    // rank-1 rollout tables cut into two K slices (0 / 1: one table).  CONV_1x3_ROLL launches: a block contracts half of the
    // 128-channel chunks and writes slice s at out + s * B * L * 4 * cout.  k_conv_wino24s: rrow / rcol are read as the sum of the
    // two tables (slice stride B * h * 4 * cout / B * w * 4 * cout floats)
    int r1_slices;
    const GnbArgs* gnb;                       // HOST pointer, 3x3 mixed-Winograd launches only: see GnbArgs (null: plain convolution)
};
// CONV_1x3_ROLL: the forward rollout tables — args.cout = the convolution's cout, out [B][pos][4 variants][cout] (k_rank1<true>)
enum ConvKind { CONV_3x3 = 0, CONV_1x1 = 1, CONV_1x3_VEC = 2, CONV_5x5 = 3, CONV_1x3_ROLL = 4 };
// Enqueue all jobs (same B/cin/cout/kind) as ONE launch.  cin must be a multiple of 32.
int launch_conv(ConvKind kind, ConvArgs& a, hipStream_t st);
// Debug/triangulation path: a plain one-thread-per-output direct convolution (no MFMA, no LDS).
// Selected with S3D_CONV_IMPL=naive; never the default.
int launch_conv_naive(ConvKind kind, ConvArgs& a, hipStream_t st);
bool conv_use_naive();
// every launcher notes the kernel it dispatched (thread-local; read back by s3d_unet::timed_conv for s3d_unet_profile_kernel)
void conv_note_kernel(const char* name);
const char* conv_last_kernel();
// MFMA flops a launch really issues / its direct-convolution flop count (1 for the direct kernels, 4/9 for Winograd F(2x2,3x3))
double conv_exec_fraction(ConvKind kind, const ConvArgs& a);

// Packed TriplaneConv parameters: offsets (in floats) into a staging image that is uploaded as one buffer.
struct ConvW {
    size_t dense[3] = {0, 0, 0};      // [taps][cout][cin_own]
    size_t bias[3] = {0, 0, 0};
    size_t rrow[3] = {0, 0, 0};       // rank-1 weights for the row-varying mean vector  [3 taps][ceil(cout/8)*24][C], row (co/8)*24 + o*8 + co%8
    size_t rcol[3] = {0, 0, 0};       // rank-1 weights for the column-varying mean vector
    size_t rrow_f[3] = {0, 0, 0};     // the same two in the fragment order of k_rank1b (0 = not packed: cin not a multiple of 128, or the training tier's image)
    size_t rcol_f[3] = {0, 0, 0};
    size_t wino[3] = {0, 0, 0};       // 3x3: G g G^T in fragment order (0 = not packed)
    size_t wino24s[3] = {0, 0, 0};    // 3x3: G2 g G4^T (mixed F(2x4,3x3)) in the fragment order of k_conv_wino24s / k_conv_wino24w
    int cin = 0, cout = 0, k = 0;
    bool rollout = false;
    // training handles: the per-step device repack rewrites only the images of the kernel forms selected when the plan was built
    // (s3d_pack.hip); true = of this 3x3 layer's images only `wino24s` follows the parameters, `dense` / `wino` hold load-time
    // values.  A launch that would read one of those fails loudly (Fwd::conv) instead of computing with stale weights.
    bool only24_current = false;
};
size_t push(std::vector<float>& stage, const float* src, size_t n);
// Fragment-order image of the rank-1 weights (k_rank1b, s3d_conv.hip): float index of W[tap][o][co][c] — group g = co / 32 and
// K quarter w = (c % 128) / 32 select a stream of steps (chunk, tap, k8), three 1-KB fragments (o) per step, lane = half * 32 +
// co % 32 holds channels c0 + half * 4 + {0..3} of its output channel.  cin % 128 == 0; ceil(cout / 32) * cin * 288 floats.
__host__ __device__ inline size_t rank1_frag_index(int cin, int tap, int o, int co, int c) {
    const int g = co >> 5, i = co & 31, chunk = c >> 7, cc = c & 127, w = cc >> 5, k8 = (cc >> 3) & 3, half = (cc >> 2) & 1, e = cc & 3;
    const size_t step = size_t(g * 4 + w) * (cin / 128 * 12) + chunk * 12 + tap * 4 + k8;
    return ((step * 3 + o) * 64 + half * 32 + i) * 4 + e;
}
inline size_t rank1_frag_floats(int cin, int cout) { return size_t((cout + 31) / 32) * cin * 288; }
void pack_tconv_raw(std::vector<float>& stage, const float* const W[3], const float* const bias[3], int cin, int cout,
                    int k, bool roll, ConvW& cw);

// ------------------------------------------------------------------ small kernels (s3d_kernels.hip)
// NCHW <-> NHWC plane repacks used only by the leaf-operator test entry points.
int launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int h, int w, hipStream_t st);
int launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int h, int w, hipStream_t st);

// in_conv: composed NCHW x [B,Cin,H+D,W+D] -> three NHWC planes through a 1x1 conv (Cin small).
// wT: [3][Cin][Cout] (transposed), bias [3][Cout]
// part != null (only when in_conv_gn_parts() says the shape supports it, with exactly those part counts): the kernel
// also writes the GroupNorm partials of its output
struct GnPartials;
bool in_conv_gn_parts(const Geo& g, int Cin, int Cout, int nparts[3]);
int launch_in_conv(const float* x, int B, int Cin, int H, int W, int D, const float* wT, const float* bias,
                   int Cout, Tri& out, hipStream_t st, const GnPartials* part = nullptr);

// GroupNorm statistics.  Stage 1: producers emit partial {sum, sumsq} in double, indexed
//   p[(((b*3 + plane) * nsub + sub) * maxparts + part) * 2 + {0,1}],  sub = channel / sg
// where a "part" is a pixel chunk (launch_gn_partials) or one wave's 32-pixel tile of a convolution epilogue and a
// "subgroup" is sg consecutive channels of one group.  Stage 2 (launch_gn_finalize) adds the parts in index order
// and writes {mean, rstd} per (b, plane, group): deterministic, no float atomics.
constexpr int kGnChunks = 256;    // parts per (b, plane) written by launch_gn_partials (128: +0.7 % per step, 512: +0.4 %)
struct GnStats { float* mr; };    // [B][3][32 groups][2] = {mean, rstd}
inline int gn_subgroup(int C) {   // largest power of two dividing C/32, at most 32
    int cg = C / 32, sg = 1;
    while (sg < 32 && cg % (sg * 2) == 0) sg *= 2;
    return sg;
}
int launch_gn_partials(const Tri& x, int B, GnPartials out, hipStream_t st);   // out: maxparts=kGnChunks, nsub=32
int launch_gn_finalize(const GnPartials& part, const Geo& g, int C, int B, GnStats out, hipStream_t st);
// GroupNorm statistics of the VIRTUAL tensor [bilinear2x(u) | sk] (TriplaneUpsample2x + concat, unet_triplane.py:106-124,
// 501-503) without materialising it: launch_gn_partials_up reduces the upsampled half per subgroup of `sg` channels
// (out: nparts from gn_up_parts, nsub = u.C / sg), the skip half brings the partials its producing convolution left
// (same subgroup size), and launch_gn_finalize_cat adds both in sub / part order.
void gn_up_parts(const Geo& out_g, int nparts[3]);             // parts per plane launch_gn_partials_up writes (8x8 output tiles)
int launch_gn_partials_up(const Tri& u, int B, int sg, GnPartials out, hipStream_t st);
int launch_gn_finalize_cat(const GnPartials& pu, const GnPartials& ps, const Geo& g, int C, int B, GnStats out, hipStream_t st);
// how many parts per plane a convolution epilogue writes for a given geometry (must match s3d_conv.hip's tiling)
void conv_gn_parts(ConvKind kind, const Geo& g, int nparts[3], bool wino24 = false);
// Winograd F(2x2,3x3) path for the 3x3 convolutions (s3d_wino.hip); S3D_WINO=0 selects the direct kernel
bool conv_use_wino();
void wino_gn_parts(const Geo& g, int nparts[3]);
size_t pack_wino_weights(std::vector<float>& stage, const float* W, int cout, int ctot, int cin);
int launch_conv_wino(ConvArgs& a, hipStream_t st);
double wino_exec_fraction();
// mixed Winograd F(2x4,3x3) (s3d_wino24.hip): the default 3x3 kernel of the inference forward (S3D_WINO=4 / 2 / 0 select the others)
bool conv_use_wino24();
bool conv_wino24_channels(int cin, int cout);       // the mixed kernel takes every 3x3 launch of these widths
size_t wino24_packed_floats(int cout, int cin);
size_t pack_wino24s_weights(std::vector<float>& stage, const float* W, int cout, int ctot, int cin);
// k_conv_wino24s (32 output channels per block) or, for launches of several rounds of blocks, k_conv_wino24w (64): the two are
// bit-identical, so the choice may depend on the batch size (S3D_WINO24W=0 never / =1 whenever the widths allow)
int launch_conv_wino24s(ConvArgs& a, hipStream_t st);
int launch_conv_wino24_narrow(ConvArgs& a, hipStream_t st);     // k_conv_wino24s, whatever the launch size
int launch_conv_wino24_wide(ConvArgs& a, hipStream_t st);       // k_conv_wino24w (cout % 64 == 0)
int launch_conv_wino24_glds(ConvArgs& a, hipStream_t st);       // k_conv_wino24g: halo by LDS-DMA, persistent blocks (S3D_WINO24G=1)

// GroupNorm-apply (+FiLM) + SiLU, writing y and (optionally) row/col partial sums of y for the rollout means.
struct ActArgs {
    const float* gamma[3];   // [C]
    const float* beta[3];
    const float* film;       // [B][film_stride]: scale at [0,C), shift at [C,2C); or null
    int film_stride;
};
constexpr int kActRows = 8, kActCols = 8;     // tile of the act kernel (measured: 8x16 +1 %, 4x16 +3 %, 16x8 +15 % per step; 4x8 on the
                                              // half-resolution planes only, 384 blocks instead of 192: -0.7 us per call, not kept)
struct MeanPartials {        // per plane: rowpart [B][ntc][h][C] (sum over a tile's columns), colpart [B][ntr][w][C]
    float* rowpart[3];
    float* colpart[3];
};
// stats.p == nullptr: identity (no norm, no SiLU) — used by the leaf-operator entry point to get the rollout
// means of a raw input.
// stats_part != null (with stats.mr == null): the kernel adds the producer's partial sums itself (gn_act_can_add_parts)
int launch_gn_act(const Tri& x, int B, GnStats stats, const ActArgs& a, Tri& y, const MeanPartials* mp,
                  hipStream_t st, const GnPartials* stats_part = nullptr);
bool gn_act_can_add_parts(const GnPartials& part, int C);
bool gn_parts_in_consumer(long long consumer_blocks);      // the consumer's own blocks add them (small launches) / k_gn_finalize_as does
int gn_act_threads(int C);                                  // block size of k_gn_act for C channels
long long gn_act_blocks(const Geo& g, int B);               // its grid
int launch_gn_finalize_as(const GnPartials& part, const Geo& g, int C, int B, int threads, GnStats out, hipStream_t st);
// the same on the virtual concat [bilinear2x(u) | sk] (y.C = u.C + sk.C, y.g = sk.g = 2 * u.g)
int launch_gn_act_cat(const Tri& u, const Tri& sk, int B, GnStats stats, const ActArgs& a, Tri& y, const MeanPartials* mp,
                      hipStream_t st);
// finalize the six mean vectors: rowmean[p] [B][h][C], colmean[p] [B][w][C]
struct MeanVecs { float* rowmean[3]; float* colmean[3]; };
struct MeanFinArgs {          // per vector v = 2 * plane + is_col (blockIdx.y): everything a block needs comes from wave-uniform scalar loads
    const float* src[6];      // tile partials [B][nt][len][C]
    float* dst[6];            // the mean vector [B][len][C]
    int len[6], nt[6];        // positions along the kept axis, tile partials per position
    float inv[6];             // 1 / length of the summed-out axis
    int C, cq, B;
};
MeanFinArgs means_finalize_args(const Geo& g, int C, int B, const MeanPartials& mp, const MeanVecs& mv);
int launch_means_finalize(const Geo& g, int C, int B, const MeanPartials& mp, MeanVecs mv, hipStream_t st);

// how many K slices (1 or 2) the rank-1 tables of a rollout convolution with `cin` own channels are cut into when its consumer is
// k_conv_wino24s (S3D_RANK1_SLICES=0: never): two from 256 channels on
int conv_rank1_slices(int cin);

// part != null: pixel-chunk form that also emits the GroupNorm partials of y (kGnChunks parts per plane, 32 groups)
int launch_avgpool(const Tri& x, int B, Tri& y, hipStream_t st, const GnPartials* part = nullptr);
// bilinear (align_corners=False) resize into a channel slice of a wider NHWC tensor
int launch_bilinear(const float* in, int B, int C, int hi, int wi, float* out, int ho, int wo, int out_cstride,
                    int out_coff, hipStream_t st);
// exact-2x bilinear upsample of u fused with the concat [up(u) | skip] for all three planes (out.C = u.C + sk.C)
bool upcat_gn_parts(const Geo& out_g, int C, int nparts[3]);      // part counts of the form that also emits GroupNorm partials
int launch_upcat(const Tri& u, const Tri& sk, int B, Tri& out, hipStream_t st, const GnPartials* part = nullptr);
int launch_copy_slice(const float* in, int B, int C, int h, int w, float* out, int out_cstride, int out_coff,
                      hipStream_t st);

// out head: GN + SiLU + 1x1 conv (C -> Cout small) + compose into NCHW [B,Cout,H+D,W+D] with zero corner
// fuse != null: + the sampler update of the step on the model output (in the same launch when out_head_fuses_sampler(); then
// `out` may be null and the model output is never stored)
// carry (fused steps only): the head also evaluates the NEXT step's in_conv — TriplaneConv(in, ch, 1, is_rollout =
// False), a pointwise map of x_{t-1}, unet_triplane.py:378, 482 — on the x_{t-1} values it has just formed, with k_in_conv_lds's
// arithmetic, thread mapping and GroupNorm-partial order (same bits), into the workspace tensors the next forward will read
struct InConvCarry { const float* wT; const float* bias; float* out[3]; double* part; int maxparts, Cin; };
int launch_out_head(const Tri& x, int B, GnStats stats, const ActArgs& a, const float* w /*[3][Cout][C]*/,
                    const float* bias /*[3][Cout]*/, int Cout, int H, int W, int D, float* out, hipStream_t st,
                    const s3d_sampler_args* fuse = nullptr, const GnPartials* part = nullptr, const InConvCarry* carry = nullptr);
bool out_head_can_carry(int C, int Cin, int Cout_head);  // the pixel-chunk head of this width can run the in_conv tail
bool out_head_fuses_sampler(int C, int Cout, int B);     // the update happens inside the head's launch (else: head, then k_sampler)
bool out_head_px_takes(int C, int Cout);                 // the pixel-chunk head serves this width (it can add GroupNorm partials itself)
// part (with stats.mr == null): the head adds its input's GroupNorm partials itself — no k_gn_finalize launch before it
bool out_head_adds_parts(const GnPartials& part, int C, int Cout);
long long out_head_px_blocks(const Geo& g, int B);         // grid of the pixel-chunk head

// small dense layers for the timestep path: y[b][o] = act_out( sum_i f(in[b][i]) * W[o][i] + bias[o] )
// in_mode 0: plain, 1: SiLU(in), 2: in is t[b] -> sinusoidal embedding of width I (cos | sin)
int launch_linear(const float* in, int B, int I, const float* W, const float* bias, int O, float* out,
                  int in_mode, int out_silu, hipStream_t st);

int launch_sampler(const s3d_sampler_args& a, hipStream_t st);

}  // namespace s3d
