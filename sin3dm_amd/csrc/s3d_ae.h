// s3d_ae.h — launchers of the auto-encoder training tier (s3d_ae_kernels.hip, s3d_decoder.hip) used by s3d_ae.hip.
#pragma once
#include "s3d_common.h"

namespace s3d {

// the input volume reduced to three 2-D projections (see s3d_ae_kernels.hip): P[p] [2h][2w][4 taps][C]
struct EncDesc {
    const float* P[3];
    Geo g;                 // feature-map plane sizes (h, w per plane)
    float inv_len[3];      // 1 / length of the averaged axis (D, W, H)
    int C, CO;             // input channels of the volume, encoder feature channels (geo + tex)
};
int launch_project(const float* vol, int C, int X2, int Y2, int Z2, float* const P[3], hipStream_t st);
int launch_enc_pack(const float* wgeo, const float* wtex, int geo, int tex, int C, float* Wp, hipStream_t st);
int launch_enc_fwd(const EncDesc& e, const float* Wp, const float* bias, float* const pre[3], hipStream_t st);
int launch_enc_norm(bool backward, float* const x[3], float* const y[3], float* mr, float* const dy[3], float* const dx[3],
                    const Geo& g, int CO, void* ws, hipStream_t st);       // ws (backward only): enc_norm_bwd_ws_bytes(CO)
size_t enc_norm_bwd_ws_bytes(int CO);
size_t enc_wgrad_ws_floats(int C, int CO);
int launch_enc_wgrad(const EncDesc& e, float* const dpre[3], int geo, int tex, float* ws, float* dwgeo, float* dbgeo, float* dwtex,
                     float* dbtex, hipStream_t st);

// the three planes of a net in one launch each
int launch_slice_pad_nhwc3(const float* const in[3], float* const out[3], const size_t hw[3], int CT, int c0, int cin, hipStream_t st);
int launch_unslice_nhwc3(const float* const dx[3], float* const dfeat[3], const size_t hw[3], int CT, int c0, int cin, hipStream_t st);
int launch_mr_from_partials3(double* const part[3], int nchunks, int C, const size_t hw[3], float eps, float* mr, hipStream_t st);   // mr [3][C][2]
// InstanceNorm2d(affine) + SiLU of a net's three planes + their {mean, rstd} (mr [3][C][2]): partials, statistics, apply — three launches
int launch_inorm_silu3(float* const x[3], double* const part[3], const float* const gamma[3], const float* const beta[3], float* const y[3],
                       const size_t hw[3], int C, float eps, float* mr, hipStream_t st);
int launch_colsum3(const float* const x[3], const size_t rows[3], int C, float* ws, float* const out[3], float* const out2[3], hipStream_t st);   // ws: 3 * colsum_ws_floats(C)
// InstanceNorm2d(affine, eps) + SiLU of one NHWC plane (s3d_decoder.hip); part: kInNormChunks*C*2 doubles
constexpr int kInNormChunks = 64;
int launch_inorm_silu(const float* x, double* part, const float* gamma, const float* beta, float* y, int hw, int C, float eps,
                      hipStream_t st);

struct PointSet { const float* pts; long long N, Np; float aabb[6]; };   // Np = N rounded up to the GEMM row tile
int launch_gather(const PointSet& ps, const float* const feat[2][3], const int ph[3], const int pw[3], int C, int nnets,
                  float* const X[2], hipStream_t st);
struct GatherArgs {
    const float* pts; long long N, Np;
    float amin[3], ainv[3];                     // x_n = 2 (x - amin) * ainv - 1
    const float* feat[2][3]; float* dfeat[2][3];
    int ph[3], pw[3];
    float* X[2]; const float* dX[2];            // [Np][C]
    int C, nnets;
};
struct ScatterSortedArgs {
    GatherArgs g;
    const unsigned* order[3]; const unsigned* start[3];            // sorted point ids / first position of every cell
    long long begin[4];
};
struct ScatterPlan { ScatterSortedArgs args; bool ready = false; };     // the point order of one batch (launch_scatter_prepare)
// backward of launch_gather: dfeat (every element written) from dX; ws: scatter_ws_bytes(...) bytes.  No float atomics.
size_t scatter_ws_bytes(long long Np, const int ph[3], const int pw[3]);
int launch_scatter(const PointSet& ps, float* const dfeat[2][3], const int ph[3], const int pw[3], int C, int nnets,
                   const float* const dX[2], void* ws, hipStream_t st);
// the same in two halves: the point order depends on the points alone (21 small dependent launches) and may be enqueued early
int launch_scatter_prepare(const PointSet& ps, const int ph[3], const int pw[3], int C, int nnets, void* ws, ScatterPlan& plan, hipStream_t st);
int launch_scatter_apply(ScatterPlan& plan, float* const dfeat[2][3], const float* const dX[2], hipStream_t st);

// dpre = dact[:, coff:coff+C] * (act > 0) and colsum[C] = its column sums (ws: colsum_ws_floats(C))
int launch_relu_bwd(const float* dact, int dstride, int coff, const float* act, float* dpre, long long rows, int C, float* ws,
                    float* colsum, hipStream_t st);
int launch_add_slice(const float* a, const float* b, int bstride, int coff, float* out, long long rows, int C, hipStream_t st);
size_t colsum_ws_floats(int C);
int launch_colsum(const float* x, long long rows, int C, float* ws, float* out, hipStream_t st, float* out2 = nullptr);   // out2: a second parameter with the same gradient
int launch_last_fwd(const float* h, const float* W, const float* b, int I, int O, int sigm, float* pred, int pstride, int ooff,
                    long long N, hipStream_t st);
size_t last_bwd_ws_floats(int I, int O);
int launch_last_bwd(const float* dout, int dstride, int ooff, const float* W, const float* h, int I, int O, long long Np, float* dh,
                    float* ws, float* dW, float* db, hipStream_t st);
// losses[3] = {sdf_loss, tex_loss, #points inside the texture band}; dout [Np][1+TC] or null; ws: 192 floats
int launch_ae_loss(const float* pred, const float* sdf, const float* tex, long long N, long long Np, int TC, int sdf_mode, int tex_mode,
                   float band, float tex_weight, float* ws, float* losses, float* dout, hipStream_t st);

}  // namespace s3d
