// s3d_bwd.h — launchers of the backward kernels (s3d_bwd.hip), used by the training driver (s3d_train.hip).
#pragma once
#include "s3d_common.h"
#include <vector>

namespace s3d {

// row / column sums of dy with the three edge variants: R[p] [B][h][3][C], Cs[p] [B][w][3][C]
int launch_edge_sums(const Tri& dy, int B, float* const R[3], float* const Cs[3], hipStream_t st, hipEvent_t stop = nullptr);   // stop: recorded with the launch's completion
// dbias[p][C] from the row sums; per_sample [B][stride] (+= over planes) or null
int launch_bias_grad(float* const R[3], const Geo& g, int C, int B, float* const dbias[3], float* per_sample,
                     int per_sample_stride, hipStream_t st);

struct SlotWgradArgs {
    const float* rowvec[3]; const float* colvec[3];   // the conv's row-/column-varying mean vectors [B][h|w][C]
    const float* R[3]; const float* Cs[3];            // edge sums of dy
    float* dW[3];                                     // OIHW [cout][3C][3][3] gradient of each plane's conv weight
    Geo g; int B, C, cout;
};
int launch_slot_wgrad(const SlotWgradArgs& s, hipStream_t st);

struct WgradArgs {
    Tri dy;            // [B][h][w][cout]
    Tri a;             // conv input; a.C is its channel stride, channels [0, cin) are used
    float* part[3];    // split-K partials, wgrad_part_floats() each
    float* dW[3];      // OIHW [cout][ctot][taps]; own channels [0, cin) are written
    int B, cin, cout, ctot, taps, ksplit;
    int cin_store = 0;   // > 0: only input channels [0, cin_store) are written (the rest are zero padding of `a`)
    int nplanes = 3;     // jobs actually present in dy/a/part/dW
};
int wgrad_ksplit(const Geo& g, int B, int cin, int cout, int taps);
bool wgrad_uses_wino();          // the 3x3 weight gradient goes to k_wgrad_wino (option WGRAD_WINO)
size_t wgrad_part_floats(int ksplit, int cin, int cout, int taps);
// The small tail launches of a convolution's backward — the split-K reduction of its weight gradient and the bias gradient from
// the row sums — have no consumer inside the pass: a DeferredTail collects them (operands live in the pass's arena) and
// flush() runs each kind as ONE launch over all collected jobs (same arithmetic per job: same bits).  A training step has
// ten of each, 6-10 us of dependent-launch latency apiece.
struct DeferredTail {
    struct Red { const float* part[3]; float* dW[3]; int ksplit, cout, cin, ctot, taps, cin_store, nplanes, wino; };
    struct Bias { const float* R[3]; float* dbias[3]; float* per_sample; int per_sample_stride; int h[3]; int B, C; };
    std::vector<Red> red;
    std::vector<Bias> bias;
    int flush(hipStream_t st);
};
// tail != null: the reduction launch is queued there instead of being enqueued behind the partial kernel
int launch_wgrad(const WgradArgs& w, hipStream_t st, DeferredTail* tail = nullptr);
int launch_bias_grad_deferred(float* const R[3], const Geo& g, int C, int B, float* const dbias[3], float* per_sample,
                              int per_sample_stride, DeferredTail& tail);

struct GnActBwd {
    Tri x, dy, dx;
    const float* const* rowadd; const float* const* coladd;   // arrays of 3 or null
    const Tri* add;
    const float* gamma[3]; const float* beta[3];
    float* dgamma[3]; float* dbeta[3];
    GnStats stats;
    const float* film; float* dfilm; int film_stride;
    float* ws;                                                // gn_bwd_ws_floats(B, C)
    int B;
    int ngroups = 32;                                         // GroupNorm32; C for an InstanceNorm
    // the two per-channel sums were left by the input-gradient convolution that produced dy (k_conv_wino24s_gnb): per-tile records in
    // the GroupNorm-partial layout with one subgroup per channel — no read pass over (x, dy) then
    const GnPartials* conv_part = nullptr;
};
size_t gn_bwd_ws_floats(int B, int C);
int launch_gn_act_bwd(const GnActBwd& s, hipStream_t st);

struct SmallOuter {
    const float* s; int S;          // composed map [B][S][H+D][W+D]
    Tri v;                          // NHWC planes, C channels
    const float* Wt; Tri* dv;       // optional dv = Wt^T s
    float* outer[3]; int outer_transposed;
    float* const* ssum; float* const* vsum;
    float* ws;                      // small_outer_ws_floats(S, C)
    int H, W, D, C, B;
};
size_t small_outer_ws_floats(int S, int C);
int launch_small_outer(const SmallOuter& s, hipStream_t st);

int launch_pool_bwd_add(const Tri* dpool, const Tri* dskip, int skip_coff, int B, Tri& out, hipStream_t st);
int launch_bilinear_bwd(const float* dout, int B, int C, int ho, int wo, int out_cstride, int out_coff, float* din, int hi,
                        int wi, hipStream_t st);
// the three planes of a triplane in one launch (same per-element arithmetic)
int launch_bilinear_bwd3(const float* const dout[3], int B, int C, const int ho[3], const int wo[3], int out_cstride, int out_coff,
                         float* const din[3], const int hi[3], const int wi[3], hipStream_t st);
// y = L(f(in)): dW/db (when dW != null) and dx = (dy W) * f'(in) (when dx != null); dy rows are dy_stride apart
int launch_linear_bwd(const float* dy, int dy_stride, const float* in, int B, int I, const float* W, int O, int in_mode, float* dW,
                      float* db, float* dx, hipStream_t st);

// dW / db of nseg linears on one input whose dy rows are the column ranges [seg_begin[k], seg_begin[k+1]) of one matrix (contiguous, from 0)
int launch_linear_bwd_w_multi(const float* dy, int dy_stride, const float* in, int B, int I, int in_mode, int nseg, const int* seg_begin,
                              float* const* dW, float* const* db, hipStream_t st);

// x0_bs / tgt_bs: elements between the batch rows of x0 / tgt (0: one row shared by the batch; < 0: dense)
int launch_q_sample(const float* x0, const float* eps, const float* sa, const float* sb, const int64_t* t, long long per, int B,
                    float* xt, hipStream_t st, long long x0_bs = -1);
constexpr int kMseWsFloats = 3 * 32;     // per sample
int launch_mse_terms(const float* out, const float* tgt, int B, int C, int H, int W, int D, float* ws, float* terms, hipStream_t st, long long tgt_bs = -1);
int launch_mse_grad(const float* out, const float* tgt, const float* wgt, int B, int C, int H, int W, int D, float* dout, hipStream_t st,
                    long long tgt_bs = -1, int wcols = 3, float wdiv = 1.0f);
int launch_adamw_ema(float* p, const float* g, float* m, float* v, float* const* ema, const float* ema_rate, int n_ema, long long n,
                     float lr, float b1, float b2, float eps, float wd, int step, hipStream_t st);

}  // namespace s3d
