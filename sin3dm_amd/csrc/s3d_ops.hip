// s3d_ops.hip — leaf-operator entry points (NCHW in / NCHW out) so tests can pin each kernel against the
// reference op it replaces.  They allocate, run and synchronise: test plumbing, not the hot path.
#include "s3d_common.h"

using namespace s3d;

namespace {

struct Scratch {                      // frees everything at scope exit
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T>
    int alloc(T** out, size_t n) {
        void* p = nullptr;
        S3D_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return 0;
    }
};

int tri_from_nchw(Scratch& sc, const float* const in[3], int B, int C, const Geo& g, Tri& t, hipStream_t st) {
    t.C = C; t.g = g;
    for (int p = 0; p < 3; ++p) {
        S3D_TRY(sc.alloc(&t.p[p], size_t(B) * C * g.h[p] * g.w[p]));
        S3D_TRY(launch_nchw_to_nhwc(in[p], t.p[p], B, C, g.h[p], g.w[p], st));
    }
    return 0;
}
int tri_alloc(Scratch& sc, int B, int C, const Geo& g, Tri& t) {
    t.C = C; t.g = g;
    for (int p = 0; p < 3; ++p) S3D_TRY(sc.alloc(&t.p[p], size_t(B) * C * g.h[p] * g.w[p]));
    return 0;
}
int tri_to_nchw(const Tri& t, int B, float* const out[3], hipStream_t st) {
    for (int p = 0; p < 3; ++p) S3D_TRY(launch_nhwc_to_nchw(t.p[p], out[p], B, t.C, t.g.h[p], t.g.w[p], st));
    return 0;
}

}  // namespace

extern "C" {

int s3d_op_triplane_conv(const float* const in[3], float* const out[3], int B, int C, int H, int W, int D, int Cout,
                         int ksize, int is_rollout, const float* const weight[3], const float* const bias[3],
                         void* stream) {
    S3D_CHECK(in && out && weight && bias, S3D_ERR_INVALID, "op_triplane_conv: null argument");
    S3D_CHECK(ksize == 1 || ksize == 3, S3D_ERR_INVALID, "op_triplane_conv: kernel size %d", ksize);
    S3D_CHECK(!is_rollout || ksize == 3, S3D_ERR_INVALID, "op_triplane_conv: rollout needs a 3x3 kernel");
    S3D_CHECK(C % 32 == 0, S3D_ERR_INVALID, "op_triplane_conv: C=%d must be a multiple of 32", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scratch sc;
    const Geo g = Geo::from_hwd(H, W, D);
    Tri x, y, o;
    S3D_TRY(tri_from_nchw(sc, in, B, C, g, x, st));
    S3D_TRY(tri_alloc(sc, B, Cout, g, o));

    std::vector<float> stage;
    ConvW cw;
    pack_tconv_raw(stage, weight, bias, C, Cout, ksize, is_rollout != 0, cw);
    float* wdev = nullptr;
    S3D_TRY(sc.alloc(&wdev, stage.size()));
    S3D_HIP(hipMemcpyAsync(wdev, stage.data(), stage.size() * sizeof(float), hipMemcpyHostToDevice, st));

    float* tab_row[3] = {nullptr, nullptr, nullptr};
    float* tab_col[3] = {nullptr, nullptr, nullptr};
    if (is_rollout) {
        S3D_TRY(tri_alloc(sc, B, C, g, y));
        MeanPartials mp; MeanVecs mv;
        for (int p = 0; p < 3; ++p) {
            const int h = g.h[p], w = g.w[p];
            const int ntc = (w + kActCols - 1) / kActCols, ntr = (h + kActRows - 1) / kActRows;
            S3D_TRY(sc.alloc(&mp.rowpart[p], size_t(B) * ntc * h * C));
            S3D_TRY(sc.alloc(&mp.colpart[p], size_t(B) * ntr * w * C));
            S3D_TRY(sc.alloc(&mv.rowmean[p], size_t(B) * h * C));
            S3D_TRY(sc.alloc(&mv.colmean[p], size_t(B) * w * C));
            S3D_TRY(sc.alloc(&tab_row[p], size_t(B) * h * 4 * Cout));
            S3D_TRY(sc.alloc(&tab_col[p], size_t(B) * w * 4 * Cout));
        }
        ActArgs aa; memset(&aa, 0, sizeof aa);
        S3D_TRY(launch_gn_act(x, B, GnStats{nullptr}, aa, y, &mp, st));          // identity copy + axis sums
        S3D_TRY(launch_means_finalize(g, C, B, mp, mv, st));
        ConvArgs ca; memset(&ca, 0, sizeof ca);
        ca.B = B; ca.cin = C; ca.cout = 4 * Cout; ca.njobs = 6;
        const float* rowvec[3] = {mv.rowmean[1], mv.rowmean[0], mv.colmean[0]};
        const float* colvec[3] = {mv.rowmean[2], mv.colmean[2], mv.colmean[1]};
        for (int p = 0; p < 3; ++p) {
            ConvJob& jr = ca.job[2 * p];
            jr.in = rowvec[p]; jr.wgt = wdev + cw.rrow[p]; jr.out = tab_row[p]; jr.h = 1; jr.w = g.h[p];
            ConvJob& jc = ca.job[2 * p + 1];
            jc.in = colvec[p]; jc.wgt = wdev + cw.rcol[p]; jc.out = tab_col[p]; jc.h = 1; jc.w = g.w[p];
        }
        S3D_TRY(launch_conv(CONV_1x3_VEC, ca, st));
    }
    ConvArgs ca; memset(&ca, 0, sizeof ca);
    ca.B = B; ca.cin = C; ca.cout = Cout; ca.njobs = 3;
    for (int p = 0; p < 3; ++p) {
        ConvJob& J = ca.job[p];
        J.in = x.p[p]; J.wgt = wdev + cw.dense[p]; J.bias = wdev + cw.bias[p];
        J.wgt_wino = ksize == 3 ? wdev + cw.wino[p] : nullptr;
        J.rrow = tab_row[p]; J.rcol = tab_col[p]; J.out = o.p[p]; J.h = g.h[p]; J.w = g.w[p];
    }
    S3D_TRY(launch_conv(ksize == 3 ? CONV_3x3 : CONV_1x1, ca, st));
    S3D_TRY(tri_to_nchw(o, B, out, st));
    S3D_HIP(hipStreamSynchronize(st));
    return 0;
}

int s3d_op_triplane_norm_silu(const float* const in[3], float* const out[3], int B, int C, int H, int W, int D,
                              const float* const gamma[3], const float* const beta[3], void* stream) {
    S3D_CHECK(in && out && gamma && beta, S3D_ERR_INVALID, "op_triplane_norm_silu: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scratch sc;
    const Geo g = Geo::from_hwd(H, W, D);
    Tri x, y;
    S3D_TRY(tri_from_nchw(sc, in, B, C, g, x, st));
    S3D_TRY(tri_alloc(sc, B, C, g, y));
    GnPartials part;
    S3D_TRY(sc.alloc(&part.p, size_t(B) * 3 * kGnChunks * 64));
    part.maxparts = kGnChunks; part.nsub = 32;
    for (int p = 0; p < 3; ++p) part.nparts[p] = kGnChunks;
    GnStats stats;
    S3D_TRY(sc.alloc(&stats.mr, size_t(B) * 3 * 64));
    ActArgs aa; memset(&aa, 0, sizeof aa);
    for (int p = 0; p < 3; ++p) {
        float *gd, *bd;
        S3D_TRY(sc.alloc(&gd, C)); S3D_TRY(sc.alloc(&bd, C));
        S3D_HIP(hipMemcpyAsync(gd, gamma[p], C * sizeof(float), hipMemcpyHostToDevice, st));
        S3D_HIP(hipMemcpyAsync(bd, beta[p], C * sizeof(float), hipMemcpyHostToDevice, st));
        aa.gamma[p] = gd; aa.beta[p] = bd;
    }
    S3D_TRY(launch_gn_partials(x, B, part, st));
    S3D_TRY(launch_gn_finalize(part, g, C, B, stats, st));
    S3D_TRY(launch_gn_act(x, B, stats, aa, y, nullptr, st));
    S3D_TRY(tri_to_nchw(y, B, out, st));
    S3D_HIP(hipStreamSynchronize(st));
    return 0;
}

int s3d_op_triplane_resample(const float* const in[3], float* const out[3], int B, int C, const int hi[3],
                             const int wi[3], const int ho[3], const int wo[3], int mode, void* stream) {
    S3D_CHECK(in && out && hi && wi && ho && wo, S3D_ERR_INVALID, "op_triplane_resample: null argument");
    S3D_CHECK(C % 4 == 0, S3D_ERR_INVALID, "op_triplane_resample: C=%d must be a multiple of 4", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scratch sc;
    Geo gi{{hi[0], hi[1], hi[2]}, {wi[0], wi[1], wi[2]}};
    Geo go = mode == 0 ? gi.half() : Geo{{ho[0], ho[1], ho[2]}, {wo[0], wo[1], wo[2]}};
    Tri x, y;
    S3D_TRY(tri_from_nchw(sc, in, B, C, gi, x, st));
    S3D_TRY(tri_alloc(sc, B, C, go, y));
    if (mode == 0) S3D_TRY(launch_avgpool(x, B, y, st));
    else
        for (int p = 0; p < 3; ++p)
            S3D_TRY(launch_bilinear(x.p[p], B, C, gi.h[p], gi.w[p], y.p[p], go.h[p], go.w[p], C, 0, st));
    S3D_TRY(tri_to_nchw(y, B, out, st));
    S3D_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
