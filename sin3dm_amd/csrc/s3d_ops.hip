// s3d_ops.hip — leaf-operator entry points (NCHW in / NCHW out) so tests can pin each kernel against the
// reference op it replaces.  They allocate, run and synchronise: test plumbing, not the hot path.
#include "s3d_model.h"

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
        ca.B = B; ca.cin = C; ca.cout = Cout; ca.njobs = 6;
        const float* rowvec[3] = {mv.rowmean[1], mv.rowmean[0], mv.colmean[0]};
        const float* colvec[3] = {mv.rowmean[2], mv.colmean[2], mv.colmean[1]};
        for (int p = 0; p < 3; ++p) {
            ConvJob& jr = ca.job[2 * p];
            jr.in = rowvec[p]; jr.wgt = wdev + cw.rrow[p]; jr.out = tab_row[p]; jr.h = 1; jr.w = g.h[p];
            jr.wgt_r1f = cw.rrow_f[p] ? wdev + cw.rrow_f[p] : nullptr;
            ConvJob& jc = ca.job[2 * p + 1];
            jc.in = colvec[p]; jc.wgt = wdev + cw.rcol[p]; jc.out = tab_col[p]; jc.h = 1; jc.w = g.w[p];
            jc.wgt_r1f = cw.rcol_f[p] ? wdev + cw.rcol_f[p] : nullptr;
        }
        S3D_TRY(launch_conv(CONV_1x3_ROLL, ca, st));
    }
    ConvArgs ca; memset(&ca, 0, sizeof ca);
    ca.B = B; ca.cin = C; ca.cout = Cout; ca.njobs = 3;
    for (int p = 0; p < 3; ++p) {
        ConvJob& J = ca.job[p];
        J.in = x.p[p]; J.wgt = wdev + cw.dense[p]; J.bias = wdev + cw.bias[p];
        J.wgt_wino = ksize == 3 ? wdev + cw.wino[p] : nullptr;
        J.wgt_wino24s = ksize == 3 && cw.wino24s[p] ? wdev + cw.wino24s[p] : nullptr;
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

// timestep_embedding (src/diffusion/nn.py:103-121) alone: the embedding stage of k_linear against an identity weight
// (one non-zero product per output, the rest exact zeros).
int s3d_op_timestep_embed(const float* t, int B, int dim, float* out, void* stream) {
    S3D_CHECK(t && out && B >= 1 && dim >= 1, S3D_ERR_INVALID, "op_timestep_embed: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scratch sc;
    std::vector<float> eye(size_t(dim) * dim, 0.f);
    for (int i = 0; i < dim; ++i) eye[size_t(i) * dim + i] = 1.f;
    float* w = nullptr;
    S3D_TRY(sc.alloc(&w, eye.size()));
    S3D_HIP(hipMemcpyAsync(w, eye.data(), eye.size() * sizeof(float), hipMemcpyHostToDevice, st));
    S3D_TRY(launch_linear(t, B, dim, w, nullptr, dim, out, 2, 0, st));
    S3D_HIP(hipStreamSynchronize(st));
    return 0;
}

// One TriplaneResBlock (src/diffusion/unet_triplane.py:175-311) through exactly the code path of the model's blocks
// (Fwd::resblock): parameters by their state-dict names relative to the block, host pointers, PyTorch layouts.
int s3d_op_triplane_resblock(const float* const in[3], float* const out[3], const float* emb, int B, int C, int Cout,
                             int H, int W, int D, int emb_dim, int use_scale_shift_norm, int is_rollout,
                             const char* const* names, const float* const* tensors, int n_tensors, void* stream) {
    S3D_CHECK(in && out && emb && names && tensors, S3D_ERR_INVALID, "op_triplane_resblock: null argument");
    S3D_CHECK(C % 32 == 0 && Cout % 32 == 0, S3D_ERR_INVALID, "op_triplane_resblock: GroupNorm32 needs channels %% 32 == 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::map<std::string, const float*> P;
    for (int i = 0; i < n_tensors; ++i) P[names[i]] = tensors[i];
    auto get = [&](const std::string& k) -> const float* { auto it = P.find(k); return it == P.end() ? nullptr : it->second; };
    const bool ssn = use_scale_shift_norm != 0, roll = is_rollout != 0;
    s3d_unet m;
    memset(&m.cfg, 0, sizeof m.cfg);
    m.cfg.use_scale_shift_norm = ssn; m.cfg.is_rollout = roll;
    ResBlockW rb;
    rb.C = C; rb.Cout = Cout; rb.has_skip = C != Cout; rb.film_off = 0;
    const int eo = ssn ? 2 * Cout : Cout;
    m.film_total = eo;
    std::vector<float>& stg = m.stage;
    const char* missing = nullptr;
    auto need = [&](const std::string& k) -> const float* { const float* p = get(k); if (!p && !missing) missing = strdup(k.c_str()); return p; };
    auto norm = [&](const std::string& pre, int c, NormW& nw) {
        for (int p = 0; p < 3; ++p) {
            const float* g = need(pre + ".norm_" + kPlane[p] + ".weight"); const float* b = need(pre + ".norm_" + kPlane[p] + ".bias");
            if (g && b) { nw.gamma[p] = push(stg, g, c); nw.beta[p] = push(stg, b, c); }
        }
    };
    auto tconv = [&](const std::string& pre, int cin, int cout, int k, bool r, ConvW& cw) {
        const float* Wp[3]; const float* bp[3];
        bool ok = true;
        for (int p = 0; p < 3; ++p) {
            Wp[p] = need(pre + ".conv_" + kPlane[p] + ".weight"); bp[p] = need(pre + ".conv_" + kPlane[p] + ".bias");
            ok = ok && Wp[p] && bp[p];
        }
        if (ok) pack_tconv_raw(stg, Wp, bp, cin, cout, k, r, cw);
    };
    norm("in_layers.0", C, rb.n1);
    tconv("in_layers.2", C, Cout, 3, roll, rb.c1);
    norm("out_layers.0", Cout, rb.n2);
    tconv("out_layers.2", Cout, Cout, 3, roll, rb.c2);
    if (rb.has_skip) tconv("skip_connection", C, Cout, 1, false, rb.skip);
    const float* ew = need("emb_layers.1.weight"); const float* eb = need("emb_layers.1.bias");
    if (missing) { set_error("op_triplane_resblock: parameter '%s' missing", missing); free(const_cast<char*>(missing)); return S3D_ERR_MISSING; }
    const size_t film_w = push(stg, ew, size_t(eo) * emb_dim), film_b = push(stg, eb, eo);
    S3D_TRY(upload(m.wbuf, stg.data(), stg.size() * sizeof(float)));
    Scratch sc;
    const Geo g = Geo::from_hwd(H, W, D);
    Tri x, o;
    S3D_TRY(tri_from_nchw(sc, in, B, C, g, x, st));
    float* film = nullptr;
    S3D_TRY(sc.alloc(&film, size_t(B) * eo));
    S3D_TRY(launch_linear(emb, B, emb_dim, m.dev(film_w), m.dev(film_b), eo, film, 1, 0, st));     // emb_layers = SiLU -> Linear
    Fwd f{&m, B, st, film};
    m.arena.measuring = true; m.arena.high = 0; m.arena.reset();
    S3D_TRY(f.resblock(rb, x, o, false));
    m.arena.measuring = false;
    S3D_TRY(m.arena.buf.reserve(m.arena.high));
    m.arena.reset();
    S3D_TRY(f.resblock(rb, x, o, false));
    S3D_TRY(tri_to_nchw(o, B, out, st));
    S3D_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
