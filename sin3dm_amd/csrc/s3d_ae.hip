// s3d_ae.hip — auto-encoder training tier: AutoEncoderGroupSkip encode + decode + losses and
// their backward pass on the MI355X, behind s3d_ae_* (include/sin3dm_hip.h).
//
// Reference: AutoEncoderGroupSkip (src/encoding/networks.py:122-220), TriplaneGroupResnetBlock / DecoderMLPSkipConcat
// (src/encoding/blocks.py:65-91, 189-256), ShapeAutoEncoder._forward_batch / update_network (src/encoding/model.py:178-237).
//
// One training iteration (what `net(input_grid, pts)` + `loss.backward()` do there):
//   encode   the Conv3d + axis means collapse to three 2-D convolutions over projections of the (fixed) input volume
//            (s3d_ae_kernels.hip) -> InstanceNorm2d -> tanh(x/2): the 12-channel triplane latent
//   planes   per feature group (geo 4 ch, tex 8 ch): conv5x5 -> InstanceNorm(affine) -> SiLU -> conv5x5 + 1x1 shortcut
//            on the MFMA convolution kernels (s3d_conv.hip)
//   points   bilinear border-clamped gather of the three planes, summed -> [N][64] per group
//   MLPs     DecoderMLPSkipConcat as 1x1 convolutions over the point list (fp32 MFMA); activations kept for the backward
//   losses   weighted-L1 sdf + masked texture loss (k_loss_*)
//   backward the same operators transposed: MLP dgrad (1x1 conv with W^T), MLP wgrad (k_wgrad_mfma<1>), scatter-add of the
//            point gradients onto the planes (points sorted by plane cell, then a per-pixel gather in that fixed order:
//            no float atomics, unlike PyTorch's grid_sampler backward), 5x5 dgrad / wgrad (k_wgrad_mfma<25>), InstanceNorm backward
//            (the GroupNorm backward with one channel per group), tanh/InstanceNorm backward and the encoder's weight
//            gradient as a 2-D correlation with the projections.
// Parameters: one flat fp32 device vector in the order [geo_encoder, geo_convs, geo_decoder | tex_encoder, tex_convs,
// tex_decoder] — the reference's two AdamW parameter groups (networks.py:146-150) are the two halves.
#include <memory>

#include "s3d_ae.h"
#include "s3d_bwd.h"
#include "s3d_model.h"

namespace s3d {

static const char* kAePl[3] = {"xy", "xz", "yz"};

struct AeSpec { std::string name; std::vector<int64_t> shape; size_t numel() const { size_t n = 1; for (auto d : shape) n *= size_t(d); return n; } };

struct AeNet {
    int cin = 0, nout = 0;                                   // real feature channels (4 / 8), outputs (1 / tex_channels)
    size_t f_in_w, f_in_b, f_out_w, f_out_b, f_sc_w, f_sc_b, f_gamma[3], f_beta[3], f_mw[6], f_mb[6];   // flat offsets
    size_t in_dense[3], out_dense[3], sc_dense[3], in_T[3], out_T[3], sc_T[3], mT[5];                   // packed offsets
    int I[6], O[6];
};

}  // namespace s3d

using namespace s3d;

struct s3d_ae {
    s3d_decoder_cfg cfg;
    int geo, tex, up, hid, TC, C, nnets = 2;
    std::vector<AeSpec> specs;
    std::vector<size_t> flat_off;
    int64_t tex_begin = 0;
    float* flat = nullptr;
    AeNet net[2];
    size_t f_enc_w[2], f_enc_b[2];
    // packed image
    DevBuf pbuf, descs_dev;
    std::vector<PackDesc> descs;
    int pack_blocks = 0;
    size_t wp_off = 0, encb_off = 0, psize = 0;
    // volume projections
    DevBuf proj;
    EncDesc enc{};
    bool has_volume = false;
    Arena arena;
    // backward pass: weight-gradient launches (no consumer before the optimizer) on a handle-owned low-priority side stream, beside
    // the chain of input gradients — as in the denoiser's backward pass (s3d_train.hip: Bwd::edge / join)
    hipStream_t side = nullptr;
    hipStream_t chain2 = nullptr;       // the texture MLP's backward chain beside the geometry MLP's (the two nets only meet at the scatter)
    std::vector<hipEvent_t> events;
    ~s3d_ae() {
        for (auto e : events) (void)hipEventDestroy(e);
        if (side) (void)hipStreamDestroy(side);
        if (chain2) (void)hipStreamDestroy(chain2);
    }
    const float* P(size_t off) const { return static_cast<const float*>(pbuf.p) + off; }
    size_t off_of(const std::string& n) const {
        for (size_t i = 0; i < specs.size(); ++i) if (specs[i].name == n) return flat_off[i];
        return size_t(-1);
    }
};

namespace s3d {

static void ae_specs(s3d_ae* a) {
    const int up = a->up, hid = a->hid;
    auto add = [&](const std::string& n, std::vector<int64_t> s) { a->specs.push_back({n, std::move(s)}); };
    const char* pre[2] = {"geo", "tex"};
    const int cin[2] = {a->geo, a->tex}, cout[2] = {1, a->TC}, encin[2] = {1, a->C};
    for (int k = 0; k < 2; ++k) {
        if (k == 1) { int64_t n = 0; for (auto& s : a->specs) n += int64_t(s.numel()); a->tex_begin = n; }
        const std::string e = std::string(pre[k]) + "_encoder", cv = std::string(pre[k]) + "_convs", ml = std::string(pre[k]) + "_decoder";
        add(e + ".weight", {cin[k], encin[k], 4, 4, 4}); add(e + ".bias", {cin[k]});
        add(cv + ".in_layers.0.weight", {3 * up, cin[k], 5, 5}); add(cv + ".in_layers.0.bias", {3 * up});
        for (int p = 0; p < 3; ++p) { add(cv + ".norm_" + kAePl[p] + ".weight", {up}); add(cv + ".norm_" + kAePl[p] + ".bias", {up}); }
        add(cv + ".out_layers.1.weight", {3 * up, up, 5, 5}); add(cv + ".out_layers.1.bias", {3 * up});
        add(cv + ".shortcut.weight", {3 * up, cin[k], 1, 1}); add(cv + ".shortcut.bias", {3 * up});
        const char* ln[6] = {".first_layers.0", ".first_layers.2", ".first_layers.4", ".second_layers.0", ".second_layers.2", ".second_layers.4"};
        const int I[6] = {up, hid, hid, up + hid, hid, hid}, O[6] = {hid, hid, hid, hid, hid, cout[k]};
        for (int l = 0; l < 6; ++l) { add(ml + ln[l] + ".weight", {O[l], I[l]}); add(ml + ln[l] + ".bias", {O[l]}); }
    }
    a->flat_off.resize(a->specs.size());
    size_t off = 0;
    for (size_t i = 0; i < a->specs.size(); ++i) { a->flat_off[i] = off; off += a->specs[i].numel(); }
}

static int ae_plan(s3d_ae* a) {
    const int up = a->up, hid = a->hid;
    size_t ps = 0;
    auto palloc = [&](size_t n) { size_t o = (ps + 63) & ~size_t(63); ps = o + n; return o; };
    auto add = [&](int kind, size_t src, size_t dst, long long n, int cout = 0, int ctot = 0, int cin = 0, int taps = 0, int slot = 0) {
        PackDesc d{};
        d.kind = kind; d.cout = cout; d.ctot = ctot; d.cin = cin; d.slot = slot; d.taps = taps; d.src = (long long)src; d.dst = (long long)dst; d.n = n;
        a->descs.push_back(d);
    };
    a->descs.clear();
    const char* pre[2] = {"geo", "tex"};
    const int cin[2] = {a->geo, a->tex}, cout[2] = {1, a->TC};
    a->encb_off = palloc(a->geo + a->tex);
    a->wp_off = palloc(size_t(3) * (a->geo + a->tex) * 16 * 4 * a->C);
    for (int k = 0; k < 2; ++k) {
        AeNet& N = a->net[k];
        N.cin = cin[k]; N.nout = cout[k];
        const std::string e = std::string(pre[k]) + "_encoder", cv = std::string(pre[k]) + "_convs", ml = std::string(pre[k]) + "_decoder";
        a->f_enc_w[k] = a->off_of(e + ".weight"); a->f_enc_b[k] = a->off_of(e + ".bias");
        add(PK_COPY, a->f_enc_b[k], a->encb_off + (k == 0 ? 0 : a->geo), cin[k]);
        N.f_in_w = a->off_of(cv + ".in_layers.0.weight"); N.f_in_b = a->off_of(cv + ".in_layers.0.bias");
        N.f_out_w = a->off_of(cv + ".out_layers.1.weight"); N.f_out_b = a->off_of(cv + ".out_layers.1.bias");
        N.f_sc_w = a->off_of(cv + ".shortcut.weight"); N.f_sc_b = a->off_of(cv + ".shortcut.bias");
        for (int p = 0; p < 3; ++p) {
            N.f_gamma[p] = a->off_of(cv + ".norm_" + kAePl[p] + ".weight"); N.f_beta[p] = a->off_of(cv + ".norm_" + kAePl[p] + ".bias");
            const size_t win = N.f_in_w + size_t(p) * up * cin[k] * 25, wout = N.f_out_w + size_t(p) * up * up * 25, wsc = N.f_sc_w + size_t(p) * up * cin[k];
            N.in_dense[p] = palloc(size_t(25) * up * 32); add(PK_DENSE_PAD, win, N.in_dense[p], (long long)25 * up * cin[k], up, cin[k], cin[k], 25, 32);
            N.in_T[p] = palloc(size_t(25) * 32 * up);     add(PK_DENSE_T_PAD, win, N.in_T[p], (long long)25 * up * cin[k], up, cin[k], cin[k], 25, 32);
            N.out_dense[p] = palloc(size_t(25) * up * up); add(PK_DENSE, wout, N.out_dense[p], (long long)up * up, up, up, up, 25);
            N.out_T[p] = palloc(size_t(25) * up * up);    add(PK_DENSE_T, wout, N.out_T[p], (long long)up * up, up, up, up, 25);
            N.sc_dense[p] = palloc(size_t(up) * 32);      add(PK_DENSE_PAD, wsc, N.sc_dense[p], (long long)up * cin[k], up, cin[k], cin[k], 1, 32);
            N.sc_T[p] = palloc(size_t(32) * up);          add(PK_DENSE_T_PAD, wsc, N.sc_T[p], (long long)up * cin[k], up, cin[k], cin[k], 1, 32);
        }
        const char* ln[6] = {".first_layers.0", ".first_layers.2", ".first_layers.4", ".second_layers.0", ".second_layers.2", ".second_layers.4"};
        const int I[6] = {up, hid, hid, up + hid, hid, hid}, O[6] = {hid, hid, hid, hid, hid, cout[k]};
        for (int l = 0; l < 6; ++l) {
            N.I[l] = I[l]; N.O[l] = O[l];
            N.f_mw[l] = a->off_of(ml + ln[l] + ".weight"); N.f_mb[l] = a->off_of(ml + ln[l] + ".bias");
            if (l < 5) { N.mT[l] = palloc(size_t(I[l]) * O[l]); add(PK_TRANS2D, N.f_mw[l], N.mT[l], (long long)I[l] * O[l], O[l], 0, I[l]); }
        }
    }
    a->psize = ps;
    S3D_TRY(a->pbuf.reserve(ps * sizeof(float)));
    S3D_HIP(hipMemset(a->pbuf.p, 0, ps * sizeof(float)));
    S3D_TRY(finalize_pack_plan(a->descs, a->descs_dev, a->pack_blocks));
    return 0;
}

struct AeRun {
    s3d_ae* a; hipStream_t st; float* grads;
    hipStream_t sw = nullptr;             // weight gradients; `side` says whether it is a stream of its own (the caller's stream may be the null stream)
    bool side = false;
    size_t ev_next = 0;
    Arena& ar() { return a->arena; }
    bool meas() { return a->arena.measuring; }
    int edge(hipStream_t from, hipStream_t to) {           // `to` waits for what has been enqueued on `from` so far
        if (!side || from == to) return 0;
        if (ev_next == a->events.size()) {
            hipEvent_t e = nullptr;
            S3D_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            a->events.push_back(e);
        }
        hipEvent_t e = a->events[ev_next++];
        S3D_HIP(hipEventRecord(e, from));
        S3D_HIP(hipStreamWaitEvent(to, e, 0));
        return 0;
    }
    const float* F(size_t off) const { return a->flat + off; }
    float* G(size_t off) const { return grads ? grads + off : reinterpret_cast<float*>(uintptr_t(256)); }

    int conv(ConvKind kind, int cin, int cout, const Geo& g, float* const in[3], const float* const wgt[3], const float* const bias[3],
             float* const res[3], float* const out[3], int njobs = 3, bool relu = false, hipStream_t on = nullptr, bool use_on = false) {
        if (meas()) return 0;
        ConvArgs ca; memset(&ca, 0, sizeof ca);
        ca.B = 1; ca.cin = cin; ca.cout = cout; ca.njobs = njobs; ca.relu = relu ? 1 : 0;
        for (int p = 0; p < njobs; ++p) {
            ConvJob& J = ca.job[p];
            J.in = in[p]; J.wgt = wgt[p]; J.bias = bias ? bias[p] : nullptr; J.res = res ? res[p] : nullptr; J.out = out[p];
            J.h = g.h[p]; J.w = g.w[p];
        }
        return launch_conv(kind, ca, use_on ? on : st);
    }
    // from / use_from: the stream dy was produced on when it is not the caller's (the second MLP chain)
    int wgrad(int taps, int cin, int cin_store, int cout, const Geo& g, float* const dy[3], int a_cstride, float* const act[3],
              float* const dW[3], int nplanes = 3, hipStream_t from = nullptr, bool use_from = false) {
        WgradArgs w;
        w.dy.C = cout; w.a.C = a_cstride; w.dy.g = w.a.g = g;
        for (int p = 0; p < 3; ++p) { w.dy.p[p] = p < nplanes ? dy[p] : nullptr; w.a.p[p] = p < nplanes ? act[p] : nullptr; }
        w.B = 1; w.cin = cin; w.cout = cout; w.ctot = cin_store; w.taps = taps; w.cin_store = cin_store; w.nplanes = nplanes;
        w.ksplit = wgrad_ksplit(g, 1, cin, cout, taps);
        for (int p = 0; p < nplanes; ++p) { w.part[p] = ar().alloc<float>(wgrad_part_floats(w.ksplit, cin, cout, taps)); w.dW[p] = dW[p]; }
        if (meas()) return 0;
        hipStream_t src = use_from ? from : st;
        hipStream_t on = side ? sw : src;
        S3D_TRY(edge(src, on));                            // dy and the activation are final in the order of the stream that produced them
        return launch_wgrad(w, on, &tail);                 // (the split-K partials stay in the arena; every reduction of the pass in ONE launch at its end)
    }
    DeferredTail tail;
};

// encode: projections -> pre-activation planes -> InstanceNorm + tanh.  pre/feat: NHWC [hw][CO] per plane
static int ae_encode(AeRun& R, float* pre[3], float* feat[3], float* mr) {
    s3d_ae* a = R.a;
    if (R.meas()) return 0;
    S3D_TRY(launch_enc_pack(R.F(a->f_enc_w[0]), R.F(a->f_enc_w[1]), a->geo, a->tex, a->C, static_cast<float*>(a->pbuf.p) + a->wp_off, R.st));
    S3D_TRY(launch_enc_fwd(a->enc, a->P(a->wp_off), a->P(a->encb_off), pre, R.st));
    return launch_enc_norm(false, pre, feat, mr, nullptr, nullptr, a->enc.g, a->enc.CO, nullptr, R.st);
}

// Forward + (when grads != null) backward of one batch of points.  pred: [N][1+TC] output.
static int ae_step(s3d_ae* a, const float* pts, const float* sdf, const float* tex, long long N, const float aabb[6],
                   const s3d_ae_loss_cfg& lc, float* losses, float* pred_out, float* grads, hipStream_t st) {
    AeRun R{a, st, grads};
    Arena& ar = a->arena;
    const bool meas = ar.measuring;
    ar.reset();
    if (!meas && grads && opt_on(OPT_BWD_SIDE)) {
        if (!a->side) {
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            S3D_HIP(hipStreamCreateWithPriority(&a->side, hipStreamNonBlocking, least));
        }
        if (!a->chain2) S3D_HIP(hipStreamCreateWithFlags(&a->chain2, hipStreamNonBlocking));
        R.sw = a->side;
        R.side = true;
    }
    const int up = a->up, hid = a->hid, CO = a->geo + a->tex, S = 1 + a->TC;
    const Geo g = a->enc.g;
    const long long Np = (N + 63) / 64 * 64;
    const bool bwd = grads != nullptr || meas;              // the measuring pass sizes the workspace for both
    size_t hw[3];
    for (int p = 0; p < 3; ++p) hw[p] = size_t(g.h[p]) * g.w[p];

    // ---- encode
    float *pre[3], *feat[3];
    for (int p = 0; p < 3; ++p) { pre[p] = ar.alloc<float>(hw[p] * CO); feat[p] = ar.alloc<float>(hw[p] * CO); }
    float* enc_mr = ar.alloc<float>(size_t(3) * CO * 2);
    S3D_TRY(ae_encode(R, pre, feat, enc_mr));

    // ---- plane blocks
    struct NetAct { float *x[3], *a1[3], *y[3], *s[3], *f[3], *mr; double* part[3]; } A[2];
    // (training with the side streams on: the texture net's plane blocks run as a second chain beside the geometry net's — they
    // share nothing but the encoder's feature planes — here, and again in the backward pass)
    if (!meas && R.side) S3D_TRY(R.edge(st, a->chain2));
    for (int n = 0; n < 2; ++n) {
        const AeNet& N_ = a->net[n];
        NetAct& T = A[n];
        const bool second = n == 1 && R.side && !meas;
        hipStream_t cs = second ? a->chain2 : st;
        for (int p = 0; p < 3; ++p) {
            T.x[p] = ar.alloc<float>(hw[p] * 32); T.a1[p] = ar.alloc<float>(hw[p] * up); T.y[p] = ar.alloc<float>(hw[p] * up);
            T.s[p] = ar.alloc<float>(hw[p] * up); T.f[p] = ar.alloc<float>(hw[p] * up);
            T.part[p] = ar.alloc<double>(size_t(kInNormChunks) * up * 2);
        }
        T.mr = ar.alloc<float>(size_t(3) * up * 2);
        const float *w_in[3], *b_in[3], *w_out[3], *b_out[3], *w_sc[3], *b_sc[3];
        for (int p = 0; p < 3; ++p) {
            w_in[p] = a->P(N_.in_dense[p]); b_in[p] = R.F(N_.f_in_b + size_t(p) * up);
            w_out[p] = a->P(N_.out_dense[p]); b_out[p] = R.F(N_.f_out_b + size_t(p) * up);
            w_sc[p] = a->P(N_.sc_dense[p]); b_sc[p] = R.F(N_.f_sc_b + size_t(p) * up);
        }
        if (!meas) S3D_TRY(launch_slice_pad_nhwc3(feat, T.x, hw, CO, n == 0 ? 0 : a->geo, N_.cin, cs));
        S3D_TRY(R.conv(CONV_5x5, 32, up, g, T.x, w_in, b_in, nullptr, T.a1, 3, false, cs, true));
        if (!meas) {
            const float *gam[3], *bet[3];
            for (int p = 0; p < 3; ++p) { gam[p] = R.F(N_.f_gamma[p]); bet[p] = R.F(N_.f_beta[p]); }
            S3D_TRY(launch_inorm_silu3(T.a1, T.part, gam, bet, T.y, hw, up, 1e-6f, T.mr, cs));
        }
        S3D_TRY(R.conv(CONV_1x1, 32, up, g, T.x, w_sc, b_sc, nullptr, T.s, 3, false, cs, true));
        S3D_TRY(R.conv(CONV_5x5, up, up, g, T.y, w_out, b_out, T.s, T.f, 3, false, cs, true));
    }
    if (!meas && R.side) S3D_TRY(R.edge(a->chain2, st));          // the gather reads both nets' planes
    // (measured and not kept: the backward scatter's point sort — 21 small dependent launches that depend on the points alone —
    // enqueued here on the second chain's stream, beside the MLPs' forward pass: every one of them waits for a slot behind the
    // 8192-block launches of that pass, the chain arrives late at its backward half, 5.00 -> 5.30 ms/iteration)
    PointSet ps; ps.pts = pts; ps.N = N; ps.Np = Np;
    for (int k = 0; k < 6; ++k) ps.aabb[k] = aabb[k];

    // ---- points: gather, MLPs
    float *X0[2], *H[2][5], *CAT[2];
    for (int n = 0; n < 2; ++n) {
        X0[n] = ar.alloc<float>(size_t(Np) * up);
        for (int l = 0; l < 5; ++l) H[n][l] = ar.alloc<float>(size_t(Np) * hid);
        CAT[n] = ar.alloc<float>(size_t(Np) * (up + hid));
    }
    float* pred = ar.alloc<float>(size_t(Np) * S);
    Geo gp; for (int p = 0; p < 3; ++p) { gp.h[p] = p < 2 ? int(Np / 64) : 0; gp.w[p] = p < 2 ? 64 : 0; }   // the point list as two "planes" (geo, tex)
    auto lin = [&](int l, float* const in[2], float* const out[2]) -> int {       // both groups in one launch
        const float *w[3], *b[3];
        float *i3[3], *o3[3];
        for (int n = 0; n < 2; ++n) { w[n] = R.F(a->net[n].f_mw[l]); b[n] = R.F(a->net[n].f_mb[l]); i3[n] = in[n]; o3[n] = out[n]; }
        return R.conv(CONV_1x1, a->net[0].I[l], a->net[0].O[l], gp, i3, w, b, nullptr, o3, 2, /*relu=*/true);
    };
    if (!meas) {
        const float* fp[2][3];
        for (int n = 0; n < 2; ++n) for (int p = 0; p < 3; ++p) fp[n][p] = A[n].f[p];
        S3D_TRY(launch_gather(ps, fp, g.h, g.w, up, 2, X0, st));
    }
    {
        float* h0[2] = {H[0][0], H[1][0]}; float* h1[2] = {H[0][1], H[1][1]}; float* h2[2] = {H[0][2], H[1][2]};
        float* h3[2] = {H[0][3], H[1][3]}; float* h4[2] = {H[0][4], H[1][4]};
        S3D_TRY(lin(0, X0, h0)); S3D_TRY(lin(1, h0, h1)); S3D_TRY(lin(2, h1, h2));
        if (!meas)
            for (int n = 0; n < 2; ++n) {
                S3D_TRY(launch_copy_slice(X0[n], 1, up, int(Np / 64), 64, CAT[n], up + hid, 0, st));
                S3D_TRY(launch_copy_slice(H[n][2], 1, hid, int(Np / 64), 64, CAT[n], up + hid, up, st));
            }
        S3D_TRY(lin(3, CAT, h3)); S3D_TRY(lin(4, h3, h4));
        if (!meas)
            for (int n = 0; n < 2; ++n)
                S3D_TRY(launch_last_fwd(H[n][4], R.F(a->net[n].f_mw[5]), R.F(a->net[n].f_mb[5]), hid, a->net[n].nout, n == 1, pred, S,
                                        n == 0 ? 0 : 1, N, st));
    }
    if (!meas && pred_out) S3D_HIP(hipMemcpyAsync(pred_out, pred, size_t(N) * S * sizeof(float), hipMemcpyDeviceToDevice, st));

    // ---- losses and d loss / d (pre-activation outputs)
    float* loss_ws = ar.alloc<float>(256);
    float* loss3 = ar.alloc<float>(4);
    float* dout = bwd ? ar.alloc<float>(size_t(Np) * S) : nullptr;
    if (!meas && sdf) {
        S3D_TRY(launch_ae_loss(pred, sdf, tex, N, Np, a->TC, lc.sdf_loss, lc.tex_loss, lc.sdf_threshold * lc.tex_threshold_ratio,
                               lc.tex_weight, loss_ws, loss3, grads ? dout : nullptr, st));
        if (losses) S3D_HIP(hipMemcpyAsync(losses, loss3, 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (!bwd) return 0;

    // ================================================================= backward
    // ---- MLPs
    float* dX0[2];
    float* cws = ar.alloc<float>(colsum_ws_floats(std::max(hid, up)));
    float* cws2 = ar.alloc<float>(3 * colsum_ws_floats(up));       // the plane blocks' bias sums (side stream, three planes per launch) have a workspace of their own
    // The two MLPs' backward chains are independent until the scatter: the texture net's runs on a second stream beside the
    // geometry net's (side-stream builds only) — each alternates a bandwidth-bound ReLU backward with an MFMA-bound 1x1 dgrad, so
    // the two chains want different things at most times.  Same kernels on the same operands: same bits.
    float* cws_n[2] = {cws, ar.alloc<float>(colsum_ws_floats(std::max(hid, up)))};
    if (!meas && R.side) S3D_TRY(R.edge(st, a->chain2));
    // (issued layer by layer for BOTH nets, not net after net: the two chains then advance together however far the host is ahead)
    struct Step { int l; float* act; float* in; int in_stride; float* din; const float* dact; int dstride, coff; };
    Step steps[2][5];
    float *dXa_n[2], *dCAT_n[2];
    hipStream_t cs_n[2];
    for (int n = 0; n < 2; ++n) {
        const AeNet& N_ = a->net[n];
        cs_n[n] = n == 1 && R.side && !meas ? a->chain2 : st;      // this net's chain
        float* dH = ar.alloc<float>(size_t(Np) * hid);            // gradient of a hidden activation (reused)
        float* dCAT = ar.alloc<float>(size_t(Np) * (up + hid));
        float* dXa = ar.alloc<float>(size_t(Np) * up);
        dX0[n] = ar.alloc<float>(size_t(Np) * up);
        dXa_n[n] = dXa; dCAT_n[n] = dCAT;
        float* lws = ar.alloc<float>(last_bwd_ws_floats(hid, N_.nout));
        if (!meas) S3D_TRY(launch_last_bwd(dout, S, n == 0 ? 0 : 1, R.F(N_.f_mw[5]), H[n][4], hid, N_.nout, Np, dH, lws, R.G(N_.f_mw[5]), R.G(N_.f_mb[5]), cs_n[n]));
        // hidden layers 4..0: dP = dH * relu'(H_l) ; db_l = colsum(dP) ; dW_l = dP^T in_l ; d in_l = dP W_l
        const Step st5[5] = {{4, H[n][4], H[n][3], hid, dH, dH, hid, 0},
                             {3, H[n][3], CAT[n], up + hid, dCAT, dH, hid, 0},
                             {2, H[n][2], H[n][1], hid, dH, dCAT, up + hid, up},
                             {1, H[n][1], H[n][0], hid, dH, dH, hid, 0},
                             {0, H[n][0], X0[n], up, dXa, dH, hid, 0}};
        for (int k = 0; k < 5; ++k) steps[n][k] = st5[k];
    }
    for (int kk = 0; kk < 10; ++kk) {
            const int k = kk / 2, n = kk % 2;
            const AeNet& N_ = a->net[n];
            const Step& s = steps[n][k];
            hipStream_t cs = cs_n[n];
            const int I = N_.I[s.l], O = N_.O[s.l];
            // gradient of the layer's pre-activation: one buffer per layer — the weight gradient that reads it runs on the side
            // stream while this chain has moved on
            float* dP = ar.alloc<float>(size_t(Np) * hid);
            if (!meas) S3D_TRY(launch_relu_bwd(s.dact, s.dstride, s.coff, s.act, dP, Np, O, cws_n[n], R.G(N_.f_mb[s.l]), cs));
            Geo g1; for (int p = 0; p < 3; ++p) { g1.h[p] = p == 0 ? int(Np / 64) : 0; g1.w[p] = p == 0 ? 64 : 0; }
            float* dy3[3] = {dP, nullptr, nullptr}; float* a3[3] = {s.in, nullptr, nullptr}; float* dw3[3] = {R.G(N_.f_mw[s.l]), nullptr, nullptr};
            S3D_TRY(R.wgrad(1, I, I, O, g1, dy3, s.in_stride, a3, dw3, 1, cs, true));
            const float* wT[3] = {a->P(N_.mT[s.l]), nullptr, nullptr};
            float* o3[3] = {s.din, nullptr, nullptr};
            S3D_TRY(R.conv(CONV_1x1, O, I, g1, dy3, wT, nullptr, nullptr, o3, 1, false, cs, true));
        }
    for (int n = 0; n < 2; ++n)
        if (!meas) S3D_TRY(launch_add_slice(dXa_n[n], dCAT_n[n], up + hid, 0, dX0[n], Np, up, cs_n[n]));
    if (!meas && R.side) S3D_TRY(R.edge(a->chain2, st));          // the scatter reads both nets' point gradients
    // ---- scatter the point gradients onto the planes
    float* dF[2][3];
    for (int n = 0; n < 2; ++n) for (int p = 0; p < 3; ++p) dF[n][p] = ar.alloc<float>(hw[p] * up);
    char* sws = ar.alloc<char>(scatter_ws_bytes(Np, g.h, g.w));
    if (!meas) {
        const float* dx[2] = {dX0[0], dX0[1]};
        S3D_TRY(launch_scatter(ps, dF, g.h, g.w, up, 2, dx, sws, st));
    }
    // ---- plane blocks
    float* dfeat[3];
    for (int p = 0; p < 3; ++p) dfeat[p] = ar.alloc<float>(hw[p] * CO);
    if (!meas && R.side) S3D_TRY(R.edge(st, a->chain2));          // (the scatter's plane gradients are final)
    for (int n = 0; n < 2; ++n) {
        const AeNet& N_ = a->net[n];
        NetAct& T = A[n];
        const bool second = n == 1 && R.side && !meas;
        hipStream_t cs = second ? a->chain2 : st;                  // this net's chain (see the forward pass)
        float *d_y[3], *d_xs[3], *d_a1[3], *d_x[3];
        for (int p = 0; p < 3; ++p) {
            d_y[p] = ar.alloc<float>(hw[p] * up); d_xs[p] = ar.alloc<float>(hw[p] * 32);
            d_a1[p] = ar.alloc<float>(hw[p] * up); d_x[p] = ar.alloc<float>(hw[p] * 32);
        }
        const float *w_outT[3], *w_scT[3], *w_inT[3];
        float *dw_out[3], *dw_sc[3], *dw_in[3];
        for (int p = 0; p < 3; ++p) {
            w_outT[p] = a->P(N_.out_T[p]); w_scT[p] = a->P(N_.sc_T[p]); w_inT[p] = a->P(N_.in_T[p]);
            dw_out[p] = R.G(N_.f_out_w + size_t(p) * up * up * 25); dw_sc[p] = R.G(N_.f_sc_w + size_t(p) * up * N_.cin);
            dw_in[p] = R.G(N_.f_in_w + size_t(p) * up * N_.cin * 25);
        }
        // bias gradients are column sums nobody inside the pass reads: side stream (the workspace `cws2` belongs to it in this phase)
        hipStream_t sb = R.side ? R.sw : st;
        if (!meas) {
            S3D_TRY(R.edge(cs, sb));
            float *ob[3], *sb2[3];                     // out conv and shortcut share dy: identical bias gradients, one pass, two outputs
            for (int p = 0; p < 3; ++p) { ob[p] = R.G(N_.f_out_b + size_t(p) * up); sb2[p] = R.G(N_.f_sc_b + size_t(p) * up); }
            S3D_TRY(launch_colsum3(dF[n], hw, up, cws2, ob, sb2, sb));
        }
        S3D_TRY(R.conv(CONV_5x5, up, up, g, dF[n], w_outT, nullptr, nullptr, d_y, 3, false, cs, true));
        S3D_TRY(R.wgrad(25, up, up, up, g, dF[n], up, T.y, dw_out, 3, cs, true));
        S3D_TRY(R.conv(CONV_1x1, up, 32, g, dF[n], w_scT, nullptr, nullptr, d_xs, 3, false, cs, true));
        S3D_TRY(R.wgrad(1, 32, N_.cin, up, g, dF[n], 32, T.x, dw_sc, 3, cs, true));
        {   // InstanceNorm(affine) + SiLU backward = GroupNorm backward with one channel per group
            GnActBwd s;
            s.x.C = s.dy.C = s.dx.C = up; s.x.g = s.dy.g = s.dx.g = g;
            for (int p = 0; p < 3; ++p) {
                s.x.p[p] = T.a1[p]; s.dy.p[p] = d_y[p]; s.dx.p[p] = d_a1[p];
                s.gamma[p] = R.F(N_.f_gamma[p]); s.beta[p] = R.F(N_.f_beta[p]); s.dgamma[p] = R.G(N_.f_gamma[p]); s.dbeta[p] = R.G(N_.f_beta[p]);
            }
            s.rowadd = nullptr; s.coladd = nullptr; s.add = nullptr; s.stats.mr = T.mr; s.film = nullptr; s.dfilm = nullptr; s.film_stride = 0;
            s.B = 1; s.ngroups = up;
            s.ws = ar.alloc<float>(gn_bwd_ws_floats(1, up));
            if (!meas) S3D_TRY(launch_gn_act_bwd(s, cs));
        }
        if (!meas) {
            S3D_TRY(R.edge(cs, sb));                    // d_a1 is final
            float* ib[3];
            for (int p = 0; p < 3; ++p) ib[p] = R.G(N_.f_in_b + size_t(p) * up);
            S3D_TRY(launch_colsum3(d_a1, hw, up, cws2, ib, nullptr, sb));
        }
        S3D_TRY(R.wgrad(25, 32, N_.cin, up, g, d_a1, 32, T.x, dw_in, 3, cs, true));
        S3D_TRY(R.conv(CONV_5x5, up, 32, g, d_a1, w_inT, nullptr, d_xs, d_x, 3, false, cs, true));
        if (!meas)
            S3D_TRY(launch_unslice_nhwc3(d_x, dfeat, hw, CO, n == 0 ? 0 : a->geo, N_.cin, cs));
    }
    if (!meas && R.side) S3D_TRY(R.edge(a->chain2, st));          // the encoder's backward reads both nets' slices of dfeat
    // ---- encoder
    float* dpre[3];
    for (int p = 0; p < 3; ++p) dpre[p] = ar.alloc<float>(hw[p] * CO);
    float* ews = ar.alloc<float>(std::max(enc_wgrad_ws_floats(a->C, CO), enc_norm_bwd_ws_bytes(CO) / sizeof(float)));
    if (!meas) {
        S3D_TRY(launch_enc_norm(true, pre, feat, enc_mr, dfeat, dpre, g, CO, ews, st));        // (its chunk sums are consumed before the weight gradient reuses ews)
        S3D_TRY(launch_enc_wgrad(a->enc, dpre, a->geo, a->tex, ews, R.G(a->f_enc_w[0]), R.G(a->f_enc_b[0]), R.G(a->f_enc_w[1]),
                                 R.G(a->f_enc_b[1]), st));
        S3D_TRY(R.tail.flush(R.side ? R.sw : st));        // the 16 weight-gradient reductions (they were 16 launches on the stream the iteration waits for)
        S3D_TRY(R.edge(R.sw, st));                        // every gradient of the pass is final in the order of the caller's stream (no-op in line)
    }
    return 0;
}

static int ae_run(s3d_ae* a, const float* pts, const float* sdf, const float* tex, long long N, const float aabb[6],
                  const s3d_ae_loss_cfg& lc, float* losses, float* pred, float* grads, hipStream_t st) {
    a->arena.measuring = true; a->arena.high = 0;
    int rc = ae_step(a, pts, sdf, tex, N, aabb, lc, losses, pred, grads, st);
    a->arena.measuring = false;
    if (rc) return rc;
    if (a->arena.high > a->arena.buf.cap) {
        S3D_HIP(hipStreamSynchronize(st));
        S3D_TRY(a->arena.buf.reserve(a->arena.high + (a->arena.high >> 3)));
    }
    return ae_step(a, pts, sdf, tex, N, aabb, lc, losses, pred, grads, st);
}

}  // namespace s3d

extern "C" {

int s3d_ae_create(const s3d_decoder_cfg* cfg, s3d_ae** out) {
    S3D_CHECK(cfg && out, S3D_ERR_INVALID, "ae_create: null argument");
    S3D_CHECK(cfg->mlp_hidden_layers == 4, S3D_ERR_UNSUPPORTED, "ae: mlp_hidden_layers=%d (only the default 4 is built)", cfg->mlp_hidden_layers);
    S3D_CHECK(cfg->feat_channel_up % 32 == 0 && cfg->mlp_hidden_channels % 32 == 0 && cfg->feat_channel_up > 0 && cfg->mlp_hidden_channels > 0,
              S3D_ERR_UNSUPPORTED, "ae training: feat_channel_up=%d and mlp_hidden_channels=%d must be multiples of 32",
              cfg->feat_channel_up, cfg->mlp_hidden_channels);
    S3D_CHECK(cfg->geo_feat_channels >= 1 && cfg->tex_feat_channels >= 1 && cfg->geo_feat_channels + cfg->tex_feat_channels <= 12 &&
              cfg->geo_feat_channels % 4 == 0 && cfg->tex_feat_channels % 4 == 0,
              S3D_ERR_UNSUPPORTED, "ae training: feature groups must be multiples of 4 channels, at most 12 together");
    S3D_CHECK(cfg->tex_channels >= 1 && cfg->tex_channels <= 3, S3D_ERR_UNSUPPORTED, "ae: tex_channels must be 1..3");
    std::unique_ptr<s3d_ae> a(new s3d_ae());
    a->cfg = *cfg;
    a->geo = cfg->geo_feat_channels; a->tex = cfg->tex_feat_channels; a->up = cfg->feat_channel_up; a->hid = cfg->mlp_hidden_channels;
    a->TC = cfg->tex_channels; a->C = 1 + cfg->tex_channels;
    ae_specs(a.get());
    *out = a.release();
    return 0;
}
void s3d_ae_destroy(s3d_ae* a) { delete a; }
int s3d_ae_num_params(const s3d_ae* a) { return a ? int(a->specs.size()) : S3D_ERR_INVALID; }
int s3d_ae_param_info(const s3d_ae* a, int i, const char** name, int64_t shape[5], int* ndim, int64_t* offset) {
    S3D_CHECK(a && i >= 0 && i < int(a->specs.size()), S3D_ERR_INVALID, "ae param_info: index %d out of range", i);
    if (name) *name = a->specs[i].name.c_str();
    if (ndim) *ndim = int(a->specs[i].shape.size());
    if (shape) for (size_t k = 0; k < a->specs[i].shape.size(); ++k) shape[k] = a->specs[i].shape[k];
    if (offset) *offset = int64_t(a->flat_off[i]);
    return 0;
}
int64_t s3d_ae_param_numel(const s3d_ae* a, int64_t* tex_group_begin) {
    if (!a) return S3D_ERR_INVALID;
    if (tex_group_begin) *tex_group_begin = a->tex_begin;
    return int64_t(a->flat_off.back() + a->specs.back().numel());
}
int s3d_ae_attach(s3d_ae* a, float* params, int64_t numel) {
    S3D_CHECK(a && params, S3D_ERR_INVALID, "ae_attach: null argument");
    S3D_CHECK(numel == s3d_ae_param_numel(a, nullptr), S3D_ERR_INVALID, "ae_attach: %lld floats given, the model has %lld",
              (long long)numel, (long long)s3d_ae_param_numel(a, nullptr));
    a->flat = params;
    return ae_plan(a);
}
int s3d_ae_repack(s3d_ae* a, void* stream) {
    S3D_CHECK(a && a->flat, S3D_ERR_INVALID, "ae_repack: call s3d_ae_attach first");
    return launch_repack_generic(static_cast<const PackDesc*>(a->descs_dev.p), int(a->descs.size()), a->pack_blocks, a->flat,
                                 static_cast<float*>(a->pbuf.p), static_cast<float*>(a->pbuf.p), static_cast<hipStream_t>(stream));
}
int s3d_ae_set_volume(s3d_ae* a, const float* vol, int C, int X2, int Y2, int Z2, void* stream) {
    S3D_CHECK(a && vol, S3D_ERR_INVALID, "ae_set_volume: null argument");
    S3D_CHECK(C == a->C, S3D_ERR_INVALID, "ae_set_volume: %d channels, the encoder takes %d (sdf + texture)", C, a->C);
    S3D_CHECK(X2 >= 2 && Y2 >= 2 && Z2 >= 2 && X2 % 2 == 0 && Y2 % 2 == 0 && Z2 % 2 == 0, S3D_ERR_INVALID,
              "ae_set_volume: the volume must be twice the feature-map size, got (%d,%d,%d)", X2, Y2, Z2);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n[3] = {size_t(X2) * Y2 * 4 * C, size_t(X2) * Z2 * 4 * C, size_t(Y2) * Z2 * 4 * C};
    S3D_HIP(hipStreamSynchronize(st));
    S3D_TRY(a->proj.reserve((n[0] + n[1] + n[2]) * sizeof(float)));
    float* P[3] = {static_cast<float*>(a->proj.p), static_cast<float*>(a->proj.p) + n[0], static_cast<float*>(a->proj.p) + n[0] + n[1]};
    S3D_TRY(launch_project(vol, C, X2, Y2, Z2, P, st));
    const int H = X2 / 2, W = Y2 / 2, D = Z2 / 2;
    a->enc.g = Geo::from_hwd(H, W, D);
    for (int p = 0; p < 3; ++p) a->enc.P[p] = P[p];
    a->enc.inv_len[0] = 1.f / D; a->enc.inv_len[1] = 1.f / W; a->enc.inv_len[2] = 1.f / H;
    a->enc.C = C; a->enc.CO = a->geo + a->tex;
    a->has_volume = true;
    return 0;
}
int s3d_ae_encode(s3d_ae* a, float* xy, float* xz, float* yz, void* stream) {
    S3D_CHECK(a && xy && xz && yz, S3D_ERR_INVALID, "ae_encode: null argument");
    S3D_CHECK(a->flat && a->has_volume, S3D_ERR_INVALID, "ae_encode: attach parameters and set the volume first");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Geo g = a->enc.g; const int CO = a->enc.CO;
    float* out[3] = {xy, xz, yz};
    for (int pass = 0; pass < 2; ++pass) {
        Arena& ar = a->arena;
        ar.measuring = pass == 0;
        if (pass == 0) ar.high = 0;
        else if (ar.high > ar.buf.cap) { S3D_HIP(hipStreamSynchronize(st)); S3D_TRY(ar.buf.reserve(ar.high)); }
        ar.reset();
        float *pre[3], *feat[3];
        for (int p = 0; p < 3; ++p) { pre[p] = ar.alloc<float>(size_t(g.h[p]) * g.w[p] * CO); feat[p] = ar.alloc<float>(size_t(g.h[p]) * g.w[p] * CO); }
        float* mr = ar.alloc<float>(size_t(3) * CO * 2);
        AeRun R{a, st, nullptr};
        S3D_TRY(ae_encode(R, pre, feat, mr));
        if (pass == 1) for (int p = 0; p < 3; ++p) S3D_TRY(launch_nhwc_to_nchw(feat[p], out[p], 1, CO, g.h[p], g.w[p], st));
    }
    a->arena.measuring = false;
    return 0;
}
int s3d_ae_forward(s3d_ae* a, const float* pts, int64_t N, const float aabb[6], float* pred, void* stream) {
    S3D_CHECK(a && pts && aabb && pred && N >= 1, S3D_ERR_INVALID, "ae_forward: bad argument");
    S3D_CHECK(a->flat && a->has_volume, S3D_ERR_INVALID, "ae_forward: attach parameters and set the volume first");
    s3d_ae_loss_cfg lc{};
    return ae_run(a, pts, nullptr, nullptr, N, aabb, lc, nullptr, pred, nullptr, static_cast<hipStream_t>(stream));
}
int s3d_ae_loss_grads(s3d_ae* a, const float* pts, const float* sdf, const float* tex, int64_t N, const float aabb[6],
                      const s3d_ae_loss_cfg* cfg, float* losses, float* pred, float* grads, void* stream) {
    S3D_CHECK(a && pts && sdf && tex && aabb && cfg && losses && grads && N >= 1, S3D_ERR_INVALID, "ae_loss_grads: bad argument");
    S3D_CHECK(a->flat && a->has_volume, S3D_ERR_INVALID, "ae_loss_grads: attach parameters and set the volume first");
    S3D_CHECK(cfg->sdf_loss >= 0 && cfg->sdf_loss <= 1 && cfg->tex_loss >= 0 && cfg->tex_loss <= 2, S3D_ERR_INVALID, "ae_loss_grads: unknown loss type");
    return ae_run(a, pts, sdf, tex, N, aabb, *cfg, losses, pred, grads, static_cast<hipStream_t>(stream));
}

}  // extern "C"
