// s3d_unet.hip — host side of the denoiser: parameter registry, weight repacking and the kernel
// sequence of TriplaneUNetModelSmall.forward (src/diffusion/unet_triplane.py:465-510).
#include "s3d_model.h"


namespace s3d {

static void add_lin(std::vector<ParamSpec>& s, const std::string& p, int o, int i) {
    s.push_back({p + ".weight", {o, i}});
    s.push_back({p + ".bias", {o}});
}
static void add_tconv(std::vector<ParamSpec>& s, const std::string& p, int cin, int cout, int k, bool roll) {
    for (int i = 0; i < 3; ++i) {
        s.push_back({p + ".conv_" + kPlane[i] + ".weight", {cout, roll ? 3 * cin : cin, k, k}});
        s.push_back({p + ".conv_" + kPlane[i] + ".bias", {cout}});
    }
}
static void add_tnorm(std::vector<ParamSpec>& s, const std::string& p, int c) {
    for (int i = 0; i < 3; ++i) {
        s.push_back({p + ".norm_" + kPlane[i] + ".weight", {c}});
        s.push_back({p + ".norm_" + kPlane[i] + ".bias", {c}});
    }
}
static void add_resblock(std::vector<ParamSpec>& s, const std::string& p, int c, int cout, int ted, bool ssn, bool roll) {
    add_tnorm(s, p + ".in_layers.0", c);
    add_tconv(s, p + ".in_layers.2", c, cout, 3, roll);
    add_lin(s, p + ".emb_layers.1", ssn ? 2 * cout : cout, ted);
    add_tnorm(s, p + ".out_layers.0", cout);
    add_tconv(s, p + ".out_layers.2", cout, cout, 3, roll);
    if (c != cout) add_tconv(s, p + ".skip_connection", c, cout, 1, false);
}

// Mirrors TriplaneUNetModelSmall.__init__ (src/diffusion/unet_triplane.py:346-449).
static int build_specs(s3d_unet* m) {
    const s3d_unet_cfg& c = m->cfg;
    const int mc = c.model_channels, ted = 4 * mc;
    const bool ssn = c.use_scale_shift_norm != 0, roll = c.is_rollout != 0;
    auto& s = m->specs;
    add_lin(s, "time_embed.0", ted, mc);
    add_lin(s, "time_embed.2", ted, ted);
    int ch = c.channel_mult[0] * mc;
    add_tconv(s, "in_conv.0", c.in_channels, ch, 1, false);
    std::vector<int> chans{ch};
    int film = 0;
    for (int level = 0; level < c.n_levels; ++level) {
        const int cout = c.channel_mult[level] * mc;
        ResBlockW rb;
        rb.prefix = "input_blocks." + std::to_string(level) + "." + std::to_string(level == 0 ? 0 : 1);
        rb.C = ch; rb.Cout = cout; rb.has_skip = ch != cout; rb.film_off = film;
        film += ssn ? 2 * cout : cout;
        add_resblock(s, rb.prefix, ch, cout, ted, ssn, roll);
        m->in_blocks.push_back(rb);
        ch = cout;
        chans.push_back(ch);
    }
    for (int oi = 0; oi < c.n_levels; ++oi) {
        const int level = c.n_levels - 1 - oi;
        int ich = chans.back(); chans.pop_back();
        if (oi == 0) ich = 0;
        const int cout = c.channel_mult[level] * mc;
        ResBlockW rb;
        rb.prefix = "output_blocks." + std::to_string(oi) + ".0";
        rb.C = ch + ich; rb.Cout = cout; rb.has_skip = rb.C != cout; rb.film_off = film;
        rb.c_up = ich > 0 ? ch : 0;
        film += ssn ? 2 * cout : cout;
        add_resblock(s, rb.prefix, rb.C, cout, ted, ssn, roll);
        m->out_blocks.push_back(rb);
        ch = cout;
    }
    m->film_total = film;
    add_tnorm(s, "out.0", ch);
    add_tconv(s, "out.2", c.channel_mult[0] * mc, c.out_channels, 1, false);
    return 0;
}

// ------------------------------------------------------------------ packing
size_t push(std::vector<float>& st, const float* src, size_t n) {
    size_t off = (st.size() + 63) & ~size_t(63);      // 256-byte aligned starts
    st.resize(off + n);
    if (src) memcpy(st.data() + off, src, n * sizeof(float));
    return off;
}
static const std::vector<float>& H(const s3d_unet* m, const std::string& name) { return m->host.at(name); }

// (taps of the summed-out axis that stay inside the image, by edge variant — edge_variant in s3d_conv.hip: interior
// {1,1,1}, first {0,1,1}, last {1,1,0}, single {0,1,0}; k_rank1<true> forms the four sums from the per-tap products)

// Pack one TriplaneConv.  Rollout channel blocks (src/diffusion/unet_triplane.py:37-46):
//   plane xy: A = mean_d(yz) varies along columns (w), B = mean_d(xz) varies along rows (h)
//   plane xz: A = mean_w(xy) varies along rows (h),    B = mean_w(yz) varies along columns (d)
//   plane yz: A = mean_h(xy) varies along rows (w),    B = mean_h(xz) varies along columns (d)
void pack_tconv_raw(std::vector<float>& stage, const float* const Wp[3], const float* const bp[3], int cin, int cout,
                    int k, bool roll, ConvW& cw) {
    cw.cin = cin; cw.cout = cout; cw.k = k; cw.rollout = roll;
    const int taps = k * k, ctot = roll ? 3 * cin : cin;
    for (int p = 0; p < 3; ++p) {
        const float* W = Wp[p];                                   // [cout][ctot][k][k]
        cw.bias[p] = push(stage, bp[p], cout);
        cw.dense[p] = push(stage, nullptr, size_t(taps) * cout * cin);
        float* d = stage.data() + cw.dense[p];
        for (int t = 0; t < taps; ++t)
            for (int co = 0; co < cout; ++co)
                for (int c = 0; c < cin; ++c) d[(size_t(t) * cout + co) * cin + c] = W[(size_t(co) * ctot + c) * taps + t];
        if (k == 3) cw.wino[p] = pack_wino_weights(stage, W, cout, ctot, cin);
        if (k == 3 && cin % 32 == 0) {
            cw.wino24s[p] = pack_wino24s_weights(stage, W, cout, ctot, cin);
        }
        if (!roll) continue;
        const bool a_is_col = (p == 0);          // slot A column-varying only for xy; slot B is the other kind
        for (int slot = 1; slot <= 2; ++slot) {
            const bool col_varying = (slot == 1) ? a_is_col : !a_is_col;
            // per tap t along the vector's own axis and tap o of the summed-out axis: row (co/8)*24 + o*8 + co%8 (k_rank1<true>)
            const int n3 = (cout + 7) / 8 * 24;
            size_t off = push(stage, nullptr, size_t(3) * n3 * cin);
            float* r = stage.data() + off;
            std::fill(r, r + size_t(3) * n3 * cin, 0.f);
            for (int t = 0; t < 3; ++t)
                for (int o = 0; o < 3; ++o)
                    for (int co = 0; co < cout; ++co) {
                        const int kh = col_varying ? o : t, kw = col_varying ? t : o;
                        float* dst = r + (size_t(t) * n3 + (co / 8) * 24 + o * 8 + (co & 7)) * cin;
                        for (int c = 0; c < cin; ++c) dst[c] = W[(size_t(co) * ctot + slot * cin + c) * 9 + kh * 3 + kw];
                    }
            if (col_varying) cw.rcol[p] = off; else cw.rrow[p] = off;
            if (cin % 128 == 0) {                            // the same values in the fragment order of k_rank1b (batch >= 2)
                const size_t off_f = push(stage, nullptr, rank1_frag_floats(cin, cout));
                float* f = stage.data() + off_f;
                std::fill(f, f + rank1_frag_floats(cin, cout), 0.f);
                const float* r2 = stage.data() + off;        // (push may have moved the staging image)
                for (int t = 0; t < 3; ++t)
                    for (int o = 0; o < 3; ++o)
                        for (int co = 0; co < cout; ++co) {
                            const float* src = r2 + (size_t(t) * n3 + (co / 8) * 24 + o * 8 + (co & 7)) * cin;
                            for (int c = 0; c < cin; ++c) f[rank1_frag_index(cin, t, o, co, c)] = src[c];
                        }
                if (col_varying) cw.rcol_f[p] = off_f; else cw.rrow_f[p] = off_f;
            }
        }
    }
}
static void pack_tconv(s3d_unet* m, const std::string& prefix, int cin, int cout, int k, bool roll, ConvW& cw) {
    const float* W[3]; const float* bv[3];
    for (int p = 0; p < 3; ++p) {
        W[p] = H(m, prefix + ".conv_" + kPlane[p] + ".weight").data();
        bv[p] = H(m, prefix + ".conv_" + kPlane[p] + ".bias").data();
    }
    pack_tconv_raw(m->stage, W, bv, cin, cout, k, roll, cw);
}
static void pack_norm(s3d_unet* m, const std::string& prefix, int c, NormW& nw) {
    for (int p = 0; p < 3; ++p) {
        nw.gamma[p] = push(m->stage, H(m, prefix + ".norm_" + kPlane[p] + ".weight").data(), c);
        nw.beta[p] = push(m->stage, H(m, prefix + ".norm_" + kPlane[p] + ".bias").data(), c);
    }
}

int pack_all(s3d_unet* m) {
    for (const auto& sp : m->specs)
        S3D_CHECK(m->host.count(sp.name), S3D_ERR_MISSING, "parameter '%s' was never set (load_state_dict incomplete)", sp.name.c_str());
    const s3d_unet_cfg& c = m->cfg;
    const int mc = c.model_channels, ted = 4 * mc;
    const bool ssn = c.use_scale_shift_norm != 0, roll = c.is_rollout != 0;
    m->stage.clear();
    m->te0_w = push(m->stage, H(m, "time_embed.0.weight").data(), size_t(ted) * mc);
    m->te0_b = push(m->stage, H(m, "time_embed.0.bias").data(), ted);
    m->te2_w = push(m->stage, H(m, "time_embed.2.weight").data(), size_t(ted) * ted);
    m->te2_b = push(m->stage, H(m, "time_embed.2.bias").data(), ted);
    // all emb_layers.1 stacked into one [film_total][4mc] matrix -> one launch per forward
    m->film_w = push(m->stage, nullptr, size_t(m->film_total) * ted);
    m->film_b = push(m->stage, nullptr, m->film_total);
    auto pack_block = [&](ResBlockW& rb) {
        const int eo = ssn ? 2 * rb.Cout : rb.Cout;
        memcpy(m->stage.data() + m->film_w + size_t(rb.film_off) * ted, H(m, rb.prefix + ".emb_layers.1.weight").data(),
               size_t(eo) * ted * sizeof(float));
        memcpy(m->stage.data() + m->film_b + rb.film_off, H(m, rb.prefix + ".emb_layers.1.bias").data(), eo * sizeof(float));
        pack_norm(m, rb.prefix + ".in_layers.0", rb.C, rb.n1);
        pack_tconv(m, rb.prefix + ".in_layers.2", rb.C, rb.Cout, 3, roll, rb.c1);
        pack_norm(m, rb.prefix + ".out_layers.0", rb.Cout, rb.n2);
        pack_tconv(m, rb.prefix + ".out_layers.2", rb.Cout, rb.Cout, 3, roll, rb.c2);
        if (rb.has_skip) pack_tconv(m, rb.prefix + ".skip_connection", rb.C, rb.Cout, 1, false, rb.skip);
        if (rb.has_skip && rb.c_up > 0) {                       // the 1x1 skip split by input half (Fwd::resblock_cat)
            const int cs = rb.C - rb.c_up;
            rb.skip_a = ConvW(); rb.skip_b = ConvW();
            rb.skip_a.cin = rb.c_up; rb.skip_b.cin = cs;
            rb.skip_a.cout = rb.skip_b.cout = rb.Cout; rb.skip_a.k = rb.skip_b.k = 1;
            for (int p = 0; p < 3; ++p) {
                const float* W = H(m, rb.prefix + ".skip_connection.conv_" + kPlane[p] + ".weight").data();     // [cout][C]
                rb.skip_a.dense[p] = push(m->stage, nullptr, size_t(rb.Cout) * rb.c_up);
                rb.skip_b.dense[p] = push(m->stage, nullptr, size_t(rb.Cout) * cs);
                float* da = m->stage.data() + rb.skip_a.dense[p];
                float* db = m->stage.data() + rb.skip_b.dense[p];
                for (int co = 0; co < rb.Cout; ++co) {
                    memcpy(da + size_t(co) * rb.c_up, W + size_t(co) * rb.C, rb.c_up * sizeof(float));
                    memcpy(db + size_t(co) * cs, W + size_t(co) * rb.C + rb.c_up, cs * sizeof(float));
                }
                rb.skip_a.bias[p] = rb.skip_b.bias[p] = rb.skip.bias[p];
            }
        }
    };
    for (auto& rb : m->in_blocks) pack_block(rb);
    for (auto& rb : m->out_blocks) pack_block(rb);
    // in_conv: transposed [3][Cin][Cout]
    {
        const int ci = c.in_channels, co = c.channel_mult[0] * mc;
        m->in_wT = push(m->stage, nullptr, size_t(3) * ci * co);
        m->in_b = push(m->stage, nullptr, size_t(3) * co);
        for (int p = 0; p < 3; ++p) {
            const auto& W = H(m, std::string("in_conv.0.conv_") + kPlane[p] + ".weight");
            const auto& bv = H(m, std::string("in_conv.0.conv_") + kPlane[p] + ".bias");
            for (int i = 0; i < ci; ++i)
                for (int o = 0; o < co; ++o) m->stage[m->in_wT + (size_t(p) * ci + i) * co + o] = W[size_t(o) * ci + i];
            memcpy(m->stage.data() + m->in_b + size_t(p) * co, bv.data(), co * sizeof(float));
        }
    }
    {
        const int ci = c.channel_mult[0] * mc, co = c.out_channels;
        pack_norm(m, "out.0", ci, m->out_norm);
        m->out_w = push(m->stage, nullptr, size_t(3) * co * ci);
        m->out_b = push(m->stage, nullptr, size_t(3) * co);
        for (int p = 0; p < 3; ++p) {
            memcpy(m->stage.data() + m->out_w + size_t(p) * co * ci, H(m, std::string("out.2.conv_") + kPlane[p] + ".weight").data(),
                   size_t(co) * ci * sizeof(float));
            memcpy(m->stage.data() + m->out_b + size_t(p) * co, H(m, std::string("out.2.conv_") + kPlane[p] + ".bias").data(),
                   co * sizeof(float));
        }
    }
    S3D_TRY(upload(m->wbuf, m->stage.data(), m->stage.size() * sizeof(float)));
    m->stage.clear(); m->stage.shrink_to_fit();
    m->packed = true;
    return 0;
}

// ------------------------------------------------------------------ forward

int run_forward(s3d_unet* m, const float* x, const float* t, int B, int H, int W, int D, float* out, hipStream_t st,
                Tape* tape, const float* ext_film, int ext_film_stride, const s3d_sampler_args* fuse, int carry_flags) {
    const s3d_unet_cfg& c = m->cfg;
    const int mc = c.model_channels, ted = 4 * mc;
    Fwd f{m, B, st, nullptr};
    f.tape = tape;
    Arena& ar = m->arena;
    const bool meas = ar.measuring;
    ar.reset();
    if (tape) { *tape = Tape(); tape->B = B; tape->H = H; tape->W = W; tape->D = D; tape->x = x; tape->t = t; }

    // emb = time_embed(timestep_embedding(t)) ; all blocks' emb_layers in one stacked linear
    float* e1 = ar.alloc<float>(size_t(B) * ted);
    float* emb = ar.alloc<float>(size_t(B) * ted);
    float* film = ar.alloc<float>(size_t(B) * m->film_total);
    f.film = film;
    if (ext_film) { f.film = ext_film; f.film_stride = ext_film_stride; }
    if (!meas && !ext_film) {
        // training keeps the pre-activation of time_embed.0 (SiLU is applied on the way into the next layer instead)
        S3D_TRY(launch_linear(t, B, mc, m->dev(m->te0_w), m->dev(m->te0_b), ted, e1, 2, tape ? 0 : 1, st));
        S3D_TRY(launch_linear(e1, B, ted, m->dev(m->te2_w), m->dev(m->te2_b), ted, emb, tape ? 1 : 0, 0, st));
        S3D_TRY(launch_linear(emb, B, ted, m->dev(m->film_w), m->dev(m->film_b), m->film_total, film, 1, 0, st));
    }
    if (tape) { tape->pre1 = e1; tape->emb = emb; tape->film = film; }

    Geo g0 = Geo::from_hwd(H, W, D);
    Tri h = f.alloc_tri(c.channel_mult[0] * mc, g0);
    // The previous fused step's output head may have left THIS step's in_conv here (s3d_unet_step_film_carry; the workspace is
    // bump-allocated in a fixed order, so the tensor and its GroupNorm partials sit where this forward allocates them): taken when
    // the caller vouches for the input (S3D_CARRY_IN) and it is that step's sample, same shape, same lane, nothing in between.
    const long long ckey[4] = {B, H, W, D};
    const bool take = !meas && !tape && (carry_flags & S3D_CARRY_IN) && m->carry.valid && m->carry.sample == x &&
                      m->carry.key[0] == ckey[0] && m->carry.key[1] == ckey[1] && m->carry.key[2] == ckey[2] && m->carry.key[3] == ckey[3];
    if (!meas) m->carry.valid = false;                             // consumed, or void: one step only
    Tri h0 = h;
    GnPartials part0{nullptr, 0, {0, 0, 0}, 0};
    {
        int np[3];
        if (in_conv_gn_parts(h.g, c.in_channels, h.C, np)) {        // the kernel leaves the GroupNorm partials of its output
            const Fwd::ChunkStats cs = f.chunk_stats(np);
            part0 = cs.part;
            if (!meas && !take) S3D_TRY(launch_in_conv(x, B, c.in_channels, H, W, D, m->dev(m->in_wT), m->dev(m->in_b), h.C, h, st, &cs.part));
            S3D_TRY(f.finish(cs, h));
        } else if (!meas) S3D_TRY(launch_in_conv(x, B, c.in_channels, H, W, D, m->dev(m->in_wT), m->dev(m->in_b), h.C, h, st));
    }
    if (tape) tape->h0 = h;

    std::vector<Tri> hs;
    for (int level = 0; level < c.n_levels; ++level) {
        if (level != 0) {                                           // TriplaneDownsample2x (:127-145)
            Tri d = f.alloc_tri(h.C, h.g.half());
            for (int p = 0; p < 3; ++p)
                S3D_CHECK(d.g.h[p] > 0 && d.g.w[p] > 0, S3D_ERR_INVALID, "plane too small to downsample (level %d)", level);
            const Fwd::ChunkStats cs = f.chunk_stats();
            if (!meas) S3D_TRY(launch_avgpool(h, B, d, st, &cs.part));
            S3D_TRY(f.finish(cs, d));
            h = d;
        }
        Tri o;
        // the deepest output goes straight into a norm; the others become the skip half of a concat later: in the
        // inference forward their producing convolution leaves the GroupNorm partials with them (Fwd::resblock_cat)
        // (... and the deepest one's too: the GN-act kernel of the first output block adds them itself)
        S3D_TRY(f.resblock(m->in_blocks[level], h, o, level == c.n_levels - 1, tape ? (level == c.n_levels - 1 ? -1 : 0) : 2));
        if (tape) { f.last_rb.index = level; f.last_rb.is_out = false; tape->in_rb.push_back(f.last_rb); }
        h = o;
        hs.push_back(o);
    }
    for (int oi = 0; oi < c.n_levels; ++oi) {
        const int level = c.n_levels - 1 - oi;
        Tri inp;
        if (tape) tape->up_src.push_back(oi == 0 ? Tri() : h);
        if (oi == 0) { inp = hs.back(); hs.pop_back(); }
        else {
            // previous block's output `h` -> TriplaneUpsample2x (:106-124) -> resize to the skip's size when it
            // differs (:494-499) -> concat [h, skip] (:501-503), written straight into the concat buffer
            Tri sk = hs.back(); hs.pop_back();
            const Geo up = h.g.twice();
            {
                const ResBlockW& rb = m->out_blocks[oi];
                const bool vcat_on = opt_on(OPT_VCAT);
                const int Cc = h.C + sk.C;
                if (vcat_on && !tape && up == sk.g && sk.part.p && rb.has_skip && rb.c_up == h.C && rb.skip_a.dense[0] &&
                    gn_subgroup(Cc) == gn_subgroup(sk.C) && sk.part.nsub == sk.C / gn_subgroup(Cc) && h.C % gn_subgroup(Cc) == 0) {
                    Tri o;
                    // the last block feeds the output head, which adds the partial sums itself when it can (else: k_gn_finalize)
                    S3D_TRY(f.resblock_cat(rb, h, sk, o, oi == c.n_levels - 1 ? (out_head_px_takes(rb.Cout, c.out_channels) ? 2 : 1) : 0));
                    h = o;
                    continue;
                }
            }
            inp = f.alloc_tri(h.C + sk.C, sk.g);
            if (up == sk.g) {                                       // the common case: one fused pass for all planes
                int np[3];
                if (upcat_gn_parts(inp.g, inp.C, np)) {
                    const Fwd::ChunkStats cs = f.chunk_stats(np);
                    if (!meas) S3D_TRY(launch_upcat(h, sk, B, inp, st, &cs.part));
                    S3D_TRY(f.finish(cs, inp));
                } else if (!meas) S3D_TRY(launch_upcat(h, sk, B, inp, st));
            } else
            for (int p = 0; p < 3; ++p) {
                const bool same = up.h[p] == sk.g.h[p] && up.w[p] == sk.g.w[p];
                if (same) {
                    if (!meas) S3D_TRY(launch_bilinear(h.p[p], B, h.C, h.g.h[p], h.g.w[p], inp.p[p], up.h[p], up.w[p], inp.C, 0, st));
                } else {
                    S3D_CHECK(c.is_rollout, S3D_ERR_UNSUPPORTED,
                              "TriplaneUNetModelSmallRaw has no skip-size resize: plane sizes must be divisible by 2^levels");
                    float* tmp = ar.alloc<float>(size_t(B) * up.h[p] * up.w[p] * h.C);
                    if (!meas) {
                        S3D_TRY(launch_bilinear(h.p[p], B, h.C, h.g.h[p], h.g.w[p], tmp, up.h[p], up.w[p], h.C, 0, st));
                        S3D_TRY(launch_bilinear(tmp, B, h.C, up.h[p], up.w[p], inp.p[p], sk.g.h[p], sk.g.w[p], inp.C, 0, st));
                    }
                }
                if (!meas) S3D_TRY(launch_copy_slice(sk.p[p], B, sk.C, sk.g.h[p], sk.g.w[p], inp.p[p], inp.C, h.C, st));
            }
        }
        Tri o;
        // ... and so does the last one (out head) — which adds the partials itself when it can (inference): partials only
        const bool last = oi == c.n_levels - 1;
        const int head_parts = last && !tape && out_head_px_takes(m->out_blocks[oi].Cout, c.out_channels) ? 2 : -1;
        S3D_TRY(f.resblock(m->out_blocks[oi], inp, o, last, head_parts));
        if (tape) { f.last_rb.index = oi; f.last_rb.is_out = true; tape->out_rb.push_back(f.last_rb); tape->cat_in.push_back(inp); }
        h = o;
        (void)level;
    }
    // the decoder's Upsample of the LAST level-0 block does not exist (level > 0 only), so h is at full size
    GnStats stats{nullptr};
    const bool head_few = !tape && !h.gn && h.part.p && out_head_adds_parts(h.part, h.C, c.out_channels);
    const bool head_adds = head_few && gn_parts_in_consumer(out_head_px_blocks(h.g, B));
    if (head_few && !head_adds) {                           // (bit-identical to the in-head addition: the head's 256-thread blocks)
        stats.mr = ar.alloc<float>(size_t(B) * 3 * 64);
        if (!meas) S3D_TRY(launch_gn_finalize_as(h.part, h.g, h.C, B, 256, stats, st));
    } else if (!head_adds) S3D_TRY(f.stats_of(h, stats));
    if (tape) { tape->head_in = h; tape->head_stats = stats; tape->arena_off = ar.off; tape->valid = !meas; }
    // a fused step on a width the pixel-chunk head does not take: the model output goes through a workspace buffer
    if (fuse && !out && !out_head_fuses_sampler(h.C, c.out_channels, B)) out = ar.alloc<float>(size_t(B) * c.out_channels * (H + D) * (W + D));
    if (!meas) {
        ActArgs aa;
        for (int p = 0; p < 3; ++p) { aa.gamma[p] = m->dev(m->out_norm.gamma[p]); aa.beta[p] = m->dev(m->out_norm.beta[p]); }
        aa.film = nullptr; aa.film_stride = 0;
        // S3D_CARRY_OUT: the head also runs the NEXT step's in_conv on the x_{t-1} it forms (unet_triplane.py:378, 482: a pointwise
        // TriplaneConv) into h0 / part0 — dead since this step's first block consumed them
        InConvCarry cy;
        const bool give = (carry_flags & S3D_CARRY_OUT) && fuse && fuse->sample && fuse->mode != S3D_STEP_MEAN_ONLY && !tape && part0.p &&
                          h.C == h0.C && out_head_fuses_sampler(h.C, c.out_channels, B) && out_head_can_carry(h.C, c.in_channels, c.out_channels);
        if (give) {
            cy.wT = m->dev(m->in_wT); cy.bias = m->dev(m->in_b); cy.part = part0.p; cy.maxparts = part0.maxparts; cy.Cin = c.in_channels;
            for (int p = 0; p < 3; ++p) cy.out[p] = h0.p[p];
        }
        S3D_TRY(launch_out_head(h, B, stats, aa, m->dev(m->out_w), m->dev(m->out_b), c.out_channels, H, W, D, out, st, fuse, head_adds ? &h.part : nullptr,
                                give ? &cy : nullptr));
        if (give) { m->carry.valid = true; m->carry.sample = fuse->sample; for (int k = 0; k < 4; ++k) m->carry.key[k] = ckey[k]; }
    }
    return 0;
}

}  // namespace s3d

extern "C" {

int s3d_unet_create(const s3d_unet_cfg* cfg, s3d_unet** out) {
    S3D_CHECK(cfg && out, S3D_ERR_INVALID, "unet_create: null argument");
    S3D_CHECK(cfg->num_res_blocks == 1, S3D_ERR_UNSUPPORTED,
              "num_res_blocks=%d: the reference constructor only succeeds for 1 (unet_triplane.py:383-419)", cfg->num_res_blocks);
    S3D_CHECK(cfg->n_levels >= 1 && cfg->n_levels <= 8, S3D_ERR_INVALID, "channel_mult must have 1..8 entries");
    S3D_CHECK(cfg->model_channels > 0 && cfg->model_channels % 32 == 0, S3D_ERR_INVALID,
              "model_channels=%d must be a positive multiple of 32 (GroupNorm32(32, C))", cfg->model_channels);
    S3D_CHECK(cfg->in_channels > 0 && cfg->out_channels > 0, S3D_ERR_INVALID, "in/out channels must be positive");
    for (int i = 0; i < cfg->n_levels; ++i) S3D_CHECK(cfg->channel_mult[i] >= 1, S3D_ERR_INVALID, "channel_mult entries must be >= 1");
    std::unique_ptr<s3d_unet> m(new s3d_unet());
    m->cfg = *cfg;
    S3D_TRY(build_specs(m.get()));
    *out = m.release();
    return 0;
}

void s3d_unet_destroy(s3d_unet* m) { delete m; }

int s3d_unet_num_params(const s3d_unet* m) { return m ? int(m->specs.size()) : S3D_ERR_INVALID; }

int s3d_unet_param_info(const s3d_unet* m, int i, const char** name, int64_t shape[4], int* ndim) {
    S3D_CHECK(m && i >= 0 && i < int(m->specs.size()), S3D_ERR_INVALID, "param_info: index %d out of range", i);
    const ParamSpec& sp = m->specs[i];
    if (name) *name = sp.name.c_str();
    if (ndim) *ndim = int(sp.shape.size());
    if (shape) for (size_t k = 0; k < sp.shape.size(); ++k) shape[k] = sp.shape[k];
    return 0;
}

int s3d_unet_set_param(s3d_unet* m, const char* name, const float* data, const int64_t* shape, int ndim) {
    S3D_CHECK(m && name && data && shape, S3D_ERR_INVALID, "set_param: null argument");
    for (const auto& sp : m->specs) {
        if (sp.name != name) continue;
        bool ok = int(sp.shape.size()) == ndim;
        for (int k = 0; ok && k < ndim; ++k) ok = sp.shape[k] == shape[k];
        S3D_CHECK(ok, S3D_ERR_INVALID, "size mismatch for %s", name);
        m->host[sp.name].assign(data, data + sp.numel());
        m->packed = false;
        return 0;
    }
    set_error("unexpected key '%s' in state_dict", name);
    return S3D_ERR_INVALID;
}

static int forward_impl(s3d_unet* m, const float* x, const float* t, int B, int H, int W, int D, float* out, void* stream,
                        const float* ext_film, int ext_film_stride, const s3d_sampler_args* fuse = nullptr, int carry_flags = 0);

int s3d_unet_forward(s3d_unet* m, const float* x, const float* t, int B, int H, int W, int D, float* out, void* stream) {
    S3D_CHECK(m && x && t && out, S3D_ERR_INVALID, "unet_forward: null argument");
    return forward_impl(m, x, t, B, H, W, D, out, stream, nullptr, 0);
}

int s3d_unet_film_width(const s3d_unet* m) { return m ? m->film_total : S3D_ERR_INVALID; }

int s3d_unet_select_lane(s3d_unet* m, int lane) {
    S3D_CHECK(m && lane >= 0 && lane < S3D_MAX_LANES, S3D_ERR_INVALID, "unet_select_lane: lane must be in [0, %d)", S3D_MAX_LANES);
    if (lane == m->cur_lane) return 0;
    S3D_CHECK(!m->tape.valid, S3D_ERR_INVALID, "unet_select_lane: a forward_train is waiting for its backward (training runs on lane 0)");
    const size_t need = size_t(std::max(lane, m->cur_lane)) + 1;
    while (m->lanes.size() < need) m->lanes.emplace_back(new s3d_unet::LaneState());
    m->swap_lane(*m->lanes[m->cur_lane]);          // park the selected lane ...
    m->swap_lane(*m->lanes[lane]);                 // ... and bring the requested one in
    m->cur_lane = lane;
    return 0;
}

int s3d_unet_current_lane(const s3d_unet* m) { return m ? m->cur_lane : S3D_ERR_INVALID; }

int s3d_unet_film(s3d_unet* m, const float* t, int n, float* film, void* stream) {
    S3D_CHECK(m && t && film && n >= 1, S3D_ERR_INVALID, "unet_film: bad argument");
    if (!m->packed) S3D_TRY(pack_all(m));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int mc = m->cfg.model_channels, ted = 4 * mc;
    const size_t need = size_t(2) * n * ted * sizeof(float);
    if (need > m->film_ws.cap) { S3D_HIP(hipStreamSynchronize(st)); S3D_TRY(m->film_ws.reserve(need)); }
    float* e1 = static_cast<float*>(m->film_ws.p);
    float* emb = e1 + size_t(n) * ted;
    S3D_TRY(launch_linear(t, n, mc, m->dev(m->te0_w), m->dev(m->te0_b), ted, e1, 2, 1, st));
    S3D_TRY(launch_linear(e1, n, ted, m->dev(m->te2_w), m->dev(m->te2_b), ted, emb, 0, 0, st));
    S3D_TRY(launch_linear(emb, n, ted, m->dev(m->film_w), m->dev(m->film_b), m->film_total, film, 1, 0, st));
    return 0;
}

int s3d_unet_forward_film(s3d_unet* m, const float* x, const float* film, int film_stride, int B, int H, int W, int D, float* out,
                          void* stream) {
    S3D_CHECK(m && x && film && out, S3D_ERR_INVALID, "unet_forward_film: null argument");
    S3D_CHECK(film_stride == 0 || film_stride == m->film_total, S3D_ERR_INVALID, "unet_forward_film: film_stride must be 0 or %d", m->film_total);
    return forward_impl(m, x, nullptr, B, H, W, D, out, stream, film, film_stride);
}

int s3d_unet_step_film(s3d_unet* m, const float* film, int film_stride, int B, int H, int W, int D, const s3d_sampler_args* step,
                       float* model_out, void* stream) {
    S3D_CHECK(m && film && step, S3D_ERR_INVALID, "unet_step_film: null argument");
    S3D_CHECK(film_stride == 0 || film_stride == m->film_total, S3D_ERR_INVALID, "unet_step_film: film_stride must be 0 or %d", m->film_total);
    S3D_CHECK(step->x && step->t && step->tables && step->pred_xstart && (step->mode == S3D_STEP_MEAN_ONLY || step->sample) &&
                  (step->mode != S3D_STEP_DDPM || step->noise),
              S3D_ERR_INVALID, "unet_step_film: incomplete sampler arguments");
    S3D_CHECK(step->batch == B && step->per_sample == (long long)m->cfg.out_channels * (H + D) * (W + D) && m->cfg.in_channels == m->cfg.out_channels,
              S3D_ERR_INVALID, "unet_step_film: the step's shape is not the model's");
    return forward_impl(m, step->x, nullptr, B, H, W, D, model_out, stream, film, film_stride, step, 0);
}

int s3d_unet_step_film_carry(s3d_unet* m, const float* film, int film_stride, int B, int H, int W, int D, const s3d_sampler_args* step,
                             float* model_out, void* stream, int carry_flags) {
    S3D_CHECK(m && film && step, S3D_ERR_INVALID, "unet_step_film_carry: null argument");
    S3D_CHECK((carry_flags & ~(S3D_CARRY_OUT | S3D_CARRY_IN)) == 0, S3D_ERR_INVALID, "unet_step_film_carry: unknown flag bits %d", carry_flags);
    S3D_CHECK(film_stride == 0 || film_stride == m->film_total, S3D_ERR_INVALID, "unet_step_film_carry: film_stride must be 0 or %d", m->film_total);
    S3D_CHECK(step->x && step->t && step->tables && step->pred_xstart && (step->mode == S3D_STEP_MEAN_ONLY || step->sample) &&
                  (step->mode != S3D_STEP_DDPM || step->noise),
              S3D_ERR_INVALID, "unet_step_film_carry: incomplete sampler arguments");
    S3D_CHECK(step->batch == B && step->per_sample == (long long)m->cfg.out_channels * (H + D) * (W + D) && m->cfg.in_channels == m->cfg.out_channels,
              S3D_ERR_INVALID, "unet_step_film_carry: the step's shape is not the model's");
    return forward_impl(m, step->x, nullptr, B, H, W, D, model_out, stream, film, film_stride, step, carry_flags);
}

}  // extern "C"

static int forward_impl(s3d_unet* m, const float* x, const float* t, int B, int H, int W, int D, float* out, void* stream,
                        const float* ext_film, int ext_film_stride, const s3d_sampler_args* fuse, int carry_flags) {
    S3D_CHECK(B >= 1 && H >= 1 && W >= 1 && D >= 1, S3D_ERR_INVALID, "unet_forward: B,H,W,D must be >= 1");
    if (!m->packed) { m->drop_carries(); S3D_TRY(pack_all(m)); }      // (new weights: a carried in_conv was formed with the old ones)
    m->tape.valid = false;                    // the workspace is shared with the training tape
    hipStream_t st = static_cast<hipStream_t>(stream);
    // pass 1: measure the workspace; grow it if needed (synchronising only when it really grows).  The walk is pure host
    // work and depends only on the shapes: skipped when they repeat (every step of a sampling loop).
    const long long key[4] = {B, H, W, D | (fuse && !out ? 1LL << 40 : 0)};      // (a fused step without a caller buffer may need one from the workspace)
    const bool same = m->inf_key[0] == key[0] && m->inf_key[1] == key[1] && m->inf_key[2] == key[2] && m->inf_key[3] == key[3];
    int rc = 0;
    if (!same || m->inf_high > m->arena.buf.cap) {
        m->arena.measuring = true;
        m->arena.high = 0;
        rc = run_forward(m, x, t, B, H, W, D, out, st, nullptr, ext_film, ext_film_stride, fuse);
        m->arena.measuring = false;
        if (rc) return rc;
        m->inf_high = m->arena.high;
        for (int k = 0; k < 4; ++k) m->inf_key[k] = key[k];
        if (m->arena.high > m->arena.buf.cap) {
            S3D_HIP(hipStreamSynchronize(st));
            S3D_TRY(m->arena.buf.reserve(m->arena.high + (m->arena.high >> 3)));
            m->carry.valid = false;                                // the workspace moved
        }
    }
    m->prof_now = m->prof_every > 0 && (m->fwd_count % m->prof_every) == 0;
    ++m->fwd_count;
    if (m->prof_now) ++m->prof_forwards;
    rc = run_forward(m, x, t, B, H, W, D, out, st, nullptr, ext_film, ext_film_stride, fuse, carry_flags);
    m->prof_now = false;
    return rc;
}

extern "C" {

int s3d_unet_profile(s3d_unet* m, int every) {
    S3D_CHECK(m && every >= 0, S3D_ERR_INVALID, "unet_profile: bad argument");
    m->prof_every = every;
    m->fwd_count = 0;
    if (every > 0)                                                    // a new measurement names its own kernels (of the classes it brackets)
        for (int c = 0; c < S3D_PROF_CLASSES; ++c) if ((m->prof_mask >> c) & 1) m->prof_kernel[c].clear();
    return 0;
}

int s3d_unet_profile_classes(s3d_unet* m, int mask) {
    S3D_CHECK(m && mask >= 0 && mask < (1 << S3D_PROF_CLASSES), S3D_ERR_INVALID, "unet_profile_classes: bad argument");
    m->prof_mask = mask;
    return 0;
}

int s3d_unet_profile_read(s3d_unet* m, s3d_profile* out) {
    S3D_CHECK(m && out, S3D_ERR_INVALID, "unet_profile_read: null argument");
    for (auto& r : m->prof_recs) {
        if (r.e0 && r.e1) {
            S3D_HIP(hipEventSynchronize(r.e1));
            float ms = 0.f;
            S3D_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
            out->ms[r.cls] += ms;
            out->flops[r.cls] += r.flops;
            out->mfma_flops[r.cls] += r.mfma_flops;
            out->launches[r.cls] += 1;
        }
        if (r.e0) m->prof_pool.push_back(r.e0);
        if (r.e1) m->prof_pool.push_back(r.e1);
    }
    m->prof_recs.clear();
    out->forwards += m->prof_forwards;
    m->prof_forwards = 0;
    return 0;
}

const char* s3d_unet_profile_kernel(const s3d_unet* m, int cls) {
    return m && cls >= 0 && cls < S3D_PROF_CLASSES ? m->prof_kernel[cls].c_str() : "";
}

}  // extern "C"
