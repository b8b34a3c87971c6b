/*
 * sin3dm_oracle.c — CPU restatement of the Sin3DM denoising path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (sin3dm_amd/) never does and fails loudly when its HIP library is missing.
 *
 * Plain C restatement of what the reference (pure Python/PyTorch, /root/reference/src) computes,
 * op by op, in the reference's own NCHW layout and with the reference's dense rollout concat
 * (no rank-1 shortcut here: the oracle is deliberately the literal algorithm).
 * Parity status: PINNED — checked in tests/test_oracle_golden.py against golden vectors produced
 * by importing the reference itself (tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ parameter table ------- */
typedef struct {
    int n;
    const char* const* names;
    const float* const* ptrs;
} orc_params;

static const float* P(const orc_params* t, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static const float* P(const orc_params* t, const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    for (int i = 0; i < t->n; ++i)
        if (strcmp(t->names[i], buf) == 0) return t->ptrs[i];
    fprintf(stderr, "sin3dm_oracle: missing parameter '%s'\n", buf);
    abort();
}
static int has_param(const orc_params* t, const char* name) {
    for (int i = 0; i < t->n; ++i)
        if (strcmp(t->names[i], name) == 0) return 1;
    return 0;
}

static float* falloc(size_t n) {
    float* p = (float*)malloc((n ? n : 1) * sizeof(float));
    if (!p) { fprintf(stderr, "sin3dm_oracle: out of memory\n"); abort(); }
    return p;
}

/* ------------------------------------------------------------------ leaf ops -------------- */

/* nn.Conv2d: cross-correlation, zero padding k/2, stride 1, + bias; weights OIHW.
 * src/diffusion/unet_triplane.py:27-29 (TriplaneConv) ; src/encoding/blocks.py:204-233 (groups handled by caller). */
ORC_API void orc_conv2d(const float* x, const float* w, const float* bias, float* y,
                        int B, int Cin, int H, int W, int Cout, int K) {
    const int pad = K / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co) {
            float* yo = y + ((size_t)b * Cout + co) * H * W;
            const float bv = bias ? bias[co] : 0.f;
            for (int i = 0; i < H * W; ++i) yo[i] = bv;
            for (int ci = 0; ci < Cin; ++ci) {
                const float* xi = x + ((size_t)b * Cin + ci) * H * W;
                const float* wk = w + ((size_t)co * Cin + ci) * K * K;
                for (int kh = 0; kh < K; ++kh)
                    for (int kw = 0; kw < K; ++kw) {
                        const float wv = wk[kh * K + kw];
                        const int dh = kh - pad, dw = kw - pad;
                        const int h0 = dh < 0 ? -dh : 0, h1 = dh > 0 ? H - dh : H;
                        const int w0 = dw < 0 ? -dw : 0, w1 = dw > 0 ? W - dw : W;
                        for (int h = h0; h < h1; ++h) {
                            float* yr = yo + (size_t)h * W;
                            const float* xr = xi + (size_t)(h + dh) * W + dw;
                            for (int ww = w0; ww < w1; ++ww) yr[ww] += wv * xr[ww];
                        }
                    }
            }
        }
}

/* GroupNorm32(32, C): biased variance over (C/G)*HW, eps inside sqrt, affine, fp32.
 * src/diffusion/nn.py:17-19, 93-100.  Statistics accumulated in double (ATen uses a cascade sum). */
ORC_API void orc_group_norm(const float* x, const float* gamma, const float* beta, float* y,
                            int B, int C, int HW, int G, float eps) {
    const int cg = C / G;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g) {
            const float* xp = x + ((size_t)b * C + (size_t)g * cg) * HW;
            float* yp = y + ((size_t)b * C + (size_t)g * cg) * HW;
            const size_t n = (size_t)cg * HW;
            double s = 0, ss = 0;
            for (size_t i = 0; i < n; ++i) s += xp[i];
            const double mean = s / (double)n;
            for (size_t i = 0; i < n; ++i) { double d = xp[i] - mean; ss += d * d; }
            const float rstd = (float)(1.0 / sqrt(ss / (double)n + (double)eps));
            const float mu = (float)mean;
            for (int c = 0; c < cg; ++c) {
                const float ga = gamma ? gamma[g * cg + c] : 1.f, be = beta ? beta[g * cg + c] : 0.f;
                for (int i = 0; i < HW; ++i) yp[(size_t)c * HW + i] = (xp[(size_t)c * HW + i] - mu) * rstd * ga + be;
            }
        }
}

/* SiLU = x * sigmoid(x).  src/diffusion/nn.py:12-14 ; src/encoding/blocks.py:94-96 */
static inline float silu1(float v) { return v / (1.f + expf(-v)); }
ORC_API void orc_silu(const float* x, float* y, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) y[i] = silu1(x[i]);
}

/* F.avg_pool2d(k=2, s=2): floor(H/2) x floor(W/2), trailing odd row/col dropped.
 * src/diffusion/unet_triplane.py:137-139 */
ORC_API void orc_avgpool2(const float* x, float* y, int BC, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < BC; ++c)
        for (int h = 0; h < Ho; ++h)
            for (int w = 0; w < Wo; ++w) {
                const float* p = x + ((size_t)c * H + 2 * h) * W + 2 * w;
                y[((size_t)c * Ho + h) * Wo + w] = (p[0] + p[1] + p[W] + p[W + 1]) * 0.25f;
            }
}

/* F.interpolate(mode='bilinear', align_corners=False) for scale_factor=2 and for size=...:
 * src = scale*(dst+0.5)-0.5 clamped at 0, scale = in/out (== 1/scale_factor for the exact 2x case).
 * src/diffusion/unet_triplane.py:116-118 and :494-499 */
ORC_API void orc_bilinear(const float* x, float* y, int BC, int Hi, int Wi, int Ho, int Wo) {
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < BC; ++c)
        for (int h = 0; h < Ho; ++h) {
            float fh = sh * ((float)h + 0.5f) - 0.5f; if (fh < 0) fh = 0;
            int h0 = (int)fh; if (h0 > Hi - 1) h0 = Hi - 1;
            const int h1 = h0 + (h0 < Hi - 1 ? 1 : 0);
            const float lh1 = fh - (float)h0, lh0 = 1.f - lh1;
            for (int w = 0; w < Wo; ++w) {
                float fw = sw * ((float)w + 0.5f) - 0.5f; if (fw < 0) fw = 0;
                int w0 = (int)fw; if (w0 > Wi - 1) w0 = Wi - 1;
                const int w1 = w0 + (w0 < Wi - 1 ? 1 : 0);
                const float lw1 = fw - (float)w0, lw0 = 1.f - lw1;
                const float* p = x + (size_t)c * Hi * Wi;
                y[((size_t)c * Ho + h) * Wo + w] =
                    lh0 * (lw0 * p[h0 * Wi + w0] + lw1 * p[h0 * Wi + w1]) +
                    lh1 * (lw0 * p[h1 * Wi + w0] + lw1 * p[h1 * Wi + w1]);
            }
        }
}

/* nn.Linear: y[b,o] = sum_i x[b,i] w[o,i] + bias[o] */
ORC_API void orc_linear(const float* x, const float* w, const float* bias, float* y, int B, int I, int O) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < O; ++o) {
            float acc = bias ? bias[o] : 0.f;
            const float* xr = x + (size_t)b * I;
            const float* wr = w + (size_t)o * I;
            for (int i = 0; i < I; ++i) acc += xr[i] * wr[i];
            y[(size_t)b * O + o] = acc;
        }
}

/* timestep_embedding(t, dim): half=dim/2, f_i=exp(-ln(1e4)*i/half), [cos | sin].
 * src/diffusion/nn.py:103-121 */
ORC_API void orc_timestep_embedding(const float* t, float* emb, int B, int dim) {
    const int half = dim / 2;
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < half; ++i) {
            const float f = expf(-logf(10000.f) * (float)i / (float)half);
            const float a = t[b] * f;
            emb[(size_t)b * dim + i] = cosf(a);
            emb[(size_t)b * dim + half + i] = sinf(a);
        }
        if (dim & 1) emb[(size_t)b * dim + dim - 1] = 0.f;
    }
}

/* ------------------------------------------------------------------ triplane ops ---------- */
typedef struct { float* p[3]; int C; int h[3], w[3]; } tri;   /* planes xy[H,W], xz[H,D], yz[W,D], NCHW */

static tri tri_alloc(int B, int C, int H, int W, int D) {
    tri t; t.C = C;
    t.h[0] = H; t.w[0] = W; t.h[1] = H; t.w[1] = D; t.h[2] = W; t.w[2] = D;
    for (int i = 0; i < 3; ++i) t.p[i] = falloc((size_t)B * C * t.h[i] * t.w[i]);
    return t;
}
static tri tri_alloc_hw(int B, int C, const int* h, const int* w) {
    tri t; t.C = C;
    for (int i = 0; i < 3; ++i) { t.h[i] = h[i]; t.w[i] = w[i]; t.p[i] = falloc((size_t)B * C * h[i] * w[i]); }
    return t;
}
static void tri_free(tri* t) { for (int i = 0; i < 3; ++i) { free(t->p[i]); t->p[i] = NULL; } }

/* decompose_featmaps: xy=[..,:H,:W], xz=[..,:H,W:], yz=[..,H:,:W]^T.  src/utils/triplane_util.py:20-25 */
ORC_API void orc_decompose(const float* comp, float* xy, float* xz, float* yz, int BC, int H, int W, int D) {
    const int Wc = W + D, Hc = H + D;
    for (int c = 0; c < BC; ++c) {
        const float* s = comp + (size_t)c * Hc * Wc;
        for (int h = 0; h < H; ++h) {
            for (int w = 0; w < W; ++w) xy[((size_t)c * H + h) * W + w] = s[(size_t)h * Wc + w];
            for (int d = 0; d < D; ++d) xz[((size_t)c * H + h) * D + d] = s[(size_t)h * Wc + W + d];
        }
        for (int w = 0; w < W; ++w)
            for (int d = 0; d < D; ++d) yz[((size_t)c * W + w) * D + d] = s[(size_t)(H + d) * Wc + w];
    }
}
/* compose_featmaps: [[xy | xz],[yz^T | 0]].  src/utils/triplane_util.py:7-17 */
ORC_API void orc_compose(const float* xy, const float* xz, const float* yz, float* comp, int BC, int H, int W, int D) {
    const int Wc = W + D, Hc = H + D;
    memset(comp, 0, (size_t)BC * Hc * Wc * sizeof(float));
    for (int c = 0; c < BC; ++c) {
        float* s = comp + (size_t)c * Hc * Wc;
        for (int h = 0; h < H; ++h) {
            for (int w = 0; w < W; ++w) s[(size_t)h * Wc + w] = xy[((size_t)c * H + h) * W + w];
            for (int d = 0; d < D; ++d) s[(size_t)h * Wc + W + d] = xz[((size_t)c * H + h) * D + d];
        }
        for (int w = 0; w < W; ++w)
            for (int d = 0; d < D; ++d) s[(size_t)(H + d) * Wc + w] = yz[((size_t)c * W + w) * D + d];
    }
}

/* mean over the last (dim=-1) or second-to-last (dim=-2) axis of [BC,h,w] */
static void mean_last(const float* x, float* m, int BC, int h, int w) {      /* -> [BC,h] */
    for (size_t i = 0; i < (size_t)BC * h; ++i) {
        double s = 0; for (int j = 0; j < w; ++j) s += x[i * w + j];
        m[i] = (float)(s / w);
    }
}
static void mean_rows(const float* x, float* m, int BC, int h, int w) {      /* -> [BC,w] */
    for (int c = 0; c < BC; ++c)
        for (int j = 0; j < w; ++j) {
            double s = 0; for (int i = 0; i < h; ++i) s += x[((size_t)c * h + i) * w + j];
            m[(size_t)c * w + j] = (float)(s / h);
        }
}

/* TriplaneConv.forward.  src/diffusion/unet_triplane.py:31-60.
 * Rollout concat order (:37-46):
 *   xy_h = [xy, mean_d(yz)^T expanded over h, mean_d(xz) expanded over w]
 *   xz_h = [xz, mean_w(xy) expanded over d,  mean_w(yz) (dim=-2) expanded over h]
 *   yz_h = [yz, mean_h(xy)^T expanded over d, mean_h(xz) (dim=-2) expanded over w]          */
static tri triplane_conv(const orc_params* pr, const char* prefix, const tri* in, int B, int Cout, int K, int rollout) {
    static const char* pn[3] = {"xy", "xz", "yz"};
    const int C = in->C;
    tri out = tri_alloc_hw(B, Cout, in->h, in->w);
    const int H = in->h[0], W = in->w[0], D = in->w[1];
    for (int p = 0; p < 3; ++p) {
        const int h = in->h[p], w = in->w[p];
        const float* wt = P(pr, "%s.conv_%s.weight", prefix, pn[p]);
        const float* bs = P(pr, "%s.conv_%s.bias", prefix, pn[p]);
        if (!rollout) {
            orc_conv2d(in->p[p], wt, bs, out.p[p], B, C, h, w, Cout, K);
            continue;
        }
        float* cat = falloc((size_t)B * 3 * C * h * w);
        float* va = falloc((size_t)B * C * (H + W + D));
        float* vb = falloc((size_t)B * C * (H + W + D));
        /* which vectors, and along which axis of the target plane they vary */
        int a_var_row, b_var_row;   /* 1: vector indexed by target row (const along cols); 0: by target col */
        if (p == 0) {        /* xy[h,w] */
            mean_last(in->p[2], va, B * C, W, D);  a_var_row = 0;   /* mean_d yz[w,d] -> [w], const in h */
            mean_last(in->p[1], vb, B * C, H, D);  b_var_row = 1;   /* mean_d xz[h,d] -> [h], const in w */
        } else if (p == 1) { /* xz[h,d] */
            mean_last(in->p[0], va, B * C, H, W);  a_var_row = 1;   /* mean_w xy[h,w] -> [h], const in d */
            mean_rows(in->p[2], vb, B * C, W, D);  b_var_row = 0;   /* mean_w yz[w,d] -> [d], const in h */
        } else {             /* yz[w,d] */
            mean_rows(in->p[0], va, B * C, H, W);  a_var_row = 1;   /* mean_h xy[h,w] -> [w] = rows of yz */
            mean_rows(in->p[1], vb, B * C, H, D);  b_var_row = 0;   /* mean_h xz[h,d] -> [d], const in w(row) */
        }
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < C; ++c) {
                const float* src = in->p[p] + ((size_t)b * C + c) * h * w;
                float* d0 = cat + ((size_t)b * 3 * C + c) * h * w;
                float* d1 = cat + ((size_t)b * 3 * C + C + c) * h * w;
                float* d2 = cat + ((size_t)b * 3 * C + 2 * C + c) * h * w;
                memcpy(d0, src, (size_t)h * w * sizeof(float));
                const float* A = va + ((size_t)b * C + c) * (a_var_row ? h : w);
                const float* Bv = vb + ((size_t)b * C + c) * (b_var_row ? h : w);
                for (int i = 0; i < h; ++i)
                    for (int j = 0; j < w; ++j) {
                        d1[(size_t)i * w + j] = a_var_row ? A[i] : A[j];
                        d2[(size_t)i * w + j] = b_var_row ? Bv[i] : Bv[j];
                    }
            }
        orc_conv2d(cat, wt, bs, out.p[p], B, 3 * C, h, w, Cout, K);
        free(cat); free(va); free(vb);
    }
    return out;
}

/* TriplaneNorm (+ optional FiLM) + TriplaneSiLU.  src/diffusion/unet_triplane.py:63-95, 285-297 */
static tri triplane_norm_silu(const orc_params* pr, const char* prefix, const tri* in, int B,
                              const float* scale, const float* shift /* [B,C] or NULL */) {
    static const char* pn[3] = {"xy", "xz", "yz"};
    const int C = in->C;
    tri out = tri_alloc_hw(B, C, in->h, in->w);
    for (int p = 0; p < 3; ++p) {
        const int hw = in->h[p] * in->w[p];
        orc_group_norm(in->p[p], P(pr, "%s.norm_%s.weight", prefix, pn[p]), P(pr, "%s.norm_%s.bias", prefix, pn[p]),
                       out.p[p], B, C, hw, 32, 1e-5f);
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < C; ++c) {
                float* y = out.p[p] + ((size_t)b * C + c) * hw;
                if (scale) {
                    const float sc = 1.f + scale[(size_t)b * C + c], sf = shift[(size_t)b * C + c];
                    for (int i = 0; i < hw; ++i) y[i] = silu1(y[i] * sc + sf);
                } else
                    for (int i = 0; i < hw; ++i) y[i] = silu1(y[i]);
            }
    }
    return out;
}

/* TriplaneResBlock._forward.  src/diffusion/unet_triplane.py:269-311 */
static tri triplane_resblock(const orc_params* pr, const char* prefix, const tri* x, const float* emb /*[B,4mc]*/,
                             int B, int ted, int Cout, int ssn, int rollout) {
    char buf[200];
    const int C = x->C;
    snprintf(buf, sizeof buf, "%s.in_layers.0", prefix);
    tri a = triplane_norm_silu(pr, buf, x, B, NULL, NULL);
    snprintf(buf, sizeof buf, "%s.in_layers.2", prefix);
    tri h = triplane_conv(pr, buf, &a, B, Cout, 3, rollout);
    tri_free(&a);
    /* emb_layers = SiLU -> Linear (:232-238, 281) */
    const int eo = ssn ? 2 * Cout : Cout;
    float* es = falloc((size_t)B * ted);
    for (size_t i = 0; i < (size_t)B * ted; ++i) es[i] = silu1(emb[i]);
    float* eout = falloc((size_t)B * eo);
    orc_linear(es, P(pr, "%s.emb_layers.1.weight", prefix), P(pr, "%s.emb_layers.1.bias", prefix), eout, B, ted, eo);
    free(es);
    snprintf(buf, sizeof buf, "%s.out_layers.0", prefix);
    tri n;
    if (ssn) {      /* scale, shift = chunk(emb_out, 2, dim=1); h = norm(h)*(1+scale)+shift (:285-297) */
        float* sc = falloc((size_t)B * Cout); float* sf = falloc((size_t)B * Cout);
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < Cout; ++c) { sc[b * Cout + c] = eout[(size_t)b * eo + c]; sf[b * Cout + c] = eout[(size_t)b * eo + Cout + c]; }
        n = triplane_norm_silu(pr, buf, &h, B, sc, sf);
        free(sc); free(sf);
    } else {        /* h = h + emb_out ; out_layers(h) (:298-306) */
        for (int p = 0; p < 3; ++p) {
            const int hw = h.h[p] * h.w[p];
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < Cout; ++c) {
                    float* y = h.p[p] + ((size_t)b * Cout + c) * hw;
                    const float e = eout[(size_t)b * eo + c];
                    for (int i = 0; i < hw; ++i) y[i] += e;
                }
        }
        n = triplane_norm_silu(pr, buf, &h, B, NULL, NULL);
    }
    free(eout);
    tri_free(&h);
    snprintf(buf, sizeof buf, "%s.out_layers.2", prefix);
    tri o = triplane_conv(pr, buf, &n, B, Cout, 3, rollout);
    tri_free(&n);
    /* skip connection: identity or 1x1 TriplaneConv without rollout (:257-262, 308-311) */
    if (C == Cout) {
        for (int p = 0; p < 3; ++p) {
            const size_t nel = (size_t)B * Cout * o.h[p] * o.w[p];
            for (size_t i = 0; i < nel; ++i) o.p[p][i] += x->p[p][i];
        }
    } else {
        snprintf(buf, sizeof buf, "%s.skip_connection", prefix);
        tri s = triplane_conv(pr, buf, x, B, Cout, 1, 0);
        for (int p = 0; p < 3; ++p) {
            const size_t nel = (size_t)B * Cout * o.h[p] * o.w[p];
            for (size_t i = 0; i < nel; ++i) o.p[p][i] += s.p[p][i];
        }
        tri_free(&s);
    }
    return o;
}

typedef struct {
    int in_channels, model_channels, out_channels;
    int n_levels; int channel_mult[8];
    int use_scale_shift_norm; int rollout;
} orc_unet_cfg;

/* TriplaneUNetModelSmall.forward / ...SmallRaw.forward.  src/diffusion/unet_triplane.py:465-510 (:664-702)
 * x, out: [B, C, H+D, W+D] ; t: [B] (already mapped to the original schedule index, as float). */
ORC_API void orc_unet_forward(const orc_unet_cfg* cfg, const orc_params* pr, const float* x, const float* t,
                              int B, int H, int W, int D, float* out) {
    const int mc = cfg->model_channels, ted = 4 * mc, nl = cfg->n_levels;
    char buf[200];
    /* emb = time_embed(timestep_embedding(t, mc))  (:477, :371-375) */
    float* e0 = falloc((size_t)B * mc);
    orc_timestep_embedding(t, e0, B, mc);
    float* e1 = falloc((size_t)B * ted);
    orc_linear(e0, P(pr, "time_embed.0.weight"), P(pr, "time_embed.0.bias"), e1, B, mc, ted);
    for (size_t i = 0; i < (size_t)B * ted; ++i) e1[i] = silu1(e1[i]);
    float* emb = falloc((size_t)B * ted);
    orc_linear(e1, P(pr, "time_embed.2.weight"), P(pr, "time_embed.2.bias"), emb, B, ted, ted);
    free(e0); free(e1);

    tri xin = tri_alloc(B, cfg->in_channels, H, W, D);
    orc_decompose(x, xin.p[0], xin.p[1], xin.p[2], B * cfg->in_channels, H, W, D);
    int ch = cfg->channel_mult[0] * mc;
    tri h = triplane_conv(pr, "in_conv.0", &xin, B, ch, 1, 0);           /* :482 */
    tri_free(&xin);

    tri hs[8];
    for (int level = 0; level < nl; ++level) {                            /* :484-486 */
        if (level != 0) {                                                  /* TriplaneDownsample2x :127-145 */
            int nh[3], nw[3];
            for (int p = 0; p < 3; ++p) { nh[p] = h.h[p] / 2; nw[p] = h.w[p] / 2; }
            tri d = tri_alloc_hw(B, h.C, nh, nw);
            for (int p = 0; p < 3; ++p) orc_avgpool2(h.p[p], d.p[p], B * h.C, h.h[p], h.w[p]);
            h = d;                          /* the un-pooled tensor stays alive as hs[level-1] */
        }
        const int cout = cfg->channel_mult[level] * mc;
        snprintf(buf, sizeof buf, "input_blocks.%d.%d", level, level == 0 ? 0 : 1);
        tri o = triplane_resblock(pr, buf, &h, emb, B, ted, cout, cfg->use_scale_shift_norm, cfg->rollout);
        tri_free(&h);                      /* in_conv output (level 0) or the pooled temporary */
        h = o; hs[level] = o; ch = cout;
    }
    int top = nl;                                                          /* :488-505 */
    for (int oi = 0; oi < nl; ++oi) {
        const int level = nl - 1 - oi;
        const int cout = cfg->channel_mult[level] * mc;
        tri inp;
        if (oi == 0) { inp = hs[--top]; }
        else {
            tri sk = hs[--top];
            tri cur = h;
            /* resize to the skip's spatial size when it differs (:494-499; absent in ...SmallRaw) */
            tri rs = tri_alloc_hw(B, cur.C, sk.h, sk.w);
            for (int p = 0; p < 3; ++p) {
                if (cur.h[p] != sk.h[p] || cur.w[p] != sk.w[p])
                    orc_bilinear(cur.p[p], rs.p[p], B * cur.C, cur.h[p], cur.w[p], sk.h[p], sk.w[p]);
                else memcpy(rs.p[p], cur.p[p], (size_t)B * cur.C * cur.h[p] * cur.w[p] * sizeof(float));
            }
            inp = tri_alloc_hw(B, cur.C + sk.C, sk.h, sk.w);               /* th.cat([h, skip], dim=1) :501-503 */
            for (int p = 0; p < 3; ++p) {
                const size_t hw = (size_t)sk.h[p] * sk.w[p];
                for (int b = 0; b < B; ++b) {
                    memcpy(inp.p[p] + (size_t)b * inp.C * hw, rs.p[p] + (size_t)b * cur.C * hw, cur.C * hw * sizeof(float));
                    memcpy(inp.p[p] + ((size_t)b * inp.C + cur.C) * hw, sk.p[p] + (size_t)b * sk.C * hw, sk.C * hw * sizeof(float));
                }
            }
            tri_free(&rs); tri_free(&cur); tri_free(&sk);
        }
        snprintf(buf, sizeof buf, "output_blocks.%d.0", oi);
        tri o = triplane_resblock(pr, buf, &inp, emb, B, ted, cout, cfg->use_scale_shift_norm, cfg->rollout);
        tri_free(&inp);                    /* the popped hs entry (oi == 0) or the concat buffer */
        h = o;
        if (level > 0) {                                                   /* TriplaneUpsample2x :106-124 */
            int nh[3], nw[3];
            for (int p = 0; p < 3; ++p) { nh[p] = h.h[p] * 2; nw[p] = h.w[p] * 2; }
            tri u = tri_alloc_hw(B, h.C, nh, nw);
            for (int p = 0; p < 3; ++p) orc_bilinear(h.p[p], u.p[p], B * h.C, h.h[p], h.w[p], nh[p], nw[p]);
            tri_free(&h); h = u;
        }
    }
    /* out = TriplaneNorm -> SiLU -> 1x1 TriplaneConv ; compose (:507-508) */
    tri a = triplane_norm_silu(pr, "out.0", &h, B, NULL, NULL);
    tri_free(&h);
    tri y = triplane_conv(pr, "out.2", &a, B, cfg->out_channels, 1, 0);
    tri_free(&a);
    orc_compose(y.p[0], y.p[1], y.p[2], out, B * cfg->out_channels, H, W, D);
    tri_free(&y);
    free(emb);
}

/* ------------------------------------------------------------------ diffusion process ----- */
/* Schedule tables in float64.  src/diffusion/gaussian_diffusion.py:19-43 (linear betas), :119-170 (tables),
 * src/diffusion/respace.py:72-86 (respaced betas from kept alphas_cumprod).
 * tables: 8 rows of length T: betas, alphas_cumprod, alphas_cumprod_prev, sqrt_recip, sqrt_recipm1,
 *         posterior_variance, posterior_mean_coef1, posterior_mean_coef2 */
ORC_API void orc_linear_betas(double* betas, int T) {
    const double scale = 1000.0 / T, b0 = scale * 0.0001, b1 = scale * 0.02;
    for (int i = 0; i < T; ++i) betas[i] = T == 1 ? b0 : b0 + (b1 - b0) * (double)i / (double)(T - 1);
}
ORC_API void orc_tables_from_betas(const double* betas, int T, double* tab /* [8][T] */) {
    double* ac = tab + 1 * T; double* acp = tab + 2 * T;
    double cum = 1.0;
    for (int i = 0; i < T; ++i) {
        tab[i] = betas[i];
        acp[i] = cum;
        cum *= (1.0 - betas[i]);
        ac[i] = cum;
    }
    for (int i = 0; i < T; ++i) {
        tab[3 * T + i] = sqrt(1.0 / ac[i]);
        tab[4 * T + i] = sqrt(1.0 / ac[i] - 1.0);
        tab[5 * T + i] = betas[i] * (1.0 - acp[i]) / (1.0 - ac[i]);
        tab[6 * T + i] = betas[i] * sqrt(acp[i]) / (1.0 - ac[i]);
        tab[7 * T + i] = (1.0 - acp[i]) * sqrt(1.0 - betas[i]) / (1.0 - ac[i]);
    }
}
ORC_API int orc_respace_betas(const double* base_betas, int T, const uint8_t* keep, double* new_betas, int64_t* tmap) {
    double cum = 1.0, last = 1.0; int n = 0;
    for (int i = 0; i < T; ++i) {
        cum *= (1.0 - base_betas[i]);
        if (keep[i]) { new_betas[n] = 1.0 - cum / last; last = cum; tmap[n] = i; ++n; }
    }
    return n;
}

/* p_mean_variance + p_sample.  src/diffusion/gaussian_diffusion.py:233-327, 396-440.
 * mean_eps == 0: model_out is the x0 prediction (ModelMeanType.START_X, :307-308); 1: it is the noise prediction and
 * x0 = sqrt_recip_alphas_cumprod[t] x - sqrt_recipm1_alphas_cumprod[t] model_out (EPSILON, :309-312, :329-335).
 * var_small == 0: FIXED_LARGE, variance [posterior_variance[1], betas[1:]] (:282-285); 1: FIXED_SMALL, the posterior variance with
 * its clipped log (:286-289, :163-165).  clip: process_xstart's clamp (:294-299).
 * Coefficients are gathered from float64 tables then cast to float (:944).  mean_out (optional): the model mean. */
/* The reference evaluates these expressions as separate fp32 tensor operations: no fused multiply-add (GNU C contracts a * b + c
 * by default; the EPSILON branch subtracts two products of magnitude ~157 at t = 999, where a contracted fma shows) */
#define ORC_NO_FMA __attribute__((optimize("fp-contract=off")))
ORC_NO_FMA ORC_API void orc_p_sample_update_ex(const float* model_out, const float* x, const float* eps, float* sample,
                                    float* pred_xstart, float* mean_out, size_t n, const double* tab, int T, int t, int clip,
                                    int mean_eps, int var_small) {
    const float c1 = (float)tab[6 * T + t], c2 = (float)tab[7 * T + t];
    const float sr = (float)tab[3 * T + t], srm1 = (float)tab[4 * T + t];
    double var;
    if (var_small) var = tab[5 * T + (t == 0 ? 1 : t)];              /* posterior_log_variance_clipped */
    else var = t == 0 ? tab[5 * T + 1] : tab[t];                      /* [posterior_variance[1], betas[1:]] */
    const float logvar = (float)log(var);
    const float sigma = expf(0.5f * logvar);
    const float mask = t != 0 ? 1.f : 0.f;
    for (size_t i = 0; i < n; ++i) {
        float x0 = mean_eps ? sr * x[i] - srm1 * model_out[i] : model_out[i];
        if (clip) x0 = x0 < -1.f ? -1.f : (x0 > 1.f ? 1.f : x0);
        const float mean = c1 * x0 + c2 * x[i];
        sample[i] = mean + mask * sigma * eps[i];
        pred_xstart[i] = x0;
        if (mean_out) mean_out[i] = mean;
    }
}
ORC_API void orc_p_sample_update(const float* model_out, const float* x, const float* eps, float* sample,
                                 float* pred_xstart, size_t n, const double* tab, int T, int t, int clip) {
    orc_p_sample_update_ex(model_out, x, eps, sample, pred_xstart, NULL, n, tab, T, t, clip, 0, 0);
}
/* ddim_sample.  src/diffusion/gaussian_diffusion.py:538-600 (+ _predict_eps_from_xstart :346-350); mean_eps as above
 * (the variance type does not enter the DDIM update) */
ORC_NO_FMA ORC_API void orc_ddim_update_ex(const float* model_out, const float* x, const float* noise, float* sample,
                                float* pred_xstart, size_t n, const double* tab, int T, int t, int clip, float eta, int mean_eps) {
    const float sr = (float)tab[3 * T + t], srm1 = (float)tab[4 * T + t];
    const float ab = (float)tab[1 * T + t], abp = (float)tab[2 * T + t];
    const float sigma = eta * sqrtf((1.f - abp) / (1.f - ab)) * sqrtf(1.f - ab / abp);
    const float mask = t != 0 ? 1.f : 0.f;
    const float ca = sqrtf(abp), cb = sqrtf(1.f - abp - sigma * sigma);
    for (size_t i = 0; i < n; ++i) {
        float x0 = mean_eps ? sr * x[i] - srm1 * model_out[i] : model_out[i];
        if (clip) x0 = x0 < -1.f ? -1.f : (x0 > 1.f ? 1.f : x0);
        const float e = (sr * x[i] - x0) / srm1;
        sample[i] = x0 * ca + cb * e + mask * sigma * noise[i];
        pred_xstart[i] = x0;
    }
}
ORC_API void orc_ddim_update(const float* model_out, const float* x, const float* noise, float* sample,
                             float* pred_xstart, size_t n, const double* tab, int T, int t, int clip, float eta) {
    orc_ddim_update_ex(model_out, x, noise, sample, pred_xstart, n, tab, T, t, clip, eta, 0);
}

/* ------------------------------------------------------------------ autoencoder decode ---- */
/* InstanceNorm2d(C, eps=1e-6, affine=True), instance statistics always.  src/encoding/blocks.py:219-221 */
static void instance_norm(const float* x, const float* g, const float* b, float* y, int C, int HW, float eps) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float* xp = x + (size_t)c * HW; float* yp = y + (size_t)c * HW;
        double s = 0, ss = 0;
        for (int i = 0; i < HW; ++i) s += xp[i];
        const double mean = s / HW;
        for (int i = 0; i < HW; ++i) { double d = xp[i] - mean; ss += d * d; }
        const float rstd = (float)(1.0 / sqrt(ss / HW + (double)eps)), mu = (float)mean;
        for (int i = 0; i < HW; ++i) yp[i] = (xp[i] - mu) * rstd * g[c] + b[c];
    }
}

/* TriplaneGroupResnetBlock(Cin, up, ks=5, input_norm=False, input_act=False).forward on one triplane:
 * grouped conv (groups=3) over zero-padded channel-stacked planes == three independent per-plane convs
 * (padding region is zero, exactly like the conv's own zero padding).  src/encoding/blocks.py:237-256
 * planes: [1,Cin,h,w] each; out planes [1,up,h,w]. prefix = "geo_convs" | "tex_convs". */
ORC_API void orc_ae_plane_block(const orc_params* pr, const char* prefix, const float* const* planes,
                                const int* ph, const int* pw, int Cin, int up, float* const* outs) {
    static const char* pn[3] = {"xy", "xz", "yz"};
    const float* w_in = P(pr, "%s.in_layers.0.weight", prefix);
    const float* b_in = P(pr, "%s.in_layers.0.bias", prefix);
    const float* w_out = P(pr, "%s.out_layers.1.weight", prefix);
    const float* b_out = P(pr, "%s.out_layers.1.bias", prefix);
    const float* w_sc = P(pr, "%s.shortcut.weight", prefix);
    const float* b_sc = P(pr, "%s.shortcut.bias", prefix);
    for (int p = 0; p < 3; ++p) {
        const int h = ph[p], w = pw[p], hw = h * w;
        float* a = falloc((size_t)up * hw); float* n = falloc((size_t)up * hw); float* s = falloc((size_t)up * hw);
        orc_conv2d(planes[p], w_in + (size_t)p * up * Cin * 25, b_in + p * up, a, 1, Cin, h, w, up, 5);
        instance_norm(a, P(pr, "%s.norm_%s.weight", prefix, pn[p]), P(pr, "%s.norm_%s.bias", prefix, pn[p]), n, up, hw, 1e-6f);
        orc_silu(n, n, (size_t)up * hw);
        orc_conv2d(n, w_out + (size_t)p * up * up * 25, b_out + p * up, outs[p], 1, up, h, w, up, 5);
        orc_conv2d(planes[p], w_sc + (size_t)p * up * Cin, b_sc + p * up, s, 1, Cin, h, w, up, 1);
        for (size_t i = 0; i < (size_t)up * hw; ++i) outs[p][i] += s[i];
        free(a); free(n); free(s);
    }
}

/* F.grid_sample(bilinear, padding_mode='border', align_corners=False) of one point on one [C,h,w] plane,
 * accumulated into feat[C].  Coordinate u indexes rows (h axis), v columns: the reference flips the pair
 * before grid_sample (src/encoding/networks.py:182-190), so x[..., coords[0]] -> rows. */
static void sample_plane_acc(const float* fm, int C, int h, int w, float u, float v, float* feat) {
    float fy = ((u + 1.f) * (float)h - 1.f) * 0.5f, fx = ((v + 1.f) * (float)w - 1.f) * 0.5f;
    fy = fy < 0.f ? 0.f : (fy > (float)(h - 1) ? (float)(h - 1) : fy);
    fx = fx < 0.f ? 0.f : (fx > (float)(w - 1) ? (float)(w - 1) : fx);
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = y0 + 1, x1 = x0 + 1;
    const float ty = fy - (float)y0, tx = fx - (float)x0;
    const float w00 = (1.f - ty) * (1.f - tx), w01 = (1.f - ty) * tx, w10 = ty * (1.f - tx), w11 = ty * tx;
    const int y1c = y1 > h - 1 ? h - 1 : y1, x1c = x1 > w - 1 ? w - 1 : x1;   /* weight is 0 whenever clamped */
    for (int c = 0; c < C; ++c) {
        const float* p = fm + (size_t)c * h * w;
        feat[c] += w00 * p[y0 * w + x0] + w01 * p[y0 * w + x1c] + w10 * p[y1c * w + x0] + w11 * p[y1c * w + x1c];
    }
}

/* DecoderMLPSkipConcat.forward for one point.  src/encoding/blocks.py:65-91 (ReLU MLP, cat([x, h])) */
static void mlp_skip_concat(const orc_params* pr, const char* prefix, const float* x, int cin, int hid, int nh,
                            int cout, float* out) {
    float buf0[1024], buf1[1024], cat[2048];
    const int n = nh / 2;
    const float* cur = x; int curd = cin;
    float* nxt = buf0;
    for (int l = 0; l <= n; ++l) {
        const float* w = P(pr, "%s.first_layers.%d.weight", prefix, 2 * l);
        const float* b = P(pr, "%s.first_layers.%d.bias", prefix, 2 * l);
        for (int o = 0; o < hid; ++o) {
            float acc = b[o];
            for (int i = 0; i < curd; ++i) acc += cur[i] * w[(size_t)o * curd + i];
            nxt[o] = acc > 0.f ? acc : 0.f;
        }
        cur = nxt; curd = hid; nxt = (nxt == buf0) ? buf1 : buf0;
    }
    memcpy(cat, x, cin * sizeof(float));
    memcpy(cat + cin, cur, hid * sizeof(float));
    cur = cat; curd = cin + hid; nxt = buf0;
    for (int l = 0; l < n; ++l) {
        const float* w = P(pr, "%s.second_layers.%d.weight", prefix, 2 * l);
        const float* b = P(pr, "%s.second_layers.%d.bias", prefix, 2 * l);
        for (int o = 0; o < hid; ++o) {
            float acc = b[o];
            for (int i = 0; i < curd; ++i) acc += cur[i] * w[(size_t)o * curd + i];
            nxt[o] = acc > 0.f ? acc : 0.f;
        }
        cur = nxt; curd = hid; nxt = (nxt == buf0) ? buf1 : buf0;
    }
    const float* w = P(pr, "%s.second_layers.%d.weight", prefix, 2 * n);
    const float* b = P(pr, "%s.second_layers.%d.bias", prefix, 2 * n);
    for (int o = 0; o < cout; ++o) {
        float acc = b[o];
        for (int i = 0; i < curd; ++i) acc += cur[i] * w[(size_t)o * curd + i];
        out[o] = acc;
    }
}

/* AutoEncoderGroupSkip.decode.  src/encoding/networks.py:192-220.
 * pts [N,3]; planes xy[1,12,H,W], xz[1,12,H,D], yz[1,12,W,D]; aabb[6]; out [N,4] = (sdf, sigmoid(rgb)).
 * The plane convs are evaluated once here (the reference redoes them per 16 384-point chunk with
 * identical results: src/encoding/model.py:327-330). */
ORC_API void orc_ae_decode(const orc_params* pr, const float* pts, int N, const float* xy, const float* xz,
                           const float* yz, int H, int W, int D, const float* aabb, int geo_dim, int tex_dim,
                           int up, int hid, int nh, float* out) {
    const int ph[3] = {H, H, W}, pw[3] = {W, D, D};
    const float* full[3] = {xy, xz, yz};
    const int ctot = geo_dim + tex_dim;
    (void)ctot;
    float* geo_in[3]; float* tex_in[3]; float* geo_f[3]; float* tex_f[3];
    for (int p = 0; p < 3; ++p) {
        const size_t hw = (size_t)ph[p] * pw[p];
        geo_in[p] = falloc(geo_dim * hw); tex_in[p] = falloc(tex_dim * hw);
        memcpy(geo_in[p], full[p], geo_dim * hw * sizeof(float));                       /* fm[:, :geo] */
        memcpy(tex_in[p], full[p] + geo_dim * hw, tex_dim * hw * sizeof(float));        /* fm[:, geo:] */
        geo_f[p] = falloc(up * hw); tex_f[p] = falloc(up * hw);
    }
    orc_ae_plane_block(pr, "geo_convs", (const float* const*)geo_in, ph, pw, geo_dim, up, geo_f);
    orc_ae_plane_block(pr, "tex_convs", (const float* const*)tex_in, ph, pw, tex_dim, up, tex_f);
    static const int cl[3][2] = {{0, 1}, {0, 2}, {1, 2}};
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float q[3], hg[256], ht[256], o[4];
        for (int k = 0; k < 3; ++k) q[k] = 2.f * (pts[(size_t)i * 3 + k] - aabb[k]) / (aabb[3 + k] - aabb[k]) - 1.f;
        for (int c = 0; c < up; ++c) { hg[c] = 0.f; ht[c] = 0.f; }
        for (int p = 0; p < 3; ++p) {
            sample_plane_acc(geo_f[p], up, ph[p], pw[p], q[cl[p][0]], q[cl[p][1]], hg);
            sample_plane_acc(tex_f[p], up, ph[p], pw[p], q[cl[p][0]], q[cl[p][1]], ht);
        }
        mlp_skip_concat(pr, "geo_decoder", hg, up, hid, nh, 1, o);
        mlp_skip_concat(pr, "tex_decoder", ht, up, hid, nh, 3, o + 1);
        out[(size_t)i * 4] = o[0];
        for (int k = 1; k < 4; ++k) out[(size_t)i * 4 + k] = 1.f / (1.f + expf(-o[k]));
    }
    for (int p = 0; p < 3; ++p) { free(geo_in[p]); free(tex_in[p]); free(geo_f[p]); free(tex_f[p]); }
}

/* Exposed single-op wrappers used by the leaf-level golden checks */
ORC_API void orc_triplane_conv(const orc_params* pr, const char* prefix, const float* xy, const float* xz,
                               const float* yz, int B, int C, int H, int W, int D, int Cout, int K, int rollout,
                               float* oxy, float* oxz, float* oyz) {
    tri in; in.C = C; in.p[0] = (float*)xy; in.p[1] = (float*)xz; in.p[2] = (float*)yz;
    in.h[0] = H; in.w[0] = W; in.h[1] = H; in.w[1] = D; in.h[2] = W; in.w[2] = D;
    tri o = triplane_conv(pr, prefix, &in, B, Cout, K, rollout);
    float* dst[3] = {oxy, oxz, oyz};
    for (int p = 0; p < 3; ++p) { memcpy(dst[p], o.p[p], (size_t)B * Cout * o.h[p] * o.w[p] * sizeof(float)); }
    tri_free(&o);
}
ORC_API void orc_triplane_norm_silu(const orc_params* pr, const char* prefix, const float* xy, const float* xz,
                                    const float* yz, int B, int C, int H, int W, int D,
                                    float* oxy, float* oxz, float* oyz) {
    tri in; in.C = C; in.p[0] = (float*)xy; in.p[1] = (float*)xz; in.p[2] = (float*)yz;
    in.h[0] = H; in.w[0] = W; in.h[1] = H; in.w[1] = D; in.h[2] = W; in.w[2] = D;
    tri o = triplane_norm_silu(pr, prefix, &in, B, NULL, NULL);
    float* dst[3] = {oxy, oxz, oyz};
    for (int p = 0; p < 3; ++p) { memcpy(dst[p], o.p[p], (size_t)B * C * o.h[p] * o.w[p] * sizeof(float)); }
    tri_free(&o);
}
ORC_API void orc_triplane_resblock(const orc_params* pr, const char* prefix, const float* xy, const float* xz,
                                   const float* yz, const float* emb, int B, int C, int H, int W, int D, int ted,
                                   int Cout, int ssn, int rollout, float* oxy, float* oxz, float* oyz) {
    tri in; in.C = C; in.p[0] = (float*)xy; in.p[1] = (float*)xz; in.p[2] = (float*)yz;
    in.h[0] = H; in.w[0] = W; in.h[1] = H; in.w[1] = D; in.h[2] = W; in.w[2] = D;
    tri o = triplane_resblock(pr, prefix, &in, emb, B, ted, Cout, ssn, rollout);
    float* dst[3] = {oxy, oxz, oyz};
    for (int p = 0; p < 3; ++p) { memcpy(dst[p], o.p[p], (size_t)B * Cout * o.h[p] * o.w[p] * sizeof(float)); }
    tri_free(&o);
}
ORC_API int orc_has_param(const orc_params* pr, const char* name) { return has_param(pr, name); }

/* ------------------------------------------------------------------------------------------------
 * Marching cubes (the checker of sin3dm_amd/csrc/s3d_mc.hip).  Stands for `mcubes.marching_cubes(np.pad(grid, 1,
 * constant_values=pad_value), iso)` followed by `v -= 1` (src/encoding/utils3d.py:196-203).  PyMCubes is not in this
 * environment: PARITY WITH IT IS UNPINNED.  The case table is generated by tools/gen_mc_tables.py; vertices are emitted
 * per cut edge in (axis, vertex index) order and triangles per cell in index order, which is also the device order.
 * ------------------------------------------------------------------------------------------------ */
#include "_gen/s3d_mc_tables.h"      /* generated by oracle/Makefile (tools/gen_mc_tables.py): the checker's own copy */
static float mc_s(const float* v, int X, int Y, int Z, int stride, int pad, float padv, int x, int y, int z) {
    x -= pad; y -= pad; z -= pad;
    if (x < 0 || y < 0 || z < 0 || x >= X || y >= Y || z >= Z) return padv;
    return v[(((size_t)x * Y + y) * Z + z) * stride];
}
/* two-call protocol: verts == NULL counts only.  Returns 0. */
ORC_API int orc_marching_cubes(const float* v, int X, int Y, int Z, int stride, float iso, int pad, float padv,
                               float* verts, int32_t* tris, int64_t* n_verts, int64_t* n_tris) {
    const int Xp = X + 2 * pad, Yp = Y + 2 * pad, Zp = Z + 2 * pad;
    const int64_t NV = (int64_t)Xp * Yp * Zp;
    int32_t* vid = (int32_t*)malloc(sizeof(int32_t) * 3 * NV);
    int64_t nv = 0, nt = 0;
    for (int axis = 0; axis < 3; ++axis)
        for (int64_t i = 0; i < NV; ++i) {
            const int z = (int)(i % Zp), y = (int)((i / Zp) % Yp), x = (int)(i / ((int64_t)Zp * Yp));
            const int x1 = x + (axis == 0), y1 = y + (axis == 1), z1 = z + (axis == 2);
            vid[axis * NV + i] = -1;
            if (x1 >= Xp || y1 >= Yp || z1 >= Zp) continue;
            const float v0 = mc_s(v, X, Y, Z, stride, pad, padv, x, y, z), v1 = mc_s(v, X, Y, Z, stride, pad, padv, x1, y1, z1);
            if ((v0 < iso) == (v1 < iso)) continue;
            if (verts) {
                const float t = (iso - v0) / (v1 - v0);
                float p[3] = {(float)(x - pad), (float)(y - pad), (float)(z - pad)};
                p[axis] += t;
                verts[nv * 3] = p[0]; verts[nv * 3 + 1] = p[1]; verts[nv * 3 + 2] = p[2];
            }
            vid[axis * NV + i] = (int32_t)nv++;
        }
    for (int64_t i = 0; i < NV; ++i) {
        const int z = (int)(i % Zp), y = (int)((i / Zp) % Yp), x = (int)(i / ((int64_t)Zp * Yp));
        if (x + 1 >= Xp || y + 1 >= Yp || z + 1 >= Zp) continue;
        int cs = 0;
        for (int c = 0; c < 8; ++c) cs |= (mc_s(v, X, Y, Z, stride, pad, padv, x + (c & 1), y + ((c >> 1) & 1), z + ((c >> 2) & 1)) < iso) << c;
        for (int k = 0; k < 3 * MC_NTRI[cs]; ++k) {
            if (tris) {
                const int e = MC_TRI[cs][k], c0 = MC_EDGE[e][0], axis = MC_EDGE_AXIS[e];
                const int64_t lin = ((int64_t)(x + (c0 & 1)) * Yp + (y + ((c0 >> 1) & 1))) * Zp + (z + ((c0 >> 2) & 1));
                tris[nt * 3 + k] = vid[axis * NV + lin];
            }
        }
        nt += MC_NTRI[cs];
    }
    free(vid);
    *n_verts = nv; *n_tris = nt;
    return 0;
}

/* Connected components of an indexed triangle mesh by union-find; labels[v] = smallest vertex index of v's component.
 * Checker of s3d_mesh_components (the pcu.connected_components step of sdfgrid_to_mesh, src/encoding/utils3d.py:204-208). */
static int32_t uf_find(int32_t* p, int32_t v) { while (p[v] != v) { p[v] = p[p[v]]; v = p[v]; } return v; }
ORC_API void orc_mesh_components(const int32_t* tris, int64_t nt, int64_t nv, int32_t* labels) {
    for (int64_t i = 0; i < nv; ++i) labels[i] = (int32_t)i;
    for (int64_t i = 0; i < nt; ++i)
        for (int k = 1; k < 3; ++k) {
            int32_t a = uf_find(labels, tris[i * 3]), b = uf_find(labels, tris[i * 3 + k]);
            if (a < b) labels[b] = a; else if (b < a) labels[a] = b;          /* the smaller index stays the root */
        }
    for (int64_t i = 0; i < nv; ++i) labels[i] = uf_find(labels, (int32_t)i);
}
