// s3d_sampler.h — the element-wise sampler update, shared by the stand-alone kernel (k_sampler, s3d_sampler_step) and the
// output head that applies it to the 12 model outputs it holds for a pixel (k_out_head_px<.., true>, SURVEY.md §2b K8 + K9).
// p_mean_variance (START_X | EPSILON, FIXED_* variance, clip) + p_sample / ddim_sample:
// src/diffusion/gaussian_diffusion.py:233-327, 329-335, 346-350, 396-440, 538-600.  Coefficients are the fp32 casts of the
// float64 tables, gathered per sample by t (as _extract_into_tensor does, :934-947).
// The reference evaluates these expressions as separate fp32 tensor operations, each rounded: every product, sum and quotient
// below is an explicit round-to-nearest operation in the reference's order — no fused multiply-add (the EPSILON branch
// subtracts two products of magnitude sqrt(1/alphas_cumprod) ~ 157 at t = 999: a contracted fma differs from the reference in
// the fourth decimal of a clamped x0).
#pragma once
#include "s3d_common.h"

namespace s3d {

// (HIP's __fmul_rn / __fadd_rn are plain `*` / `+` unless OCML_BASIC_ROUNDED_OPERATIONS is defined, i.e. still contractable:
// the functions below switch contraction off with `#pragma clang fp contract(off)`, which travels with the operations when they
// are inlined into a kernel.)
struct SamplerCoef {          // per-sample scalars of one step (gathered once per thread / block)
    float sr, srm1, c1, c2, sig_ddpm, nz;
    float sigma, ca, cb;      // ddim: sigma_t, sqrt(alpha_bar_prev), sqrt(1 - alpha_bar_prev - sigma^2)
};

__device__ __forceinline__ SamplerCoef sampler_coef(const s3d_sampler_args& a, int t) {
#pragma clang fp contract(off)
    SamplerCoef c;
    c.sr = a.tables[S3D_TAB_SQRT_RECIP * a.T + t]; c.srm1 = a.tables[S3D_TAB_SQRT_RECIPM1 * a.T + t];
    c.c1 = a.tables[S3D_TAB_COEF1 * a.T + t]; c.c2 = a.tables[S3D_TAB_COEF2 * a.T + t];
    c.nz = t != 0 ? 1.f : 0.f;
    c.sig_ddpm = c.nz * expf(0.5f * a.tables[S3D_TAB_LOGVAR * a.T + t]);      // nonzero_mask * exp(0.5 * log_variance)
    const float ab = a.tables[S3D_TAB_ACP * a.T + t], abp = a.tables[S3D_TAB_ACP_PREV * a.T + t];
    // eta * sqrt((1 - abp) / (1 - ab)) * sqrt(1 - ab / abp)      (:579-583)
    c.sigma = (a.eta * sqrtf((1.f - abp) / (1.f - ab))) * sqrtf(1.f - ab / abp);
    c.ca = sqrtf(abp);
    c.cb = sqrtf((1.f - abp) - c.sigma * c.sigma);
    return c;
}

// element i of the step: mo = the model's output there, xt = x_t[i], nv = the step's eps there (0 when the step has none).
// Writes sample / pred_xstart / mean as the mode asks; returns the value written to `sample` (x_{t-1}; 0 for MEAN_ONLY).
__device__ __forceinline__ float sampler_element(const s3d_sampler_args& a, const SamplerCoef& c, long long i, float mo, float xt, float nv) {
#pragma clang fp contract(off)
    float x0 = mo;
    float xprev = 0.f;
    if (a.mean_type == S3D_MEAN_EPSILON) {                                                                  // _predict_xstart_from_eps (:329-335)
        const float p1 = c.sr * xt, p2 = c.srm1 * mo;
        x0 = p1 - p2;
    }
    if (a.clip_denoised) x0 = fminf(fmaxf(x0, -1.f), 1.f);
    if (a.mode == S3D_STEP_DDIM) {
        if (a.y0 && a.mask) {                                                                               // in-painting (:568-577)
            const float m = a.mask[i];
            const float q1 = m * a.y0[i], q2 = (1.f - m) * x0;
            const float mixed = q1 + q2;
            if (a.is_mask_t0) x0 = mixed;
            else { const float r1 = mixed * c.nz, r2 = x0 * (1.f - c.nz); x0 = r1 + r2; }
        }
        const float p1 = c.sr * xt;
        const float eps = (p1 - x0) / c.srm1;                                                               // _predict_eps_from_xstart (:346-350)
        const float m1 = x0 * c.ca, m2 = c.cb * eps;
        const float mean_pred = m1 + m2;
        const float nzs = (c.nz * c.sigma) * nv;
        xprev = mean_pred + nzs;
        a.sample[i] = xprev;
        a.pred_xstart[i] = x0;
    } else {
        const float m1 = c.c1 * x0, m2 = c.c2 * xt;                                                         // q_posterior_mean_variance (:218-221)
        const float mean = m1 + m2;
        if (a.mean) a.mean[i] = mean;
        a.pred_xstart[i] = x0;
        if (a.mode == S3D_STEP_DDPM) { const float nzs = c.sig_ddpm * nv; xprev = mean + nzs; a.sample[i] = xprev; }
    }
    return xprev;
}
__device__ __forceinline__ float sampler_element(const s3d_sampler_args& a, const SamplerCoef& c, long long i, float mo) {
    return sampler_element(a, c, i, mo, a.x[i], a.noise ? a.noise[i] : 0.f);
}

}  // namespace s3d
