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

struct SamplerCoef {          // per-sample scalars of one step (gathered once per thread / block)
    float sr, srm1, c1, c2, sig_ddpm, nz;
    float sigma, ca, cb;      // ddim: sigma_t, sqrt(alpha_bar_prev), sqrt(1 - alpha_bar_prev - sigma^2)
};

__device__ __forceinline__ SamplerCoef sampler_coef(const s3d_sampler_args& a, int t) {
    SamplerCoef c;
    c.sr = a.tables[S3D_TAB_SQRT_RECIP * a.T + t]; c.srm1 = a.tables[S3D_TAB_SQRT_RECIPM1 * a.T + t];
    c.c1 = a.tables[S3D_TAB_COEF1 * a.T + t]; c.c2 = a.tables[S3D_TAB_COEF2 * a.T + t];
    c.nz = t != 0 ? 1.f : 0.f;
    c.sig_ddpm = __fmul_rn(c.nz, expf(__fmul_rn(0.5f, a.tables[S3D_TAB_LOGVAR * a.T + t])));      // nonzero_mask * exp(0.5 * log_variance)
    const float ab = a.tables[S3D_TAB_ACP * a.T + t], abp = a.tables[S3D_TAB_ACP_PREV * a.T + t];
    // eta * sqrt((1 - abp) / (1 - ab)) * sqrt(1 - ab / abp)      (:579-583)
    c.sigma = __fmul_rn(__fmul_rn(a.eta, __fsqrt_rn(__fdiv_rn(__fsub_rn(1.f, abp), __fsub_rn(1.f, ab)))), __fsqrt_rn(__fsub_rn(1.f, __fdiv_rn(ab, abp))));
    c.ca = __fsqrt_rn(abp);
    c.cb = __fsqrt_rn(__fsub_rn(__fsub_rn(1.f, abp), __fmul_rn(c.sigma, c.sigma)));
    return c;
}

// element i of the step: mo = the model's output there.  Writes sample / pred_xstart / mean as the mode asks.
__device__ __forceinline__ void sampler_element(const s3d_sampler_args& a, const SamplerCoef& c, long long i, float mo) {
    const float xt = a.x[i];
    float x0 = mo;
    if (a.mean_type == S3D_MEAN_EPSILON) x0 = __fsub_rn(__fmul_rn(c.sr, xt), __fmul_rn(c.srm1, mo));      // _predict_xstart_from_eps (:329-335)
    if (a.clip_denoised) x0 = fminf(fmaxf(x0, -1.f), 1.f);
    if (a.mode == S3D_STEP_DDIM) {
        if (a.y0 && a.mask) {                                                                               // in-painting (:568-577)
            const float m = a.mask[i];
            const float mixed = __fadd_rn(__fmul_rn(m, a.y0[i]), __fmul_rn(__fsub_rn(1.f, m), x0));
            x0 = a.is_mask_t0 ? mixed : __fadd_rn(__fmul_rn(mixed, c.nz), __fmul_rn(x0, __fsub_rn(1.f, c.nz)));
        }
        const float eps = __fdiv_rn(__fsub_rn(__fmul_rn(c.sr, xt), x0), c.srm1);                            // _predict_eps_from_xstart (:346-350)
        const float mean_pred = __fadd_rn(__fmul_rn(x0, c.ca), __fmul_rn(c.cb, eps));
        const float nv = a.noise ? a.noise[i] : 0.f;
        a.sample[i] = __fadd_rn(mean_pred, __fmul_rn(__fmul_rn(c.nz, c.sigma), nv));
        a.pred_xstart[i] = x0;
    } else {
        const float mean = __fadd_rn(__fmul_rn(c.c1, x0), __fmul_rn(c.c2, xt));                             // q_posterior_mean_variance (:218-221)
        if (a.mean) a.mean[i] = mean;
        a.pred_xstart[i] = x0;
        if (a.mode == S3D_STEP_DDPM) a.sample[i] = __fadd_rn(mean, __fmul_rn(c.sig_ddpm, a.noise[i]));
    }
}

}  // namespace s3d
