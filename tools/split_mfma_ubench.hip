// Upper bound of a split-operand (3 x bf16) emulation of the fp32 MFMA in a Winograd-like k-loop:
// no memory traffic, three waves per SIMD, the VALU work of the real loop (input transform) plus the operand split.
//   fp32 : per 16 channels and wave  4 freq x 8 v_mfma_f32_32x32x2_f32 (64 cycles each)   + 64 VALU
//   bf16 : per 16 channels and wave  4 freq x 6 v_mfma_f32_32x32x16_bf16 (32 cycles each) + 64 VALU + split of 32 values
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void fake_transform(f32x4 (&v)[8], float s) {      // 64 VALU like B^T d B on 8 channels x 2 halves
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = v[k] * s + v[(k + 1) & 7]; v[k] = v[k] - v[(k + 3) & 7]; }
}
// v = h + l + m exactly: three bf16 pieces by truncation (the pieces are the three bytes of the mantissa)
__device__ __forceinline__ void split3(const f32x4& a, const f32x4& b, bf16x8& h, bf16x8& l, bf16x8& m) {
    unsigned hh[8], ll[8], mm[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? a[e] : b[e - 4];
        const unsigned vb = __builtin_bit_cast(unsigned, v);
        const float hf = __builtin_bit_cast(float, vb & 0xFFFF0000u);
        const float r1 = v - hf;
        const unsigned r1b = __builtin_bit_cast(unsigned, r1);
        const float lf = __builtin_bit_cast(float, r1b & 0xFFFF0000u);
        const float r2 = r1 - lf;
        hh[e] = vb >> 16; ll[e] = r1b >> 16; mm[e] = __builtin_bit_cast(unsigned, r2) >> 16;
    }
    u32x4 ph, pl, pm;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ph[e] = hh[2 * e] | (hh[2 * e + 1] << 16); pl[e] = ll[2 * e] | (ll[2 * e + 1] << 16); pm[e] = mm[2 * e] | (mm[2 * e + 1] << 16); }
    h = __builtin_bit_cast(bf16x8, ph); l = __builtin_bit_cast(bf16x8, pl); m = __builtin_bit_cast(bf16x8, pm);
}
template <int MODE>
__global__ __launch_bounds__(256, 3) void k_loop(float* out, int iters, float s) {
    f32x16 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = f32x4{float(threadIdx.x + k), 1.f, 2.f, 3.f} * 1e-3f;
    // weights: fixed registers (their loads are not what is measured)
    f32x4 w32[4][2];
    bf16x8 wh[4], wl[4], wm[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        w32[f][0] = f32x4{1.f + f, 2.f, 3.f, 4.f} * 1e-2f; w32[f][1] = w32[f][0] * 0.5f;
        split3(w32[f][0], w32[f][1], wh[f], wl[f], wm[f]);
    }
    for (int it = 0; it < iters; ++it) {
        fake_transform(v, s);
        if (MODE == 0) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[f][e], w32[f][0][e], acc[f], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[4 + f][e], w32[f][1][e], acc[f], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                bf16x8 h, l, m;
                split3(v[f], v[4 + f], h, l, m);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, wh[f], acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, wl[f], acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l, wh[f], acc[f], 0, 0, 0);
                if (MODE == 2) {
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l, wl[f], acc[f], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, wm[f], acc[f], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m, wh[f], acc[f], 0, 0, 0);
                }
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[f][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
    float* out; CK(hipMalloc(&out, 768 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    const char* names[3] = {"fp32 MFMA (32 x 32x32x2 per 16 channels)", "bf16 x3 products (12 x 32x32x16)", "bf16 x6 products (24 x 32x32x16)"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(k_loop<0>, dim3(768), dim3(256), 0, 0, out, iters, 0.999f);
            else if (mode == 1) hipLaunchKernelGGL(k_loop<1>, dim3(768), dim3(256), 0, 0, out, iters, 0.999f);
            else hipLaunchKernelGGL(k_loop<2>, dim3(768), dim3(256), 0, 0, out, iters, 0.999f);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // fp32-equivalent flops: 4 freq x 32x32 tile x 16 channels x 2 per wave and iteration
            const double flop = double(768) * 4 * iters * 4 * 32 * 32 * 16 * 2;
            if (rep) printf("%-44s %8.1f us for %d k16-steps: %6.1f TF fp32-equivalent, %.0f cycles per step and SIMD (3 waves) at 2.0 GHz\n", names[mode], ms * 1e3, iters, flop / ms / 1e9, ms * 1e-3 / iters * 2.0e9);
        }
    }
    return 0;
}
