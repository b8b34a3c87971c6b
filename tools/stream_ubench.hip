// What a plain streaming kernel achieves on this box (calibration for the elementwise kernels of the step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_read(const float4* __restrict__ in, float* out, size_t n) {
    float4 s = make_float4(0, 0, 0, 0);
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) { float4 v = in[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = 1.f;
}
__global__ void k_write(float4* __restrict__ out, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) out[i] = make_float4(1, 2, 3, 4);
}
__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) out[i] = in[i];
}
__global__ void k_empty() {}
int main() {
    const size_t bytes = size_t(25) << 20, n = bytes / 16;
    float4 *a, *b; float* o; float4* junk;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64)); CK(hipMalloc(&junk, size_t(512) << 20));
    CK(hipMemset(a, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int blocks : {512, 2048, 8192, 24576}) {
        for (int mode = 0; mode < 3; ++mode) {
            double hot = 0, cold = 0;
            for (int rep = 0; rep < 6; ++rep) {
                const bool c = rep >= 3;
                if (c) hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, junk, (size_t(512) << 20) / 16);
                else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, o, n);
                else if (mode == 1) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n);
                else hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                (c ? cold : hot) += ms * 1e3 / 3;
            }
            printf("%s 25 MB, %5d blocks: after a tiny kernel %.1f us, after a 512 MB write %.1f us\n", mode == 0 ? "read " : (mode == 1 ? "write" : "copy "), blocks, hot, cold);
        }
    }
    // dependent empty kernels: the floor of a launch
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("100 dependent empty kernels: %.2f us each\n", ms * 10);
    return 0;
}
