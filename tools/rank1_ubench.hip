// Stand-alone timing of the rank-1 table kernel (tuning only): hot (back-to-back) and cold (caches swept between launches).
#include "../sin3dm_amd/csrc/s3d_common.h"
#include "ub_stubs.h"
namespace s3d { void set_error(const char*, ...) {} const char* get_error() { return ""; } bool conv_use_wino() { return false; } void wino_gn_parts(const Geo&, int*) {}
  int launch_conv_wino(ConvArgs&, hipStream_t) { return 0; } }
#include "../sin3dm_amd/csrc/s3d_conv.hip"
#include <vector>
#include <cstdlib>
using namespace s3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_touch(const float4* __restrict__ p, size_t n, float* out) {
    float4 s = make_float4(0, 0, 0, 0);
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) { const float4 v = p[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = 1.f;
}
static void run(int C, int cout, int L) {
    const int N = 4 * cout;
    float *v, *w, *out, *junk;
    CK(hipMalloc(&v, size_t(6) * L * C * 4)); CK(hipMalloc(&w, size_t(6) * 3 * N * C * 4)); CK(hipMalloc(&out, size_t(6) * L * N * 4));
    const size_t jb = size_t(512) << 20; CK(hipMalloc(&junk, jb));
    CK(hipMemset(v, 0, size_t(6) * L * C * 4)); CK(hipMemset(w, 0, size_t(6) * 3 * N * C * 4));
    ConvArgs a; memset(&a, 0, sizeof a);
    a.B = 1; a.cin = C; a.cout = N; a.njobs = 6;
    for (int j = 0; j < 6; ++j) { a.job[j].in = v + size_t(j) * L * C; a.job[j].wgt = w + size_t(j) * 3 * N * C; a.job[j].out = out + size_t(j) * L * N; a.job[j].h = 1; a.job[j].w = L; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_rank1(a, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) launch_rank1(a, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double hot = ms * 1e3 / 20;
    double cold = 0;
    for (int i = 0; i < 5; ++i) {
        CK(hipMemsetAsync(junk, i, jb, 0));
        CK(hipEventRecord(e0, 0)); launch_rank1(a, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); cold += ms * 1e3 / 5;
    }
    // warm: weights streamed once by 64 blocks (whatever XCD they land on), then 50 MB of other traffic, then the kernel
    double warm = 0;
    for (int i = 0; i < 5; ++i) {
        CK(hipMemsetAsync(junk, i, jb, 0));
        hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, 0, reinterpret_cast<const float4*>(w), size_t(6) * 3 * N * C / 4, out);
        CK(hipMemcpyAsync(junk, junk + (size_t(64) << 20) / 4, size_t(25) << 20, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e0, 0)); launch_rank1(a, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); warm += ms * 1e3 / 5;
    }
    printf("rank1 C=%3d cout=%3d L=%3d: hot %.1f us, cold (after a 512 MB memset) %.1f us, weights touched once before 50 MB of other traffic %.1f us\n", C, cout, L, hot, cold, warm);
    CK(hipFree(v)); CK(hipFree(w)); CK(hipFree(out)); CK(hipFree(junk));
}
int main() { run(128, 128, 128); run(256, 256, 64); run(128, 256, 64); run(384, 128, 128); return 0; }
