// Stand-alone timing harness for the MFMA convolution (tuning only; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DS3D_ABLATE=k] tools/conv_ubench.hip -o /tmp/ub && /tmp/ub
#include "../sin3dm_amd/csrc/s3d_common.h"
#include "ub_stubs.h"
namespace s3d { void set_error(const char*, ...) {} const char* get_error() { return ""; }
  // (the Winograd side of launch_conv is not linked into this harness)
  bool conv_use_wino24() { return false; } bool conv_use_wino() { return false; } void wino_gn_parts(const Geo&, int*) {} double wino_exec_fraction() { return 1.0; }
  int launch_conv_wino24s(ConvArgs&, hipStream_t) { return -1; } int launch_conv_wino(ConvArgs&, hipStream_t) { return -1; } }
#include "../sin3dm_amd/csrc/s3d_conv.hip"
#include <vector>
#include <cstdlib>
using namespace s3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <class CFG>
static double run(const char* name, int cin, int cout, int hw, int B, int iters) {
    size_t npix = size_t(3) * hw * hw * B;
    float *in, *wgt, *out;
    CK(hipMalloc(&in, npix * cin * 4)); CK(hipMalloc(&wgt, size_t(3) * CFG::KH * CFG::KW * cout * cin * 4)); CK(hipMalloc(&out, npix * cout * 4));
    std::vector<float> h(npix * cin); for (auto& v : h) v = float(rand()) / RAND_MAX - 0.5f;
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hw_(size_t(3) * CFG::KH * CFG::KW * cout * cin); for (auto& v : hw_) v = float(rand()) / RAND_MAX - 0.5f;
    CK(hipMemcpy(wgt, hw_.data(), hw_.size() * 4, hipMemcpyHostToDevice));
    ConvArgs a; memset(&a, 0, sizeof a);
    a.B = B; a.cin = cin; a.cout = cout; a.njobs = 3;
    for (int p = 0; p < 3; ++p) {
        a.job[p].in = in + size_t(p) * hw * hw * B * cin; a.job[p].wgt = wgt + size_t(p) * CFG::KH * CFG::KW * cout * cin;
        a.job[p].out = out + size_t(p) * hw * hw * B * cout; a.job[p].h = hw; a.job[p].w = hw;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_cfg<CFG>(a, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch_cfg<CFG>(a, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    double fl = 2.0 * CFG::KH * CFG::KW * cin * cout * npix;
    printf("ABL=%d %-28s cin=%4d cout=%4d hw=%3d B=%d: %8.1f us  %6.1f TF\n", S3D_ABLATE, name, cin, cout, hw, B, us, fl / us / 1e6);
    CK(hipFree(in)); CK(hipFree(wgt)); CK(hipFree(out));
    return us;
}
int main() {
    // the 1x1 launches of the scored step (plain epilogue: no residual here)
    run<ConvCfg<8, 8, 1, 1, 2, 2, 1, 1, 1, 2, false>>("1x1 8x8 px x64 PF=2", 128, 128, 128, 1, 20);
    run<ConvCfg<8, 8, 1, 1, 2, 2, 1, 1, 1, 2, false>>("1x1 8x8 px x64 PF=2", 256, 128, 64, 1, 20);
    run<ConvCfg<8, 8, 1, 1, 2, 2, 1, 1, 1, 2, false>>("1x1 8x8 px x64 PF=2", 128, 256, 64, 1, 20);
    if (getenv("UB_1X1_ONLY")) return 0;
    run<ConvCfg<8, 8, 3, 3, 2, 2, 1, 1, 2>>("8x8 px x64 KS=2", 128, 128, 128, 1, 20);
    run<ConvCfg<8, 8, 3, 3, 2, 2, 1, 1, 4>>("8x8 px x64 KS=4", 128, 128, 128, 1, 20);
    run<ConvCfg<8, 16, 3, 3, 4, 1, 1, 2, 2>>("8x16 px x64 KS=2", 128, 128, 128, 1, 20);
    run<ConvCfg<8, 8, 3, 3, 2, 2, 1, 1, 4>>("8x8 px x64 KS=4", 256, 256, 64, 1, 20);
    run<ConvCfg<8, 8, 3, 3, 2, 2, 1, 1, 4>>("8x8 px x64 KS=4", 128, 128, 128, 8, 5);
    run<ConvCfg<8, 16, 3, 3, 4, 1, 1, 2, 2>>("8x16 px x64 KS=2", 128, 128, 128, 8, 5);
    run<ConvCfg<8, 8, 3, 3, 2, 2, 1, 1>>("8x8 px x64 (2x2 waves)", 128, 128, 128, 1, 20);
    run<ConvCfg<8, 16, 3, 3, 4, 1, 1, 2>>("8x16 px x64 (4x1 waves)", 128, 128, 128, 1, 20);
    run<ConvCfg<8, 8, 3, 3, 2, 2, 1, 1>>("8x8 px x64 (2x2 waves)", 256, 256, 64, 1, 20);
    run<ConvCfg<8, 8, 3, 3, 2, 2, 1, 1>>("8x8 px x64 (2x2 waves)", 128, 128, 128, 8, 5);
    run<ConvCfg<8, 16, 3, 3, 4, 1, 1, 2>>("8x16 px x64 (4x1 waves)", 128, 128, 128, 8, 5);
    return 0;
}
