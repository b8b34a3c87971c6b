// Stand-alone timing of the 3x3 weight-gradient kernel k_wgrad_wino (+ its split-K reduction) on the training tier's layer shapes
// (profiles/r06_wgrad.txt):  us per launch, MFMA rate, and — with -DWGW_TIMING — where a block's time goes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wgrad_ubench.hip -o tools/ub_wgrad        (+ -DWGW_TIMING -o tools/ub_wgrad_t)
#include "../sin3dm_amd/csrc/s3d_common.h"
#include "ub_stubs.h"
namespace s3d { void set_error(const char*, ...) {} const char* get_error() { return ""; } }
#include "../sin3dm_amd/csrc/s3d_bwd.hip"
#include <chrono>
#include <cstring>
#include <algorithm>
using namespace s3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#ifdef WGW_TIMING
static unsigned long long* g_tb = nullptr;
#endif
static void run(int cin, int cout, int H, int W, int D, int B, int iters) {
    const Geo g = Geo::from_hwd(H, W, D);
    const size_t npix = g.pixels() * B;
    float *a, *dy, *part, *dW;
    CK(hipMalloc(&a, npix * cin * 4)); CK(hipMalloc(&dy, npix * cout * 4));
    std::vector<float> h(npix * std::max(cin, cout));
    for (auto& v : h) v = float(rand()) / RAND_MAX - 0.5f;
    CK(hipMemcpy(a, h.data(), npix * cin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, h.data(), npix * cout * 4, hipMemcpyHostToDevice));
    WgradArgs w;
    w.B = B; w.cin = cin; w.cout = cout; w.ctot = cin; w.taps = 9; w.nplanes = 3;
    w.ksplit = wgrad_ksplit(g, B, cin, cout, 9);
    const size_t pf = wgrad_part_floats(w.ksplit, cin, cout, 9);
    CK(hipMalloc(&part, 3 * pf * 4)); CK(hipMalloc(&dW, size_t(3) * cout * cin * 9 * 4));
    w.dy.C = cout; w.dy.g = g; w.a.C = cin; w.a.g = g;
    size_t off = 0;
    for (int p = 0; p < 3; ++p) {
        w.dy.p[p] = dy + off * cout; w.a.p[p] = a + off * cin; off += size_t(g.h[p]) * g.w[p] * B;
        w.part[p] = part + p * pf; w.dW[p] = dW + size_t(p) * cout * cin * 9;
    }
    auto launch = [&]() { return launch_wgrad(w, 0, nullptr); };
    std::vector<float> ref;
    for (int mode = 2; mode >= 1; --mode) {              // 2: k_wgrad_wino (register staging, the default), 1: k_wgrad_wino_dma
    setenv("S3D_WGRAD_WINO", mode == 2 ? "1" : "2", 1);
    {
        CK(hipMemset(dW, 0xFF, size_t(3) * cout * cin * 9 * 4));
        launch(); CK(hipDeviceSynchronize());
        std::vector<float> o(size_t(3) * cout * cin * 9);
        CK(hipMemcpy(o.data(), dW, o.size() * 4, hipMemcpyDeviceToHost));
        if (mode == 2) ref.swap(o);
        else printf("    k_wgrad_wino_dma vs k_wgrad_wino: %s\n", memcmp(o.data(), ref.data(), o.size() * 4) == 0 ? "bit-identical" : "MISMATCH");
    }
    const auto w0 = std::chrono::steady_clock::now();
    do { for (int i = 0; i < 10; ++i) launch(); CK(hipDeviceSynchronize()); } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 0.3);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, fl = 2.0 * 9 * cin * cout * double(npix);
    const int blocks = 3 * (cout / 32) * (cin / 32) * w.ksplit;
    printf("%s cin=%4d cout=%4d planes (%d,%d,%d) B=%d ksplit=%3d blocks=%5d: %7.1f us (kernel + reduction)  direct-equiv %6.1f TF  executed %6.1f TF (%.3f of 157.3)\n",
           mode == 2 ? "k_wgrad_wino     3x3 Winograd F(2x2)" : "k_wgrad_wino_dma 3x3 Winograd F(2x2)", cin, cout, H, W, D, B, w.ksplit, blocks, us, fl / us / 1e6, fl * 4 / 9 / us / 1e6, fl * 4 / 9 / us / 1e6 / 157.3);
#ifdef WGW_TIMING
    if (mode == 2) {
        launch(); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(size_t(blocks) * 8);
        CK(hipMemcpy(t.data(), g_tb, t.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        double stage = 0, mfma = 0, pre = 0, post = 0, regions = 0, dur = 0;
        for (int i = 0; i < blocks; ++i) { t0 = std::min(t0, t[i * 8]); t1 = std::max(t1, t[i * 8 + 1]); }
        for (int i = 0; i < blocks; ++i) {
            stage += t[i * 8 + 2] * 0.01; mfma += t[i * 8 + 3] * 0.01; regions += double(t[i * 8 + 5]);
            post += double(t[i * 8 + 1] - t[i * 8 + 4]) * 0.01; dur += double(t[i * 8 + 1] - t[i * 8]) * 0.01;
            pre += double(t[i * 8 + 4] - t[i * 8]) * 0.01 - (t[i * 8 + 2] + t[i * 8 + 3]) * 0.01;
        }
        const double mf_region = 64.0 * 64 / 2400.0;      // 64 MFMAs of 64 cycles per wave and region at 2.4 GHz, us
        printf("    k_wgrad_wino span %.1f us; per block: %.1f regions, %.1f us = setup %.1f + regions [request -> operands in LDS %.2f | LDS -> last MFMA issued %.2f per region; "
               "MFMA time of one wave per region %.2f us, of the SIMD's three %.2f] + epilogue (dg = G^T dU G, partial stores) %.1f\n",
               (t1 - t0) * 0.01, regions / blocks, dur / blocks, pre / blocks, stage / regions, mfma / regions, mf_region, 3 * mf_region, post / blocks);
    }
#endif
    }
    CK(hipFree(a)); CK(hipFree(dy)); CK(hipFree(part)); CK(hipFree(dW));
}
int main(int argc, char** argv) {
#ifdef WGW_TIMING
    CK(hipMalloc(&g_tb, size_t(1 << 15) * 64)); CK(hipMemcpyToSymbol(HIP_SYMBOL(s3d::g_wgwtime), &g_tb, sizeof g_tb));
#endif
    // the 64-channel UNet of BASELINE configs[3] at the towerruins size (92, 128, 92), batch 4: its four 3x3 layer shapes
    run(64, 64, 92, 128, 92, 4, 20);
    run(64, 128, 46, 64, 46, 4, 20);
    run(128, 128, 46, 64, 46, 4, 20);
    run(192, 64, 92, 128, 92, 4, 20);
    // the 128-channel UNet at 128^3, batch 1 and 4
    run(128, 128, 128, 128, 128, 1, 20);
    run(128, 128, 128, 128, 128, 4, 10);
    return 0;
}
