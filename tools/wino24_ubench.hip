// Stand-alone timing of the 3x3 kernels on the network's layer shapes (tuning / profiles/r04_wino_ubench.txt):
//   k_conv_wino4 (F(2x2), s3d_wino.hip), k_conv_wino24s and k_conv_wino24w (F(2x4), s3d_wino24.hip), random data, three square planes.
//   -DW24_TIMING: per-block phase stamps (+ -DW24_WHERE_ID and UB_WHERE=1: block durations per XCD / per CU — its own build, it slows k_conv_wino24s); -DW24W_RING=N: weight-fragment ring depth of the wide kernel
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino24_ubench.hip -o tools/ub_wino24
#include "../sin3dm_amd/csrc/s3d_common.h"
#include "ub_stubs.h"
namespace s3d { void set_error(const char*, ...) {} const char* get_error() { return ""; } bool conv_use_naive() { return false; } void conv_note_kernel(const char*) {} const char* conv_last_kernel() { return ""; }
  size_t push(std::vector<float>& st, const float* src, size_t n) { size_t off = (st.size() + 63) & ~size_t(63); st.resize(off + n); if (src) memcpy(st.data() + off, src, n * 4); return off; } }
#include "../sin3dm_amd/csrc/s3d_wino.hip"
#include "../sin3dm_amd/csrc/s3d_wino24.hip"
#include <chrono>
#include <cstring>
#include <map>
#include <algorithm>
#include <cstdlib>
using namespace s3d;
static double g_warm_s = 0.3;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#ifdef W24_TIMING
static unsigned long long* g_tb = nullptr;
static unsigned* g_idb = nullptr;
#endif
static void run(int cin, int cout, int hw, int B, int iters, bool extras) {
    const size_t npix = size_t(3) * hw * hw * B;
    float *in, *wgt, *out, *res, *tab;
    const size_t wsz = std::max(wino24_packed_floats(cout, cin), size_t((cout + 31) / 32) * (cin / 8) * 16 * 256);   // the F(2x4) and F(2x2) images
    CK(hipMalloc(&in, npix * cin * 4)); CK(hipMalloc(&wgt, 3 * wsz * 4)); CK(hipMalloc(&out, npix * cout * 4)); CK(hipMalloc(&res, npix * cout * 4));
    CK(hipMalloc(&tab, size_t(B) * hw * 4 * cout * 4));
    std::vector<float> h(npix * cin); for (auto& v : h) v = float(rand()) / RAND_MAX - 0.5f;
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hw_(3 * wsz); for (auto& v : hw_) v = float(rand()) / RAND_MAX - 0.5f;
    CK(hipMemcpy(wgt, hw_.data(), hw_.size() * 4, hipMemcpyHostToDevice));
    { std::vector<float> r(npix * cout); for (auto& v : r) v = float(rand()) / RAND_MAX - 0.5f; CK(hipMemcpy(res, r.data(), r.size() * 4, hipMemcpyHostToDevice));
      std::vector<float> t(size_t(B) * hw * 4 * cout); for (auto& v : t) v = float(rand()) / RAND_MAX - 0.5f; CK(hipMemcpy(tab, t.data(), t.size() * 4, hipMemcpyHostToDevice)); }
    const char* names[4] = {"wino4   F(2x2) 8x16 px x 32", "wino24s F(2x4) 8x16 px x 32", "wino24w F(2x4) 8x16 px x 64", "wino24g F(2x4) LDS-DMA persist"};
    const double frac[4] = {4.0 / 9, 1.0 / 3, 1.0 / 3, 1.0 / 3};
    std::vector<float> ref_out;
    // GroupNorm partial records of the epilogue (one per tile and subgroup), compared bit for bit as well
    const int tiles = ((hw + 7) / 8) * ((hw + 15) / 16), sg = cout >= 32 ? gn_subgroup(cout) : 1, nsub = cout / sg;
    const size_t gn_n = size_t(B) * 3 * nsub * tiles * 2;
    double* gn; CK(hipMalloc(&gn, gn_n * 8));
    std::vector<double> ref_gn;
    for (int k = 0; k < 4; ++k) {
        if (k == 2 && cout % 64) continue;
        if (getenv("UB_ONLY") && !strchr(getenv("UB_ONLY"), '0' + k) && k != 1) continue;
        ConvArgs a; memset(&a, 0, sizeof a);
        a.B = B; a.cin = cin; a.cout = cout; a.njobs = 3;
        for (int p = 0; p < 3; ++p) {
            a.job[p].in = in + size_t(p) * hw * hw * B * cin; a.job[p].wgt = wgt + p * wsz;
            a.job[p].out = out + size_t(p) * hw * hw * B * cout; a.job[p].h = hw; a.job[p].w = hw;
            if (extras) { a.job[p].res = res + size_t(p) * hw * hw * B * cout; a.job[p].rrow = tab; a.job[p].rcol = tab; }
            if (extras && k >= 1) { a.job[p].gn_part = gn + size_t(p) * tiles * nsub * 2; a.gn_sg = sg; a.gn_nsub = nsub; a.gn_maxparts = tiles; }
        }
        auto launch = [&]() { return k == 0 ? launch_conv_wino(a, 0) : (k == 1 ? launch_conv_wino24_narrow(a, 0) : (k == 2 ? launch_conv_wino24_wide(a, 0) : launch_conv_wino24_glds(a, 0))); };
        if (k >= 1) {                // the wide form must reproduce k_conv_wino24s bit for bit
            CK(hipMemset(out, 0xFF, npix * cout * 4)); CK(hipMemset(gn, 0xFF, gn_n * 8));
            launch(); CK(hipDeviceSynchronize());
            std::vector<double> og(gn_n); CK(hipMemcpy(og.data(), gn, gn_n * 8, hipMemcpyDeviceToHost));
            if (k == 1) ref_gn = og;
            else if (extras) {
                size_t nb_ = 0, first_ = 0;
                for (size_t i = 0; i < gn_n; ++i) if (memcmp(&og[i], &ref_gn[i], 8)) { if (!nb_) first_ = i; ++nb_; }
                if (nb_) {
                    printf("    GroupNorm partials MISMATCH: %zu of %zu doubles, first at %zu: %g vs %g\n", nb_, gn_n, first_, og[first_], ref_gn[first_]);
                    std::vector<float> oo(npix * cout); CK(hipMemcpy(oo.data(), out, oo.size() * 4, hipMemcpyDeviceToHost));
                    for (int sub = 0; sub < 3 && sub < nsub; ++sub) {          // plane 0, sample 0, tile 0: the sums from the output tensor itself
                        double s1 = 0, s2 = 0;
                        for (int y = 0; y < 8 && y < hw; ++y) for (int x = 0; x < 16 && x < hw; ++x) for (int c = sub * sg; c < (sub + 1) * sg; ++c) { const double v = oo[(size_t(y) * hw + x) * cout + c]; s1 += v; s2 += v * v; }
                        const size_t at = (size_t(sub) * tiles + 0) * 2;
                        printf("      tile 0 subgroup %d: from the output %g %g | this kernel %g %g | wino24s %g %g\n", sub, s1, s2, og[at], og[at + 1], ref_gn[at], ref_gn[at + 1]);
                    }
                }
                else printf("    GroupNorm partials: bit-identical\n");
            }
            std::vector<float> o(npix * cout);
            CK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
            if (k == 1) ref_out.swap(o);
            else {
                size_t nbad = 0, first = 0;
                for (size_t i = 0; i < o.size(); ++i) if (memcmp(&o[i], &ref_out[i], 4)) { if (!nbad) first = i; ++nbad; }
                printf("    %s vs wino24s: %s", k == 2 ? "wino24w" : "wino24g", nbad ? "MISMATCH" : "bit-identical\n");
                if (nbad) printf(" (%zu of %zu elements, first at %zu: %g vs %g)\n", nbad, o.size(), first, o[first], ref_out[first]);
            }
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        {   // steady state: the chip is power-limited under this kernel (profiles/r04_clock.txt) — measure after ~0.3 s of it, not on a cold ramp
            const auto w0 = std::chrono::steady_clock::now();
            do { for (int i = 0; i < 10; ++i) launch(); CK(hipDeviceSynchronize()); } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < g_warm_s);
        }
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, fl = 2.0 * 9 * cin * cout * npix;
        int blocks = 0; for (int p = 0; p < 3; ++p) blocks += a.job[p].tiles_per_img * a.job[p].n_tiles_n * B;
#ifdef W24_TIMING
        if (k >= 1) {   // per-block phase times of one launch (wall_clock64 ticks of 10 ns)
            unsigned long long* tb = g_tb;
            launch(); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> t(size_t(blocks) * 8);
            CK(hipMemcpy(t.data(), tb, t.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t5 = 0;
            for (int i = 0; i < blocks; ++i) { t0 = std::min(t0, t[i * 8]); t5 = std::max(t5, t[i * 8 + 5]); }
            double ph[2][5] = {{0}}; int n[2] = {0, 0};
            for (int i = 0; i < blocks; ++i) {
                const int late = (t[i * 8] - t0) > 300;
                for (int q = 0; q < 5; ++q) ph[late][q] += (t[i * 8 + q + 1] - t[i * 8 + q]) * 0.01;
                ++n[late];
            }
            if (k == 1 && getenv("UB_WHERE")) {   // where does a block's time depend on? per XCD and per CU durations of the launch's blocks
                std::vector<unsigned> id(blocks);
                CK(hipMemcpy(id.data(), g_idb, size_t(blocks) * 4, hipMemcpyDeviceToHost));
                double xs[8] = {0}, xe[8] = {0}; int xn[8] = {0};
                std::map<unsigned, std::vector<double>> cu;
                for (int i = 0; i < blocks; ++i) {
                    const int x = (id[i] >> 16) & 7; const double d = double(t[i * 8 + 5] - t[i * 8]) * 0.01;
                    xs[x] += d; xe[x] = std::max(xe[x], double(t[i * 8 + 5] - t0) * 0.01); ++xn[x];
                    cu[(id[i] >> 16 & 7) << 16 | (id[i] & 0xFF00)].push_back(d);
                }
                printf("    per XCD: block duration mean / last end:");
                for (int x = 0; x < 8; ++x) printf("  [%d] %.1f / %.1f", x, xn[x] ? xs[x] / xn[x] : 0.0, xe[x]);
                std::vector<double> cm;
                for (auto& kv : cu) { double s_ = 0; for (double v : kv.second) s_ += v; cm.push_back(s_ / kv.second.size()); }
                std::sort(cm.begin(), cm.end());
                printf("\n    per CU (%zu CUs seen): mean block duration min %.1f  10%% %.1f  median %.1f  90%% %.1f  max %.1f us; blocks per CU min %zu max %zu\n", cm.size(), cm.front(), cm[cm.size() / 10],
                       cm[cm.size() / 2], cm[cm.size() * 9 / 10], cm.back(),
                       std::min_element(cu.begin(), cu.end(), [](auto& a, auto& b) { return a.second.size() < b.second.size(); })->second.size(),
                       std::max_element(cu.begin(), cu.end(), [](auto& a, auto& b) { return a.second.size() < b.second.size(); })->second.size());
            }
            printf("    %s phases, span %.1f us:", k == 1 ? "wino24s" : (k == 2 ? "wino24w" : "wino24g"), (t5 - t0) * 0.01);
            for (int l = 0; l < 2; ++l)
                if (n[l]) printf("  [%s %d blocks] halo->LDS %.1f | first operands %.1f | k-loop %.1f | barrier + share images + operand loads %.1f | finish + stores %.1f us",
                                 l ? "later" : "first-wave", n[l], ph[l][0] / n[l], ph[l][1] / n[l], ph[l][2] / n[l], ph[l][3] / n[l], ph[l][4] / n[l]);
            printf("\n");
            // a grid of ONE block per CU (and, wide form, exactly two): what a block's phases take when it has the CU's matrix pipe to itself
            for (int per_cu = 1; k != 3 && per_cu <= (k == 2 ? 2 : 3); ++per_cu) {
                const int nb = 256 * per_cu;
                if (nb > blocks) break;
                CK(hipMemset(tb, 0, size_t(blocks) * 64));
                for (int rep = 0; rep < 3; ++rep) {
                    if (k == 1) hipLaunchKernelGGL(k_conv_wino24s, dim3(nb), dim3(256), 0, 0, a); else hipLaunchKernelGGL(k_conv_wino24w, dim3(nb), dim3(256), 0, 0, a);
                }
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(t.data(), tb, size_t(nb) * 64, hipMemcpyDeviceToHost));
                double q[5] = {0, 0, 0, 0, 0}, cs = 0, ws = 0;
                for (int i = 0; i < nb; ++i) { for (int z = 0; z < 5; ++z) q[z] += (t[i * 8 + z + 1] - t[i * 8 + z]) * 0.01; cs += double(t[i * 8 + 7] - t[i * 8 + 6]); ws += double(t[i * 8 + 3] - t[i * 8 + 2]); }
                printf("      %d block(s) per CU (%d blocks): halo->LDS %.1f | first operands %.1f | k-loop %.1f (%.0f MHz in it; MFMA time of one block at that clock %.1f us) | images %.1f | finish %.1f us\n",
                       per_cu, nb, q[0] / nb, q[1] / nb, q[2] / nb, cs / ws * 100.0, double(cin / 16) * (k == 2 ? 96 : 48) * 32 / (cs / ws * 100.0), q[3] / nb, q[4] / nb);
            }
        }
#endif
        printf("%s cin=%4d cout=%4d hw=%3d B=%d extras=%d blocks=%5d: %8.1f us  direct-equiv %6.1f TF  executed %6.1f TF (%.3f of 157.3)\n",
               names[k], cin, cout, hw, B, extras, blocks, us, fl / us / 1e6, fl * frac[k] / us / 1e6, fl * frac[k] / us / 1e6 / 157.3);
    }
    CK(hipFree(gn)); CK(hipFree(in)); CK(hipFree(wgt)); CK(hipFree(out)); CK(hipFree(res)); CK(hipFree(tab));
}
int main(int argc, char** argv) {          // arguments: indices of the cases to run (default: all)
#ifdef W24_TIMING
    CK(hipMalloc(&g_tb, size_t(1 << 16) * 64)); CK(hipMemcpyToSymbol(HIP_SYMBOL(s3d::g_w24time), &g_tb, sizeof g_tb));
    CK(hipMalloc(&g_idb, size_t(1 << 16) * 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(s3d::g_w24id), &g_idb, sizeof g_idb));
#endif
    if (getenv("UB_WARM_S")) g_warm_s = atof(getenv("UB_WARM_S"));
    const int NC = 25;
    const int cases[NC][5] = {{128, 128, 128, 1, 20},     // 0 input_blocks.0 / output_blocks.1.0.out_layers.2
                              {128, 256, 64, 1, 20},      // 1 input_blocks.1.1.in_layers.2
                              {256, 256, 64, 1, 20},      // 2 the three half-resolution 256 -> 256 layers
                              {384, 128, 128, 1, 20},     // 3 output_blocks.1.0.in_layers.2
                              {128, 128, 128, 8, 5},      // 4 batch 8 (BASELINE config 3)
                              {256, 256, 64, 8, 5},       // 5
                              {384, 128, 256, 1, 5},      // 6 (256,256,128)-sized planes (config 5), square stand-in
                              {128, 128, 128, 2, 10}, {128, 256, 64, 2, 10}, {256, 256, 64, 2, 10}, {384, 128, 128, 2, 10},     // 7-10 batch 2
                              {128, 128, 128, 4, 10}, {128, 256, 64, 4, 10}, {256, 256, 64, 4, 10}, {384, 128, 128, 4, 10},     // 11-14 batch 4
                              {128, 256, 64, 8, 5}, {384, 128, 128, 8, 5},                                                      // 15-16 the other two shapes at batch 8
                              {64, 64, 96, 4, 10}, {64, 128, 48, 4, 10},                                                       // 17-18 the training tier's widths (64-channel UNet, batch 4)
                              {32, 32, 20, 2, 10}, {96, 32, 20, 2, 10}, {32, 64, 10, 3, 10}, {160, 96, 52, 2, 10},            // 19-22 two- and six-piece items, ragged planes, odd widths
                              {64, 64, 200, 2, 5}, {96, 36, 100, 3, 5}};                                                      // 23-24 ... in persistent launches                                                      // 17-18 the training tier's widths (64-channel UNet, batch 4)
    for (int c = 0; c < NC; ++c) {
        bool on = argc < 2;
        for (int i = 1; i < argc; ++i) on |= atoi(argv[i]) == c;
        if (on) run(cases[c][0], cases[c][1], cases[c][2], cases[c][3], cases[c][4], true);
    }
    return 0;
}
