// Stand-alone timing harness for the Winograd convolution (tuning only).
#include "../sin3dm_amd/csrc/s3d_common.h"
#include "ub_stubs.h"
namespace s3d { void set_error(const char*, ...) {} const char* get_error() { return ""; } bool conv_use_naive() { return false; } void conv_note_kernel(const char*) {} const char* conv_last_kernel() { return ""; }
  size_t push(std::vector<float>& st, const float* src, size_t n) { size_t off = (st.size() + 63) & ~size_t(63); st.resize(off + n); if (src) memcpy(st.data() + off, src, n * 4); return off; } }
#include "../sin3dm_amd/csrc/s3d_wino.hip"
#include <cstdlib>
using namespace s3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static void run(int cin, int cout, int hw, int B, int iters, bool extras) {
    size_t npix = size_t(3) * hw * hw * B;
    float *in, *wgt, *out, *res, *tab;
    size_t wsz = size_t((cout + 31) / 32) * (cin / 8) * 16 * 256;
    CK(hipMalloc(&in, npix * cin * 4)); CK(hipMalloc(&wgt, 3 * wsz * 4)); CK(hipMalloc(&out, npix * cout * 4)); CK(hipMalloc(&res, npix * cout * 4));
    CK(hipMalloc(&tab, size_t(B) * hw * 4 * cout * 4));
    std::vector<float> h(npix * cin); for (auto& v : h) v = float(rand()) / RAND_MAX - 0.5f;
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hw_(3 * wsz); for (auto& v : hw_) v = float(rand()) / RAND_MAX - 0.5f;
    CK(hipMemcpy(wgt, hw_.data(), hw_.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(res, 0, npix * cout * 4)); CK(hipMemset(tab, 0, size_t(B) * hw * 4 * cout * 4));
    ConvArgs a; memset(&a, 0, sizeof a);
    a.B = B; a.cin = cin; a.cout = cout; a.njobs = 3;
    for (int p = 0; p < 3; ++p) {
        a.job[p].in = in + size_t(p) * hw * hw * B * cin; a.job[p].wgt = wgt + p * wsz;
        a.job[p].out = out + size_t(p) * hw * hw * B * cout; a.job[p].h = hw; a.job[p].w = hw;
        if (extras) { a.job[p].res = res + size_t(p) * hw * hw * B * cout; a.job[p].rrow = tab; a.job[p].rcol = tab; }
    }
#ifdef W_TIMING
    unsigned long long* tb; CK(hipMalloc(&tb, size_t(1 << 16) * 64)); CK(hipMemcpyToSymbol(HIP_SYMBOL(s3d::g_wtime), &tb, sizeof tb));
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_conv_wino(a, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch_conv_wino(a, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    double fl = 2.0 * 9 * cin * cout * npix;
    printf("wino cin=%4d cout=%4d hw=%3d B=%d extras=%d: %8.1f us  direct-equiv %6.1f TF  executed %6.1f TF\n", cin, cout, hw, B, extras, us, fl / us / 1e6, fl * 4 / 9 / us / 1e6);
#ifdef W_TIMING
    {   // per-block phase times of one launch (wall_clock64 ticks of 10 ns)
        int blocks = 0; for (int p = 0; p < 3; ++p) blocks += a.job[p].tiles_per_img * a.job[p].n_tiles_n * B;
        launch_conv_wino(a, 0); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(size_t(blocks) * 8);
        CK(hipMemcpy(t.data(), tb, t.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t3 = 0;
        for (int i = 0; i < blocks; ++i) { t0 = std::min(t0, t[i * 8]); t3 = std::max(t3, t[i * 8 + 3]); }
        double sp[2] = {0, 0}, sk[2] = {0, 0}, se[2] = {0, 0}, e1[2] = {0, 0}, e2[2] = {0, 0}, q1[2] = {0, 0}, q2[2] = {0, 0}; int n[2] = {0, 0};
        std::vector<int> percu(8 * 64, 0);
        for (int i = 0; i < blocks; ++i) {
            const int late = (t[i * 8] - t0) > 300;                        // started more than 3 us after the first block
            sp[late] += (t[i * 8 + 1] - t[i * 8]) * 0.01; sk[late] += (t[i * 8 + 2] - t[i * 8 + 1]) * 0.01; se[late] += (t[i * 8 + 3] - t[i * 8 + 2]) * 0.01; ++n[late];
            q1[late] += (t[i * 8 + 7] - t[i * 8]) * 0.01; q2[late] += (t[i * 8 + 4] - t[i * 8 + 7]) * 0.01;
            e1[late] += (t[i * 8 + 5] - t[i * 8 + 2]) * 0.01; e2[late] += (t[i * 8 + 6] - t[i * 8 + 5]) * 0.01;
        }
        printf("    timing: span %.1f us; first-wave blocks %d: prologue %.1f k-loop %.1f epilogue %.1f us; later blocks %d: prologue %.1f k-loop %.1f epilogue %.1f us\n",
               (t3 - t0) * 0.01, n[0], sp[0] / std::max(n[0], 1), sk[0] / std::max(n[0], 1), se[0] / std::max(n[0], 1), n[1], sp[1] / std::max(n[1], 1), sk[1] / std::max(n[1], 1), se[1] / std::max(n[1], 1));
        printf("    epilogue split (first / later): k-loop-end barrier %.1f / %.1f, transform + exchange + operand loads %.1f / %.1f us\n", e1[0] / std::max(n[0], 1), e1[1] / std::max(n[1], 1), e2[0] / std::max(n[0], 1), e2[1] / std::max(n[1], 1));
        printf("    prologue split (first / later): setup + halo landed in LDS %.1f / %.1f, barrier %.1f / %.1f us\n", q1[0] / std::max(n[0], 1), q1[1] / std::max(n[1], 1), q2[0] / std::max(n[0], 1), q2[1] / std::max(n[1], 1));
        // start-time histogram in 10 us bins
        int hist[16] = {0}; for (int i = 0; i < blocks; ++i) hist[std::min<unsigned long long>(15, (t[i * 8] - t0) / 1000)]++;
        printf("    block starts per 10 us:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\n");
        CK(hipFree(tb));
    }
#endif
    CK(hipFree(in)); CK(hipFree(wgt)); CK(hipFree(out)); CK(hipFree(res)); CK(hipFree(tab));
}
int main() {
    printf("W_ABL=%d S3D_WINO=%s\n", W_ABL, getenv("S3D_WINO") ? getenv("S3D_WINO") : "(default 4)");
    for (int c : {128, 256, 512}) run(c, 64, 128, 1, 10, false);     // 384 blocks (wino2) / 192 (wino1)
    for (int c : {128, 256, 512}) run(c, 128, 128, 1, 10, false);    // 768 / 384
    for (int co : {64, 128, 256, 512}) run(256, co, 64, 1, 10, false);   // half resolution: 96 / 192 / 384 / 768 blocks
    run(128, 256, 64, 1, 10, false);
    run(128, 128, 128, 8, 3, false);
    run(128, 128, 128, 1, 10, true);
    return 0;
}
