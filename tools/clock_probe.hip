// What clock and power does the 3x3 kernel really run at?  (every "k-loop at its MFMA-bound time" statement
// in DESIGN.md assumed ~2.0 GHz; the data sheet says 2.4.)
//   * a sibling thread samples rocm_smi's gpu_metrics (gfx clock of every XCD, socket power, hot-spot temperature, throttle
//     status) every 10 ms while the main thread loops a kernel for ~2 s;
//   * inside the kernel: clock64() (s_memtime, the shader-clock counter) against wall_clock64() (the constant 100 MHz counter)
//     over the k-loop of every block (W24_TIMING slots 2/3 and 6/7) -> the clock the CU saw while it issued the MFMAs;
//   * a load-free loop of v_mfma_f32_16x16x4_f32 (8 passes = 32 cycles each, three waves per SIMD, independent accumulators):
//     MFMAs per second / 32 is the matrix pipe's clock whatever the counters mean.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DW24_TIMING tools/clock_probe.hip -o tools/ub_clock -L/opt/rocm/lib -lrocm_smi64 -lpthread
#include "../sin3dm_amd/csrc/s3d_common.h"
#include "ub_stubs.h"
namespace s3d { void set_error(const char*, ...) {} const char* get_error() { return ""; } bool conv_use_naive() { return false; } void conv_note_kernel(const char*) {} const char* conv_last_kernel() { return ""; }
  bool conv_use_wino24() { return true; }
  size_t push(std::vector<float>& st, const float* src, size_t n) { size_t off = (st.size() + 63) & ~size_t(63); st.resize(off + n); if (src) memcpy(st.data() + off, src, n * 4); return off; } }
#include "../sin3dm_amd/csrc/s3d_wino24.hip"
#include <rocm_smi/rocm_smi.h>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>
using namespace s3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Sample { double t; int gfx[8]; int ngfx; int cur; double power, avg_power; int temp; unsigned throttle; };
static std::atomic<bool> g_run{false};
static std::vector<Sample> g_samples;
static int g_smi = -1;
static void sampler() {
    const auto t0 = std::chrono::steady_clock::now();
    while (g_run.load()) {
        rsmi_gpu_metrics_t m; memset(&m, 0, sizeof m);
        if (g_smi >= 0 && rsmi_dev_gpu_metrics_info_get(uint32_t(g_smi), &m) == RSMI_STATUS_SUCCESS) {
            Sample s; s.t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            s.ngfx = 0;
            for (int i = 0; i < 8; ++i) if (m.current_gfxclks[i] != 0xFFFF && m.current_gfxclks[i] != 0) s.gfx[s.ngfx++] = m.current_gfxclks[i];
            s.cur = m.current_gfxclk; s.power = m.current_socket_power; s.avg_power = m.average_socket_power; s.temp = m.temperature_hotspot; s.throttle = m.throttle_status;
            g_samples.push_back(s);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
}
static void report(const char* what, double secs) {
    if (g_samples.empty()) { printf("%-52s no rocm_smi samples\n", what); return; }
    double lo = 1e9, hi = 0, sum = 0, pw = 0, pwmax = 0; long n = 0; int tmax = 0; unsigned thr = 0;
    const size_t skip = g_samples.size() / 5;                // let the clocks settle: ignore the first fifth
    for (size_t i = skip; i < g_samples.size(); ++i) {
        const Sample& s = g_samples[i];
        const int* v = s.ngfx ? s.gfx : &s.cur; const int nv = s.ngfx ? s.ngfx : 1;
        for (int k = 0; k < nv; ++k) { if (v[k] == 0xFFFF) continue; lo = std::min(lo, double(v[k])); hi = std::max(hi, double(v[k])); sum += v[k]; ++n; }
        const double p = s.power != 0xFFFF ? s.power : s.avg_power;
        pw += p; pwmax = std::max(pwmax, p); tmax = std::max(tmax, s.temp); thr |= s.throttle;
    }
    printf("%-52s %4.1f s, %4zu samples: gfxclk min/mean/max %4.0f / %4.0f / %4.0f MHz (%d XCD readings each), socket power mean %4.0f max %4.0f W, hot spot %d C, throttle bits 0x%x\n",
           what, secs, g_samples.size() - skip, lo, n ? sum / n : 0.0, hi, g_samples.back().ngfx, pw / double(g_samples.size() - skip), pwmax, tmax, thr);
}

typedef float v4 __attribute__((ext_vector_type(4)));
// load-free MFMA loop: 12 independent accumulator chains per wave (the k-step of k_conv_wino24s), stamps around the loop
__global__ __launch_bounds__(256, 3) void k_mfma_only(unsigned long long* stamps, float* out, int iters) {
    v4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = v4{0.f, 0.f, 0.f, 0.f};
    const float a = float(threadIdx.x) * 1e-3f, b = 1e-3f;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = t;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = w1 - w0; stamps[blockIdx.x * 2 + 1] = c1 - c0; }
}

template <class F>
static double loop_for(double secs, F launch) {
    g_samples.clear(); g_run = true;
    std::thread th(sampler);
    const auto t0 = std::chrono::steady_clock::now();
    double el = 0;
    while (el < secs) { for (int i = 0; i < 50; ++i) launch(); CK(hipDeviceSynchronize()); el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    g_run = false; th.join();
    return el;
}

int main() {
    // the rocm_smi index of HIP device 0 (matched by PCI address: the box may hold more GPUs than this process sees)
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    if (rsmi_init(0) == RSMI_STATUS_SUCCESS) {
        uint32_t n = 0; rsmi_num_monitor_devices(&n);
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t bdf = 0;
            if (rsmi_dev_pci_id_get(i, &bdf) != RSMI_STATUS_SUCCESS) continue;
            if (int((bdf >> 8) & 0xFF) == prop.pciBusID && int((bdf >> 32) & 0xFFFFFFFF) == prop.pciDomainID) g_smi = int(i);
        }
        if (g_smi < 0 && n == 1) g_smi = 0;
        printf("device %s, %d CUs, HIP clockRate %d kHz; rocm_smi device %d of %u\n", prop.name, prop.multiProcessorCount, prop.clockRate, g_smi, n);
        if (g_smi >= 0) {
            rsmi_frequencies_t f; memset(&f, 0, sizeof f);
            if (rsmi_dev_gpu_clk_freq_get(uint32_t(g_smi), RSMI_CLK_TYPE_SYS, &f) == RSMI_STATUS_SUCCESS) {
                printf("sclk levels:");
                for (uint32_t i = 0; i < f.num_supported; ++i) printf(" %.0f%s", f.frequency[i] / 1e6, i == f.current ? "*" : "");
                printf(" MHz\n");
            }
            uint64_t cap = 0;
            if (rsmi_dev_power_cap_get(uint32_t(g_smi), 0, &cap) == RSMI_STATUS_SUCCESS) printf("power cap %.0f W\n", cap / 1e6);
        }
    } else printf("rsmi_init failed: no clock / power samples\n");

    unsigned long long* tb; CK(hipMalloc(&tb, size_t(1 << 16) * 64)); CK(hipMemcpyToSymbol(HIP_SYMBOL(s3d::g_w24time), &tb, sizeof tb));
    // idle
    { g_samples.clear(); g_run = true; std::thread th(sampler); std::this_thread::sleep_for(std::chrono::milliseconds(500)); g_run = false; th.join(); report("idle", 0.5); }

    // --- MFMA only
    {
        unsigned long long* st; float* out; CK(hipMalloc(&st, 768 * 16)); CK(hipMalloc(&out, 768 * 256 * 4));
        const int iters = 4000;                               // 4000 x 48 MFMAs x 32 cycles x 3 waves per SIMD = 18.4 M cycles per launch
        const double el = loop_for(2.0, [&]() { hipLaunchKernelGGL(k_mfma_only, dim3(768), dim3(256), 0, 0, st, out, iters); });
        report("load-free v_mfma_f32_16x16x4_f32 loop (3 waves/SIMD)", el);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_mfma_only, dim3(768), dim3(256), 0, 0, st, out, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(768 * 2); CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        double wsum = 0, csum = 0; for (int i = 0; i < 768; ++i) { wsum += double(h[2 * i]); csum += double(h[2 * i + 1]); }
        const double mfma_cycles = double(iters) * 48 * 32 * 3;      // per SIMD and launch
        printf("    per launch %.1f us; MFMA issue cycles per SIMD %.0f -> matrix-pipe clock >= %.0f MHz (if the pipe never idles); in-kernel clock64/wall_clock64 = %.3f -> %.0f MHz if clock64 counts shader cycles\n",
               ms * 100.0, mfma_cycles, mfma_cycles / (ms * 100.0), csum / wsum, csum / wsum * 100.0);
        printf("    executed %.1f TF fp32 (peak 157.3 at 2400 MHz x 256 CUs x 256 flop/cycle)\n", 768.0 * 4 * iters * 48 * 2048 / (ms * 100.0) / 1e6);
    }

    // --- the real kernel
    auto conv = [&](int cin, int cout, int hw, int B, bool wide) {
        const size_t npix = size_t(3) * hw * hw * B;
        float *in, *wgt, *out, *res, *tab;
        const size_t wsz = wino24_packed_floats(cout, cin);
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
            a.job[p].res = res + size_t(p) * hw * hw * B * cout; a.job[p].rrow = tab; a.job[p].rcol = tab;
        }
        auto launch = [&]() { wide ? launch_conv_wino24_wide(a, 0) : launch_conv_wino24_narrow(a, 0); };
        char what[128]; snprintf(what, sizeof what, "%s %d->%d @%d^2 B=%d", wide ? "k_conv_wino24w" : "k_conv_wino24s", cin, cout, hw, B);
        const double el = loop_for(2.0, launch);
        report(what, el);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int blocks = 0; for (int p = 0; p < 3; ++p) blocks += a.job[p].tiles_per_img * a.job[p].n_tiles_n * B;
        std::vector<unsigned long long> t(size_t(blocks) * 8);
        CK(hipMemcpy(t.data(), tb, t.size() * 8, hipMemcpyDeviceToHost));
        double wsum = 0, csum = 0, kl = 0;
        for (int i = 0; i < blocks; ++i) { wsum += double(t[i * 8 + 3] - t[i * 8 + 2]); csum += double(t[i * 8 + 7] - t[i * 8 + 6]); kl += double(t[i * 8 + 3] - t[i * 8 + 2]) * 0.01; }
        const double us = ms * 1e3 / 20, fl = 2.0 * 9 * cin * cout * npix / 3.0;
        const double mfma_cycles_block = double(cin / 16) * (wide ? 96 : 48) * 32;         // per wave = per SIMD share of one block
        printf("    %.1f us per launch, %d blocks, executed %.1f TF (%.3f of 157.3); k-loop mean %.2f us per block = %.0f MFMA cycles -> one block alone would need %.0f MHz to be MFMA-bound, %d co-resident blocks share the pipe; in-kernel clock64/wall_clock64 over the k-loops = %.3f -> %.0f MHz\n",
               us, blocks, fl / us / 1e6, fl / us / 1e6 / 157.3, kl / blocks, mfma_cycles_block, mfma_cycles_block / (kl / blocks), wide ? 2 : 3, csum / wsum, csum / wsum * 100.0);
        CK(hipFree(in)); CK(hipFree(wgt)); CK(hipFree(out)); CK(hipFree(res)); CK(hipFree(tab));
    };
    conv(128, 128, 128, 1, false);
    conv(128, 128, 128, 8, false);
    conv(128, 128, 128, 8, true);
    conv(256, 256, 64, 8, false);
    conv(256, 256, 64, 8, true);
    return 0;
}
