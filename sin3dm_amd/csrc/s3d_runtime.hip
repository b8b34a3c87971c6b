// s3d_runtime.hip — error state, device buffers and the generic C-ABI entry points.
#include "s3d_common.h"
#include <atomic>

namespace s3d {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    release();
    S3D_HIP(hipMalloc(&p, bytes));
    cap = bytes;
    return 0;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}
int upload(DevBuf& dst, const void* host, size_t bytes) {
    S3D_TRY(dst.reserve(bytes));
    S3D_HIP(hipMemcpy(dst.p, host, bytes, hipMemcpyHostToDevice));
    return 0;
}

// ---- options
struct OptDef { const char* name; bool is_impl; };
static const OptDef kOpts[OPT_COUNT] = {{"WINO", false}, {"WINO24W", false}, {"VCAT", false}, {"WGRAD_WINO", false}, {"RANK1_SLICES", false},
                                        {"RANK1_BATCH", false}, {"CONV_IMPL", true}, {"CONV1X1_T", false}, {"GN_FUSED", false}, {"BWD_SIDE", false}, {"GNB_FUSED", false}, {"WINO24G", false}, {"EDGE_SIGNAL", false}};
static std::atomic<int> g_opt[OPT_COUNT];
static std::atomic<int> g_opt_state[OPT_COUNT];          // 0: not looked at yet, 1: resolved (environment or unset), 2: set through the ABI
static int parse_opt(int o, const char* v) { return kOpts[o].is_impl ? (strcmp(v, "naive") == 0 ? 1 : 0) : atoi(v); }
int opt(Opt o) {
    if (g_opt_state[o].load(std::memory_order_acquire) == 0) {
        const std::string env = std::string("S3D_") + kOpts[o].name;
        const char* e = getenv(env.c_str());
        int expect = 0;
        const int v = e ? parse_opt(o, e) : kOptUnset;
        // (a concurrent s3d_set_option wins: its state is 2)
        if (g_opt_state[o].compare_exchange_strong(expect, -1, std::memory_order_acq_rel)) {
            g_opt[o].store(v, std::memory_order_relaxed);
            g_opt_state[o].store(1, std::memory_order_release);
        } else while (g_opt_state[o].load(std::memory_order_acquire) < 0) {}
    }
    return g_opt[o].load(std::memory_order_relaxed);
}
static int find_opt(const char* name) {
    if (!name) return -1;
    if (strncmp(name, "S3D_", 4) == 0) name += 4;
    for (int o = 0; o < OPT_COUNT; ++o) if (strcmp(name, kOpts[o].name) == 0) return o;
    return -1;
}
int device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    n = 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
    return n;
}

}  // namespace s3d

extern "C" {

int s3d_set_option(const char* name, const char* value) {
    using namespace s3d;
    const int o = find_opt(name);
    S3D_CHECK(o >= 0, S3D_ERR_INVALID, "set_option: unknown option '%s'", name ? name : "(null)");
    g_opt[o].store(value && *value ? parse_opt(o, value) : kOptUnset, std::memory_order_relaxed);
    g_opt_state[o].store(2, std::memory_order_release);
    return 0;
}

int s3d_get_option(const char* name, int* value) {
    using namespace s3d;
    const int o = find_opt(name);
    S3D_CHECK(o >= 0 && value, S3D_ERR_INVALID, "get_option: unknown option '%s'", name ? name : "(null)");
    *value = opt(Opt(o));
    return 0;
}

int s3d_abi_version(void) { return S3D_ABI_VERSION; }
const char* s3d_last_error(void) { return s3d::get_error(); }

int s3d_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        s3d::set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return S3D_ERR_HIP;
    }
    return n;
}

int s3d_sampler_step(const s3d_sampler_args* a, void* stream) {
    using namespace s3d;
    S3D_CHECK(a != nullptr, S3D_ERR_INVALID, "sampler_step: null args");
    S3D_CHECK(a->model_out && a->x && a->t && a->tables && a->pred_xstart, S3D_ERR_INVALID,
              "sampler_step: model_out, x, t, tables and pred_xstart are required");
    S3D_CHECK(a->mode == S3D_STEP_DDPM || a->mode == S3D_STEP_DDIM || a->mode == S3D_STEP_MEAN_ONLY, S3D_ERR_INVALID,
              "sampler_step: bad mode %d", a->mode);
    S3D_CHECK(a->mode == S3D_STEP_MEAN_ONLY || a->sample, S3D_ERR_INVALID, "sampler_step: sample output required");
    S3D_CHECK(a->mode != S3D_STEP_DDPM || a->noise, S3D_ERR_INVALID, "sampler_step: DDPM step needs noise");
    S3D_CHECK(a->mode != S3D_STEP_DDIM || a->eta == 0.f || a->noise, S3D_ERR_INVALID, "sampler_step: eta>0 needs noise");
    S3D_CHECK((a->y0 == nullptr) == (a->mask == nullptr), S3D_ERR_INVALID, "sampler_step: y0 and mask go together");
    S3D_CHECK(a->T > 0 && a->batch >= 0 && a->per_sample >= 0, S3D_ERR_INVALID, "sampler_step: bad sizes");
    return launch_sampler(*a, static_cast<hipStream_t>(stream));
}

}  // extern "C"
