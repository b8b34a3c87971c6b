// s3d_runtime.hip — error state, device buffers and the generic C-ABI entry points.
#include "s3d_common.h"

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

}  // namespace s3d

extern "C" {

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
