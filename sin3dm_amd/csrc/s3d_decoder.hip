// s3d_decoder.hip — AutoEncoderGroupSkip.decode on MI355X (placeholder until the fused kernels land).
#include "s3d_common.h"
using namespace s3d;
struct s3d_decoder { s3d_decoder_cfg cfg; };
extern "C" {
int s3d_decoder_create(const s3d_decoder_cfg*, s3d_decoder**) { set_error("decoder: not built yet"); return S3D_ERR_UNSUPPORTED; }
void s3d_decoder_destroy(s3d_decoder* d) { delete d; }
int s3d_decoder_num_params(const s3d_decoder*) { return S3D_ERR_UNSUPPORTED; }
int s3d_decoder_param_info(const s3d_decoder*, int, const char**, int64_t*, int*) { return S3D_ERR_UNSUPPORTED; }
int s3d_decoder_set_param(s3d_decoder*, const char*, const float*, const int64_t*, int) { return S3D_ERR_UNSUPPORTED; }
int s3d_decoder_prepare_triplane(s3d_decoder*, const float*, const float*, const float*, int, int, int, void*) { return S3D_ERR_UNSUPPORTED; }
int s3d_decoder_decode_points(s3d_decoder*, const float*, int64_t, const float*, int, float*, void*) { return S3D_ERR_UNSUPPORTED; }
int s3d_decoder_grid_dims(const float*, int, int*) { return S3D_ERR_UNSUPPORTED; }
int s3d_decoder_decode_grid(s3d_decoder*, int, const float*, float*, void*) { return S3D_ERR_UNSUPPORTED; }
}
