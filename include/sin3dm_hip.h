/*
 * sin3dm_hip.h — C ABI of libsin3dm_hip.so, the MI355X (gfx950) implementation of the Sin3DM
 * denoising path.  Plain pointers and sizes only; no torch types.  The reference has no FFI of its
 * own (it is pure Python/PyTorch): each entry point below names the reference Python
 * interface it stands under (paths relative to /root/reference/).  INTEGRATION.md shows the ctypes
 * stub that binds them.
 *
 * Conventions
 *   - return 0 on success, a negative s3d_status otherwise; the message is s3d_last_error()
 *     (thread-local).  Nothing throws or aborts across this boundary.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every compute call only enqueues work on it and never synchronises.
 *   - all tensors are fp32 device pointers owned by the caller, in the REFERENCE layouts
 *     (NCHW composed triplanes, [N,3] points ...); the library re-lays them out internally (NHWC).
 *   - parameters are copied and repacked into library-owned device memory; workspaces belong to the
 *     handle and grow on demand.  Handles are not thread-safe: one host thread per handle.
 */
#ifndef SIN3DM_HIP_H
#define SIN3DM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3D_ABI_VERSION 4
#define S3D_API __attribute__((visibility("default")))

typedef enum {
    S3D_OK = 0,
    S3D_ERR_INVALID = -1,     /* bad argument / shape (the shim raises AssertionError/ValueError) */
    S3D_ERR_MISSING = -2,     /* forward called with parameters not all set                      */
    S3D_ERR_HIP = -3,         /* a HIP runtime call failed                                        */
    S3D_ERR_UNSUPPORTED = -4  /* a configuration the reference itself cannot construct/run        */
} s3d_status;

S3D_API int s3d_abi_version(void);

/* Process-wide options: which of several kernel forms the library launches.  Every form is parity-tested (tests/test_hip_parity.py,
 * test_hip_train.py run the golden vectors under each); "bit-identical" forms differ in launch shape only.  name: with or without
 * the "S3D_" prefix; value: a decimal integer ("naive" / anything else for CONV_IMPL), "" or NULL = back to the library's own choice.
 * Until s3d_set_option is called for an option, the environment variable S3D_<NAME> (read once, at the option's first use) supplies
 * its value — the way earlier rounds selected forms; the call is the documented way.  Options are read at every launch; set them
 * before creating handles: a training handle's repack plan (s3d_unet_train_attach) keeps only the weight images of the forms
 * selected THEN current.
 *   WINO          24 (default) mixed Winograd F(2x4,3x3) | 4, 2: F(2x2) | 0: direct MFMA convolution          (rounding differs)
 *   WINO24W       unset: by launch size | 0 never | 1 always the 64-output-channel block                    (bit-identical)
 *   WINO24G       1: the mixed Winograd kernel with its halo staged by LDS-DMA and persistent blocks (default: off — measured on
 *                 par with or behind the register-staged kernels, profiles/r06_wino_glds.txt)                (bit-identical)
 *   VCAT          0: materialise upsample + concat in the output blocks (default: virtual concat)            (rounding differs)
 *   WGRAD_WINO    0: direct 3x3 weight gradient (rounding differs) | default (1): Winograd F(2x2) | 2: the same with the operands of
 *                 half regions double-buffered by LDS-DMA (measured slower, profiles/r06_wgrad.txt)         (1, 2: bit-identical)
 *   RANK1_SLICES  0: one K slice of the rollout tables (default: two from 256 channels)                      (rounding differs)
 *   RANK1_BATCH   0: k_rank1, one sample per block (default: k_rank1b / two-sample blocks)                   (bit-identical)
 *   CONV_IMPL     "naive": one-thread-per-output reference kernels (tests)                                   (rounding differs)
 *   CONV1X1_T     unset: by launch size | 0 never | 1 always the transposed-accumulator 1x1 epilogue         (bit-identical)
 *   GN_FUSED      unset: by launch size | 0: GroupNorm partials added ahead of the consumer | 1: inside it   (bit-identical)
 *   BWD_SIDE      0: every launch of a training step stays on the caller's stream (default: weight gradients on a side stream, the
 *                 auto-encoder's two nets as two chains)                                                     (bit-identical)
 *   GNB_FUSED     0: the GroupNorm backward's two per-channel sums always from its own read pass (default: from the epilogue of
 *                 the input-gradient convolution in front of it where that is the mixed Winograd kernel)     (rounding differs)
 *   EDGE_SIGNAL   how the backward pass's side stream learns that the edge sums are done: 1 an event on the launch's own completion
 *                 signal (default outside profilers), 0 a plain event record behind it (default under rocprofv3, whose tracing
 *                 makes the first form crawl); committed traces state which form they ran                    (bit-identical)
 * s3d_get_option: the current value, -1 when unset. */
S3D_API int s3d_set_option(const char* name, const char* value);
S3D_API int s3d_get_option(const char* name, int* value);
S3D_API const char* s3d_last_error(void);
/* number of visible HIP devices, or a negative s3d_status: lets the shim fail loudly without torch */
S3D_API int s3d_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Denoiser: TriplaneUNetModelSmall / TriplaneUNetModelSmallRaw
 *   constructor  src/diffusion/unet_triplane.py:346-449 (Raw: :544-646)
 *   forward      src/diffusion/unet_triplane.py:465-510 (Raw: :664-702)
 * ------------------------------------------------------------------------------------------ */
typedef struct s3d_unet s3d_unet;

typedef struct {
    int32_t in_channels;            /* 12 = fdim_geo + fdim_tex   (src/utils/parser_util.py:131-132) */
    int32_t model_channels;         /* 64 default, 128 for BASELINE config 2                          */
    int32_t out_channels;
    int32_t num_res_blocks;         /* only 1 is constructible in the reference       */
    int32_t n_levels;               /* len(channel_mult)                                              */
    int32_t channel_mult[8];
    int32_t use_scale_shift_norm;   /* FiLM h*(1+scale)+shift (default True) vs h+emb                 */
    int32_t is_rollout;             /* 1: TriplaneUNetModelSmall, 0: ...SmallRaw                      */
} s3d_unet_cfg;

S3D_API int s3d_unet_create(const s3d_unet_cfg* cfg, s3d_unet** out);
S3D_API void s3d_unet_destroy(s3d_unet* m);

/* Number of parameter tensors the configuration expects and the i-th one's state_dict name/shape
 * (the names of nn.Module.state_dict(), e.g. "input_blocks.1.1.in_layers.2.conv_xz.weight"). */
S3D_API int s3d_unet_num_params(const s3d_unet* m);
S3D_API int s3d_unet_param_info(const s3d_unet* m, int i, const char** name, int64_t shape[4], int* ndim);

/* load_state_dict, one tensor at a time: `data` is a HOST fp32 pointer in the PyTorch layout
 * (conv OIHW, linear [out,in]); the tensor is repacked for the kernels and uploaded. */
S3D_API int s3d_unet_set_param(s3d_unet* m, const char* name, const float* data, const int64_t* shape, int ndim);

/* forward(x, timesteps, H, W, D): x,out [B,C,H+D,W+D] device fp32; t [B] device fp32 (already in
 * the original 0..999 index space, i.e. after _WrappedModel, src/diffusion/respace.py:123-128). */
S3D_API int s3d_unet_forward(s3d_unet* m, const float* x, const float* t, int B, int H, int W, int D,
                     float* out, void* stream);

/* The timestep path alone and a forward that takes its result.  emb -> FiLM only depends on the timestep values (and the
 * weights): a sampling loop that knows them on the host (GaussianDiffusion._loop) computes each table once and reuses it
 * for every sample.  s3d_unet_film: timestep_embedding -> time_embed -> all emb_layers (nn.py:103-121,
 * unet_triplane.py:371-375, 232-238, 477, 281) for n timestep values t (device fp32) -> film [n][s3d_unet_film_width()].
 * s3d_unet_forward_film: forward() with that table; film_stride = width (one row per sample) or 0 (all samples share row 0).
 * Streams: every call on one handle uses handle-owned scratch (the workspace arena, s3d_unet_film's hidden vectors) without
 * internal events — issue the calls of a handle from ONE stream at a time, and make a stream that consumes a film table
 * produced on another stream wait for it (the Python mirror records an event with each cached table and does so). */
S3D_API int s3d_unet_film_width(const s3d_unet* m);

/* Workspace lanes: several INDEPENDENT sample chains through one handle, each on its own HIP stream (hardware queue).  The
 * reference makes N samples by batching them (src/sample.py:33-38); at small batch a denoising step is a chain of one-round,
 * latency-bound launches, and a second independent chain fills what the first leaves idle (no event edge is needed between
 * independent samples).  The packed weights are shared; everything a forward WRITES — the activation workspace, s3d_unet_film's
 * scratch, the measured-shape key — exists once per lane.  s3d_unet_select_lane(m, k) makes lane k (0 <= k < S3D_MAX_LANES;
 * created on first use; lane 0 always exists and is selected at creation) the target of the following inference calls on this
 * handle (s3d_unet_forward / _film / _forward_film / _step_film): one stream per lane at a time, calls of different lanes may
 * be in flight on different streams together.  Still ONE host thread per handle.  Training (s3d_unet_forward_train /
 * _backward) runs on lane 0; a weight update (s3d_unet_set_param, s3d_unet_repack) must not overlap ANY lane's work in flight. */
#define S3D_MAX_LANES 16
S3D_API int s3d_unet_select_lane(s3d_unet* m, int lane);
S3D_API int s3d_unet_current_lane(const s3d_unet* m);
S3D_API int s3d_unet_film(s3d_unet* m, const float* t, int n, float* film, void* stream);
S3D_API int s3d_unet_forward_film(s3d_unet* m, const float* x, const float* film, int film_stride, int B, int H, int W, int D,
                          float* out, void* stream);

/* Live kernel timing for bench.py's roofline line: HIP events are recorded on `stream` around every
 * MFMA convolution launch of each `every`-th forward (0 = off).  s3d_unet_profile_read waits for the
 * recorded events, ADDS their durations to `out` (caller zero-initialises) and recycles them.
 * flops = algorithmic flops of the launches, 2*taps*cin*cout*pixels; mfma_flops = what the
 * matrix cores really multiply for them (Winograd F(2x2,3x3): 4/9 of the direct count, F(2x4,3x3): 1/3, F(4x4,3x3): 1/4).
 * s3d_unet_profile_kernel: names of the kernels the timed launches of class cls actually dispatched since profiling was switched
 * on (" + "-joined when a class used more than one, e.g. both blockings of the mixed Winograd kernel; "" if none):
 * bench.py labels its roofline line with it instead of deriving a name from environment switches.
 * s3d_unet_profile_classes: which launch classes of a profiled forward are bracketed (bit 0: 3x3, bit 1: 1x1, bit 2: rank-1;
 * default 7).  An event pair costs the step ~3 us (38 pairs' worth per profiled step with all classes): bench.py brackets only
 * the dominant class inside its timed region and the other two in a short pass of its own afterwards. */
#define S3D_PROF_CLASSES 4
typedef struct {
    double ms[S3D_PROF_CLASSES];   /* [0] dense 3x3 (the dominant kernel; training: forward + dgrad), [1] 1x1 skip convs, [2] rank-1
                                      rollout vector convs, [3] training only: the 3x3 weight-gradient launches (k_wgrad_wino; on the
                                      backward pass's side stream their time includes sharing the chip with the main chain) */
    double flops[S3D_PROF_CLASSES];
    int64_t launches[S3D_PROF_CLASSES];
    int64_t forwards;      /* forwards that were instrumented */
    double mfma_flops[S3D_PROF_CLASSES];
} s3d_profile;
S3D_API int s3d_unet_profile(s3d_unet* m, int every);
S3D_API int s3d_unet_profile_classes(s3d_unet* m, int mask);   /* bit c = class c of s3d_profile */
S3D_API int s3d_unet_profile_read(s3d_unet* m, s3d_profile* out);
S3D_API const char* s3d_unet_profile_kernel(const s3d_unet* m, int cls);
/* ------------------------------------------------------------------------------------------
 * Sampler update: GaussianDiffusion.p_mean_variance + p_sample / ddim_sample
 *   src/diffusion/gaussian_diffusion.py:233-327, 396-440, 538-600
 * One fused element-wise kernel per step.  `tables` is a device fp32 array [S3D_TAB_ROWS][T]
 * (the float64 schedule tables gathered-then-cast exactly like _extract_into_tensor, :934-947).
 * ------------------------------------------------------------------------------------------ */
enum { S3D_TAB_SQRT_RECIP = 0, S3D_TAB_SQRT_RECIPM1, S3D_TAB_COEF1, S3D_TAB_COEF2, S3D_TAB_LOGVAR,
       S3D_TAB_ACP, S3D_TAB_ACP_PREV, S3D_TAB_ROWS };
enum { S3D_STEP_DDPM = 0, S3D_STEP_DDIM = 1, S3D_STEP_MEAN_ONLY = 2 };
enum { S3D_MEAN_START_X = 0, S3D_MEAN_EPSILON = 1 };

typedef struct {
    int32_t mode;                /* S3D_STEP_*                                                     */
    int32_t mean_type;           /* S3D_MEAN_*  (predict_xstart=True -> START_X)                    */
    int32_t clip_denoised;       /* clamp x0 to [-1,1]                                              */
    int32_t is_mask_t0;          /* ddim in-painting branch (:568-577)                              */
    float eta;                   /* ddim eta                                                        */
    int32_t T;                   /* table length                                                    */
    int64_t batch;               /* B                                                               */
    int64_t per_sample;          /* C*(H+D)*(W+D)                                                   */
    const float* model_out;      /* [B, per_sample]                                                 */
    const float* x;              /* x_t                                                             */
    const float* noise;          /* eps ~ N(0,1); may be NULL for MEAN_ONLY or ddim eta == 0        */
    const int64_t* t;            /* [B] device int64 indices into the (respaced) tables             */
    const float* tables;         /* [S3D_TAB_ROWS][T] device fp32                                   */
    const float* y0;             /* optional in-painting target / mask (both or neither)            */
    const float* mask;
    float* sample;               /* out: x_{t-1}   (NULL for MEAN_ONLY)                              */
    float* pred_xstart;          /* out                                                             */
    float* mean;                 /* out, optional (posterior mean; DDPM / MEAN_ONLY)                */
} s3d_sampler_args;

S3D_API int s3d_sampler_step(const s3d_sampler_args* a, void* stream);

/* One whole denoising step of the sampling loops (p_sample_loop_progressive / ddim_sample_loop_progressive,
 * src/diffusion/gaussian_diffusion.py:488-536, 687-734): the UNet forward on x_t = step->x with the FiLM row(s) of
 * s3d_unet_film, its output head (unet_triplane.py:441-445, 507-508) and the sampler update above in the head's launch
 * ("K8: fuse with K9") — the model output is not stored unless model_out != NULL (step->model_out is
 * ignored).  Same arithmetic in the same order as s3d_unet_forward_film followed by s3d_sampler_step: identical bits. */
S3D_API int s3d_unet_step_film(s3d_unet* m, const float* film, int film_stride, int B, int H, int W, int D,
                       const s3d_sampler_args* step, float* model_out, void* stream);
/* The same step when the caller knows what follows it (round 6; SURVEY.md section 2b lists K8 + K9 + K2 as one fusion).  in_conv is
 * TriplaneConv(in_channels, ch, 1, padding=0, is_rollout=False) (src/diffusion/unet_triplane.py:378, applied first thing in forward,
 * :482): a pointwise map of x_t with no timestep in it.  In a sampling loop the next step's x_t is this step's sample (:533-534), so
 *   - S3D_CARRY_OUT: the output head also evaluates the NEXT step's in_conv (+ its GroupNorm partial sums) on the x_{t-1} values it has
 *                  just formed and leaves it in the lane's workspace — same products, same order, same thread mapping as the in_conv
 *                  kernel: the same bits;
 *   - S3D_CARRY_IN: the caller vouches that step->x is the previous step's sample, UNMODIFIED since that call: its in_conv launch is
 *                  skipped when the previous call on this lane was an S3D_CARRY_OUT step of the same shape whose sample is step->x and
 *                  nothing (another forward, a parameter change, a workspace reallocation) came between; otherwise the flag is ignored.
 * carry_flags = 0 is s3d_unet_step_film.  Results are identical bits either way (tests/test_hip_parity.py). */
enum { S3D_CARRY_OUT = 1, S3D_CARRY_IN = 2 };
S3D_API int s3d_unet_step_film_carry(s3d_unet* m, const float* film, int film_stride, int B, int H, int W, int D,
                       const s3d_sampler_args* step, float* model_out, void* stream, int carry_flags);

/* ------------------------------------------------------------------------------------------
 * Leaf operators, exported so the parity tests can pin each kernel to the reference op it replaces
 * (SURVEY.md §8c item 3).  Planes are NCHW device tensors xy[B,C,H,W], xz[B,C,H,D], yz[B,C,W,D].
 * ------------------------------------------------------------------------------------------ */
/* TriplaneConv.forward  src/diffusion/unet_triplane.py:31-60; weights: 3 host OIHW tensors + biases */
S3D_API int s3d_op_triplane_conv(const float* const in[3], float* const out[3], int B, int C, int H, int W, int D,
                         int Cout, int ksize, int is_rollout, const float* const weight[3],
                         const float* const bias[3], void* stream);
/* TriplaneNorm + TriplaneSiLU  src/diffusion/unet_triplane.py:63-95; gamma/beta: host [C] per plane */
S3D_API int s3d_op_triplane_norm_silu(const float* const in[3], float* const out[3], int B, int C, int H, int W, int D,
                              const float* const gamma[3], const float* const beta[3], void* stream);
/* TriplaneDownsample2x / TriplaneUpsample2x / size-targeted bilinear resize
 * src/diffusion/unet_triplane.py:106-145, 494-499.  mode 0: avg-pool 2x2, 1: bilinear to (ho[i], wo[i]) */
S3D_API int s3d_op_triplane_resample(const float* const in[3], float* const out[3], int B, int C, const int hi[3],
                             const int wi[3], const int ho[3], const int wo[3], int mode, void* stream);
/* timestep_embedding  src/diffusion/nn.py:103-121; t: device [B] float, out: device [B][dim] = cos | sin */
S3D_API int s3d_op_timestep_embed(const float* t, int B, int dim, float* out, void* stream);
/* TriplaneResBlock._forward  src/diffusion/unet_triplane.py:269-311 (constructor :175-267), run through the model's own
 * block code.  emb: device [B][emb_dim] (the time_embed output); parameters by state-dict name relative to the block
 * ("in_layers.0.norm_xy.weight", "in_layers.2.conv_xy.weight", "emb_layers.1.weight", "out_layers.0...",
 * "out_layers.2...", "skip_connection.conv_xy.weight" when C != Cout), host pointers in PyTorch layouts */
S3D_API int s3d_op_triplane_resblock(const float* const in[3], float* const out[3], const float* emb, int B, int C,
                             int Cout, int H, int W, int D, int emb_dim, int use_scale_shift_norm, int is_rollout,
                             const char* const* names, const float* const* tensors, int n_tensors, void* stream);

/* ------------------------------------------------------------------------------------------
 * Decoder: AutoEncoderGroupSkip.decode + ShapeAutoEncoder.decode_batch/decode_grid
 *   src/encoding/networks.py:192-220 ; src/encoding/model.py:319-349 ; src/encoding/utils3d.py:13-25
 * ------------------------------------------------------------------------------------------ */
typedef struct s3d_decoder s3d_decoder;

typedef struct {
    int32_t geo_feat_channels;   /* 4   */
    int32_t tex_feat_channels;   /* 8   */
    int32_t feat_channel_up;     /* 64  */
    int32_t mlp_hidden_channels; /* 256 */
    int32_t mlp_hidden_layers;   /* 4   */
    int32_t tex_channels;        /* 3   */
} s3d_decoder_cfg;

S3D_API int s3d_decoder_create(const s3d_decoder_cfg* cfg, s3d_decoder** out);
S3D_API void s3d_decoder_destroy(s3d_decoder* d);
S3D_API int s3d_decoder_num_params(const s3d_decoder* d);
S3D_API int s3d_decoder_param_info(const s3d_decoder* d, int i, const char** name, int64_t shape[4], int* ndim);
S3D_API int s3d_decoder_set_param(s3d_decoder* d, const char* name, const float* data, const int64_t* shape, int ndim);
/* Runs geo_convs / tex_convs once for a triplane (the reference redoes them for every 16 384-point
 * chunk, src/encoding/model.py:327-330 -> networks.py:203-212, with identical results).
 * xy [1,Cg+Ct,H,W], xz [1,Cg+Ct,H,D], yz [1,Cg+Ct,W,D] device fp32. */
S3D_API int s3d_decoder_prepare_triplane(s3d_decoder* d, const float* xy, const float* xz, const float* yz,
                                 int H, int W, int D, void* stream);
/* net.decode(points, feat_maps, aabb): pts [N,3] device, aabb[6] host -> out [N, 1+tex_channels]
 * = (sdf, sigmoid(rgb)); clamp_color != 0 additionally applies decode_batch's clamp(0,1) (:332). */
S3D_API int s3d_decoder_decode_points(s3d_decoder* d, const float* pts, int64_t N, const float aabb[6],
                              int clamp_color, float* out, void* stream);
/* decode_grid: cell-centred grid over aabb with res_i = floor(reso*size_i/max(size)); writes
 * out [res0,res1,res2, 1+tex_channels] and the three resolutions. */
S3D_API int s3d_decoder_grid_dims(const float aabb[6], int reso, int dims[3]);
S3D_API int s3d_decoder_decode_grid(s3d_decoder* d, int reso, const float aabb[6], float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training tier: what the reference gets from autograd + torch.optim for
 * GaussianDiffusion.training_losses (src/diffusion/gaussian_diffusion.py:771-856) and
 * TrainLoop.run_step (src/diffusion/train_util.py:163-247).
 *
 * Parameters live in ONE caller-owned flat fp32 device vector: the state-dict tensors in the order of
 * s3d_unet_param_info(), each in its PyTorch layout, tightly packed (6 989 860 floats at 64 channels).
 * Gradients, Adam moments and EMA copies use the same layout, so the optimizer is one kernel and a
 * data-parallel job needs one all-reduce of the gradient vector.
 * ------------------------------------------------------------------------------------------------ */
S3D_API int64_t s3d_unet_param_numel(const s3d_unet* m);
S3D_API int s3d_unet_param_offset(const s3d_unet* m, int index, int64_t* offset);
/* Make `params` (device, numel floats) the master copy.  The library keeps the pointer (not the memory) until the
 * handle is destroyed or another vector is attached.  Call s3d_unet_repack after every change of its contents. */
S3D_API int s3d_unet_train_attach(s3d_unet* m, float* params, int64_t numel);
/* Rebuild the kernel-layout weight image (and the transposed operators of the backward pass) from the attached
 * vector: one launch on `stream`. */
S3D_API int s3d_unet_repack(s3d_unet* m, void* stream);
/* model(x, t, H, W, D) keeping the activations for one s3d_unet_backward.  Same result as s3d_unet_forward. */
S3D_API int s3d_unet_forward_train(s3d_unet* m, const float* x, const float* t, int B, int H, int W, int D, float* out,
                                   void* stream);
/* d_out: gradient of the composed output [B,Cout,H+D,W+D].  grads: numel floats, every element is overwritten
 * (= the .grad of each parameter after loss.backward() from zeroed gradients). */
S3D_API int s3d_unet_backward(s3d_unet* m, const float* d_out, float* grads, void* stream);
/* The same with progress marks for a data-parallel trainer that overlaps the gradient all-reduce with the backward pass
 * (the reference's DDP is commented out, src/diffusion/train_util.py:8-9, 98-99): events[k] (HIP events
 * owned by the caller, n_events <= 2) are recorded when a group of gradients is final — [0] out.* and output_blocks.* except
 * their emb_layers, [1] input_blocks.* except their emb_layers and in_conv.*; time_embed.* and all emb_layers.* are final when
 * the call's work on `stream` is.  (Recorded on `stream`, or — option BWD_SIDE on, the default — on the handle's side stream, which
 * finishes a group last and has been ordered behind `stream`'s progress: wait for the EVENT, as a communication stream does.) */
S3D_API int s3d_unet_backward_marked(s3d_unet* m, const float* d_out, float* grads, void* stream, void** events, int n_events);

/* q_sample (:189-207): x_t = sqrt_ac[t] * x0 + sqrt_1mac[t] * noise; tables fp32 [T] on the device, t int64 [B].
 * x0_batch_stride: elements between the batch rows of x0 — per_sample, or 0 when the batch is ONE training triplane expanded
 * (what the reference's data iterator yields, src/utils/triplane_util.py:64-69): no materialised copy of the batch is needed. */
S3D_API int s3d_train_q_sample(const float* x0, int64_t x0_batch_stride, const float* noise, const float* sqrt_ac,
                               const float* sqrt_1mac, const int64_t* t, int B, int64_t per_sample, float* x_t, void* stream);
/* terms[b][0..2] = mean((target_p - out_p)^2) over plane p = xy, xz, yz of the composed maps (:838-845), terms[b][3] =
 * (xy + xz) + yz, the reference's terms["loss"] in its order (:851); terms: [B][4]; workspace: 96*B floats;
 * target_batch_stride: C*(H+D)*(W+D), or 0 for one target shared by the batch */
S3D_API int s3d_train_mse_terms(const float* model_out, const float* target, int64_t target_batch_stride, int B, int C, int H,
                                int W, int D, float* workspace, float* terms, void* stream);
/* d_out = d( sum_{b,p} w[b][p] * terms[b][p] ) / d model_out with w[b][p] = weight[b][weight_cols == 3 ? p : 0] /
 * weight_divisor; weight: [B][weight_cols] on the device, weight_cols 1 or 3 (the trainer's loss = mean_b(weights[b] *
 * loss[b]) is weight_cols 1, weight_divisor B: train_util.py:229) */
S3D_API int s3d_train_mse_grad(const float* model_out, const float* target, int64_t target_batch_stride, const float* weight,
                               int weight_cols, float weight_divisor, int B, int C, int H, int W, int D, float* d_out,
                               void* stream);
/* torch.optim.AdamW step `step` (1-based) on the flat vectors, then update_ema (src/diffusion/nn.py:55-65) of up to
 * 4 EMA copies (device pointers in a host array). */
S3D_API int s3d_train_adamw_ema(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* const* ema,
                                const float* ema_rates, int n_ema, int64_t numel, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int step, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Auto-encoder training tier: AutoEncoderGroupSkip encode / decode / losses and their
 * gradients (src/encoding/networks.py:122-220, src/encoding/model.py:178-237).  Reuses s3d_decoder_cfg.
 * Parameters: one caller-owned flat fp32 device vector, tensors in s3d_ae_param_info() order and PyTorch layouts:
 * [geo_encoder, geo_convs, geo_decoder | tex_encoder, tex_convs, tex_decoder] — the two halves are the reference's two
 * AdamW parameter groups (networks.py:146-150); use s3d_train_adamw_ema on each half with its learning rate.
 * ------------------------------------------------------------------------------------------------ */
typedef struct s3d_ae s3d_ae;
typedef struct {
    int32_t sdf_loss;            /* 0 = l1, 1 = weightedl1 (default) */
    int32_t tex_loss;            /* 0 = l1 (default), 1 = l2, 2 = huber(delta 0.1) */
    float sdf_threshold;         /* truncation of the sdf samples */
    float tex_threshold_ratio;   /* texture loss on points with |sdf| < sdf_threshold * ratio (0.999) */
    float tex_weight;            /* 1.0 */
} s3d_ae_loss_cfg;

S3D_API int s3d_ae_create(const s3d_decoder_cfg* cfg, s3d_ae** out);
S3D_API void s3d_ae_destroy(s3d_ae* a);
S3D_API int s3d_ae_num_params(const s3d_ae* a);
S3D_API int s3d_ae_param_info(const s3d_ae* a, int i, const char** name, int64_t shape[5], int* ndim, int64_t* offset);
/* total floats; *tex_group_begin = first float of the tex_* parameter group */
S3D_API int64_t s3d_ae_param_numel(const s3d_ae* a, int64_t* tex_group_begin);
S3D_API int s3d_ae_attach(s3d_ae* a, float* params, int64_t numel);
S3D_API int s3d_ae_repack(s3d_ae* a, void* stream);              /* after every change of the attached vector */
/* The training volume [1,C,2H,2W,2D] (device, C = 1 + tex_channels: sdf, texture): reduced once to the three
 * projections the encoder needs; the volume itself is not kept. */
S3D_API int s3d_ae_set_volume(s3d_ae* a, const float* vol, int C, int X2, int Y2, int Z2, void* stream);
/* net.encode(vol): xy [1,Cg+Ct,H,W], xz [1,Cg+Ct,H,D], yz [1,Cg+Ct,W,D] */
S3D_API int s3d_ae_encode(s3d_ae* a, float* xy, float* xz, float* yz, void* stream);
/* net(vol, pts): pred [N, 1+tex_channels] (texture through the sigmoid) */
S3D_API int s3d_ae_forward(s3d_ae* a, const float* pts, int64_t N, const float aabb[6], float* pred, void* stream);
/* _forward_batch + backward: losses[2] = {sdf_loss, tex_loss} (device), pred optional, grads = d(sdf_loss+tex_loss)/d params
 * (every element overwritten).  pts [N,3], sdf [N,1], tex [N,tex_channels] device. */
S3D_API int s3d_ae_loss_grads(s3d_ae* a, const float* pts, const float* sdf, const float* tex, int64_t N, const float aabb[6],
                              const s3d_ae_loss_cfg* cfg, float* losses, float* pred, float* grads, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Iso-surface extraction: marching cubes on the device, replacing
 * `mcubes.marching_cubes(np.pad(grid, 1, constant_values=pad_value), iso)` and `v -= 1`
 * (src/encoding/utils3d.py:196-203).  Two calls because the output size is data dependent.
 * grid: value of vertex (x,y,z) at grid[((x*Y + y)*Z + z) * stride] (stride 4 reads the sdf channel of a
 * decode_grid output in place; its other channels can be interpolated onto the vertices as attributes).
 * Vertex coordinates are grid-index coordinates of the unpadded grid; triangles index the vertex array; normals
 * (right-hand rule) point from values < iso to values > iso.  Parity with PyMCubes is unpinned.
 * ------------------------------------------------------------------------------------------------ */
typedef struct s3d_mc s3d_mc;
S3D_API int s3d_mc_create(s3d_mc** out);
S3D_API void s3d_mc_destroy(s3d_mc* m);
/* pass 1: classify and scan; synchronises `stream` to return the counts.  pad = 1 surrounds the grid with pad_value. */
S3D_API int s3d_mc_count(s3d_mc* m, const float* grid, int X, int Y, int Z, int stride, float iso, int pad, float pad_value,
                         int64_t* n_verts, int64_t* n_tris, void* stream);
/* pass 2: verts [n_verts][3], attrs [n_verts][n_attr] or null (channels 1..n_attr of the grid), tris [n_tris][3] */
S3D_API int s3d_mc_extract(s3d_mc* m, float* verts, float* attrs, int n_attr, int32_t* tris, void* stream);
/* Connected components of an indexed triangle mesh (the `pcu.connected_components` step of sdfgrid_to_mesh,
 * src/encoding/utils3d.py:204-208): labels[v] = smallest vertex index of v's component.  tris [n_tris][3], labels
 * [n_verts] on the device; synchronises `stream` (iterates to a fixed point). */
S3D_API int s3d_mesh_components(const int32_t* tris, int64_t n_tris, int64_t n_verts, int32_t* labels, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIN3DM_HIP_H */
