// s3d_model.h — the denoiser handle (parameter registry, packed-weight offsets, workspace) and the forward-pass
// helpers shared by s3d_unet.hip (inference) and s3d_train.hip (training tier).
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>

#include "s3d_common.h"

namespace s3d {

static const char* kPlane[3] = {"xy", "xz", "yz"};

struct ParamSpec {
    std::string name;
    std::vector<int64_t> shape;
    size_t numel() const { size_t n = 1; for (auto d : shape) n *= size_t(d); return n; }
};

struct NormW { size_t gamma[3], beta[3]; };
struct ResBlockW {
    std::string prefix;
    int C, Cout;
    NormW n1, n2;
    ConvW c1, c2, skip;
    // output blocks that take [upsampled | skip]: the 1x1 skip_connection split by input half (inference): since a 1x1
    // convolution commutes with bilinear upsampling, skip(cat[up(h), s]) = up(W_a h) + W_b s + bias
    ConvW skip_a, skip_b;
    int c_up = 0;                      // channels of the upsampled half (0: the block's input is not a concat)
    bool has_skip;
    int film_off;                      // offset of this block's emb_layers output in the concatenated FiLM row
};

// ---- training tier (s3d_train.hip): what the forward pass leaves behind for the backward pass
struct NormTape { GnStats stats{nullptr}; MeanVecs mv{}; bool roll = false; };
struct RBTape {
    const ResBlockW* rb = nullptr;
    int index = 0;                    // position in in_blocks (is_out = false) or out_blocks
    bool is_out = false;
    Tri x, y1, h1, y2;
    NormTape n1, n2;
};
struct Tape {
    int B = 0, H = 0, W = 0, D = 0;
    const float* x = nullptr;         // composed input [B,Cin,H+D,W+D]
    const float* t = nullptr;
    float *pre1 = nullptr, *emb = nullptr, *film = nullptr;
    Tri h0;                           // in_conv output
    std::vector<RBTape> in_rb, out_rb;
    std::vector<Tri> cat_in;          // per output block: its (concatenated) input
    std::vector<Tri> up_src;          // per output block oi > 0: the previous block's output that was upsampled
    Tri head_in;
    GnStats head_stats{nullptr};
    size_t arena_off = 0;             // bump offset after the forward pass; the backward pass continues from here
    bool valid = false;
};
// training-only packed weights (offsets into s3d_unet::tbuf): the transposed operators of the backward pass
struct ConvWT {
    size_t dense_T[3] = {0, 0, 0};    // [taps][cin][cout], taps flipped: dgrad as a forward convolution
    size_t wino_T[3] = {0, 0, 0};     // the same operator in Winograd fragment order (3x3 only)
    size_t wino24s_T[3] = {0, 0, 0};  // ... and in the mixed F(2x4) order of k_conv_wino24s
    bool has_wino24s_T = false, has_wino_T = false;
    bool only24_current = false;      // of the transposed 3x3 images only wino24s_T is rewritten per step (see ConvW::only24_current)
    size_t rrow_T[3] = {0, 0, 0};     // [3 taps][cin][3*cout]: d(row-varying mean vector) from the row sums of dy
    size_t rcol_T[3] = {0, 0, 0};
};
struct ResBlockWT { ConvWT c1, c2, skip; };
enum PackKind { PK_COPY = 0, PK_TRANS2D, PK_DENSE, PK_DENSE_SLICE, PK_DENSE_T, PK_WINO, PK_WINO_T, PK_WINO24S, PK_RANK1, PK_RANK1_BWD, PK_DENSE_PAD, PK_DENSE_T_PAD, PK_WINO24S_T, PK_RANK1F };
struct PackDesc {                     // one device-side repacking job: flat reference-layout parameter -> packed image
    int kind, cout, ctot, cin, slot, taps, col_varying, to_tbuf;
    long long src, dst, n;
    int block_begin, pad;
};

}  // namespace s3d

using namespace s3d;

struct s3d_unet {
    s3d_unet_cfg cfg;
    std::vector<ParamSpec> specs;
    std::map<std::string, std::vector<float>> host;     // as handed to set_param (PyTorch layouts)
    bool packed = false;

    // packed parameters, one device allocation, offsets in floats
    DevBuf wbuf;
    std::vector<float> stage;                            // host staging of the packed image
    size_t te0_w, te0_b, te2_w, te2_b, film_w, film_b;
    int film_total = 0;
    size_t in_wT, in_b, out_w, out_b;
    NormW out_norm;
    std::vector<ResBlockW> in_blocks, out_blocks;

    Arena arena;
    DevBuf film_ws;                                      // s3d_unet_film: the two hidden vectors of the timestep MLP
    long long inf_key[4] = {-1, -1, -1, -1};             // (B,H,W,D) of the last measured inference forward ...
    size_t inf_high = 0;                                 // ... and the workspace it needs
    // Workspace lanes (s3d_unet_select_lane): independent sample chains on different HIP streams through ONE handle — the weights
    // are shared, everything a forward writes (the activation arena, the timestep MLP's scratch, the measured-shape key) exists
    // once per lane.  The fields above ARE the selected lane; the others are parked here.
    // The next step's in_conv, left in this lane's workspace by the previous fused step's output head (s3d_unet_step_film_carry):
    // valid for exactly one following step on the tensor `sample` of that shape; any other forward, a workspace reallocation or
    // a parameter change drops it.
    struct CarryState { bool valid = false; const float* sample = nullptr; long long key[4] = {-1, -1, -1, -1}; };
    CarryState carry;
    struct LaneState {
        DevBuf arena_buf, film_ws;
        size_t off = 0, high = 0, inf_high = 0;
        long long inf_key[4] = {-1, -1, -1, -1};
        CarryState carry;
    };
    std::vector<std::unique_ptr<LaneState>> lanes;       // lanes[k] holds lane k's state while another lane is selected
    int cur_lane = 0;
    void swap_lane(LaneState& L) {
        std::swap(arena.buf.p, L.arena_buf.p); std::swap(arena.buf.cap, L.arena_buf.cap);
        std::swap(arena.off, L.off); std::swap(arena.high, L.high);
        std::swap(film_ws.p, L.film_ws.p); std::swap(film_ws.cap, L.film_ws.cap);
        std::swap(inf_high, L.inf_high);
        for (int k = 0; k < 4; ++k) std::swap(inf_key[k], L.inf_key[k]);
        std::swap(carry, L.carry);
    }
    void drop_carries() { carry.valid = false; for (auto& L : lanes) if (L) L->carry.valid = false; }

    // training tier: caller-owned flat master parameters (reference layouts, specs order, tightly packed), the
    // device-side repack plan and the activation tape of the last forward_train
    float* flat = nullptr;
    int64_t flat_numel = 0;
    std::vector<size_t> flat_off;
    std::map<std::string, size_t> flat_index;            // parameter name -> offset in the flat vector
    long long train_key[4] = {-1, -1, -1, -1};           // (B,H,W,D) of the last measured forward+backward
    size_t train_high = 0;
    DevBuf tbuf, descs_dev;
    std::vector<PackDesc> descs;
    int pack_blocks = 0;
    std::vector<ResBlockWT> in_blocks_t, out_blocks_t;
    Tape tape;
    const float* tdev(size_t off) const { return static_cast<const float*>(tbuf.p) + off; }
    // backward pass: the weight-gradient launches (no consumer before the optimizer) run on a handle-owned side stream beside the
    // chain that carries the input gradients (s3d_train.hip: Bwd::fork / join)
    hipStream_t bwd_side = nullptr;       // weight gradients (low priority: they only have to be done by the end of the pass)
    std::vector<hipEvent_t> bwd_events;

    // optional live timing of the convolution launches (s3d_unet_profile)
    struct ProfRec { int cls; hipEvent_t e0, e1; double flops, mfma_flops; };
    int prof_every = 0;
    long fwd_count = 0;
    bool prof_now = false;
    int prof_mask = 15;               // launch classes (bit 0: 3x3, 1: 1x1, 2: rank-1, 3: 3x3 weight gradient) whose launches are bracketed in a profiled forward
    bool prof_train = false;          // the last training forward was a profiled one: its backward pass times its dgrad convolutions too
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    int64_t prof_forwards = 0;
    std::string prof_kernel[S3D_PROF_CLASSES];                          // every distinct kernel the timed launches of a class dispatched, " + "-joined
    void note_prof_kernel(int cls) {
        if (cls < 0 || cls >= S3D_PROF_CLASSES) return;
        const std::string n = conv_last_kernel();
        if (n.empty() || prof_kernel[cls].find(n) != std::string::npos) return;
        prof_kernel[cls] += (prof_kernel[cls].empty() ? "" : " + ") + n;
    }
    hipEvent_t prof_event() {
        hipEvent_t e = nullptr;
        if (!prof_pool.empty()) { e = prof_pool.back(); prof_pool.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        return e;
    }
    int timed_conv(int cls, ConvKind kind, ConvArgs& ca, hipStream_t st) {
        if (!prof_now || !((prof_mask >> cls) & 1)) return launch_conv(kind, ca, st);
        // algorithmic flops of the layer (direct-convolution count); the Winograd path executes 4/9 of them on the MFMA
        int taps = kind == CONV_3x3 ? 9 : (kind == CONV_1x1 ? 1 : (kind == CONV_1x3_VEC ? 3 : (kind == CONV_1x3_ROLL ? 9 : 25)));
        double pix = 0;
        for (int j = 0; j < ca.njobs; ++j) pix += double(ca.job[j].h) * ca.job[j].w;
        ProfRec r{cls, prof_event(), prof_event(), 2.0 * taps * ca.cin * ca.cout * pix * ca.B, 0.0};
        r.mfma_flops = r.flops * conv_exec_fraction(kind, ca);
        if (r.e0) (void)hipEventRecord(r.e0, st);
        int rc = launch_conv(kind, ca, st);
        note_prof_kernel(cls);
        if (r.e1) (void)hipEventRecord(r.e1, st);
        prof_recs.push_back(r);
        return rc;
    }
    template <class F>
    int timed_launch(int cls, double flops, double mfma_flops, hipStream_t st, F fn) {
        if (!prof_now || !((prof_mask >> cls) & 1)) return fn();
        ProfRec r{cls, prof_event(), prof_event(), flops, mfma_flops};
        if (r.e0) (void)hipEventRecord(r.e0, st);
        int rc = fn();
        note_prof_kernel(cls);
        if (r.e1) (void)hipEventRecord(r.e1, st);
        prof_recs.push_back(r);
        return rc;
    }
    ~s3d_unet() {
        for (auto e : bwd_events) (void)hipEventDestroy(e);
        if (bwd_side) (void)hipStreamDestroy(bwd_side);
        for (auto& r : prof_recs) { if (r.e0) (void)hipEventDestroy(r.e0); if (r.e1) (void)hipEventDestroy(r.e1); }
        for (auto e : prof_pool) (void)hipEventDestroy(e);
    }

    const float* dev(size_t off) const { return static_cast<const float*>(wbuf.p) + off; }
};

namespace s3d {

int pack_all(s3d_unet* m);
int finalize_pack_plan(std::vector<PackDesc>& descs, DevBuf& dev, int& blocks_out);   // block prefix + per-block job table, uploaded
int launch_repack_generic(const PackDesc* descs_dev, int ndesc, int blocks, const float* flat, float* wbuf, float* tbuf, hipStream_t st);
// ext_film != null: the FiLM table [B or 1][film_total] is given (s3d_unet_film), t is not read
// fuse != null: one denoising step — the sampler update is applied to the model output by the output head; `out` may then be
// null (the model output is not stored)
int run_forward(s3d_unet* m, const float* x, const float* t, int B, int H, int W, int D, float* out, hipStream_t st,
                Tape* tape, const float* ext_film = nullptr, int ext_film_stride = 0, const s3d_sampler_args* fuse = nullptr,
                int carry_flags = 0);

struct Fwd {
    s3d_unet* m;
    int B;
    hipStream_t st;
    const float* film;      // [B][film_total] (row b at film + b * film_stride)
    int film_stride = -1;   // -1: m->film_total; 0: one row shared by the whole batch (s3d_unet_forward_film)
    int fstride() const { return film_stride >= 0 ? film_stride : m->film_total; }
    Tape* tape = nullptr;   // training: record what the backward pass needs
    RBTape last_rb;         // filled by resblock() when tape is set
    Arena& ar() { return m->arena; }

    Tri alloc_tri(int C, const Geo& g) {
        Tri t; t.C = C; t.g = g;
        for (int p = 0; p < 3; ++p) t.p[p] = ar().alloc<float>(size_t(B) * g.h[p] * g.w[p] * C);
        return t;
    }
    // For a producer that writes kGnChunks GroupNorm partials of its output itself (avgpool; the chunked forms of in_conv and
    // upsample+concat were 1.5x slower than flat kernel + read pass: 384 blocks cannot keep enough loads in flight):
    // allocate the partials before the launch, finish() after it; the tensor then carries its statistics like a conv output.
    struct ChunkStats { GnPartials part; GnStats gs; };
    ChunkStats chunk_stats(const int* nparts = nullptr) {   // nparts: per-plane part counts of the producer (default kGnChunks)
        ChunkStats c;
        int mx = 0;
        for (int p = 0; p < 3; ++p) { c.part.nparts[p] = nparts ? nparts[p] : kGnChunks; mx = std::max(mx, c.part.nparts[p]); }
        c.part.maxparts = mx; c.part.nsub = 32;
        c.part.p = ar().alloc<double>(size_t(B) * 3 * mx * 64);
        c.gs.mr = ar().alloc<float>(size_t(B) * 3 * 64);
        return c;
    }
    int finish(const ChunkStats& c, Tri& t) {
        t.gn = c.gs.mr;
        if (ar().measuring) return 0;
        return launch_gn_finalize(c.part, t.g, t.C, B, c.gs, st);
    }
    // GroupNorm {mean, rstd} of x: taken from its producer's epilogue when available, otherwise one read pass.
    int stats_of(const Tri& x, GnStats& out) {
        if (x.gn) { out.mr = x.gn; return 0; }
        if (x.part.p) {                                   // only the producer's partials travel with it: add them now
            out.mr = ar().alloc<float>(size_t(B) * 3 * 64);
            if (ar().measuring) return 0;
            return launch_gn_finalize(x.part, x.g, x.C, B, out, st);
        }
        GnPartials part;
        part.p = ar().alloc<double>(size_t(B) * 3 * kGnChunks * 64);
        part.maxparts = kGnChunks; part.nsub = 32;
        for (int p = 0; p < 3; ++p) part.nparts[p] = kGnChunks;
        out.mr = ar().alloc<float>(size_t(B) * 3 * 64);
        if (ar().measuring) return 0;
        S3D_TRY(launch_gn_partials(x, B, part, st));
        return launch_gn_finalize(part, x.g, x.C, B, out, st);
    }

    // K slices of the rank-1 tables of the rollout convolution `cw` over the activated tensor y (s3d_rank1.h): 2 when that launch
    // goes to the mixed Winograd kernels (the only readers that add slices), from 256 own channels on; 1 otherwise
    int r1_slices_for(const Tri& y, const ConvW& cw) const {
        if (cw.k != 3 || !cw.rollout || conv_use_naive()) return 1;
        if (cw.wino24s[0] == 0 || !conv_wino24_channels(cw.cin, cw.cout)) return 1;
        return conv_rank1_slices(y.C);
    }
    // workspace of a rollout convolution's rank-1 terms for the activated tensor y: axis-sum partials, mean vectors, tables
    int roll_buffers(const Tri& y, const ConvW& cw, bool roll, MeanPartials& mp, MeanVecs& mv, const float* rrow[3], const float* rcol[3]) {
        for (int p = 0; p < 3; ++p) { rrow[p] = rcol[p] = nullptr; }
        if (!roll) return 0;
        size_t mtot = 0;
        for (int p = 0; p < 3; ++p) mtot += size_t(B) * (y.g.h[p] + y.g.w[p]) * y.C;
        float* means = ar().alloc<float>(mtot);            // (one allocation, as in round 3: the arena layout is unchanged)
        for (int p = 0; p < 3; ++p) {
            const int h = y.g.h[p], w = y.g.w[p];
            const int ntc = (w + kActCols - 1) / kActCols, ntr = (h + kActRows - 1) / kActRows;
            mp.rowpart[p] = ar().alloc<float>(size_t(B) * ntc * h * y.C);
            mp.colpart[p] = ar().alloc<float>(size_t(B) * ntr * w * y.C);
            mv.rowmean[p] = means; means += size_t(B) * h * y.C;
            mv.colmean[p] = means; means += size_t(B) * w * y.C;
            const int nsl = r1_slices_for(y, cw);           // two K slices when the consumer adds them (k_conv_wino24s)
            rrow[p] = ar().alloc<float>(size_t(nsl) * B * h * 4 * cw.cout);
            rcol[p] = ar().alloc<float>(size_t(nsl) * B * w * 4 * cw.cout);
        }
        return 0;
    }
    // th.mean over the axes + the six 1-D convolutions of the mean vectors (one launch each)
    int rank1_tables(const Tri& y, const ConvW& cw, const MeanPartials& mp, const MeanVecs& mv, const float* const rrow[3], const float* const rcol[3]) {
        S3D_TRY(launch_means_finalize(y.g, y.C, B, mp, mv, st));
        ConvArgs ca; memset(&ca, 0, sizeof ca);
        ca.B = B; ca.cin = y.C; ca.cout = cw.cout; ca.njobs = 6;
        ca.r1_slices = r1_slices_for(y, cw);
        // row-varying / column-varying vector of each plane
        const float* rowvec[3] = {mv.rowmean[1], mv.rowmean[0], mv.colmean[0]};   // xy<-mean_d xz ; xz<-mean_w xy ; yz<-mean_h xy
        const float* colvec[3] = {mv.rowmean[2], mv.colmean[2], mv.colmean[1]};   // xy<-mean_d yz ; xz<-mean_w yz ; yz<-mean_h xz
        for (int p = 0; p < 3; ++p) {
            ConvJob& jr = ca.job[2 * p];
            jr.in = rowvec[p]; jr.wgt = m->dev(cw.rrow[p]); jr.out = const_cast<float*>(rrow[p]); jr.h = 1; jr.w = y.g.h[p];
            jr.wgt_r1f = cw.rrow_f[p] ? m->dev(cw.rrow_f[p]) : nullptr;
            ConvJob& jc = ca.job[2 * p + 1];
            jc.in = colvec[p]; jc.wgt = m->dev(cw.rcol[p]); jc.out = const_cast<float*>(rcol[p]); jc.h = 1; jc.w = y.g.w[p];
            jc.wgt_r1f = cw.rcol_f[p] ? m->dev(cw.rcol_f[p]) : nullptr;
        }
        return m->timed_conv(2, CONV_1x3_ROLL, ca, st);
    }
    // GN (+FiLM) + SiLU of x into a new tensor; when `cw` is a rollout conv also the six mean vectors and the rank-1
    // tables its epilogue needs (rrow / rcol)
    int norm_act(const Tri& x, const NormW& nw, const float* film_ptr, const ConvW* cw, Tri& y, const float* rrow[3],
                 const float* rcol[3], NormTape* nt = nullptr) {
        const bool measuring = ar().measuring;
        GnStats stats{nullptr};
        // when x carries its producer's partial sums and they are few, the act kernel adds them itself
        // (with a tape, block 0 of each plane also writes them out for the backward pass)
        const bool few_parts = !x.gn && gn_act_can_add_parts(x.part, x.C);
        const bool add_parts = few_parts && gn_parts_in_consumer(gn_act_blocks(x.g, B));
        if (few_parts && !add_parts) {                      // many rounds of act blocks: the same sums once, ahead of them
            stats.mr = ar().alloc<float>(size_t(B) * 3 * 64);
            if (!measuring) S3D_TRY(launch_gn_finalize_as(x.part, x.g, x.C, B, gn_act_threads(x.C), stats, st));
        } else if (!add_parts) S3D_TRY(stats_of(x, stats));
        else if (nt) stats.mr = ar().alloc<float>(size_t(B) * 3 * 64);
        if (nt) { nt->stats = stats; nt->roll = cw && cw->rollout; }
        y = alloc_tri(x.C, x.g);
        ActArgs aa;
        for (int p = 0; p < 3; ++p) { aa.gamma[p] = m->dev(nw.gamma[p]); aa.beta[p] = m->dev(nw.beta[p]); }
        aa.film = film_ptr; aa.film_stride = fstride();
        const bool roll = cw && cw->rollout;
        MeanPartials mp; MeanVecs mv;
        if (roll) S3D_TRY(roll_buffers(y, *cw, true, mp, mv, rrow, rcol));
        else for (int p = 0; p < 3; ++p) { rrow[p] = rcol[p] = nullptr; }
        if (nt && roll) nt->mv = mv;
        if (measuring) return 0;
        S3D_TRY(launch_gn_act(x, B, stats, aa, y, roll ? &mp : nullptr, st, add_parts ? &x.part : nullptr));
        if (!roll) return 0;
        return rank1_tables(y, *cw, mp, mv, rrow, rcol);
    }

    // want_stats (3x3 MFMA paths only): 1 = reduce the GroupNorm statistics of the output (partials in the epilogue +
    // finalize), 2 = leave only the partials with the tensor (it is normalised later as the skip half of a concat)
    int conv(const Tri& y, const ConvW& cw, const float* bbias, const float* const rrow[3], const float* const rcol[3],
             const Tri* res, Tri& out, int want_stats = 0, hipStream_t on = nullptr, bool no_bias = false, bool res_up = false) {
        hipStream_t st = on ? on : this->st;
        out = alloc_tri(cw.cout, y.g);
        if (!(cw.k == 3 && !conv_use_naive())) want_stats = 0;
        // the mixed Winograd kernels serve every forward (the tape keeps activations, not conv internals) and, on the transposed
        // image, the backward's dgrad (s3d_train.hip:conv_bwd)
        const bool w24 = cw.k == 3 && cw.wino24s[0] != 0 && conv_wino24_channels(cw.cin, cw.cout);
        S3D_CHECK(!(cw.only24_current && (!w24 || conv_use_naive())), S3D_ERR_INVALID,
                  "conv %dx%d %d->%d: the kernel form selected now reads a weight image the training handle's repack plan does not keep "
                  "current (options changed after s3d_unet_train_attach): attach again", cw.k, cw.k, cw.cin, cw.cout);
        GnPartials part; GnStats gs{nullptr};
        if (want_stats) {
            conv_gn_parts(CONV_3x3, y.g, part.nparts, w24);
            part.maxparts = std::max(part.nparts[0], std::max(part.nparts[1], part.nparts[2]));
            part.nsub = cw.cout / gn_subgroup(cw.cout);
            part.p = ar().alloc<double>(size_t(B) * 3 * part.maxparts * part.nsub * 2);
            if (want_stats == 1) { gs.mr = ar().alloc<float>(size_t(B) * 3 * 64); out.gn = gs.mr; }
            else out.part = part;
        }
        if (ar().measuring) return 0;
        ConvArgs ca; memset(&ca, 0, sizeof ca);
        ca.B = B; ca.cin = cw.cin; ca.cout = cw.cout; ca.njobs = 3;
        if (want_stats) { ca.gn_sg = gn_subgroup(cw.cout); ca.gn_nsub = part.nsub; ca.gn_maxparts = part.maxparts; }
        ca.r1_slices = rrow && rcol && rrow[0] ? r1_slices_for(y, cw) : 1;
        for (int p = 0; p < 3; ++p) {
            ConvJob& J = ca.job[p];
            J.in = y.p[p]; J.wgt = m->dev(cw.dense[p]); J.bias = no_bias ? nullptr : m->dev(cw.bias[p]);
            J.wgt_wino = cw.k == 3 && !w24 ? m->dev(cw.wino[p]) : nullptr;     // (the repack plan keeps only the image in use current)
            J.wgt_wino24s = w24 ? m->dev(cw.wino24s[p]) : nullptr;
            J.bbias = bbias; J.bbias_stride = fstride();
            J.rrow = rrow ? rrow[p] : nullptr; J.rcol = rcol ? rcol[p] : nullptr;
            J.res = res ? res->p[p] : nullptr; J.res_up = res && res_up ? 1 : 0; J.out = out.p[p]; J.h = y.g.h[p]; J.w = y.g.w[p];
            J.gn_part = want_stats ? part.p + size_t(p) * part.maxparts * part.nsub * 2 : nullptr;
        }
        S3D_TRY(m->timed_conv(cw.k == 3 ? 0 : 1, cw.k == 3 ? CONV_3x3 : CONV_1x1, ca, st));
        if (want_stats == 1) S3D_TRY(launch_gn_finalize(part, y.g, cw.cout, B, gs, st));
        return 0;
    }

    // the same block on the VIRTUAL input [bilinear2x(u) | sk] (:494-503 + :269-311), inference only: the concat is never
    // written.  Statistics: the upsampled half by one read pass over u, the skip half from its producer's partials; the
    // first norm samples u on the fly; the 1x1 skip_connection runs per half (W_a at low resolution, then upsampled).
    int resblock_cat(const ResBlockW& rb, const Tri& u, const Tri& sk, Tri& out, int out_stats) {
        const bool ssn = m->cfg.use_scale_shift_norm != 0;
        const float* film_ptr = film ? film + rb.film_off : nullptr;
        const bool measuring = ar().measuring;
        const int C = u.C + sk.C, sg = gn_subgroup(C);
        // statistics of the virtual tensor
        GnPartials pu; GnStats stats;
        pu.nsub = u.C / sg;
        gn_up_parts(sk.g, pu.nparts);
        pu.maxparts = std::max(pu.nparts[0], std::max(pu.nparts[1], pu.nparts[2]));
        pu.p = ar().alloc<double>(size_t(B) * 3 * pu.nsub * pu.maxparts * 2);
        stats.mr = ar().alloc<float>(size_t(B) * 3 * 64);
        // skip path: z = W_a u (low resolution), up(z), then W_b sk + bias + up(z)
        Tri z, skip;
        S3D_TRY(conv(u, rb.skip_a, nullptr, nullptr, nullptr, nullptr, z, 0, nullptr, true));
        S3D_TRY(conv(sk, rb.skip_b, nullptr, nullptr, nullptr, &z, skip, 0, nullptr, false, true));     // + up(z) in the epilogue
        if (!measuring) {
            S3D_TRY(launch_gn_partials_up(u, B, sg, pu, st));
            S3D_TRY(launch_gn_finalize_cat(pu, sk.part, sk.g, C, B, stats, st));
        }
        // first norm + rollout tables on the virtual tensor
        Tri y1, h1, y2;
        y1 = alloc_tri(C, sk.g);
        const float *rr[3], *rc[3];
        {
            ActArgs aa;
            for (int p = 0; p < 3; ++p) { aa.gamma[p] = m->dev(rb.n1.gamma[p]); aa.beta[p] = m->dev(rb.n1.beta[p]); }
            aa.film = nullptr; aa.film_stride = fstride();
            const bool roll = rb.c1.rollout;
            MeanPartials mp; MeanVecs mv;
            S3D_TRY(roll_buffers(y1, rb.c1, roll, mp, mv, rr, rc));
            if (!measuring) {
                S3D_TRY(launch_gn_act_cat(u, sk, B, stats, aa, y1, roll ? &mp : nullptr, st));
                if (roll) S3D_TRY(rank1_tables(y1, rb.c1, mp, mv, rr, rc));
            }
        }
        S3D_TRY(conv(y1, rb.c1, ssn ? nullptr : film_ptr, rr, rc, nullptr, h1, 2));   // partials only (norm_act adds them)
        S3D_TRY(norm_act(h1, rb.n2, ssn ? film_ptr : nullptr, &rb.c2, y2, rr, rc, nullptr));
        S3D_TRY(conv(y2, rb.c2, nullptr, rr, rc, &skip, out, out_stats));
        return 0;
    }

    // TriplaneResBlock._forward (src/diffusion/unet_triplane.py:269-311)
    int resblock(const ResBlockW& rb, const Tri& x, Tri& out, bool out_feeds_norm, int out_stats_override = -1) {
        const bool ssn = m->cfg.use_scale_shift_norm != 0;
        const float* film_ptr = film ? film + rb.film_off : nullptr;
        Tri y1, h1, y2;
        const float *rr[3], *rc[3];
        // skip_connection(x) only depends on x (a second stream for it was measured in rounds 1, 2 and — on a low-priority hardware
        // queue of its own — 5: the event edges cost more than the overlap returns, profiles/r05_fwd_side.txt — the switch is gone)
        Tri skip;
        const Tri* res = &x;
        if (rb.has_skip) {
            S3D_TRY(conv(x, rb.skip, nullptr, nullptr, nullptr, nullptr, skip, false));
            res = &skip;
        }
        RBTape rt;
        RBTape* T = tape ? &rt : nullptr;
        S3D_TRY(norm_act(x, rb.n1, nullptr, &rb.c1, y1, rr, rc, T ? &T->n1 : nullptr));
        S3D_TRY(conv(y1, rb.c1, ssn ? nullptr : film_ptr, rr, rc, nullptr, h1, 2));   // partials only (norm_act adds them)      // (!ssn: h = h + emb_out, :298-303)
        S3D_TRY(norm_act(h1, rb.n2, ssn ? film_ptr : nullptr, &rb.c2, y2, rr, rc, T ? &T->n2 : nullptr));
        S3D_TRY(conv(y2, rb.c2, nullptr, rr, rc, res, out, out_stats_override >= 0 ? out_stats_override : (out_feeds_norm ? 1 : 0)));
        if (T) {
            rt.rb = &rb; rt.x = x; rt.y1 = y1; rt.h1 = h1; rt.y2 = y2;
            last_rb = rt;
        }
        return 0;
    }
};

}  // namespace s3d
