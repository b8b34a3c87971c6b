// s3d_train.hip — training tier of the denoiser: forward with an activation tape, the backward
// pass of TriplaneUNetModelSmall, and the C-ABI entry points around them (flat parameters, device-side repack,
// q_sample / per-plane MSE / AdamW+EMA).  Mirrors what the reference gets from autograd for
// GaussianDiffusion.training_losses (src/diffusion/gaussian_diffusion.py:771-856) + TrainLoop.run_step
// (src/diffusion/train_util.py:163-247).
#include "s3d_bwd.h"
#include "s3d_model.h"

namespace s3d {

int build_pack_plan(s3d_unet* m);
int launch_repack(s3d_unet* m, hipStream_t st);

static const char* kPl[3] = {"xy", "xz", "yz"};

struct Bwd {
    s3d_unet* m;
    int B;
    hipStream_t st;
    float* grads;                 // flat, same layout as the parameters
    float* dfilm = nullptr;       // [B][film_total]
    DeferredTail tail;            // the pass's split-K reductions and bias gradients, run in batches (before each progress mark)
    // Weight gradients (k_wgrad_wino / k_wgrad_mfma / k_slot_wgrad, their split-K reductions, the bias sums, in_conv's outer
    // products) have no consumer before the optimizer: they go to a handle-owned low-priority side stream `sw` (BWD_SIDE=0: the
    // caller's stream) and run beside the chain that carries the input gradients (dgrad -> edge sums -> mean-vector gradients ->
    // GroupNorm backward -> next dgrad), whose many one-round launches leave the matrix cores idle.  edge(a, b): stream b waits for
    // what has been enqueued on a so far (one event; the main chain only ever pays the record).  Same kernels on the same
    // operands: bit-identical gradients.  Measured (profiles/r05_train_step.txt): 3.21 -> 2.98 ms/step; a SECOND side stream for
    // each layer's edge sums -> mean-vector gradients beside its dgrad convolution (joined before the GroupNorm backward) lost
    // 1 % to the eight joins it puts on the main chain and is not kept.
    hipStream_t sw = nullptr;
    size_t ev_next = 0;
    hipEvent_t next_event() {
        if (ev_next == m->bwd_events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            m->bwd_events.push_back(e);
        }
        return m->bwd_events[ev_next++];
    }
    int edge(hipStream_t from, hipStream_t to) {
        if (from == to) return 0;
        hipEvent_t e = next_event();
        S3D_CHECK(e, S3D_ERR_HIP, "backward: hipEventCreate failed");
        S3D_HIP(hipEventRecord(e, from));
        S3D_HIP(hipStreamWaitEvent(to, e, 0));
        return 0;
    }
    int join() { return edge(sw, st); }                   // the caller's stream waits for the side stream
    Arena& ar() { return m->arena; }
    bool meas() { return m->arena.measuring; }

    float* G(const std::string& name) {
        auto it = m->flat_index.find(name);
        if (it == m->flat_index.end()) return nullptr;
        return grads ? grads + it->second : reinterpret_cast<float*>(uintptr_t(256));
    }
    Tri alloc_tri(int C, const Geo& g) {
        Tri t; t.C = C; t.g = g;
        for (int p = 0; p < 3; ++p) t.p[p] = ar().alloc<float>(size_t(B) * g.h[p] * g.w[p] * C);
        return t;
    }

    // Backward of one TriplaneConv (src/diffusion/unet_triplane.py:31-60).  dy: gradient of its output; a: its (own-
    // channel) input; nt: the rollout means of that input (null for the 1x1 / non-rollout convs).
    //  d_a      (optional) dense input gradient of the own channels (+ res), via the forward conv kernels
    //  rowadd / coladd (rollout) the gradients of the six mean vectors, routed to the plane they were taken from
    //           and divided by the averaged length: to be broadcast-added to that plane's d_a by the caller
    // gnf (optional): the GroupNorm(+FiLM)+SiLU whose backward follows this convolution's: when the dgrad launch is the mixed
    // Winograd kernel its epilogue leaves that backward's two per-channel sums in *gnf_part (k_conv_wino24s_gnb; option GNB_FUSED) —
    // the mean-vector gradients are then built BEFORE the dgrad launch (its epilogue adds their broadcast to dy, as gn_bwd does)
    struct GnFuse { const Tri* x; const GnStats* stats; const NormW* nw; const float* film; };
    int conv_bwd(const ConvW& cw, const ConvWT& wt, const std::string& prefix, const Tri& dy, const Tri& a, const NormTape* nt,
                 Tri* d_a, const Tri* res, float* rowadd[3], float* coladd[3], float* per_sample_bias,
                 const GnFuse* gnf = nullptr, GnPartials* gnf_part = nullptr) {
        const Geo& g = dy.g;
        const int cin = cw.cin, cout = cw.cout, taps = cw.k * cw.k;
        const bool roll = cw.rollout && nt && nt->roll;
        if (gnf_part) gnf_part->p = nullptr;
        const bool fuse = gnf && gnf_part && d_a && cw.k == 3 && wt.has_wino24s_T && conv_use_wino24() && !conv_use_naive() &&
                          cin % 32 == 0 && opt_on(OPT_GNB_FUSED) && gnf->x->C == cin;
        // (1) dgrad of the own channels — enqueued after (2) when its epilogue needs the mean-vector gradients
        auto dgrad = [&]() -> int {
            if (!d_a) return 0;
            *d_a = alloc_tri(cin, g);
            S3D_CHECK(!(wt.only24_current && (!conv_use_wino24() || conv_use_naive())), S3D_ERR_INVALID,
                      "backward of conv %d->%d: the kernel form selected now reads a transposed weight image the repack plan does not "
                      "keep current (options changed after s3d_unet_train_attach): attach again", cin, cout);
            GnPartials part{nullptr, 0, {0, 0, 0}, 0};
            if (fuse) {
                conv_gn_parts(CONV_3x3, g, part.nparts, true);
                part.maxparts = std::max(part.nparts[0], std::max(part.nparts[1], part.nparts[2]));
                part.nsub = cin;                                   // one record per channel and tile
                part.p = ar().alloc<double>(size_t(B) * 3 * part.maxparts * part.nsub * 2);
                *gnf_part = part;
            }
            if (meas()) return 0;
            ConvArgs ca; memset(&ca, 0, sizeof ca);
            ca.B = B; ca.cin = cout; ca.cout = cin; ca.njobs = 3;
            GnbArgs gb; memset(&gb, 0, sizeof gb);
            if (fuse) {
                ca.gn_sg = 1; ca.gn_nsub = part.nsub; ca.gn_maxparts = part.maxparts; ca.gnb = &gb;
                gb.mr = gnf->stats->mr; gb.film = gnf->film; gb.film_stride = m->film_total; gb.groups = 32;
            }
            for (int p = 0; p < 3; ++p) {
                ConvJob& J = ca.job[p];
                J.in = dy.p[p]; J.wgt = m->tdev(wt.dense_T[p]); J.wgt_wino = cw.k == 3 && wt.has_wino_T ? m->tdev(wt.wino_T[p]) : nullptr;
                J.wgt_wino24s = cw.k == 3 && wt.has_wino24s_T ? m->tdev(wt.wino24s_T[p]) : nullptr;
                J.res = res ? res->p[p] : nullptr; J.out = d_a->p[p]; J.h = g.h[p]; J.w = g.w[p];
                if (fuse) {
                    J.gn_part = part.p + size_t(p) * part.maxparts * part.nsub * 2;
                    gb.x[p] = gnf->x->p[p]; gb.gamma[p] = m->dev(gnf->nw->gamma[p]); gb.beta[p] = m->dev(gnf->nw->beta[p]);
                    gb.rowadd[p] = rowadd ? rowadd[p] : nullptr; gb.coladd[p] = coladd ? coladd[p] : nullptr;
                    gb.rowscale[p] = 1.0f / float(g.w[p]); gb.colscale[p] = 1.0f / float(g.h[p]);      // (as launch_gn_act_bwd)
                }
            }
            return m->timed_conv(cw.k == 3 ? 0 : 1, cw.k == 3 ? CONV_3x3 : CONV_1x1, ca, st);      // (events only inside a profiled step)
        };
        if (!fuse) S3D_TRY(dgrad());
        // (2) row / column sums of dy -> bias gradient, mean-slot gradients
        float* R[3]; float* Cs[3];
        for (int p = 0; p < 3; ++p) {
            R[p] = ar().alloc<float>(size_t(B) * g.h[p] * 3 * cout);
            Cs[p] = ar().alloc<float>(size_t(B) * g.w[p] * 3 * cout);
        }
        float* dbias[3]; float* dW[3];
        for (int p = 0; p < 3; ++p) {
            dbias[p] = G(prefix + ".conv_" + kPl[p] + ".bias");
            dW[p] = G(prefix + ".conv_" + kPl[p] + ".weight");
            S3D_CHECK(dbias[p] && dW[p], S3D_ERR_INVALID, "backward: unknown parameter %s", prefix.c_str());
        }
        if (!meas()) {
            // the side stream waits for the edge sums (and dy before them): the event rides on the launch's own completion signal
            // — a hipEventRecord behind it is a barrier packet of its own in front of the chain's next kernel (2.867 -> 2.850 ms/step)
            // (under rocprofv3 an event on a dispatch's completion signal makes the waiting queue crawl — a traced step took 3.8 ms
            // instead of 2.9 —: a plain record there, so that traces keep their shape.  Option EDGE_SIGNAL: 1 / 0 force either form,
            // unset = plain under a profiler's environment, on-signal otherwise; profiles/r06_train_* name the form they traced.)
            static const bool profiled = getenv("ROCP_TOOL_LIBRARIES") || getenv("HSA_TOOLS_LIB") || getenv("ROCPROFILER_LIBRARY_CTOR");
            const int edge_opt = opt(OPT_EDGE_SIGNAL);
            const bool traced = edge_opt == kOptUnset ? profiled : edge_opt == 0;
            if (traced && st != sw) {
                S3D_TRY(launch_edge_sums(dy, B, R, Cs, st));
                S3D_TRY(edge(st, sw));
                S3D_TRY(launch_bias_grad_deferred(R, g, cout, B, dbias, per_sample_bias, m->film_total, tail));
            } else {
            hipEvent_t e = st != sw ? next_event() : nullptr;
            S3D_CHECK(st == sw || e, S3D_ERR_HIP, "backward: hipEventCreate failed");
            S3D_TRY(launch_edge_sums(dy, B, R, Cs, st, e));
            if (e) S3D_HIP(hipStreamWaitEvent(sw, e, 0));
            S3D_TRY(launch_bias_grad_deferred(R, g, cout, B, dbias, per_sample_bias, m->film_total, tail));
            }
        }
        if (roll) {
            const MeanVecs& mv = nt->mv;
            const float* rowvec[3] = {mv.rowmean[1], mv.rowmean[0], mv.colmean[0]};
            const float* colvec[3] = {mv.rowmean[2], mv.colmean[2], mv.colmean[1]};
            float* d_rowvec[3]; float* d_colvec[3];
            for (int p = 0; p < 3; ++p) {
                d_rowvec[p] = ar().alloc<float>(size_t(B) * g.h[p] * cin);
                d_colvec[p] = ar().alloc<float>(size_t(B) * g.w[p] * cin);
            }
            if (!meas()) {
                SlotWgradArgs sw;
                for (int p = 0; p < 3; ++p) {
                    sw.rowvec[p] = rowvec[p]; sw.colvec[p] = colvec[p]; sw.R[p] = R[p]; sw.Cs[p] = Cs[p]; sw.dW[p] = dW[p];
                }
                sw.g = g; sw.B = B; sw.C = cin; sw.cout = cout;
                S3D_TRY(launch_slot_wgrad(sw, this->sw));
                // gradients of the six mean vectors: 1-D transposed convolutions of the edge sums (k_rank1)
                ConvArgs ca; memset(&ca, 0, sizeof ca);
                ca.B = B; ca.cin = 3 * cout; ca.cout = cin; ca.njobs = 6;
                for (int p = 0; p < 3; ++p) {
                    ConvJob& jr = ca.job[2 * p];
                    jr.in = R[p]; jr.wgt = m->tdev(wt.rrow_T[p]); jr.out = d_rowvec[p]; jr.h = 1; jr.w = g.h[p];
                    ConvJob& jc = ca.job[2 * p + 1];
                    jc.in = Cs[p]; jc.wgt = m->tdev(wt.rcol_T[p]); jc.out = d_colvec[p]; jc.h = 1; jc.w = g.w[p];
                }
                S3D_TRY(launch_conv(CONV_1x3_VEC, ca, st));
            }
            // each vector is the gradient of one axis mean of another plane; the 1/length factor is applied where it is
            // broadcast-added (launch_gn_act_bwd: rowscale = 1/w, colscale = 1/h of the receiving plane)
            rowadd[1] = d_rowvec[0]; rowadd[0] = d_rowvec[1]; coladd[0] = d_rowvec[2];
            rowadd[2] = d_colvec[0]; coladd[2] = d_colvec[1]; coladd[1] = d_colvec[2];
        } else if (rowadd) {
            for (int p = 0; p < 3; ++p) rowadd[p] = coladd[p] = nullptr;
        }
        if (fuse) S3D_TRY(dgrad());
        // (3) dense weight gradient of the own channels
        WgradArgs w;
        w.dy = dy; w.a = a; w.B = B; w.cin = cin; w.cout = cout; w.ctot = cw.rollout ? 3 * cin : cin; w.taps = taps;
        w.ksplit = wgrad_ksplit(g, B, cin, cout, taps);
        for (int p = 0; p < 3; ++p) {
            w.part[p] = ar().alloc<float>(wgrad_part_floats(w.ksplit, cin, cout, taps));
            w.dW[p] = dW[p];
        }
        if (!meas()) {
            if (taps == 9) {
                // s3d_profile class 3: direct count 2*9*cin*cout per pixel; the Winograd F(2x2,3x3) form multiplies 16 instead of 36
                // per 2x2 tile and channel pair (4/9), the direct form all of them
                double pix = 0;
                for (int p = 0; p < 3; ++p) pix += double(g.h[p]) * g.w[p];
                const double fl = 2.0 * 9 * cin * cout * pix * B;
                const bool wino = wgrad_uses_wino();
                hipStream_t on = this->sw;
                S3D_TRY(m->timed_launch(3, fl, wino ? fl * 4.0 / 9.0 : fl, on, [&] {
                    conv_note_kernel(wino ? "k_wgrad_wino Winograd F(2x2,3x3) weight gradient" : "k_wgrad_mfma<9> direct weight gradient");
                    return launch_wgrad(w, on, &tail);
                }));
            } else S3D_TRY(launch_wgrad(w, this->sw, &tail));
        }
        return 0;
    }

    int gn_bwd(const Tri& x, const GnStats& stats, const NormW& nw, const std::string& prefix, const float* film_ptr,
               float* dfilm_ptr, const Tri& dy, float* const rowadd[3], float* const coladd[3], const Tri* add, Tri& dx,
               const GnPartials* conv_part = nullptr) {
        dx = alloc_tri(x.C, x.g);
        GnActBwd s;
        s.conv_part = conv_part && conv_part->p ? conv_part : nullptr;
        s.x = x; s.dy = dy; s.dx = dx; s.add = add; s.stats = stats; s.B = B;
        const float* ra[3]; const float* ca[3];
        for (int p = 0; p < 3; ++p) { ra[p] = rowadd ? rowadd[p] : nullptr; ca[p] = coladd ? coladd[p] : nullptr; }
        s.rowadd = ra; s.coladd = ca;
        for (int p = 0; p < 3; ++p) {
            s.gamma[p] = m->dev(nw.gamma[p]); s.beta[p] = m->dev(nw.beta[p]);
            s.dgamma[p] = G(prefix + ".norm_" + kPl[p] + ".weight"); s.dbeta[p] = G(prefix + ".norm_" + kPl[p] + ".bias");
            S3D_CHECK(s.dgamma[p] && s.dbeta[p], S3D_ERR_INVALID, "backward: unknown parameter %s", prefix.c_str());
        }
        s.film = film_ptr; s.dfilm = dfilm_ptr; s.film_stride = m->film_total;
        s.ws = ar().alloc<float>(gn_bwd_ws_floats(B, x.C));
        if (meas()) return 0;
        return launch_gn_act_bwd(s, st);
    }

    // TriplaneResBlock backward (forward: src/diffusion/unet_triplane.py:269-311)
    int rb_bwd(const RBTape& T, const ResBlockW& rb, const ResBlockWT& wt, const Tri& d_out, Tri& d_x) {
        const bool ssn = m->cfg.use_scale_shift_norm != 0;
        const float* film_ptr = m->tape.film + rb.film_off;
        float* dfilm_ptr = dfilm + rb.film_off;
        Tri d_xs;
        const Tri* add = &d_out;
        if (rb.has_skip) {
            S3D_TRY(conv_bwd(rb.skip, wt.skip, rb.prefix + ".skip_connection", d_out, T.x, nullptr, &d_xs, nullptr, nullptr, nullptr, nullptr));
            add = &d_xs;
        }
        float *ra[3], *ca[3];
        Tri d_y2, d_h1, d_y1;
        GnPartials gp2, gp1;
        const GnFuse f2{&T.h1, &T.n2.stats, &rb.n2, ssn ? film_ptr : nullptr}, f1{&T.x, &T.n1.stats, &rb.n1, nullptr};
        S3D_TRY(conv_bwd(rb.c2, wt.c2, rb.prefix + ".out_layers.2", d_out, T.y2, &T.n2, &d_y2, nullptr, ra, ca, nullptr, &f2, &gp2));
        S3D_TRY(gn_bwd(T.h1, T.n2.stats, rb.n2, rb.prefix + ".out_layers.0", ssn ? film_ptr : nullptr, ssn ? dfilm_ptr : nullptr,
                       d_y2, ra, ca, nullptr, d_h1, &gp2));
        S3D_TRY(conv_bwd(rb.c1, wt.c1, rb.prefix + ".in_layers.2", d_h1, T.y1, &T.n1, &d_y1, nullptr, ra, ca, ssn ? nullptr : dfilm_ptr, &f1, &gp1));
        S3D_TRY(gn_bwd(T.x, T.n1.stats, rb.n1, rb.prefix + ".in_layers.0", nullptr, nullptr, d_y1, ra, ca, add, d_x, &gp1));
        return 0;
    }
};

// Backward of run_forward (s3d_unet.hip).  d_out: gradient of the composed output [B,Cout,H+D,W+D].
// marks (optional): HIP events recorded on `st` when a group of parameter gradients is final — [0] after the output blocks
// (out.* and output_blocks.* except the emb_layers), [1] after the input blocks and in_conv (input_blocks.* except the
// emb_layers, in_conv.*); everything else (time_embed.*, every emb_layers.*) is final when the call's work is.  A data-parallel
// trainer all-reduces the finished groups on a second stream while the rest of the backward pass runs.
static int run_backward(s3d_unet* m, const float* d_out, float* grads, hipStream_t st, hipEvent_t* marks = nullptr, int n_marks = 0) {
    const s3d_unet_cfg& c = m->cfg;
    const Tape& T = m->tape;
    const int B = T.B, mc = c.model_channels, ted = 4 * mc, nl = c.n_levels;
    Bwd b{m, B, st, grads};
    Arena& ar = m->arena;
    const bool meas = ar.measuring;
    b.sw = st;
    if (!meas && opt_on(OPT_BWD_SIDE)) {
        if (!m->bwd_side) {
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            S3D_HIP(hipStreamCreateWithPriority(&m->bwd_side, hipStreamNonBlocking, least));
        }
        b.sw = m->bwd_side;
    }
    b.dfilm = ar.alloc<float>(size_t(B) * m->film_total);

    // ---- out head: GN -> SiLU -> 1x1 conv -> compose
    Tri act = b.alloc_tri(T.head_in.C, T.head_in.g), d_act = b.alloc_tri(T.head_in.C, T.head_in.g), d_h;
    {
        float* ws = ar.alloc<float>(small_outer_ws_floats(c.out_channels, T.head_in.C));
        if (!meas) {
            ActArgs aa;
            for (int p = 0; p < 3; ++p) { aa.gamma[p] = m->dev(m->out_norm.gamma[p]); aa.beta[p] = m->dev(m->out_norm.beta[p]); }
            aa.film = nullptr; aa.film_stride = 0;
            S3D_TRY(launch_gn_act(T.head_in, B, T.head_stats, aa, act, nullptr, st));
            SmallOuter so;
            so.s = d_out; so.S = c.out_channels; so.v = act; so.Wt = m->dev(m->out_w); so.dv = &d_act; so.outer_transposed = 0;
            float* ssum[3];
            for (int p = 0; p < 3; ++p) {
                so.outer[p] = b.G(std::string("out.2.conv_") + kPl[p] + ".weight");
                ssum[p] = b.G(std::string("out.2.conv_") + kPl[p] + ".bias");
            }
            so.ssum = ssum; so.vsum = nullptr; so.ws = ws;
            so.H = T.H; so.W = T.W; so.D = T.D; so.C = T.head_in.C; so.B = B;
            S3D_TRY(launch_small_outer(so, st));
        }
        S3D_TRY(b.gn_bwd(T.head_in, T.head_stats, m->out_norm, "out.0", nullptr, nullptr, d_act, nullptr, nullptr, nullptr, d_h));
    }

    // ---- output blocks, last to first
    std::vector<Tri> dcat(nl);                     // gradient of each output block's (concatenated) input
    for (int oi = nl - 1; oi >= 0; --oi) {
        Tri d_in;
        S3D_TRY(b.rb_bwd(T.out_rb[oi], m->out_blocks[oi], m->out_blocks_t[oi], d_h, d_in));
        dcat[oi] = d_in;
        if (oi == 0) { d_h = d_in; break; }
        // input was [upsample(prev output) | skip]: route the first Cu channels back through the resampling
        const Tri& u = T.up_src[oi];
        const Geo up = u.g.twice();
        Tri d_prev = b.alloc_tri(u.C, u.g);
        if (up == d_in.g) {                                  // the common case: one launch for the three planes
            if (!meas) {
                const float* dsrc[3] = {d_in.p[0], d_in.p[1], d_in.p[2]};
                S3D_TRY(launch_bilinear_bwd3(dsrc, B, u.C, up.h, up.w, d_in.C, 0, d_prev.p, u.g.h, u.g.w, st));
            }
        } else
        for (int p = 0; p < 3; ++p) {
            const bool same = up.h[p] == d_in.g.h[p] && up.w[p] == d_in.g.w[p];
            if (same) {
                if (!meas) S3D_TRY(launch_bilinear_bwd(d_in.p[p], B, u.C, up.h[p], up.w[p], d_in.C, 0, d_prev.p[p], u.g.h[p], u.g.w[p], st));
            } else {
                float* tmp = ar.alloc<float>(size_t(B) * up.h[p] * up.w[p] * u.C);
                if (!meas) {
                    S3D_TRY(launch_bilinear_bwd(d_in.p[p], B, u.C, d_in.g.h[p], d_in.g.w[p], d_in.C, 0, tmp, up.h[p], up.w[p], st));
                    S3D_TRY(launch_bilinear_bwd(tmp, B, u.C, up.h[p], up.w[p], u.C, 0, d_prev.p[p], u.g.h[p], u.g.w[p], st));
                }
            }
        }
        d_h = d_prev;
    }

    if (!meas) { S3D_TRY(b.edge(st, b.sw)); S3D_TRY(b.tail.flush(b.sw)); }      // (the output blocks' weight / bias gradients are final at mark 0; the bias sums read the main chain's edge sums)
    // (the mark is recorded on the stream that finishes the group LAST — the side stream, which has just been made to wait for the
    // caller's stream: the caller's chain is not stalled for the side stream's backlog of weight gradients)
    if (!meas && n_marks > 0 && marks[0]) S3D_HIP(hipEventRecord(marks[0], b.sw));

    // ---- input blocks, deepest to first
    Tri d_x;                                        // gradient of the current level's resblock input
    for (int level = nl - 1; level >= 0; --level) {
        Tri d_o;
        if (level == nl - 1) d_o = d_h;
        else {
            // this level's output fed (a) the next level through avg_pool2d and (b) the concat of output block nl-1-level
            const RBTape& rt = T.in_rb[level];
            d_o = b.alloc_tri(rt.rb->Cout, rt.x.g);
            const int oi = nl - 1 - level;
            const Tri& dc = dcat[oi];
            if (!meas) S3D_TRY(launch_pool_bwd_add(&d_x, &dc, dc.C - rt.rb->Cout, B, d_o, st));
        }
        Tri d_in;
        S3D_TRY(b.rb_bwd(T.in_rb[level], m->in_blocks[level], m->in_blocks_t[level], d_o, d_in));
        d_x = d_in;
    }

    // ---- in_conv (1x1 over the composed input): weight and bias gradients only
    {
        float* ws = ar.alloc<float>(small_outer_ws_floats(c.in_channels, d_x.C));
        if (!meas) {
            SmallOuter so;
            so.s = T.x; so.S = c.in_channels; so.v = d_x; so.Wt = nullptr; so.dv = nullptr; so.outer_transposed = 1;
            float* vsum[3];
            for (int p = 0; p < 3; ++p) {
                so.outer[p] = b.G(std::string("in_conv.0.conv_") + kPl[p] + ".weight");
                vsum[p] = b.G(std::string("in_conv.0.conv_") + kPl[p] + ".bias");
            }
            so.ssum = nullptr; so.vsum = vsum; so.ws = ws;
            so.H = T.H; so.W = T.W; so.D = T.D; so.C = d_x.C; so.B = B;
            S3D_TRY(b.edge(st, b.sw));                    // d_x is final; nothing inside the pass reads in_conv's gradients
            S3D_TRY(launch_small_outer(so, b.sw));
        }
    }

    if (!meas) { S3D_TRY(b.edge(st, b.sw)); S3D_TRY(b.tail.flush(b.sw)); }      // (... the input blocks' at mark 1)
    if (!meas && c.use_scale_shift_norm == 0) S3D_TRY(b.join());      // h + emb_out: the per-sample bias sums (side stream) are the FiLM gradient the timestep MLP below reads
    if (!meas && n_marks > 1 && marks[1]) S3D_HIP(hipEventRecord(marks[1], b.sw));

    // ---- timestep MLP: film = Lf(silu(emb)), emb = L2(silu(pre1)), pre1 = L0(temb(t))
    {
        float* d_emb = ar.alloc<float>(size_t(B) * ted);
        float* d_pre1 = ar.alloc<float>(size_t(B) * ted);
        if (!meas) {
            const bool ssn = c.use_scale_shift_norm != 0;
            auto film_w = [&](const ResBlockW& rb) -> int {
                const int eo = ssn ? 2 * rb.Cout : rb.Cout;
                return launch_linear_bwd(b.dfilm + rb.film_off, m->film_total, T.emb, B, ted, nullptr, eo, 1,
                                         b.G(rb.prefix + ".emb_layers.1.weight"), b.G(rb.prefix + ".emb_layers.1.bias"), nullptr, st);
            };
            // one launch for all of them when the blocks' FiLM rows tile [0, film_total) in order (they do: build_specs)
            std::vector<const ResBlockW*> rbs;
            for (auto& rb : m->in_blocks) rbs.push_back(&rb);
            for (auto& rb : m->out_blocks) rbs.push_back(&rb);
            std::vector<int> seg; std::vector<float*> gw, gb;
            bool tiled = rbs.size() <= 16;
            int at = 0;
            for (auto* rb : rbs) {
                tiled = tiled && rb->film_off == at;
                seg.push_back(at); at += ssn ? 2 * rb->Cout : rb->Cout;
                gw.push_back(b.G(rb->prefix + ".emb_layers.1.weight")); gb.push_back(b.G(rb->prefix + ".emb_layers.1.bias"));
            }
            seg.push_back(at);
            if (tiled && at == m->film_total)
                S3D_TRY(launch_linear_bwd_w_multi(b.dfilm, m->film_total, T.emb, B, ted, 1, int(rbs.size()), seg.data(), gw.data(), gb.data(), st));
            else
                for (auto* rb : rbs) S3D_TRY(film_w(*rb));
            S3D_TRY(launch_linear_bwd(b.dfilm, m->film_total, T.emb, B, ted, m->dev(m->film_w), m->film_total, 1, nullptr, nullptr, d_emb, st));
            S3D_TRY(launch_linear_bwd(d_emb, ted, T.pre1, B, ted, m->dev(m->te2_w), ted, 1, b.G("time_embed.2.weight"),
                                      b.G("time_embed.2.bias"), d_pre1, st));
            S3D_TRY(launch_linear_bwd(d_pre1, ted, T.t, B, mc, nullptr, ted, 2, b.G("time_embed.0.weight"), b.G("time_embed.0.bias"),
                                      nullptr, st));
        }
    }
    if (!meas) S3D_TRY(b.join());                     // every gradient of the pass is final in stream order of `st`
    return 0;
}

}  // namespace s3d

using namespace s3d;

extern "C" {

int64_t s3d_unet_param_numel(const s3d_unet* m) {
    if (!m) return S3D_ERR_INVALID;
    int64_t n = 0;
    for (const auto& sp : m->specs) n += int64_t(sp.numel());
    return n;
}

int s3d_unet_param_offset(const s3d_unet* m, int i, int64_t* offset) {
    S3D_CHECK(m && offset && i >= 0 && i < int(m->specs.size()), S3D_ERR_INVALID, "param_offset: index %d out of range", i);
    int64_t n = 0;
    for (int k = 0; k < i; ++k) n += int64_t(m->specs[k].numel());
    *offset = n;
    return 0;
}

int s3d_unet_train_attach(s3d_unet* m, float* params, int64_t numel) {
    S3D_CHECK(m && params, S3D_ERR_INVALID, "train_attach: null argument");
    S3D_CHECK(numel == s3d_unet_param_numel(m), S3D_ERR_INVALID, "train_attach: %lld floats given, the model has %lld",
              (long long)numel, (long long)s3d_unet_param_numel(m));
    S3D_CHECK(m->cfg.model_channels * 2 * m->cfg.channel_mult[0] <= 1024, S3D_ERR_UNSUPPORTED, "training: model too wide");
    m->flat_off.resize(m->specs.size());
    m->flat_index.clear();
    size_t off = 0;
    for (size_t i = 0; i < m->specs.size(); ++i) { m->flat_off[i] = off; m->flat_index[m->specs[i].name] = off; off += m->specs[i].numel(); }
    // the packed layout comes from the host packer; its values are then always rebuilt from `params` on the device
    if (!m->packed) {
        for (const auto& sp : m->specs)
            if (!m->host.count(sp.name)) m->host[sp.name].assign(sp.numel(), 0.f);
        S3D_TRY(pack_all(m));
    }
    m->flat = params; m->flat_numel = numel;
    S3D_TRY(build_pack_plan(m));
    m->tape.valid = false;
    return 0;
}

int s3d_unet_repack(s3d_unet* m, void* stream) {
    S3D_CHECK(m && m->flat, S3D_ERR_INVALID, "repack: call s3d_unet_train_attach first");
    m->drop_carries();                              // (a carried in_conv was formed with the old weights)
    return launch_repack(m, static_cast<hipStream_t>(stream));
}

int s3d_unet_forward_train(s3d_unet* m, const float* x, const float* t, int B, int H, int W, int D, float* out, void* stream) {
    S3D_CHECK(m && x && t && out, S3D_ERR_INVALID, "forward_train: null argument");
    S3D_CHECK(m->flat, S3D_ERR_INVALID, "forward_train: call s3d_unet_train_attach first");
    S3D_CHECK(m->cur_lane == 0, S3D_ERR_INVALID, "forward_train: training runs on workspace lane 0 (lane %d is selected)", m->cur_lane);
    S3D_CHECK(B >= 1 && H >= 1 && W >= 1 && D >= 1, S3D_ERR_INVALID, "forward_train: B,H,W,D must be >= 1");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // measure forward + backward together: the workspace must not move between the two.  Pure host work that depends
    // only on the shapes: skipped when they repeat (every step of a training run).
    const long long key[4] = {B, H, W, D};
    const bool same = m->train_key[0] == key[0] && m->train_key[1] == key[1] && m->train_key[2] == key[2] && m->train_key[3] == key[3];
    int rc = 0;
    if (!same || m->train_high > m->arena.buf.cap) {
        m->arena.measuring = true;
        m->arena.high = 0;
        rc = run_forward(m, x, t, B, H, W, D, out, st, &m->tape);
        if (!rc) rc = run_backward(m, nullptr, nullptr, st);
        m->arena.measuring = false;
        m->tape.valid = false;
        if (rc) return rc;
        m->train_high = m->arena.high;
        for (int k = 0; k < 4; ++k) m->train_key[k] = key[k];
        if (m->arena.high > m->arena.buf.cap) {
            S3D_HIP(hipStreamSynchronize(st));
            S3D_TRY(m->arena.buf.reserve(m->arena.high + (m->arena.high >> 3)));
        }
    }
    // (s3d_unet_profile: every n-th training forward has its convolution launches bracketed by HIP events like the inference
    // forward's — bench.py --config c4 reads them; the flag stays up through the matching backward pass: its dgrad launches count)
    m->prof_now = m->prof_every > 0 && (m->fwd_count % m->prof_every) == 0;
    ++m->fwd_count;
    if (m->prof_now) ++m->prof_forwards;
    m->prof_train = m->prof_now;
    rc = run_forward(m, x, t, B, H, W, D, out, st, &m->tape);
    m->prof_now = false;
    if (rc) m->tape.valid = false;
    return rc;
}

int s3d_unet_backward(s3d_unet* m, const float* d_out, float* grads, void* stream) {
    return s3d_unet_backward_marked(m, d_out, grads, stream, nullptr, 0);
}

int s3d_unet_backward_marked(s3d_unet* m, const float* d_out, float* grads, void* stream, void** events, int n_events) {
    S3D_CHECK(m && d_out && grads && n_events >= 0 && n_events <= 2 && (n_events == 0 || events), S3D_ERR_INVALID, "backward: bad argument");
    S3D_CHECK(m->tape.valid, S3D_ERR_INVALID, "backward: no forward_train activations (call s3d_unet_forward_train first; an "
              "inference forward in between discards them)");
    m->arena.off = m->tape.arena_off;
    m->tape.valid = false;                    // the tape is consumed: gradient buffers reuse no forward memory, but one backward per forward
    hipEvent_t marks[2] = {nullptr, nullptr};
    for (int k = 0; k < n_events; ++k) marks[k] = static_cast<hipEvent_t>(events[k]);
    m->prof_now = m->prof_train;
    const int rc_b = run_backward(m, d_out, grads, static_cast<hipStream_t>(stream), marks, n_events);
    m->prof_now = m->prof_train = false;
    return rc_b;
}

int s3d_train_q_sample(const float* x0, int64_t x0_batch_stride, const float* noise, const float* sqrt_ac, const float* sqrt_1mac,
                       const int64_t* t, int B, int64_t per_sample, float* x_t, void* stream) {
    S3D_CHECK(x0 && noise && sqrt_ac && sqrt_1mac && t && x_t, S3D_ERR_INVALID, "q_sample: null argument");
    S3D_CHECK(x0_batch_stride == 0 || x0_batch_stride == per_sample, S3D_ERR_INVALID, "q_sample: x0_batch_stride must be 0 or per_sample");
    return launch_q_sample(x0, noise, sqrt_ac, sqrt_1mac, t, per_sample, B, x_t, static_cast<hipStream_t>(stream), x0_batch_stride);
}

int s3d_train_mse_terms(const float* model_out, const float* target, int64_t target_batch_stride, int B, int C, int H, int W, int D,
                        float* workspace, float* terms, void* stream) {
    S3D_CHECK(model_out && target && workspace && terms, S3D_ERR_INVALID, "mse_terms: null argument");
    S3D_CHECK(target_batch_stride == 0 || target_batch_stride == (int64_t)C * (H + D) * (W + D), S3D_ERR_INVALID, "mse_terms: target_batch_stride must be 0 or C*(H+D)*(W+D)");
    return launch_mse_terms(model_out, target, B, C, H, W, D, workspace, terms, static_cast<hipStream_t>(stream), target_batch_stride);
}

int s3d_train_mse_grad(const float* model_out, const float* target, int64_t target_batch_stride, const float* weight, int weight_cols,
                       float weight_divisor, int B, int C, int H, int W, int D, float* d_out, void* stream) {
    S3D_CHECK(model_out && target && weight && d_out, S3D_ERR_INVALID, "mse_grad: null argument");
    S3D_CHECK(weight_cols == 1 || weight_cols == 3, S3D_ERR_INVALID, "mse_grad: weight_cols must be 1 or 3");
    S3D_CHECK(target_batch_stride == 0 || target_batch_stride == (int64_t)C * (H + D) * (W + D), S3D_ERR_INVALID, "mse_grad: target_batch_stride must be 0 or C*(H+D)*(W+D)");
    return launch_mse_grad(model_out, target, weight, B, C, H, W, D, d_out, static_cast<hipStream_t>(stream), target_batch_stride,
                           weight_cols, weight_divisor);
}

int s3d_train_adamw_ema(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* const* ema,
                        const float* ema_rates, int n_ema, int64_t numel, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, void* stream) {
    S3D_CHECK(params && grads && exp_avg && exp_avg_sq && (n_ema == 0 || (ema && ema_rates)), S3D_ERR_INVALID, "adamw_ema: null argument");
    return launch_adamw_ema(params, grads, exp_avg, exp_avg_sq, ema, ema_rates, n_ema, numel, lr, beta1, beta2, eps, weight_decay,
                            step, static_cast<hipStream_t>(stream));
}

}  // extern "C"
