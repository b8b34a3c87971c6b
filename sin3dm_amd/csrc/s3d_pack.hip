// s3d_pack.hip — device-side weight repacking for the training tier.
//
// Inference packs the kernel-layout weight image on the host once (s3d_unet.hip:pack_all).  During training the
// parameters change on the device every step, so the same image — plus the transposed operators the backward pass
// needs — is rebuilt by ONE launch from the flat, reference-layout master parameter vector (OIHW conv weights,
// [out][in] linears; SURVEY.md §8b state-dict names).  Every job writes the same values the host packer would
// (sums in double, same order), which tests/test_hip_train.py checks bit-for-bit.
#include "s3d_model.h"

namespace s3d {



// a launch is ~10^7 items: 1024 per block (four per thread) and the block's job from a per-block table — with 256-item blocks
// that each binary-searched the job list (nine dependent loads) the launch was 32 rounds of blocks waiting on that chain:
// 90 us per training step (profiles/r04_train_step.txt)
constexpr int kPackItems = 1024;
__device__ __forceinline__ void repack_item(const PackDesc& d, long long i, const float* __restrict__ flat, float* __restrict__ wbuf,
                                            float* __restrict__ tbuf) {
    const float* W = flat + d.src;
    float* dst = (d.to_tbuf ? tbuf : wbuf) + d.dst;
    const int cout = d.cout, ctot = d.ctot, cin = d.cin, taps = d.taps;
    switch (d.kind) {
        case PK_COPY: dst[i] = W[i]; break;
        case PK_TRANS2D: {                       // src [cout][cin] -> dst [cin][cout]
            const int c = int(i / cout), o = int(i % cout);
            dst[i] = W[(size_t)o * cin + c];
            break;
        }
        // The tap-permuting kinds below take one (co, c) FILTER per item and move all its taps: the thread reads `taps`
        // consecutive floats (a wave: one contiguous run) and every tap's store is coalesced across the wave.  (One item per
        // element gathered its source with a stride of `taps` floats: nine times the bytes through L2 -> L1, the launch's bound.)
        case PK_DENSE: {                         // dst [(t*cout + co)*cin + c]; item = (co, c), c fastest
            const int c = int(i % cin), co = int(i / cin);
            const float* src = W + ((size_t)co * ctot + c) * taps;
            for (int t = 0; t < taps; ++t) dst[((size_t)t * cout + co) * cin + c] = src[t];
            break;
        }
        case PK_DENSE_SLICE: {                   // 1x1: dst [co*cin + c] = W[co][slot + c] (an input-channel slice, Fwd::resblock_cat)
            const int c = int(i % cin), co = int(i / cin);
            dst[i] = W[(size_t)co * ctot + d.slot + c];
            break;
        }
        case PK_DENSE_T: {                       // dst [(t*cin + c)*cout + co]: the dgrad operator as a forward conv; item = (c, co), co fastest
            const int co = int(i % cout), c = int(i / cout);
            const float* src = W + ((size_t)co * ctot + c) * taps;
            for (int t = 0; t < taps; ++t) dst[((size_t)t * cin + c) * cout + co] = src[taps - 1 - t];
            break;
        }
        case PK_DENSE_PAD: {                     // like PK_DENSE with the input channels zero-padded to d.slot per row
            const int c = int(i % cin); long long r = i / cin;
            const int co = int(r % cout), t = int(r / cout);
            dst[((size_t)t * cout + co) * d.slot + c] = W[((size_t)co * ctot + c) * taps + t];
            break;
        }
        case PK_DENSE_T_PAD: {                   // like PK_DENSE_T with the (transposed operator's) outputs padded to d.slot
            const int co = int(i % cout); long long r = i / cout;
            const int c = int(r % cin), t = int(r / cin);
            dst[((size_t)t * d.slot + c) * cout + co] = W[((size_t)co * ctot + c) * taps + (taps - 1 - t)];
            break;
        }
        case PK_WINO:
        case PK_WINO_T: {                        // item = one (n, k) filter of the (possibly transposed) operator
            // operator dims: N outputs x K inputs; forward: N = cout, K = cin; transposed: N = cin, K = cout, taps flipped
            const bool tr = d.kind == PK_WINO_T;
            const int N = tr ? cin : cout, K = tr ? cout : cin;
            const int n = int(i / K), k = int(i % K);
            double g[9];
#pragma unroll
            for (int q = 0; q < 9; ++q)
                g[q] = tr ? double(W[((size_t)k * ctot + n) * 9 + (8 - q)]) : double(W[((size_t)n * ctot + k) * 9 + q]);
            const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            double t[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 3; ++q) t[u][q] = G[u][0] * g[0 * 3 + q] + G[u][1] * g[1 * 3 + q] + G[u][2] * g[2 * 3 + q];
            const int k8t = K / 8;
            const int nt = n >> 5, jn = n & 31, k8 = k >> 3, hf = (k >> 2) & 1, e = k & 3;
            (void)N;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double uv = t[u][0] * G[v][0] + t[u][1] * G[v][1] + t[u][2] * G[v][2];
                    dst[((((size_t)nt * k8t + k8) * 16 + (u * 4 + v)) * 64 + (hf * 32 + jn)) * 4 + e] = float(uv);
                }
            break;
        }
        case PK_WINO24S_T:
        case PK_WINO24S: {                       // G2 g G4^T in the fragment order [n32][k16][24][2][64][4]
            // operator dims as in PK_WINO: forward N = cout, K = cin; transposed (dgrad) N = cin, K = cout, taps flipped.
            // item = one LANE of a fragment group: (16 outputs n) x (four input quads of one k16 step) — the thread transforms
            // the four filters (n, k0 .. k0+3) and each of its 24 stores is one float4 of a 1-KB fragment the wave writes whole.
            // (One filter per item scattered 24 four-byte stores each: the repack's bound after the tap-permuting kinds.)
            const bool tr = d.kind == PK_WINO24S_T;
            const int N = tr ? cin : cout, K = tr ? cout : cin;
            const int lane = int(i & 63);
            const long long blk = i >> 6;
            const int k16n = K / 16, nblk = int(blk / k16n), k16 = int(blk % k16n);
            const int n = nblk * 16 + (lane & 15), k0 = k16 * 16 + (lane >> 4) * 4;
            if (n >= N) break;
            const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            const double G4[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                     {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
            float out[24][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + e;
                double g[9];
#pragma unroll
                for (int q = 0; q < 9; ++q)
                    g[q] = tr ? double(W[((size_t)k * ctot + n) * 9 + (8 - q)]) : double(W[((size_t)n * ctot + k) * 9 + q]);
                double t[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 3; ++q) t[u][q] = G2[u][0] * g[0 * 3 + q] + G2[u][1] * g[1 * 3 + q] + G2[u][2] * g[2 * 3 + q];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 6; ++v)
                        out[u * 6 + v][e] = float(fma(t[u][0], G4[v][0], fma(t[u][1], G4[v][1], t[u][2] * G4[v][2])));       // as pack_wino24s_weights
            }
            float4* d4 = reinterpret_cast<float4*>(dst) + (((size_t)(nblk >> 1) * k16n + k16) * 24 * 2 + (nblk & 1)) * 64 + lane;
#pragma unroll
            for (int f = 0; f < 24; ++f) d4[(size_t)f * 2 * 64] = make_float4(out[f][0], out[f][1], out[f][2], out[f][3]);
            break;
        }
        case PK_RANK1: {                         // item (co, c), c fastest: the nine (t, o) entries dst [(t*n3 + (co/8)*24 + o*8 + co%8)*cin + c], n3 = ceil(cout/8)*24
            const int c = int(i % cin), co = int(i / cin);
            const int n3 = (cout + 7) / 8 * 24;
            const float* src = W + ((size_t)co * ctot + d.slot * cin + c) * 9;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const int kh = d.col_varying ? o : t, kw = d.col_varying ? t : o;
                    dst[((size_t)t * n3 + (co / 8) * 24 + o * 8 + (co & 7)) * cin + c] = src[kh * 3 + kw];
                }
            break;
        }
        case PK_RANK1F: {                        // the same entries in the fragment order of k_rank1b (rank1_frag_index)
            const int c = int(i % cin), co = int(i / cin);
            const float* src = W + ((size_t)co * ctot + d.slot * cin + c) * 9;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const int kh = d.col_varying ? o : t, kw = d.col_varying ? t : o;
                    dst[rank1_frag_index(cin, t, o, co, c)] = src[kh * 3 + kw];
                }
            break;
        }
        case PK_RANK1_BWD: {                     // dst [(tap*cin + c)*(3*cout) + j*cout + co]; item = (c, co), co fastest
            const int co = int(i % cout), c = int(i / cout);
            const float* src = W + ((size_t)co * ctot + d.slot * cin + c) * 9;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int dr = d.col_varying ? j : 2 - tap, dc = d.col_varying ? 2 - tap : j;
                    dst[((size_t)tap * cin + c) * (3 * cout) + j * cout + co] = src[dr * 3 + dc];
                }
            break;
        }
    }
}
__global__ __launch_bounds__(256) void k_repack(const PackDesc* __restrict__ descs, int ndesc, const float* __restrict__ flat,
                                                float* __restrict__ wbuf, float* __restrict__ tbuf) {
    const int* block_desc = reinterpret_cast<const int*>(descs + ndesc);      // (finalize_pack_plan: the table sits behind the jobs)
    const PackDesc d = descs[block_desc[blockIdx.x]];
    const long long i0 = (long long)(int(blockIdx.x) - d.block_begin) * kPackItems + threadIdx.x;
#pragma unroll
    for (int u = 0; u < kPackItems / 256; ++u) {
        const long long i = i0 + u * 256;
        if (i < d.n) repack_item(d, i, flat, wbuf, tbuf);
    }
}
// block prefix + per-block job table, uploaded behind the job list
int finalize_pack_plan(std::vector<PackDesc>& descs, DevBuf& dev, int& blocks_out) {
    int blocks = 0;
    std::vector<int> table;
    for (size_t i = 0; i < descs.size(); ++i) {
        PackDesc& d = descs[i];
        d.block_begin = blocks;
        const int nb = int((d.n + kPackItems - 1) / kPackItems);
        table.insert(table.end(), nb, int(i));
        blocks += nb;
    }
    std::vector<char> img(descs.size() * sizeof(PackDesc) + table.size() * sizeof(int));
    memcpy(img.data(), descs.data(), descs.size() * sizeof(PackDesc));
    memcpy(img.data() + descs.size() * sizeof(PackDesc), table.data(), table.size() * sizeof(int));
    S3D_TRY(upload(dev, img.data(), img.size()));
    blocks_out = blocks;
    return 0;
}

namespace {

struct Plan {
    s3d_unet* m;
    size_t tsize = 0;                                  // floats in tbuf
    size_t flat_of(const std::string& name) const {
        for (size_t i = 0; i < m->specs.size(); ++i)
            if (m->specs[i].name == name) return m->flat_off[i];
        return size_t(-1);
    }
    size_t talloc(size_t n) { size_t off = (tsize + 63) & ~size_t(63); tsize = off + n; return off; }
    void add(int kind, size_t src, size_t dst, long long n, int cout = 0, int ctot = 0, int cin = 0, int taps = 0, int slot = 0,
             int colv = 0, int to_t = 0) {
        PackDesc d{};
        d.kind = kind; d.cout = cout; d.ctot = ctot; d.cin = cin; d.slot = slot; d.taps = taps; d.col_varying = colv;
        d.to_tbuf = to_t; d.src = (long long)src; d.dst = (long long)dst; d.n = n;
        m->descs.push_back(d);
    }
    void norm(const std::string& prefix, int c, const NormW& nw) {
        static const char* kP[3] = {"xy", "xz", "yz"};
        for (int p = 0; p < 3; ++p) {
            add(PK_COPY, flat_of(prefix + ".norm_" + kP[p] + ".weight"), nw.gamma[p], c);
            add(PK_COPY, flat_of(prefix + ".norm_" + kP[p] + ".bias"), nw.beta[p], c);
        }
    }
    void tconv(const std::string& prefix, ConvW& cw, ConvWT& wt) {
        static const char* kP[3] = {"xy", "xz", "yz"};
        const int cin = cw.cin, cout = cw.cout, k = cw.k, taps = k * k, ctot = cw.rollout ? 3 * cin : cin;
        for (int p = 0; p < 3; ++p) {
            const size_t w = flat_of(prefix + ".conv_" + kP[p] + ".weight");
            add(PK_COPY, flat_of(prefix + ".conv_" + kP[p] + ".bias"), cw.bias[p], cout);
            // the per-step repack writes only the images this process's kernels read (S3D_WINO / S3D_CONV_IMPL are fixed per
            // process): with the mixed kernel active the F(2x2) images — and the dense [tap][cout][cin] images of the 3x3
            // convolutions, which only the direct and the naive kernels read — would be rewritten for nobody
            const bool fwd24 = k == 3 && cw.wino24s[p] && conv_wino24_channels(cin, cout);
            const bool bwd24 = k == 3 && cw.wino24s[p] && cout % 32 == 0 && conv_wino24_channels(cout, cin);
            // (what is skipped here is recorded: a launch that would read a skipped image fails instead of using stale weights)
            if (p == 0) { cw.only24_current = fwd24 && !conv_use_naive(); wt.only24_current = bwd24 && !conv_use_naive(); }
            if (!fwd24 || conv_use_naive()) add(PK_DENSE, w, cw.dense[p], (long long)cout * cin, cout, ctot, cin, taps);
            wt.dense_T[p] = talloc(size_t(taps) * cout * cin);
            if (!bwd24 || conv_use_naive()) add(PK_DENSE_T, w, wt.dense_T[p], (long long)cout * cin, cout, ctot, cin, taps, 0, 0, 1);
            if (k == 3) {
                if (!fwd24) add(PK_WINO, w, cw.wino[p], (long long)cout * cin, cout, ctot, cin, 9);
                if (cw.wino24s[p]) add(PK_WINO24S, w, cw.wino24s[p], (long long)((cout + 15) / 16) * (cin / 16) * 64, cout, ctot, cin, 9);
                if (!bwd24) {      // transposed operator: cin outputs (padded to 32) x cout inputs
                    wt.wino_T[p] = talloc(size_t((cin + 31) / 32) * (cout / 8) * 16 * 256);
                    wt.has_wino_T = true;
                    add(PK_WINO_T, w, wt.wino_T[p], (long long)cout * cin, cout, ctot, cin, 9, 0, 0, 1);
                } else {           // dgrad through k_conv_wino24s: K = cout in 32-channel chunks
                    wt.wino24s_T[p] = talloc(wino24_packed_floats(cin, cout));
                    wt.has_wino24s_T = true;
                    add(PK_WINO24S_T, w, wt.wino24s_T[p], (long long)((cin + 15) / 16) * (cout / 16) * 64, cout, ctot, cin, 9, 0, 0, 1);
                }
            }
            if (!cw.rollout) continue;
            const bool a_is_col = (p == 0);
            for (int slot = 1; slot <= 2; ++slot) {
                const bool colv = (slot == 1) ? a_is_col : !a_is_col;
                add(PK_RANK1, w, colv ? cw.rcol[p] : cw.rrow[p], (long long)cout * cin, cout, ctot, cin, 9, slot, colv);
                if (colv ? cw.rcol_f[p] : cw.rrow_f[p]) add(PK_RANK1F, w, colv ? cw.rcol_f[p] : cw.rrow_f[p], (long long)cout * cin, cout, ctot, cin, 9, slot, colv);
                const size_t off = talloc(size_t(3) * cin * 3 * cout);
                (colv ? wt.rcol_T[p] : wt.rrow_T[p]) = off;
                add(PK_RANK1_BWD, w, off, (long long)cin * cout, cout, ctot, cin, 9, slot, colv, 1);
            }
        }
    }
};

}  // namespace

// Build the repack plan for the current packed layout (pack_all must have run) and allocate the training image.
int build_pack_plan(s3d_unet* m) {
    S3D_CHECK(m->packed, S3D_ERR_INVALID, "build_pack_plan: parameters are not packed yet");
    const s3d_unet_cfg& c = m->cfg;
    const int mc = c.model_channels, ted = 4 * mc;
    const bool ssn = c.use_scale_shift_norm != 0;
    m->descs.clear();
    Plan P{m};
    P.add(PK_COPY, P.flat_of("time_embed.0.weight"), m->te0_w, (long long)ted * mc);
    P.add(PK_COPY, P.flat_of("time_embed.0.bias"), m->te0_b, ted);
    P.add(PK_COPY, P.flat_of("time_embed.2.weight"), m->te2_w, (long long)ted * ted);
    P.add(PK_COPY, P.flat_of("time_embed.2.bias"), m->te2_b, ted);
    m->in_blocks_t.assign(m->in_blocks.size(), ResBlockWT());
    m->out_blocks_t.assign(m->out_blocks.size(), ResBlockWT());
    auto block = [&](ResBlockW& rb, ResBlockWT& wt) {
        const int eo = ssn ? 2 * rb.Cout : rb.Cout;
        P.add(PK_COPY, P.flat_of(rb.prefix + ".emb_layers.1.weight"), m->film_w + size_t(rb.film_off) * ted, (long long)eo * ted);
        P.add(PK_COPY, P.flat_of(rb.prefix + ".emb_layers.1.bias"), m->film_b + rb.film_off, eo);
        P.norm(rb.prefix + ".in_layers.0", rb.C, rb.n1);
        P.tconv(rb.prefix + ".in_layers.2", rb.c1, wt.c1);
        P.norm(rb.prefix + ".out_layers.0", rb.Cout, rb.n2);
        P.tconv(rb.prefix + ".out_layers.2", rb.c2, wt.c2);
        if (rb.has_skip) P.tconv(rb.prefix + ".skip_connection", rb.skip, wt.skip);
        if (rb.has_skip && rb.c_up > 0 && rb.skip_a.dense[0]) {
            static const char* kQ[3] = {"xy", "xz", "yz"};
            for (int p = 0; p < 3; ++p) {
                const size_t w = P.flat_of(rb.prefix + ".skip_connection.conv_" + kQ[p] + ".weight");
                P.add(PK_DENSE_SLICE, w, rb.skip_a.dense[p], (long long)rb.Cout * rb.c_up, rb.Cout, rb.C, rb.c_up, 1, 0);
                P.add(PK_DENSE_SLICE, w, rb.skip_b.dense[p], (long long)rb.Cout * (rb.C - rb.c_up), rb.Cout, rb.C, rb.C - rb.c_up, 1, rb.c_up);
            }
        }
    };
    for (size_t i = 0; i < m->in_blocks.size(); ++i) block(m->in_blocks[i], m->in_blocks_t[i]);
    for (size_t i = 0; i < m->out_blocks.size(); ++i) block(m->out_blocks[i], m->out_blocks_t[i]);
    static const char* kP[3] = {"xy", "xz", "yz"};
    {
        const int ci = c.in_channels, co = c.channel_mult[0] * mc;
        for (int p = 0; p < 3; ++p) {
            P.add(PK_TRANS2D, P.flat_of(std::string("in_conv.0.conv_") + kP[p] + ".weight"), m->in_wT + size_t(p) * ci * co,
                  (long long)ci * co, co, 0, ci);
            P.add(PK_COPY, P.flat_of(std::string("in_conv.0.conv_") + kP[p] + ".bias"), m->in_b + size_t(p) * co, co);
        }
    }
    {
        const int ci = c.channel_mult[0] * mc, co = c.out_channels;
        P.norm("out.0", ci, m->out_norm);
        for (int p = 0; p < 3; ++p) {
            P.add(PK_COPY, P.flat_of(std::string("out.2.conv_") + kP[p] + ".weight"), m->out_w + size_t(p) * co * ci, (long long)co * ci);
            P.add(PK_COPY, P.flat_of(std::string("out.2.conv_") + kP[p] + ".bias"), m->out_b + size_t(p) * co, co);
        }
    }
    for (auto& d : m->descs) S3D_CHECK(d.src >= 0 && d.src < m->flat_numel, S3D_ERR_INVALID, "build_pack_plan: unresolved parameter");
    S3D_TRY(m->tbuf.reserve(std::max<size_t>(P.tsize, 64) * sizeof(float)));
    S3D_HIP(hipMemset(m->tbuf.p, 0, P.tsize * sizeof(float)));       // the Winograd images are zero-padded to 32 outputs
    S3D_TRY(finalize_pack_plan(m->descs, m->descs_dev, m->pack_blocks));
    return 0;
}

int launch_repack_generic(const PackDesc* descs_dev, int ndesc, int blocks, const float* flat, float* wbuf, float* tbuf, hipStream_t st) {
    if (!blocks) return 0;
    hipLaunchKernelGGL(k_repack, dim3(blocks), dim3(256), 0, st, descs_dev, ndesc, flat, wbuf, tbuf);
    S3D_HIP(hipGetLastError());
    return 0;
}

int launch_repack(s3d_unet* m, hipStream_t st) {
    S3D_CHECK(m->flat && m->pack_blocks > 0, S3D_ERR_INVALID, "repack: no flat parameter vector attached");
    hipLaunchKernelGGL(k_repack, dim3(m->pack_blocks), dim3(256), 0, st, static_cast<const PackDesc*>(m->descs_dev.p),
                       int(m->descs.size()), m->flat, static_cast<float*>(m->wbuf.p), static_cast<float*>(m->tbuf.p));
    S3D_HIP(hipGetLastError());
    return 0;
}

}  // namespace s3d
