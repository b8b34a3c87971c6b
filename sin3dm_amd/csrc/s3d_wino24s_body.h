// s3d_wino24s_body.h — the body of k_conv_wino24s, included twice by s3d_wino24.hip: W24S_GNB = 0 is the kernel of every forward
// and of plain input-gradient launches (its token stream is the kernel as it stood before this file existed: same code, same
// registers), W24S_GNB = 1 is k_conv_wino24s_gnb.
// W24S_GNB = 1 (training, s3d_train.hip): the launch is the input-gradient convolution in front of a GroupNorm(+FiLM)+SiLU backward;
// its finishing threads hold the gradient of the activated tensor and form dz = dy * silu'(z) and dz * xh there — the two
// per-channel sums that k_gn_bwd_partials otherwise gets from a read pass over (x, dy) — from the norm's input x, its statistics
// and affine / FiLM constants (GnbArgs), and leave them as per-tile records in the kernel's GroupNorm-partial layout (one record per
// CHANNEL: gn_sg = 1).  A dgrad launch has no rollout tables: those operands are compiled out, which pays for the new ones in
// registers.  The convolution's own output is the same bits as with W24S_GNB = 0.
#if W24S_GNB
__global__ __launch_bounds__(256, 3) void k_conv_wino24s_gnb(ConvArgs args, GnbArgs gb) {
#else
__global__ __launch_bounds__(256, 3) void k_conv_wino24s(ConvArgs args) {
#endif
    __shared__ __attribute__((aligned(16))) float smem[2 * C_ABUF];             // 51.8 KB: two halo buffers; four share images after the loop
    static_assert(4 * C_IMG <= 2 * C_ABUF, "LDS plan");
    W24_STAMP(0)
    int bid = blockIdx.x;
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    if (args.xcd_swizzle & 1) {
        const int chunk = int(gridDim.x) >> 3;
        if (bid < (chunk << 3)) bid = (bid & 7) * chunk + (bid >> 3);
    }
    int j = 0;
#pragma unroll
    for (int k = 1; k < kMaxConvJobs; ++k) j += (k < args.njobs && bid >= args.job[k].block_begin) ? 1 : 0;
    const ConvJob& J = args.job[j];
    int local = bid - J.block_begin;
    const int n32 = local % J.n_tiles_n; local /= J.n_tiles_n;
    const int b = local / J.tiles_per_img; local %= J.tiles_per_img;
    const int tile_idx = local;
    const int ty0 = (local / J.tiles_x) * C_TH, tx0 = (local % J.tiles_x) * C_TW;
    const int h = J.h, w = J.w, cin = args.cin, cout = args.cout;

    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);                 // row frequency of this wave
    const int t16 = lane & 15, g = lane >> 4;                               // tile of the lane, channel quad of the lane
    const int tr = t16 >> 2, tc = t16 & 3;
    const int xrow = u == 0 ? 0 : (u == 2 ? 2 : 1), yrow = u == 2 ? 1 : (u == 3 ? 3 : 2);
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(u == 1 ? 0x3F800000 : 0xBF800000));
    // byte addresses of the lane's two patch rows for the first / second 16 channels of a chunk (logical quad kk*4 + g)
    const int qx = g ^ (((tr + (xrow >> 1)) & 3) << 1), qy = g ^ (((tr + (yrow >> 1)) & 3) << 1);
    const int bx = ((2 * tr + xrow) * C_HW + 4 * tc) * C_LD * 4, by = ((2 * tr + yrow) * C_HW + 4 * tc) * C_LD * 4;
    const int ax0 = bx + 16 * qx, ax1 = bx + 16 * (qx ^ 4), ay0 = by + 16 * qy, ay1 = by + 16 * (qy ^ 4);

    const int k16_total = cin / 16;
    const float* ub = J.wgt + ((size_t(n32) * k16_total) * 48 + u * 12) * 256;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, k16_total * 48 * 1024, 0x00020000);
    const int wlane = lane * 16;
    auto wfrag = [&](int step, int s) -> f32x4 {                            // s = 2 * frequency + cout block
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (step * 48 + s) * 1024, 0));
    };
    const float* inb = J.in + size_t(b) * h * w * cin;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inb), 0, h * w * cin * 4, 0x00020000);
    // (round 3) the first weight fragments leave before the halo addressing is worked out, and that addressing avoids the
    // quarter-rate 32-bit multiplies: the byte offset of a halo item is two 24-bit multiply-adds with wave-uniform strides (the
    // launcher guarantees a plane below 2 GiB, so row stride < 2^24 holds for every supported shape) and pix / 18 is a
    // multiply-shift, exact for the 180 pixels of a halo
    f32x4 ring[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) { ring[s] = wfrag(0, s); __builtin_amdgcn_sched_barrier(0); }
    unsigned goff[C_ITEMS_PT];
    int loff[C_ITEMS_PT];
    const unsigned rowstride = unsigned(w) * unsigned(cin) * 4u, pixstride = unsigned(cin) * 4u;
    const bool small_strides = rowstride < (1u << 24) && h < (1 << 24);
#pragma unroll
    for (int it = 0; it < C_ITEMS_PT; ++it) {
        const int item = it * 256 + tid;
        const int pix = item >> 3, q = item & 7;
        const int hy = (pix * 57) >> 10, hx = pix - hy * C_HW;             // pix / 18 for pix < 192
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        const bool ok = item < C_ITEMS && gy >= 0 && gy < h && gx >= 0 && gx < w;
        const unsigned off = small_strides ? __umul24(unsigned(gy), rowstride) + __umul24(unsigned(gx), pixstride) + unsigned(q) * 16u
                                           : unsigned((gy * w + gx) * cin + q * 4) * 4u;
        goff[it] = ok ? off : 0x80000000u;
        loff[it] = pix * C_LD + ((q ^ (((hy >> 1) & 3) << 1)) << 2);
    }
    const bool last_ok = (C_ITEMS_PT - 1) * 256 + tid < C_ITEMS;
    auto item_load = [&](int it, int ch) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[it], ch * (C_KC * 4), 0));
    };
    auto item_store = [&](int it, int buf, f32x4 v) {
        if (it < C_ITEMS_PT - 1 || last_ok) *reinterpret_cast<f32x4*>(smem + buf * C_ABUF + loff[it]) = v;
    };

    f32x4 acc[6][2];
#pragma unroll
    for (int f = 0; f < 6; ++f)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[f][nb] = zero4;

    const int nchunks = cin / C_KC;
    f32x4 V[6];
    // chunk 0 into buffer 0, then the barrier; the first half of chunk 1 is requested with it but lands in buffer 1 only
    // after the first operands are built (the barrier of step (0,0) publishes it): it is off the prologue's critical path
    f32x4 pre[3];
    {
        const int c1 = nchunks > 1 ? 1 : 0;
        f32x4 h0[C_ITEMS_PT];
#pragma unroll
        for (int it = 0; it < C_ITEMS_PT; ++it) h0[it] = item_load(it, 0);
#pragma unroll
        for (int it = 0; it < 3; ++it) pre[it] = item_load(it, c1);
#pragma unroll
        for (int it = 0; it < C_ITEMS_PT; ++it) item_store(it, 0, h0[it]);
    }
    __syncthreads();
    W24_STAMP(1)
#define C_LDS4(off) (*static_cast<const f32x4*>(__builtin_assume_aligned(reinterpret_cast<const char*>(smem) + (off), 16)))
#define C_PIN(v) asm volatile("" : "+v"(v))
    {
        f32x4 t[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const f32x4 x = C_LDS4(ax0 + c * (C_LD * 4)), y = C_LDS4(ay0 + c * (C_LD * 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) t[c][e] = fmaf(sgn, y[e], x[e]);
        }
        const f32x4 s1 = t[4] - 4.f * t[2], s2 = t[3] - 4.f * t[1], s3 = t[4] - t[2], s4 = t[3] - t[1];
        V[0] = 4.f * t[0] + (t[4] - 5.f * t[2]);
        V[1] = s1 + s2; V[2] = s1 - s2;
        V[3] = s3 + 2.f * s4; V[4] = s3 - 2.f * s4;
        V[5] = 4.f * t[1] + (t[5] - 5.f * t[3]);
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) item_store(it, 1, pre[it]);

    // VERDICT r2 item 4, priced before built (tools/wino24_ubench.hip -DW24_GNSILU_PRICE; profiles/r03_gn_in_halo_price.txt): what
    // GroupNorm-apply + FiLM + SiLU on the halo items between their load and their ds_write would cost the k-loop — two fmas, exp,
    // rcp, mul and the padding select per element, constants per channel quad fetched per chunk.  (Arithmetic stand-in: results
    // are meaningless in this build.)
#ifdef W24_GNSILU_PRICE
#define W24_PRICE_GNSILU                                                                                              \
    {                                                                                                                 \
        const f32x4 gA = *reinterpret_cast<const f32x4*>(J.wgt + ((lch * 32 + (tid & 7) * 4) % cout));                \
        const f32x4 gB = *reinterpret_cast<const f32x4*>(J.wgt + ((lch * 32 + (tid & 7) * 4 + 4) % cout));            \
        _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                               \
            const bool ok_ = goff[it0 + t] != 0x80000000u;                                                            \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                           \
                float v_ = fmaf(pf[t][e], gA[e], gB[e]);                                                              \
                v_ = fmaf(v_, gB[e], gA[e]);                                                                          \
                v_ = v_ * __builtin_amdgcn_rcpf(1.0f + __expf(-v_));                                                  \
                pf[t][e] = ok_ ? v_ : 0.f;                                                                            \
            }                                                                                                         \
        }                                                                                                             \
    }
#else
#define W24_PRICE_GNSILU
#endif
    // One k-step (16 channels) = 12 groups {four MFMAs on one A operand + a piece of the other work}, pinned.
#define C_HAS_NEXT 1                                  /* 0 in a tile's last k-step: no next step's weight fragments to request */
#define C_GROUP(F, NB, WORK)                                                                                          \
    {                                                                                                                 \
        constexpr int s_ = 2 * (F) + (NB);                                                                            \
        const f32x4 bq = ring[s_ % 6];                                                                                \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][0], bq[0], acc[F][NB], 0, 0, 0);                       \
        if (s_ + 6 < 12) ring[s_ % 6] = wfrag(step, s_ + 6); else if (C_HAS_NEXT) ring[s_ % 6] = wfrag(nstep, s_ - 6);     \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][1], bq[1], acc[F][NB], 0, 0, 0);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        WORK                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][2], bq[2], acc[F][NB], 0, 0, 0);                       \
        acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][3], bq[3], acc[F][NB], 0, 0, 0);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
#define C_COMB(T, X, Y) { _Pragma("unroll") for (int e = 0; e < 4; ++e) T[e] = fmaf(sgn, Y[e], X[e]); C_PIN(T); }
    //   KK = 0: the next operands are the second 16 channels of the current buffer; halo items 3..5 of chunk c+1 go to the
    //           other buffer, then the chunk's barrier.   KK = 1: the next operands are the first 16 channels of the other
    //           buffer; halo items 0..2 of chunk c+2 go to the current buffer (whose last reads were before the barrier).
#define C_STEP(KK)                                                                                                    \
    {                                                                                                                 \
        const int step = chunk * 2 + (KK);                                                                            \
        const int nstep = step + 1 < k16_total ? step + 1 : step;                                                     \
        const int rx = ((KK) == 0 ? ax1 + cur : ax0 + (cur ^ tog)), ry = ((KK) == 0 ? ay1 + cur : ay0 + (cur ^ tog)); \
        constexpr int it0 = (KK) == 0 ? 3 : 0;                                                                        \
        const int lch = (KK) == 0 ? cn1 : cn2;                                                                        \
        f32x4 pf[3], cx0, cy0, cx1, cy1, t0, t1, t2, t3, t4, t5, s1, s2, s3, s4, v5n;                                 \
        C_GROUP(0, 0, cx0 = C_LDS4(rx); cy0 = C_LDS4(ry); cx1 = C_LDS4(rx + C_LD * 4); cy1 = C_LDS4(ry + C_LD * 4);)  \
        C_GROUP(0, 1, C_COMB(t0, cx0, cy0) C_COMB(t1, cx1, cy1)                                                       \
                      pf[0] = item_load(it0, lch); pf[1] = item_load(it0 + 1, lch); pf[2] = item_load(it0 + 2, lch);) \
        C_GROUP(1, 0, cx0 = C_LDS4(rx + 2 * C_LD * 4); cy0 = C_LDS4(ry + 2 * C_LD * 4); cx1 = C_LDS4(rx + 3 * C_LD * 4); cy1 = C_LDS4(ry + 3 * C_LD * 4);) \
        C_GROUP(1, 1, C_COMB(t2, cx0, cy0) C_COMB(t3, cx1, cy1))                                                      \
        C_GROUP(2, 0, cx0 = C_LDS4(rx + 4 * C_LD * 4); cy0 = C_LDS4(ry + 4 * C_LD * 4); cx1 = C_LDS4(rx + 5 * C_LD * 4); cy1 = C_LDS4(ry + 5 * C_LD * 4);) \
        C_GROUP(2, 1, C_COMB(t4, cx0, cy0) C_COMB(t5, cx1, cy1))                                                      \
        /* from here on V[0..2] are free: their MFMAs have been issued */                                             \
        C_GROUP(3, 0, s1 = t4 - 4.f * t2; C_PIN(s1); s2 = t3 - 4.f * t1; C_PIN(s2); V[1] = s1 + s2; C_PIN(V[1]); V[2] = s1 - s2; C_PIN(V[2]);) \
        C_GROUP(3, 1, V[0] = 4.f * t0 + (t4 - 5.f * t2); C_PIN(V[0]); s3 = t4 - t2; C_PIN(s3); s4 = t3 - t1; C_PIN(s4);) \
        C_GROUP(4, 0, V[3] = s3 + 2.f * s4; C_PIN(V[3]);)                                                             \
        C_GROUP(4, 1, v5n = 4.f * t1 + (t5 - 5.f * t3); C_PIN(v5n);)                                                  \
        C_GROUP(5, 0, V[4] = s3 - 2.f * s4; C_PIN(V[4]);)                                                             \
        C_GROUP(5, 1, ;)                                                                                              \
        V[5] = v5n;                                                                                                   \
        W24_PRICE_GNSILU                                                                                              \
        _Pragma("unroll") for (int t = 0; t < 3; ++t) item_store(it0 + t, (KK) == 0 ? (chunk + 1) & 1 : chunk & 1, pf[t]); \
        if ((KK) == 0) __syncthreads();                                                                               \
    }

    int cur = 0;                                  // byte offset of the buffer that holds the current chunk
    const int tog = C_ABUF * 4;
    W24_STAMP(2)
    __builtin_amdgcn_s_setprio(0);
    for (int chunk = 0; chunk < nchunks - 1; ++chunk) {
        const int cn1 = chunk + 1;
        const int cn2 = chunk + 2 < nchunks ? chunk + 2 : nchunks - 1;
        C_STEP(0)
        C_STEP(1)
        cur ^= tog;
    }
    // The last chunk is peeled (round 3): it has no successor to fetch, and its second step has no operands to build — the
    // registers and issue slots that frees carry the EPILOGUE's residual request (the operand that comes from HBM / the MALL),
    // which used to go out only after the last MFMA and was waited for behind the share-image barrier; the rank-1 tables
    // (L2-hot, written by the launch before) are still requested after the loop: all of them early spills 32 registers.
    {
        const int chunk = nchunks - 1;
        const int step = chunk * 2, nstep = step + 1;
        const int rx = ax1 + cur, ry = ay1 + cur;
        f32x4 cx0, cy0, cx1, cy1, t0, t1, t2, t3, t4, t5, s1, s2, s3, s4, v5n;
        C_GROUP(0, 0, cx0 = C_LDS4(rx); cy0 = C_LDS4(ry); cx1 = C_LDS4(rx + C_LD * 4); cy1 = C_LDS4(ry + C_LD * 4);)
        C_GROUP(0, 1, C_COMB(t0, cx0, cy0) C_COMB(t1, cx1, cy1))
        C_GROUP(1, 0, cx0 = C_LDS4(rx + 2 * C_LD * 4); cy0 = C_LDS4(ry + 2 * C_LD * 4); cx1 = C_LDS4(rx + 3 * C_LD * 4); cy1 = C_LDS4(ry + 3 * C_LD * 4);)
        C_GROUP(1, 1, C_COMB(t2, cx0, cy0) C_COMB(t3, cx1, cy1))
        C_GROUP(2, 0, cx0 = C_LDS4(rx + 4 * C_LD * 4); cy0 = C_LDS4(ry + 4 * C_LD * 4); cx1 = C_LDS4(rx + 5 * C_LD * 4); cy1 = C_LDS4(ry + 5 * C_LD * 4);)
        C_GROUP(2, 1, C_COMB(t4, cx0, cy0) C_COMB(t5, cx1, cy1))
        C_GROUP(3, 0, s1 = t4 - 4.f * t2; C_PIN(s1); s2 = t3 - 4.f * t1; C_PIN(s2); V[1] = s1 + s2; C_PIN(V[1]); V[2] = s1 - s2; C_PIN(V[2]);)
        C_GROUP(3, 1, V[0] = 4.f * t0 + (t4 - 5.f * t2); C_PIN(V[0]); s3 = t4 - t2; C_PIN(s3); s4 = t3 - t1; C_PIN(s4);)
        C_GROUP(4, 0, V[3] = s3 + 2.f * s4; C_PIN(V[3]);)
        C_GROUP(4, 1, v5n = 4.f * t1 + (t5 - 5.f * t3); C_PIN(v5n);)
        C_GROUP(5, 0, V[4] = s3 - 2.f * s4; C_PIN(V[4]);)
        C_GROUP(5, 1, ;)
        V[5] = v5n;
    }
    // ---- epilogue: as k_conv_wino24; lane (g, t16) holds output channel nb*16 + t16 of the tiles (tile row g, tile column r)
    const float* __restrict__ p_bias = J.bias;
    const float* __restrict__ p_bbias = J.bbias;
    const float* __restrict__ p_rcol = J.rcol;
    const float* __restrict__ p_rrow = J.rrow;
    const float* __restrict__ p_res = J.res;
    float* __restrict__ p_out = J.out;
    double* p_gn = J.gn_part;
    // the finishing threads' operands (bias, rank-1 tables, residual) are requested BEFORE the barrier and the share-image
    // writes: their L2/HBM latency runs beside both (2.3 -> 1.6 us for this phase)
    const int quad = tid & 7, xl = (tid >> 3) & 15, rsel = tid >> 7;
    const int co4 = n32 * 32 + quad * 4;
    const bool c_ok = co4 < cout;
    const int coc = c_ok ? co4 : 0;
    const int x = tx0 + xl;
    const bool x_ok = x < w && c_ok;
    const int xc = x < w ? x : 0;
    f32x4 base4 = p_bias ? *reinterpret_cast<const f32x4*>(p_bias + coc) : zero4;
    if (p_bbias) base4 += *reinterpret_cast<const f32x4*>(p_bbias + size_t(b) * J.bbias_stride + coc);
    f32x4 tcol[4], trow[4], tres[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { tcol[k] = zero4; trow[k] = zero4; tres[k] = zero4; }
    // (the table loads carry sc1 — the policy of round 3's measured kernel; each entry is read once per block)
    // args.r1_slices == 2: each table is the sum of two K slices (s3d_rank1.h) — both requested now, added behind the barrier
    const bool two = args.r1_slices == 2;
    f32x4 tcol2[4], trow2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { tcol2[k] = zero4; trow2[k] = zero4; }
    auto tload = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, kTabAux)); };
    auto request_col_tables = [&]() {
        if (p_rcol) {
            const float* base = p_rcol + size_t(b) * w * 4 * cout;
            const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, w * 4 * cout * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t trs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (two ? size_t(args.B) * w * 4 * cout : 0)), 0, w * 4 * cout * 4, 0x00020000);
            if (ty0 > 0 && ty0 + C_TH < h) {
                const unsigned off = unsigned(((xc * 4 + 0) * cout + coc) * 4);
                const f32x4 v0 = tload(trs, off);
                f32x4 v1 = zero4;
                if (two) v1 = tload(trs2, off);
    #pragma unroll
                for (int k = 0; k < 4; ++k) { tcol[k] = v0; tcol2[k] = v1; }
            } else {
    #pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = ty0 + rsel * 4 + k;
                    const unsigned off = unsigned(((xc * 4 + x_edge_variant(y < h ? y : 0, h)) * cout + coc) * 4);
                    tcol[k] = tload(trs, off);
                    if (two) tcol2[k] = tload(trs2, off);
                }
            }
        }
    };
    auto request_row_tables = [&]() {
        if (p_rrow) {
            const float* base = p_rrow + size_t(b) * h * 4 * cout;
            const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, h * 4 * cout * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t trs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (two ? size_t(args.B) * h * 4 * cout : 0)), 0, h * 4 * cout * 4, 0x00020000);
            const int vx = x_edge_variant(xc, w);
    #pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = ty0 + rsel * 4 + k;
                const unsigned off = unsigned((((y < h ? y : 0) * 4 + vx) * cout + coc) * 4);
                trow[k] = tload(trs, off);
                if (two) trow2[k] = tload(trs2, off);
            }
        }
    };
    auto request_residual = [&]() {
        if (p_res) {
    #pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = ty0 + rsel * 4 + k;
                tres[k] = *reinterpret_cast<const f32x4*>(p_res + ((size_t(b) * h + (y < h ? y : 0)) * w + xc) * cout + coc);
            }
        }
    };
#undef C_HAS_NEXT
#define C_HAS_NEXT 0
    {   // the tile's last k-step: MFMAs only; the operand requests go out behind its first groups
        const int step = (nchunks - 1) * 2 + 1, nstep = step; (void)nstep;
        C_GROUP(0, 0, ;) C_GROUP(0, 1, request_residual();) C_GROUP(1, 0, ;) C_GROUP(1, 1, ;) C_GROUP(2, 0, ;) C_GROUP(2, 1, ;)
        C_GROUP(3, 0, ;) C_GROUP(3, 1, ;) C_GROUP(4, 0, ;) C_GROUP(4, 1, ;) C_GROUP(5, 0, ;) C_GROUP(5, 1, ;)
    }
#undef C_HAS_NEXT
#undef C_STEP
#undef C_GROUP
#undef C_COMB
#undef C_LDS4
#undef C_PIN
#undef W24_PRICE_GNSILU
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    W24_STAMP(3)
    // GNB: the operands of dz — the norm's input at the thread's pixels, the broadcast gradients of the rollout means, the
    // per-channel constants (requested here, beside the barrier and the share-image writes, like the tables they replace)
#if W24S_GNB
    f32x4 gx[4], gra[4], gca = zero4, g_mean = zero4, g_rstd = zero4, g_gam = zero4, g_bet = zero4, g_sc = {1.f, 1.f, 1.f, 1.f}, g_sh = zero4;
    {
        const float* xs = gb.x[j];
        const float* ra = gb.rowadd[j];
        const float* cadd = gb.coladd[j];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = ty0 + rsel * 4 + k, yy = y < h ? y : 0;
            gx[k] = *reinterpret_cast<const f32x4*>(xs + ((size_t(b) * h + yy) * w + xc) * cout + coc);
            gra[k] = ra ? *reinterpret_cast<const f32x4*>(ra + (size_t(b) * h + yy) * cout + coc) : zero4;
        }
        if (cadd) gca = *reinterpret_cast<const f32x4*>(cadd + (size_t(b) * w + xc) * cout + coc);
        const int cg = cout / gb.groups;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float2 mr = *reinterpret_cast<const float2*>(gb.mr + ((size_t(b) * 3 + j) * gb.groups + (coc + e) / cg) * 2);
            g_mean[e] = mr.x; g_rstd[e] = mr.y;
        }
        g_gam = *reinterpret_cast<const f32x4*>(gb.gamma[j] + coc);
        g_bet = *reinterpret_cast<const f32x4*>(gb.beta[j] + coc);
        if (gb.film) {
            g_sc = *reinterpret_cast<const f32x4*>(gb.film + size_t(b) * gb.film_stride + coc) + 1.0f;
            g_sh = *reinterpret_cast<const f32x4*>(gb.film + size_t(b) * gb.film_stride + cout + coc);
        }
    }
#else
    request_row_tables();                                // (requesting the row tables a k-step early as well measured the same)
    request_col_tables();
#endif
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                     // all patch reads and halo stores of the last step are done
    {
        float* img = smem + u * C_IMG + t16;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r], m4 = acc[4][nb][r], m5 = acc[5][nb][r];
                const float p = m1 + m2, q = m1 - m2, rr = m3 + m4, s = m3 - m4;
                const int pp = (g * C_TW + 4 * r) * 32 + nb * 16;
                img[pp] = (m0 + p) + rr; img[pp + 32] = fmaf(2.f, s, q); img[pp + 64] = fmaf(4.f, rr, p); img[pp + 96] = fmaf(8.f, s, q) + m5;
            }
    }
    __syncthreads();                                     // the share images are complete
    W24_STAMP(4)
    f32x4 gs4 = zero4, gss4 = zero4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yl = rsel * 4 + k, y = ty0 + yl;
        const float* sp = smem + (yl & 1) * C_IMG + ((yl >> 1) * C_TW + xl) * 32 + quad * 4;
        const f32x4 ka = *reinterpret_cast<const f32x4*>(sp), kb = *reinterpret_cast<const f32x4*>(sp + C_IMG),
                    kc = *reinterpret_cast<const f32x4*>(sp + 2 * C_IMG);
        const f32x4 sum3 = (yl & 1) ? (ka - kb) - kc : (ka + kb) + kc;
#if W24S_GNB
        const f32x4 v = (sum3 + base4) + tres[k];        // (no tables in a dgrad launch; (x + 0) + r = x + r: the same bits)
#else
        const f32x4 v = (sum3 + base4) + (((tcol[k] + tcol2[k]) + (trow[k] + trow2[k])) + tres[k]);     // (x + 0 = x: one-slice tables add exact zeros)
#endif
        if (x_ok && y < h) {
            *reinterpret_cast<f32x4*>(p_out + ((size_t(b) * h + y) * w + x) * cout + co4) = v;
#if W24S_GNB
            // the arithmetic of s3d_bwd.hip:gn_dy4 / gn_dz, element for element
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float dy = v[e];
                dy = fmaf(gra[k][e], gb.rowscale[j], dy);
                dy = fmaf(gca[e], gb.colscale[j], dy);
                const float xh = (gx[k][e] - g_mean[e]) * g_rstd[e];
                const float z = (xh * g_gam[e] + g_bet[e]) * g_sc[e] + g_sh[e];
                const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
                const float dz = dy * sg * (1.0f + z * (1.0f - sg));
                gs4[e] += dz; gss4[e] = fmaf(dz, xh, gss4[e]);
            }
#else
            gs4 += v; gss4 += v * v;
#endif
        }
    }
    if (p_gn) {
        // one partial per BLOCK: the four waves' sums meet through LDS (wave order, double) — a quarter of the records for
        // the statistics' readers (the GN-act kernel adds them itself, s3d_kernels.hip:k_gn_act)
        __shared__ float gred[4][8][8];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) { gs4[e] += __shfl_xor(gs4[e], off, 64); gss4[e] += __shfl_xor(gss4[e], off, 64); }
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { gred[u][lane][e] = gs4[e]; gred[u][lane][4 + e] = gss4[e]; }
        }
        __syncthreads();
        if (tid < 64) {                                   // lanes 0..7 of wave 0: channel quad `quad` = lane
            double ds[4], dss[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int l8 = lane & 7;
                ds[e] = ((double(gred[0][l8][e]) + double(gred[1][l8][e])) + double(gred[2][l8][e])) + double(gred[3][l8][e]);
                dss[e] = ((double(gred[0][l8][4 + e]) + double(gred[1][l8][4 + e])) + double(gred[2][l8][4 + e])) + double(gred[3][l8][4 + e]);
            }
            const int sg = args.gn_sg;
            const int part = tile_idx;
            auto put = [&](int sub, double sv, double ssv) {
                double* dst = p_gn + ((size_t(b) * 3 * args.gn_nsub + sub) * args.gn_maxparts + part) * 2;
                dst[0] = sv; dst[1] = ssv;
            };
            if (sg >= 4) {
                double sv = (ds[0] + ds[1]) + (ds[2] + ds[3]), ssv = (dss[0] + dss[1]) + (dss[2] + dss[3]);
                for (int off = 1; off < (sg >> 2); off <<= 1) { sv += __shfl_xor(sv, off, 64); ssv += __shfl_xor(ssv, off, 64); }
                if (lane < 8 && c_ok && (co4 % sg) == 0) put(co4 / sg, sv, ssv);
            } else if (lane < 8 && c_ok) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    if (sg == 2) put((co4 + e) / 2, ds[e] + ds[e + 1], dss[e] + dss[e + 1]);
                    else { put(co4 + e, ds[e], dss[e]); put(co4 + e + 1, ds[e + 1], dss[e + 1]); }
                }
            }
        }
    }
    W24_STAMP(5)
}
