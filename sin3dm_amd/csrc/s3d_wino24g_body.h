// s3d_wino24g_body.h — k_conv_wino24g: the mixed Winograd F(2x4,3x3) kernel of s3d_wino24s_body.h with its halo staged by
// LDS-DMA (buffer_load_dwordx4 ... lds, 16 bytes per lane straight from L2 / HBM into LDS: no staging registers, no ds_write
// pass) and PERSISTENT blocks: a block walks a list of (tile, 32-output-channel) work items, and the next item's halo lands in
// LDS while the current item's output transform and stores run.  The arithmetic and its order — MFMA chains,
// share images, finishing sums, GroupNorm partials — are those of k_conv_wino24s: results are bit-identical
// (tests/test_hip_parity.py::test_switched_conv_forms_are_bit_identical_and_reported, tools/wino24_ubench.hip).
// Included twice by s3d_wino24.hip: W24G_GNB = 0 the plain kernel, = 1 with the GroupNorm-backward epilogue (k_conv_wino24s_gnb).
//
// LDS (49 KB + 1 KB, three blocks per CU): a ring of FOUR slots, each one 16-channel PIECE of the 10 x 18-pixel halo = one k-step's
// A operand: 180 pixels x 64 B = 11 520 B in a 12 288-B slot = 12 DMA pieces of 1 KB, three per wave.  An LDS-DMA writes
// M0 + lane * 16: the image must be lane-linear per wave-instruction, so there is no padding; the bank spread comes from the ORDER of
// the pixels instead (row a, column b -> pixel slot 18 a + sigma(b), sigma(b) = 4 (b & 3) + (b >> 2) for b < 16: the four tile
// columns of a patch column sit in four consecutive pixel slots) and from an XOR on the channel quad (q ^ 2 ((a >> 1) & 1)), both
// applied to the per-lane SOURCE address.  Four adjacent lanes fetch one pixel's 64 bytes (tools/glds_probe.hip: 14.4 TB/s into LDS
// chip-wide for this pattern, 17.7 for full 128-B lines, 8.8 for the bank-perfect 16-lane-strided one); a patch read is then
// conflict-free except for the two columns b = 16, 17 (one 2-way conflict in 12 reads).  Out-of-range buffer offsets (image
// border, pixel slots 180..191) land as ZEROS in LDS — the padding of the convolution (verified: tools/glds_probe.hip).
// The DMAs are issued from inline asm: hipcc would otherwise order every ds_read behind every LDS-DMA in flight (one array:
// everything may alias).  Hidden from its bookkeeping they cost nothing there — the counter is in order, so hipcc's own counted
// waits for the weight fragments only ever wait MORE than it thinks — and are retired by hand: s_waitcnt vmcnt(N) with N = the
// operations issued AFTER the pieces in question (never an over-count: a wait that is too weak is a race), then a barrier.
//
// Pipeline of one item with n = cin / 16 pieces: piece p lives in slot (p + 3) & 3.  Step d multiplies piece d's operands (in
// registers) and builds piece d + 1's from LDS; it issues the DMA of piece d + 3 (d >= 1; pieces 0..3 are in flight when the
// loop starts).  One barrier per two steps (end of the even ones) publishes pieces d + 2, d + 3 and frees the slots of d, d + 1.
// Item boundary: the share images cover slots 0..2, so the NEXT item's piece 0 goes to slot 3 as soon as the last patch reads are
// behind a barrier; its pieces 1..3 and its first weight fragments are requested once the finishing threads have read the images
// and BEFORE they store the outputs (stores share the in-order counter: requests behind them would wait for the write
// acknowledgements — what sank round 3's persistent form, profiles/r03_wino_persistent.txt).
#if W24G_GNB
__global__ __launch_bounds__(256, 3) void k_conv_wino24g_gnb(ConvArgs args, GnbArgs gb, int total_items) {
#else
__global__ __launch_bounds__(256, 3) void k_conv_wino24g(ConvArgs args, int total_items) {
#endif
    __shared__ __attribute__((aligned(16))) float smem[G_SMEM_FLOATS];          // ONE array (a second __shared__ object de-pipelines LDS-DMA kernels)
    static_assert(4 * C_IMG * 4 <= 3 * G_SLOT, "the share images must leave slot 3 free");
    const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) float*)smem));
    if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
    const int u = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);   // row frequency of this wave
    const int xrow = u == 0 ? 0 : (u == 2 ? 2 : 1), yrow = u == 2 ? 1 : (u == 3 ? 3 : 2);
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(u == 1 ? 0x3F800000 : 0xBF800000));
    const unsigned dma_dst = __builtin_amdgcn_readfirstlane(lds0 + 3 * u * 1024);
    const int cin = args.cin, cout = args.cout;
    const int k16_all = cin / 16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- work items.  Persistent launches (grid < items): every XCD owns one contiguous range of items and its blocks walk it with
    // the stride of the XCD's block count.  (Drawing the items from a ticket counter per XCD instead was built and measured: a
    // returning atomic sits in the wave's IN-ORDER vector-memory queue, every weight fragment behind it waits for it, and the k-loop
    // of the asking wave — hence of its block — grows by 1-5 us per item; profiles/r06_wino_glds.txt.)
    // One-item launches use the block order of k_conv_wino24s (contiguous per XCD).
    int first_item, range_end, per = 0;
    const bool persistent = int(gridDim.x) < total_items;                   // (then gridDim.x % 8 == 0: the launcher's choice)
    {
        const int G = int(gridDim.x);
        if (persistent) {
            const int xcd = int(blockIdx.x) & 7;
            per = G >> 3;
            const int base = total_items >> 3, rem = total_items & 7;
            const int range_begin = xcd * base + (xcd < rem ? xcd : rem);
            range_end = range_begin + base + (xcd < rem ? 1 : 0);
            first_item = range_begin + (int(blockIdx.x) >> 3);
        } else {
            int bid = blockIdx.x;
            const int chunk = G >> 3;
            if ((args.xcd_swizzle & 1) && bid < (chunk << 3)) bid = (bid & 7) * chunk + (bid >> 3);
            first_item = bid; range_end = bid + 1;
        }
    }
    // the lane id, recomputed wherever it is needed (volatile: never hoisted out of the item loop and kept — spilled — across the k-loop)
    auto lane_id_now = []() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // An item's coordinates, in two stages so that the kernel-argument loads of the first (the job's table entry: an s_load whose
    // address depends on the job) are in flight while other work runs, and are consumed by the second.
    struct Item { int j, n32, b, tile_idx, ty0, tx0; };
    struct JobRow { int j, local, n_tiles_n, tiles_per_img, tiles_x, h, w; const float* in; const float* wgt; };
    auto decode_job = [&](int bid) -> JobRow {
        int j = 0, nj = args.njobs;
        asm volatile("" : "+s"(nj));                                        // (opaque: seven `k < njobs` masks are not kept live across the k-loop)
#pragma unroll
        for (int k = 1; k < kMaxConvJobs; ++k) j += (k < nj && bid >= args.job[k].block_begin) ? 1 : 0;
        const ConvJob& J = args.job[j];
        return JobRow{j, bid - J.block_begin, J.n_tiles_n, J.tiles_per_img, J.tiles_x, J.h, J.w, J.in, J.wgt};
    };
    auto decode_tile = [&](const JobRow& r) -> Item {
        int local = r.local;
        Item it;
        it.j = r.j;
        it.n32 = local % r.n_tiles_n; local /= r.n_tiles_n;
        it.b = local / r.tiles_per_img; local %= r.tiles_per_img;
        it.tile_idx = local;
        it.ty0 = (local / r.tiles_x) * C_TH; it.tx0 = (local % r.tiles_x) * C_TW;
        return it;
    };
    auto row_desc = [&](const JobRow& r, const Item& it) -> i32x4 {
        const unsigned long long p = (unsigned long long)(r.in + size_t(it.b) * r.h * r.w * cin);
        return i32x4{int(unsigned(p)), int(unsigned(p >> 32) & 0xFFFF), int(unsigned(r.h) * unsigned(r.w) * unsigned(cin) * 4u), 0x00020000};
    };
    // a halo's DMA source: the buffer descriptor of the sample's plane (row_desc) + per-lane byte offsets of the three pieces (bit 31: out of range)
    // The lane's three DMA pieces: wave-instruction 3 u + m covers pixel slots 16 (3 u + m) .. + 15, four lanes (channel quads) each.
    // (Worked out per item from an opaque copy of the lane id: hoisted out of the item loop these values would be live — spilled —
    // across the k-loop.)
    auto halo_offsets = [&](const Item& it, int ph, int pw, unsigned (&goff)[3]) {
        int l_ = lane_id_now();      // the lane id without a live register
        asm volatile("" : "+v"(l_));
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int P = 16 * (3 * u + m) + (l_ >> 2);
            const int a = (P * 57) >> 10, s = P - a * 18;                   // P / 18 for P < 192
            const int bcol = s < 16 ? 4 * (s & 3) + (s >> 2) : s;
            const int gy = it.ty0 - 1 + a, gx = it.tx0 - 1 + bcol;
            const bool ok = P < 180 && gy >= 0 && gy < ph && gx >= 0 && gx < pw;          // (pixel slots 180..191: out of range -> zeros)
            goff[m] = ok ? unsigned((gy * pw + gx) * cin) * 4u + unsigned(((l_ & 3) ^ (((a >> 1) & 1) << 1)) << 4) : 0x80000000u;
        }
    };
#define G_DMA_PIECE(DESC, GOFF, P) { const unsigned dst_ = dma_dst + unsigned((((P) + 3) & 3) * G_SLOT);                  \
        lds_dma16(dst_, (GOFF)[0], DESC, unsigned(P) * 64u); lds_dma16(dst_ + 1024u, (GOFF)[1], DESC, unsigned(P) * 64u);   \
        lds_dma16(dst_ + 2048u, (GOFF)[2], DESC, unsigned(P) * 64u); }
#define G_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    // (the library shuffle's lane arithmetic is a loop invariant of the item loop — spilled across the k-loop; this one uses the item's opaque lane id)
    // (a function of a float BY VALUE: __builtin_bit_cast applied to a vector-element lvalue reads element 0 whatever the index)
    auto shfl_xor_f = [](float v, int lane_, int off) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane_ ^ off) << 2, __builtin_bit_cast(int, v)));
    };
#define G_SHFL_XOR(v, off) shfl_xor_f((v), etid & 63, (off))
#define C_LDS4(off) (*static_cast<const f32x4*>(__builtin_assume_aligned(reinterpret_cast<const char*>(smem) + (off), 16)))
#define C_PIN(v) asm volatile("" : "+v"(v))

    // Everything derived from the thread id is worked out per ITEM from an opaque copy (tid_): as loop invariants of the item loop the
    // compiler keeps these values live — i.e. spilled — across the k-loop, and a scratch reload is a vmcnt(0) in the middle of the pipeline.
    const JobRow row0 = decode_job(first_item);
    Item cur = decode_tile(row0);
    i32x4 desc = row_desc(row0, cur);
    unsigned goff[3];
    halo_offsets(cur, row0.h, row0.w, goff);
    bool fresh = true;                            // no piece of `cur` has been requested yet (the block's first item)
    f32x4 ring_next[6];                           // the next item's first weight fragments, requested in the current item's epilogue
#pragma unroll
    for (int s = 0; s < 6; ++s) ring_next[s] = zero4;
    int item = first_item, next_item = first_item;
    for (bool more = first_item < range_end; more; item = next_item) {
        W24G_STAMP(0)
        int k16_total = k16_all;                                            // (opaque per item: conditions on it are recomputed, not kept — spilled — as loop invariants)
        asm volatile("" : "+s"(k16_total));
        int tid = u * 64 + lane_id_now();
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int t16 = lane & 15, g = lane >> 4;                           // tile of the lane, channel quad of the lane
        const int tr = t16 >> 2, tc = t16 & 3;
        // byte addresses (within a slot) of the lane's two patch rows: columns 0..3 at +256 each from b?03, columns 4 / 5 from b?4 / b?5
        const int ax = 2 * tr + xrow, ay = 2 * tr + yrow;
        const int bx03 = (ax * 18 + tc) * 64 + ((g ^ (((ax >> 1) & 1) << 1)) << 4), by03 = (ay * 18 + tc) * 64 + ((g ^ (((ay >> 1) & 1) << 1)) << 4);
        const int s4c = tc < 3 ? tc + 1 : 16, s5c = tc < 3 ? tc + 5 : 17;   // sigma(4 tc + 4), sigma(4 tc + 5)
        const int bx4 = bx03 + (s4c - tc) * 64, bx5 = bx03 + (s5c - tc) * 64, by4 = by03 + (s4c - tc) * 64, by5 = by03 + (s5c - tc) * 64;
        const ConvJob& J = args.job[cur.j];
        const int h = J.h, w = J.w, n32 = cur.n32, b = cur.b, ty0 = cur.ty0, tx0 = cur.tx0, tile_idx = cur.tile_idx, j = cur.j;
        (void)j;
        bool has_next = false;                                              // (known from the last k-step on)
        const float* ub = J.wgt + ((size_t(n32) * k16_total) * 48 + u * 12) * 256;
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, k16_total * 48 * 1024, 0x00020000);
        const int wlane = lane * 16;
        auto wfrag = [&](int step, int s) -> f32x4 {                        // s = 2 * frequency + cout block
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (step * 48 + s) * 1024, 0));
        };
        // ---- the item's first requests.  Block's first item: pieces 0 and 1, then the weight ring.  Later items: the previous item
        // requested them behind its image reads (below) — nothing to do.  Pieces 2 and 3 leave in step 0.
        f32x4 ring[6];
        if (fresh) {
            G_DMA_PIECE(desc, goff, 0) G_DMA_PIECE(desc, goff, 1)
#pragma unroll
            for (int s = 0; s < 6; ++s) { ring[s] = wfrag(0, s); __builtin_amdgcn_sched_barrier(0); }
            G_WAIT_VM(6);                                                   // pieces 0, 1 have landed (behind them: the six fragments)
            __syncthreads();
        } else {
#pragma unroll
            for (int s = 0; s < 6; ++s) ring[s] = ring_next[s];
        }
        W24G_STAMP(1)
        f32x4 acc[6][2];
#pragma unroll
        for (int f = 0; f < 6; ++f)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[f][nb] = zero4;
        f32x4 V[6];
        {   // first operands: piece 0 (slot 3)
            const int o = 3 * G_SLOT;
            f32x4 t[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int ox = c < 4 ? bx03 + c * 256 : (c == 4 ? bx4 : bx5), oy = c < 4 ? by03 + c * 256 : (c == 4 ? by4 : by5);
                const f32x4 x = C_LDS4(ox + o), y = C_LDS4(oy + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[c][e] = fmaf(sgn, y[e], x[e]);
            }
            const f32x4 s1 = t[4] - 4.f * t[2], s2 = t[3] - 4.f * t[1], s3 = t[4] - t[2], s4_ = t[3] - t[1];
            V[0] = 4.f * t[0] + (t[4] - 5.f * t[2]);
            V[1] = s1 + s2; V[2] = s1 - s2;
            V[3] = s3 + 2.f * s4_; V[4] = s3 - 2.f * s4_;
            V[5] = 4.f * t[1] + (t[5] - 5.f * t[3]);
        }
        // One k-step (16 channels) = 12 groups {four MFMAs on one A operand + a piece of the other work}, pinned (as k_conv_wino24s).
#define C_HAS_NEXT 1
#define C_GROUP(F, NB, WORK)                                                                                          \
        {                                                                                                             \
            constexpr int s_ = 2 * (F) + (NB);                                                                        \
            const f32x4 bq = ring[s_ % 6];                                                                            \
            acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][0], bq[0], acc[F][NB], 0, 0, 0);                   \
            if (s_ + 6 < 12) ring[s_ % 6] = wfrag(step, s_ + 6); else if (C_HAS_NEXT) ring[s_ % 6] = wfrag(nstep, s_ - 6); \
            acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][1], bq[1], acc[F][NB], 0, 0, 0);                   \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            WORK                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][2], bq[2], acc[F][NB], 0, 0, 0);                   \
            acc[F][NB] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[F][3], bq[3], acc[F][NB], 0, 0, 0);                   \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }
#define C_COMB(T, X, Y) { _Pragma("unroll") for (int e = 0; e < 4; ++e) T[e] = fmaf(sgn, Y[e], X[e]); C_PIN(T); }
        // the operand-building groups of a step that reads the slot at byte offset `so` (EXTRA: the step's DMA issue, behind the first group)
#define G_BUILD(EXTRA)                                                                                                \
            C_GROUP(0, 0, { const int rx_ = bx03 + so; const int ry_ = by03 + so; cx0 = C_LDS4(rx_); cy0 = C_LDS4(ry_); cx1 = C_LDS4(rx_ + 256); cy1 = C_LDS4(ry_ + 256); }) \
            C_GROUP(0, 1, C_COMB(t0, cx0, cy0) C_COMB(t1, cx1, cy1) EXTRA)                                            \
            C_GROUP(1, 0, { const int rx_ = bx03 + so; const int ry_ = by03 + so; cx0 = C_LDS4(rx_ + 512); cy0 = C_LDS4(ry_ + 512); cx1 = C_LDS4(rx_ + 768); cy1 = C_LDS4(ry_ + 768); }) \
            C_GROUP(1, 1, C_COMB(t2, cx0, cy0) C_COMB(t3, cx1, cy1))                                                  \
            C_GROUP(2, 0, cx0 = C_LDS4(bx4 + so); cy0 = C_LDS4(by4 + so); cx1 = C_LDS4(bx5 + so); cy1 = C_LDS4(by5 + so);) \
            C_GROUP(2, 1, C_COMB(t4, cx0, cy0) C_COMB(t5, cx1, cy1))                                                  \
            /* from here on V[0..2] are free: their MFMAs have been issued */                                         \
            C_GROUP(3, 0, s1 = t4 - 4.f * t2; C_PIN(s1); s2 = t3 - 4.f * t1; C_PIN(s2); V[1] = s1 + s2; C_PIN(V[1]); V[2] = s1 - s2; C_PIN(V[2]);) \
            C_GROUP(3, 1, V[0] = 4.f * t0 + (t4 - 5.f * t2); C_PIN(V[0]); s3 = t4 - t2; C_PIN(s3); s4 = t3 - t1; C_PIN(s4);) \
            C_GROUP(4, 0, V[3] = s3 + 2.f * s4; C_PIN(V[3]);)                                                         \
            C_GROUP(4, 1, v5n = 4.f * t1 + (t5 - 5.f * t3); C_PIN(v5n);)                                              \
            C_GROUP(5, 0, V[4] = s3 - 2.f * s4; C_PIN(V[4]);)                                                         \
            C_GROUP(5, 1, ;)                                                                                          \
            V[5] = v5n;
        //   step d reads piece d + 1 = slot d & 3 and requests piece d + 3 (step 0: pieces 2 and 3); the barrier closes the even steps
#define G_STEP(D, KK)                                                                                                 \
        {                                                                                                             \
            const int step = (D);                                                                                     \
            const int nstep = step + 1;                                                                               \
            const int so = (step & 3) * G_SLOT;                                                                       \
            f32x4 cx0, cy0, cx1, cy1, t0, t1, t2, t3, t4, t5, s1, s2, s3, s4, v5n;                                    \
            G_BUILD(if (step + 3 < k16_total) G_DMA_PIECE(desc, goff, step + 3) if (step == 0) G_DMA_PIECE(desc, goff, 2)) \
            if ((KK) == 0) {                                                                                          \
                G_WAIT_VM(6); __syncthreads();                                                                        \
            }                                                                                                         \
        }
        W24G_STAMP(2)
        __builtin_amdgcn_s_setprio(0);
        for (int d = 0; d < k16_total - 2; d += 2) {
            G_STEP(d, 0)
            G_STEP(d + 1, 1)
        }
        // the last two steps, peeled as in k_conv_wino24s: step n - 2 builds the last operands (no barrier: nothing is published
        // or overwritten before the epilogue's), step n - 1 is MFMAs only
        {
            const int step = k16_total - 2, nstep = step + 1;
            const int so = (step & 3) * G_SLOT;
            f32x4 cx0, cy0, cx1, cy1, t0, t1, t2, t3, t4, t5, s1, s2, s3, s4, v5n;
            G_BUILD(;)
        }
        // ---- epilogue operands (k_conv_wino24s's, verbatim)
        const float* __restrict__ p_bias = J.bias;
        const float* __restrict__ p_bbias = J.bbias;
        const float* __restrict__ p_rcol = J.rcol;
        const float* __restrict__ p_rrow = J.rrow;
        const float* __restrict__ p_res = J.res;
        float* __restrict__ p_out = J.out;
        double* p_gn = J.gn_part;
        int etid = u * 64 + lane_id_now();       // (opaque again: the finishing threads' constants are not kept live across the k-loop either)
        asm volatile("" : "+v"(etid));
        const int quad = etid & 7, xl = (etid >> 3) & 15, rsel = etid >> 7;
        const int co4 = n32 * 32 + quad * 4;
        const bool c_ok = co4 < cout;
        const int coc = c_ok ? co4 : 0;
        const int x = tx0 + xl;
        const bool x_ok = x < w && c_ok;
        const int xc = x < w ? x : 0;
        f32x4 tcol[4], trow[4], tres[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { tcol[k] = zero4; trow[k] = zero4; tres[k] = zero4; }
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
        // the item's last k-step: MFMAs only; the residual request goes out behind its first groups, and the scalar unit works out
        // the next item (its kernel-argument loads and divisions are latency the MFMAs cover)
        Item nxt = cur;
        i32x4 ndesc = desc;
        const float* ubn = ub;                                // the next item's weight image
        int nh = h, nw = w;                                   // ... and plane size
        JobRow nrow = row0;
        {
            const int step = k16_total - 1, nstep = step; (void)nstep;
            C_GROUP(0, 0, ;) C_GROUP(0, 1, request_residual();)
            C_GROUP(1, 0, next_item = item + (persistent ? per : 1); has_next = next_item < range_end;)
            C_GROUP(1, 1, if (has_next) nrow = decode_job(next_item);)
            C_GROUP(2, 0, ;) C_GROUP(2, 1, ;)
            C_GROUP(3, 0, ;) C_GROUP(3, 1, ;)
            C_GROUP(4, 0, if (has_next) { nxt = decode_tile(nrow); ndesc = row_desc(nrow, nxt); nh = nrow.h; nw = nrow.w;
                                          ubn = nrow.wgt + ((size_t(nxt.n32) * k16_total) * 48 + u * 12) * 256; })
            C_GROUP(4, 1, ;) C_GROUP(5, 0, ;) C_GROUP(5, 1, ;)
        }
#undef C_HAS_NEXT
        if (args.xcd_swizzle & 2) __builtin_amdgcn_s_setprio(2);
        W24G_STAMP(3)
        f32x4 base4 = p_bias ? *reinterpret_cast<const f32x4*>(p_bias + coc) : zero4;
        if (p_bbias) base4 += *reinterpret_cast<const f32x4*>(p_bbias + size_t(b) * J.bbias_stride + coc);
#if W24G_GNB
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
        request_row_tables();
        request_col_tables();
#endif
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                     // all patch reads of the item are done: slots 0..2 take the share images
        {
            float* img = smem + u * C_IMG + (etid & 15);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r], m4 = acc[4][nb][r], m5 = acc[5][nb][r];
                    const float p = m1 + m2, q = m1 - m2, rr = m3 + m4, s = m3 - m4;
                    const int pp = (((etid >> 4) & 3) * C_TW + 4 * r) * 32 + nb * 16;
                    img[pp] = (m0 + p) + rr; img[pp + 32] = fmaf(2.f, s, q); img[pp + 64] = fmaf(4.f, rr, p); img[pp + 96] = fmaf(8.f, s, q) + m5;
                }
        }
        // the next item's halo offsets are worked out here, where the accumulators are dead
        unsigned ngoff[3] = {goff[0], goff[1], goff[2]};
        if (has_next) {
            __builtin_amdgcn_sched_barrier(0);
            halo_offsets(nxt, nh, nw, ngoff);
        }
        __syncthreads();                                     // the share images are complete
        W24G_STAMP(4)
        // finishing, part 1: the thread's four output vectors, from the images and the requested operands
        f32x4 vout[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yl = rsel * 4 + k;
            const float* sp = smem + (yl & 1) * C_IMG + ((yl >> 1) * C_TW + xl) * 32 + quad * 4;
            const f32x4 ka = *reinterpret_cast<const f32x4*>(sp), kb = *reinterpret_cast<const f32x4*>(sp + C_IMG),
                        kc = *reinterpret_cast<const f32x4*>(sp + 2 * C_IMG);
            const f32x4 sum3 = (yl & 1) ? (ka - kb) - kc : (ka + kb) + kc;
#if W24G_GNB
            vout[k] = (sum3 + base4) + tres[k];
#else
            vout[k] = (sum3 + base4) + (((tcol[k] + tcol2[k]) + (trow[k] + trow2[k])) + tres[k]);
#endif
            C_PIN(vout[k]);                                   // (complete HERE: sunk below the requests that follow, its operand waits would drain them)
        }
        // ... the images are consumed: the next item's remaining first pieces and weight fragments go out BEFORE this item's stores
        f32x4 ring_nx[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) ring_nx[s] = zero4;
        if (has_next) {
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            // (piece 0 could leave a barrier earlier — slot 3 is free since the patch-read barrier — but the finishing threads' operand
            // waits above are vmcnt(0) to the compiler and would wait for it)
            G_DMA_PIECE(ndesc, ngoff, 0) G_DMA_PIECE(ndesc, ngoff, 1)
            const __amdgpu_buffer_rsrc_t wn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ubn), 0, k16_total * 48 * 1024, 0x00020000);
#pragma unroll
            for (int s = 0; s < 6; ++s) { ring_nx[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wn, wlane, s * 1024, 0)); __builtin_amdgcn_sched_barrier(0); }
        }
        // finishing, part 2: stores and the GroupNorm partial sums
        f32x4 gs4 = zero4, gss4 = zero4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yl = rsel * 4 + k, y = ty0 + yl;
            const f32x4 v = vout[k];
            if (x_ok && y < h) {
                *reinterpret_cast<f32x4*>(p_out + ((size_t(b) * h + y) * w + x) * cout + co4) = v;
#if W24G_GNB
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
        float (*gred)[8][8] = reinterpret_cast<float (*)[8][8]>(smem + G_GRED_FLOAT);
        if (p_gn) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) { gs4[e] += G_SHFL_XOR(gs4[e], off); gss4[e] += G_SHFL_XOR(gss4[e], off); }
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { gred[u][lane][e] = gs4[e]; gred[u][lane][4 + e] = gss4[e]; }
            }
        }
        if (has_next) {
            // the next item's pieces 0 and 1 have landed: behind them in the counter are at least the six fragments (this item's
            // stores are younger still — never waited for here)
            G_WAIT_VM(6);
        }
        if (p_gn || has_next) __syncthreads();
        if (p_gn && u == 0) {                                 // lanes 0..7 of wave 0: channel quad `quad` = lane
            double ds[4], dss[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int l8 = lane & 7;
                ds[e] = ((double(gred[0][l8][e]) + double(gred[1][l8][e])) + double(gred[2][l8][e])) + double(gred[3][l8][e]);
                dss[e] = ((double(gred[0][l8][4 + e]) + double(gred[1][l8][4 + e])) + double(gred[2][l8][4 + e])) + double(gred[3][l8][4 + e]);
            }
            int sg = args.gn_sg;
            asm volatile("" : "+s"(sg));                      // (opaque: its division constants are not loop invariants of the item loop)
            const int part = tile_idx;
            auto shfl_xor_d = [&](double v, int off) -> double {
                const long long b_ = __builtin_bit_cast(long long, v);
                const int a_ = ((etid & 63) ^ off) << 2;
                const unsigned lo = unsigned(__builtin_amdgcn_ds_bpermute(a_, int(unsigned(b_)))), hi = unsigned(__builtin_amdgcn_ds_bpermute(a_, int(unsigned(b_ >> 32))));
                return __builtin_bit_cast(double, (long long)((unsigned long long)hi << 32 | lo));
            };
            auto put = [&](int sub, double sv, double ssv) {
                double* dst = p_gn + ((size_t(b) * 3 * args.gn_nsub + sub) * args.gn_maxparts + part) * 2;
                dst[0] = sv; dst[1] = ssv;
            };
            if (sg >= 4) {
                double sv = (ds[0] + ds[1]) + (ds[2] + ds[3]), ssv = (dss[0] + dss[1]) + (dss[2] + dss[3]);
                for (int off = 1; off < (sg >> 2); off <<= 1) { sv += shfl_xor_d(sv, off); ssv += shfl_xor_d(ssv, off); }
                if (lane < 8 && c_ok && (co4 % sg) == 0) put(co4 / sg, sv, ssv);
            } else if (lane < 8 && c_ok) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    if (sg == 2) put((co4 + e) / 2, ds[e] + ds[e + 1], dss[e] + dss[e + 1]);
                    else { put(co4 + e, ds[e], dss[e]); put(co4 + e + 1, ds[e + 1], dss[e + 1]); }
                }
            }
        }
        W24G_STAMP(5)
#pragma unroll
        for (int s = 0; s < 6; ++s) ring_next[s] = ring_nx[s];
        cur = nxt; desc = ndesc; goff[0] = ngoff[0]; goff[1] = ngoff[1]; goff[2] = ngoff[2];
        fresh = false;
        more = has_next;
    }
#undef G_STEP
#undef G_BUILD
#undef C_GROUP
#undef C_COMB
#undef C_LDS4
#undef C_PIN
#undef G_DMA_PIECE
#undef G_WAIT_VM
#undef G_SHFL_XOR
}
