// Register-pipelined implicit-GEMM kernels for gfx950 (tile_hint 15 .. 18): BK = 32, ONE workgroup barrier per K step and no
// phases.  Configurations (TCO output channels x 16 * NTILES pixel slots, NST LDS stages):
//     15: 256 x 208, 4 stages, 8 waves, one workgroup per CU      -- the MFMA-bound 3x3 layers (tiles of 196 = 49 * 4 pixels)
//     16: 256 x 224, 4 stages, 8 waves                            -- tiles of whole row pairs (2 x 112, 4 x 56, 8 x 28 pixels):
//                                                                    MaxPool2d(2,2) fused into the epilogue (pool2 = 1 / 2)
//     17: 128 x 208, 3 stages, 4 waves, TWO workgroups per CU     -- short-K / few-channel layers (1x1 convs): one workgroup's
//     18: 128 x 224, 3 stages, 4 waves                               epilogue stores run under the other's loads
//
// Why (measured with in-kernel s_memtime stamps on the staggered kernels, tools/stamps_igemm.py): per K step a wave spends
// ~200 cycles issuing its 11 ds_read_b128 (the LDS array serves the four waves of a group at once), 250-390 cycles issuing
// its four global_load_lds and ~450-520 cycles issuing 24-28 MFMAs.  With the K step cut into an L phase and an M phase
// the L phase (reads + DMA, ~560 cycles) is LONGER than the partner wave's M phase (~450), so the matrix pipe idles 40 %
// of the time however the phases are paired.  Here every wave has two register sets of fragments and, in step k,
//     * runs the MFMAs of step k on set k & 1,
//     * reads the fragments of step k+1 into the other set   -- one ds_read behind each of the first MFMAs,
//     * issues its share of the LDS-DMA of stage k+D          -- one global_load_lds every few MFMAs,
// so the LDS / texture-path work of a wave hides under its own and its SIMD partner's MFMAs.  The interleave is pinned
// with sched_group_barrier; the step is one basic block (no branches: the tail steps are separate code).
//
//   top of step k:  s_waitcnt vmcnt(..) -- this wave's pieces of stage k+1 have landed (younger stages may be in flight)
//                   s_barrier           -- everyone's pieces are visible; everyone has issued the MFMAs of step k-1, i.e. has
//                                          finished READING buffer (k-1) % NST, which the DMA of stage k+D overwrites (D = NST-1)
//
// A wave owns 64 channels x 7 (pixel group A) or NTILES - 7 (group B) columns of 16 pixels; with eight waves one wave of
// each group sits on every SIMD.  LDS image: XOR-swizzled 64-B rows, swizzle applied on the DMA's source lane (igemm.hip).
// Not available here: BatchNorm statistics, fp32 atomics (split_k needs slabs), blocked Linear weights -- yolo_igemm routes
// those to the other configurations.
#include "igemm_common.h"

namespace yolo {

template <int TCO_, int NTILES_, int NST_>
struct PipeCfg {
    static constexpr int TCO = TCO_, NTILES = NTILES_, NST = NST_, BK = 32;
    static constexpr int TPX = 16 * NTILES;             // pixel slots per tile (208 / 224)
    static constexpr int WCO = TCO / 64, NW = 2 * WCO, NTHR = NW * 64;
    static constexpr int MT = 4, NT0 = 7, NT1 = NTILES - 7;
    static constexpr int A_BYTES = TCO * BK * 2;
    static constexpr int B_BYTES = 256 * BK * 2;        // 16 KB: TPX rows used, the pad rows are loaded from one line and never read
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_PIECES = A_BYTES / 1024 / NW, B_PIECES = B_BYTES / 1024 / NW;   // LDS-DMA instructions per wave and stage
    static constexpr int LOADS = A_PIECES + B_PIECES;
    static constexpr int EP = TCO + 4;                  // fp32 epilogue row pitch (floats)
    static constexpr int PPX = NT0 * 16;                // pixels per epilogue pass (one pixel group)
    static constexpr int TABLE_BYTES = TPX * 32;
    static constexpr int MAIN_BYTES = NST * STAGE_BYTES > PPX * EP * 4 ? NST * STAGE_BYTES : PPX * EP * 4;
    static constexpr int LDS_BYTES = TABLE_BYTES + MAIN_BYTES;
    static constexpr int D = NST - 1;
    static_assert(NST == 3 || NST == 4, "three or four stages");
    static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "stage must split evenly over the waves");
};

#define GLDS16_S(base, voff, lptr) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)(base) + (unsigned long)(voff)), \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// CODES: the pooled epilogue also writes the arg-max codes (pool2 = 3, 224-pixel tiles only).  A separate instantiation: at
// 246-254 VGPRs any further epilogue code in the plain kernels makes the register allocator spill around the K loop.
template <int TCO, int NTILES, int NST, bool CODES = false>
__global__ void __launch_bounds__(TCO * 2, 2) igemm_pipe_kernel(const IgemmParams p)
{
    using C = PipeCfg<TCO, NTILES, NST>;
    constexpr int TPX = C::TPX, BK = C::BK, NW = C::NW, WCO = C::WCO, NTHR = C::NTHR, MT = C::MT, NT0 = C::NT0, NT1 = C::NT1;
    constexpr int A_BYTES = C::A_BYTES, STAGE_BYTES = C::STAGE_BYTES, LOADS = C::LOADS, EP = C::EP, PPX = C::PPX, D = C::D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long *tab = reinterpret_cast<long *>(smem);
    char *stage_base = smem + C::TABLE_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % WCO, grp = wave / WCO;          // grp 0: pixel columns 0..6, grp 1: 7..NTILES-1
    const int px_lo = grp * NT0 * 16;

#ifdef IGEMM_STAMPS
    long tstamp[5];
    tstamp[0] = __builtin_amdgcn_s_memtime();
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); tstamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
    // start skew (see igemm.hip)
    if (p.skew_phases > 1 && gridDim.x * gridDim.y > 256 && blockIdx.y == 0 && blockIdx.x < 256) {
        const int ph = (blockIdx.x >> 3) % p.skew_phases;
        if (ph) {
            const long t0 = __builtin_amdgcn_s_memtime(), wait = ph * p.skew_cycles;
            while ((long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
        }
    }

    // XCD-aware tile mapping (bijective; see igemm.hip)
    const int nwg = p.n_co_tiles * p.n_px_tiles;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int co_tile = p.px_fastest ? bid / p.n_px_tiles : bid % p.n_co_tiles;
    const int px_tile = p.px_fastest ? bid % p.n_px_tiles : bid / p.n_co_tiles;
    const int co0 = co_tile * TCO;
    const int tpv = p.tpx_valid;
    const long px0 = p.px_begin + (long)px_tile * tpv;

    // per-pixel address table: input row base / output base / aux base (elements), one pixel slot per thread.
    // Pooled epilogue (p.pool; 224-pixel tiles of whole row pairs): every 2x2 window must lie inside ONE pixel group
    // (= one epilogue slab of 112 slots).  Rows of 56 or 28 pixels do that in flat order; for rows of 112 pixels the slots of
    // a group are columns [56 g, 56 g + 56) of BOTH rows of the tile.  out_base then addresses the POOLED map, and table
    // entry 2 the un-pooled activation (pool2 = 2).
    const int pool_wl = p.pool ? (p.Wo >= 56 ? 56 : p.Wo) : 0;     // row length inside a group's slab
    if (tid < TPX) {
        int slot = tid;
        if (p.pool && p.Wo == 112) {
            const int g = tid / 112, u = tid - g * 112;
            slot = (u / 56) * 112 + g * 56 + u % 56;
        }
        long m = px0 + slot;
        const bool valid = m < p.M && tid < tpv;
        if (!valid) m = p.M - 1;
        // 32-bit divisions (the host refuses M >= 2^31): a 64-bit division is emulated in ~1 k cycles, and this table is on
        // the critical path of every tile (stamps: 2.2 k cycles with the 64-bit form)
        const unsigned mu = (unsigned)m;
        const int n = (int)(mu / (unsigned)p.HoWo);
        const int rem = (int)(mu - (unsigned)n * (unsigned)p.HoWo);
        const int oy = (int)((unsigned)rem / (unsigned)p.Wo), ox = rem - oy * p.Wo;
        tab[4 * tid] = (long)n * p.in_img_stride + (long)(oy * p.stride) * p.in_row_stride + (long)(ox * p.stride) * p.in_px_stride + p.in_off;
        const int qy = p.pool ? oy >> 1 : oy, qx = p.pool ? ox >> 1 : ox;
        tab[4 * tid + 1] = valid ? ((long)n * p.out_img_stride + (long)qy * p.out_row_stride + (long)qx * p.out_px_stride + p.out_off) : -1;
        tab[4 * tid + 2] = (long)n * p.aux_img_stride + (long)oy * p.aux_row_stride + (long)ox * p.aux_px_stride + p.aux_off;
    }
    __syncthreads();

    PSTAMP(1);
    // LDS-DMA pieces of this wave (weight pieces q = wave, wave + NW, ..; pixel pieces likewise).  Lane -> (row, 16-B chunk)
    // through the inverse swizzle (the DMA writes lane-linear), as a 32-bit BYTE offset from the operand's base pointer.
    unsigned a_voff[C::A_PIECES], b_voff[C::B_PIECES];
#pragma unroll
    for (int i = 0; i < C::A_PIECES; ++i) {
        const int pos = (i * NW + wave) * 64 + lane;
        const int R = pos >> 4, s = (pos & 15) ^ swz_key<BK, true>(R);
        const int r = R * 4 + s / 4, chunk = s % 4;
        int co = co0 + r;
        if (co >= p.Cout) co = p.Cout - 1;
        a_voff[i] = (unsigned)(((long)co * p.Ktot + chunk * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < C::B_PIECES; ++i) {
        const int pos = (i * NW + wave) * 64 + lane;
        const int R = pos >> 4, s = (pos & 15) ^ swz_key<BK, true>(R);
        const int r = R * 4 + s / 4, chunk = s % 4;
        b_voff[i] = (unsigned)((r < TPX ? tab[4 * r] + chunk * 8 : tab[0]) * 2);
    }

    // K range of this split
    const int kbeg = blockIdx.y * p.nk_per_split;
    const int kend = min(p.nk, kbeg + p.nk_per_split);
    const int nkk = kend - kbeg;
    const int cpt = p.tap_len / BK;
    // scalar staging state: byte offset of the next stage's weight columns, and (ky, kx, c0) of its tap
    unsigned a_soff = (unsigned)kbeg * (BK * 2);
    int tap = kbeg / cpt;
    int c0 = (kbeg - tap * cpt) * BK;
    int ky = tap / p.KW, kx = tap - ky * p.KW;

    auto stage = [&](int buf) {
        char *sb = stage_base + buf * STAGE_BYTES + wave * 1024;
        const unsigned b_soff = (unsigned)((ky * p.in_row_stride + kx * p.in_px_stride + c0) * 2);
        const char *wb = reinterpret_cast<const char *>(p.w) + a_soff;
        const char *xb = reinterpret_cast<const char *>(p.in) + b_soff;
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i) GLDS16_S(wb, a_voff[i], sb + i * NW * 1024);
#pragma unroll
        for (int i = 0; i < C::B_PIECES; ++i) GLDS16_S(xb, b_voff[i], sb + A_BYTES + i * NW * 1024);
        // advance to the next K step without branches (the step must stay one basic block)
        a_soff += BK * 2;
        c0 += BK;
        const int w0 = c0 == p.tap_len;
        c0 = w0 ? 0 : c0;
        kx += w0;
        const int w1 = kx == p.KW;
        kx = w1 ? 0 : kx;
        ky += w1;
    };

    // fragment read offsets (bytes inside a stage): lane l reads row (l & 15), chunk (l >> 4) of each 16-row MFMA operand
    int a_rd[MT], b_rd[NT0];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_rd[i] = lds_off<BK, true>(wco * 64 + i * 16 + (lane & 15), lane >> 4);
#pragma unroll
    for (int j = 0; j < NT0; ++j) {
        int row = px_lo + j * 16 + (lane & 15);
        if (row >= TPX) row = TPX - 1;                  // a column group B does not have (never read)
        b_rd[j] = A_BYTES + lds_off<BK, true>(row, lane >> 4);
    }

    f32x4 acc[MT][NT0];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT0; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---- prologue: stages 0 .. D-1 in flight, stage 0 landed and visible
#pragma unroll
    for (int s0 = 0; s0 < D; ++s0)
        if (s0 < nkk) stage(s0);
    if (nkk >= D) wait_vmcnt<(D - 1) * LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PSTAMP(2);

    auto run = [&](auto ntc) {
        constexpr int NTG = decltype(ntc)::value;
        bf16x8 a0[MT], b0[NT0], a1[MT], b1[NT0];
        auto rd = [&](int buf, bf16x8(&af)[MT], bf16x8(&bfr)[NT0]) {
            const char *sb = stage_base + buf * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(sb + a_rd[i]);
#pragma unroll
            for (int j = 0; j < NTG; ++j) bfr[j] = *reinterpret_cast<const bf16x8 *>(sb + b_rd[j]);
        };
        auto mm = [&](bf16x8(&af)[MT], bf16x8(&bfr)[NT0]) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTG; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        };
        constexpr int NRD = MT + NTG, NMF = MT * NTG;
        // steady-state step: reads of stage it+1, DMA of stage it+D, MFMAs of step it -- one basic block
        auto step = [&](int rbuf, int lbuf, bf16x8(&ca)[MT], bf16x8(&cb)[NT0], bf16x8(&na)[MT], bf16x8(&nb)[NT0]) {
            wait_vmcnt<(D - 2) * LOADS>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            rd(rbuf, na, nb);
            stage(lbuf);
            mm(ca, cb);
            // one ds_read behind each of the first NRD MFMAs ...
#pragma unroll
            for (int k = 0; k < NRD; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // ... then one LDS-DMA every PER MFMAs
            constexpr int PER = (NMF - NRD) / (LOADS + 1);
#pragma unroll
            for (int k = 0; k < LOADS; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD - LOADS * PER, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        // the last D steps stage nothing.  TOP >= 0: TOP younger stages than k+1 are still in flight (counted wait);
        // TOP < 0: final step, nothing to read, no barrier
        auto tail = [&](auto topc, int rbuf, bf16x8(&ca)[MT], bf16x8(&cb)[NT0], bf16x8(&na)[MT], bf16x8(&nb)[NT0]) {
            constexpr int TOP = decltype(topc)::value;
            if constexpr (TOP >= 0) {
                wait_vmcnt<TOP * LOADS>();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                rd(rbuf, na, nb);
            }
            mm(ca, cb);
            if constexpr (TOP >= 0) {
#pragma unroll
                for (int k = 0; k < NRD; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // nkk is even and >= 4 (checked by the host): the steps that stage something run in pairs (register sets 0 -> 1 -> 0), the
        // register sets and the tail are static -- no run-time parity, no merging of the two fragment sets
        rd(0, a0, b0);
        int rbuf = 1, lbuf = D;
        auto adv = [&]() {
            rbuf = rbuf + 1 == NST ? 0 : rbuf + 1;
            lbuf = lbuf + 1 == NST ? 0 : lbuf + 1;
        };
        if constexpr (D == 3) {
            for (int it = 0; it + 4 < nkk; it += 2) {
                step(rbuf, lbuf, a0, b0, a1, b1);
                adv();
                step(rbuf, lbuf, a1, b1, a0, b0);
                adv();
            }
            step(rbuf, lbuf, a0, b0, a1, b1);                                       // step nkk-4: stages the last stage
            adv();
            tail(std::integral_constant<int, 1>{}, rbuf, a1, b1, a0, b0);           // step nkk-3
            adv();
            tail(std::integral_constant<int, 0>{}, rbuf, a0, b0, a1, b1);           // step nkk-2
            adv();
            tail(std::integral_constant<int, -1>{}, rbuf, a1, b1, a0, b0);          // step nkk-1
        } else {
            for (int it = 0; it + 2 < nkk; it += 2) {                               // steps 0 .. nkk-3 stage something
                step(rbuf, lbuf, a0, b0, a1, b1);
                adv();
                step(rbuf, lbuf, a1, b1, a0, b0);
                adv();
            }
            tail(std::integral_constant<int, 0>{}, rbuf, a0, b0, a1, b1);           // step nkk-2
            adv();
            tail(std::integral_constant<int, -1>{}, rbuf, a1, b1, a0, b0);          // step nkk-1
        }
    };
    if (grp == 0) run(std::integral_constant<int, NT0>{});
    else run(std::integral_constant<int, NT1>{});

    wait_vmcnt<0>();
    __syncthreads();  // all MFMA operand reads done before the stage area is reused for the epilogue
    PSTAMP(3);

    // ---- epilogue: two passes -- group A's 112 pixel slots, group B's 16 * NT1 -- through an fp32 slab [px][co] in LDS, then
    // 16-B coalesced stores along the channel axis with the layer's epilogue applied
    float *ep = reinterpret_cast<float *>(stage_base);
    constexpr int CCH = TCO / 8;
    constexpr int PX_PER_STEP = NTHR / CCH;
    const int cc = tid % CCH;
    const int co = co0 + cc * 8;
    const bool has_bias = p.epilogue == YOLO_EPI_BIAS || p.epilogue == YOLO_EPI_BIAS_LRELU || p.epilogue == YOLO_EPI_BIAS_ADD_LRELU;
    float bias8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bias8[k] = has_bias && (co + k < p.Cout) ? p.bias[co + k] : 0.0f;
    // 0: generic loop; 1 / 2: the straight-line fast paths below (bf16 output, no pool, no slabs, channel tile inside Cout)
    const int fast_kind = (!p.out_fp32 && !p.pool && !p.slab_stride && co0 + TCO <= p.Cout && p.slope >= 0.0f && p.slope <= 1.0f)
                              ? (p.epilogue == YOLO_EPI_BIAS_LRELU ? 1 : (p.epilogue == YOLO_EPI_MUL_DLRELU ? 2 : 0)) : 0;
    void *const outp = p.slab_stride ? (void *)(reinterpret_cast<float *>(p.out) + (long)blockIdx.y * p.slab_stride) : p.out;

    auto pack8 = [](const float (&v)[8]) {
        uint4 pk;
        pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        pk.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
        pk.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        return pk;
    };

    for (int q = 0; q < 2; ++q) {
        if (q > 0) __syncthreads();
        if (grp == q) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT0; ++j) {
                    const int lp = j * 16 + (lane & 15);
                    const int cob = wco * 64 + i * 16 + 4 * (lane >> 4);
                    *reinterpret_cast<f32x4 *>(ep + lp * EP + cob) = acc[i][j];
                }
        }
        __syncthreads();
        const int pbase = q * PPX;
        const int ppx_q = q == 1 ? NT1 * 16 : PPX;
        if (p.pool) {
            // fused MaxPool2d(2,2): max over the window's four LDS rows, then bias + LeakyReLU (monotone, so
            // pool(lrelu(z + b)) == lrelu(max(z) + b)); one 16-B store per pooled pixel and channel chunk
            const int wl = pool_wl, hw = wl >> 1;
            for (int w = tid / CCH; w < ppx_q / 4; w += PX_PER_STEP) {
                const int l00 = (w / hw) * 2 * wl + (w % hw) * 2;
                const long ob = tab[4 * (pbase + l00) + 1];
                if (ob < 0 || co >= p.Cout) continue;
                float v[8];
                if (CODES && p.pool == 3) {
                    // pool2 = 3 (training): besides the pooled map, the window position of every maximum (2 bits per channel, 8
                    // channels = one ushort) -- all the backward pass needs of the un-pooled activation.  The comparison runs on
                    // the activations AS STORED (bf16), first maximum in (0,0),(0,1),(1,0),(1,1) order: exactly what
                    // yolo_maxpool2_bwd_lrelu derives from the un-pooled tensor.
                    unsigned code = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float *zp = ep + l00 * EP + cc * 8 + k;
                        float m = 0.0f;
                        unsigned am = 0;
#pragma unroll
                        for (int w4 = 0; w4 < 4; ++w4) {
                            float t = zp[((w4 >> 1) * wl + (w4 & 1)) * EP] + bias8[k];
                            t = (p.epilogue == YOLO_EPI_BIAS_LRELU && t < 0.0f) ? t * p.slope : t;
                            t = __uint_as_float((unsigned)f32_to_bf16(t) << 16);
                            if (w4 == 0 || t > m) { m = t; am = w4; }
                        }
                        v[k] = m;
                        code |= am << (2 * k);
                        __builtin_amdgcn_sched_barrier(0);     // one channel at a time: the other pixel group's accumulators are still live
                    }
                    reinterpret_cast<unsigned short *>(const_cast<bf16_t *>(p.aux))[(ob + co) >> 3] = (unsigned short)code;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float a = ep[l00 * EP + cc * 8 + k], b = ep[(l00 + 1) * EP + cc * 8 + k];
                        const float c = ep[(l00 + wl) * EP + cc * 8 + k], d = ep[(l00 + wl + 1) * EP + cc * 8 + k];
                        const float m = fmaxf(fmaxf(a, b), fmaxf(c, d)) + bias8[k];
                        v[k] = (p.epilogue == YOLO_EPI_BIAS_LRELU && m < 0.0f) ? m * p.slope : m;
                    }
                }
                *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(p.out) + ob + co) = pack8(v);
            }
            if (p.pool == 2) {
                // pool2 = 2 (training): the un-pooled activation is written as well, through the aux pointer / strides
                bf16_t *full = const_cast<bf16_t *>(p.aux);
                for (int lp = tid / CCH; lp < ppx_q; lp += PX_PER_STEP) {
                    const int px = pbase + lp;
                    if (tab[4 * px + 1] < 0 || co >= p.Cout) continue;
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float m = ep[lp * EP + cc * 8 + k] + bias8[k];
                        v[k] = (p.epilogue == YOLO_EPI_BIAS_LRELU && m < 0.0f) ? m * p.slope : m;
                    }
                    *reinterpret_cast<uint4 *>(full + tab[4 * px + 2] + co) = pack8(v);
                }
            }
            continue;
        }
        if (fast_kind) {
            // the two common epilogues -- bias + LeakyReLU (forward) and x LeakyReLU'(aux) (data gradient), bf16 out, whole channel
            // tile inside Cout -- as straight-line code: all LDS reads (and the aux loads) of the pass first, then arithmetic and
            // stores.  The generic loop below takes ~15 k cycles per tile (in-kernel stamps): it walks a chain of run-time branches
            // per pixel and exposes the LDS latency seven times per pass.
            constexpr int NI = PPX / PX_PER_STEP;           // 7 pixels per thread and pass
            auto fast = [&](auto kindc) {
                constexpr int KIND = decltype(kindc)::value;    // 1: bias + LeakyReLU, 2: multiply by LeakyReLU'(aux)
                const int lp0 = tid / CCH;
#pragma unroll
                for (int half = 0; half < 2; ++half) {          // two batches (4 + 3 pixels): the other group's accumulators are still live
                    constexpr int NB = 4;
                    long ob[NB];
                    f32x4 lo[NB], hi[NB];
                    uint4 ax[NB];
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        const int it = half * NB + i;
                        const int lp = lp0 + it * PX_PER_STEP;
                        const bool in = it < NI && lp < ppx_q;
                        const int px = pbase + (in ? lp : 0);
                        ob[i] = in ? tab[4 * px + 1] : -1;
                        lo[i] = *reinterpret_cast<const f32x4 *>(ep + (in ? lp : 0) * EP + cc * 8);
                        hi[i] = *reinterpret_cast<const f32x4 *>(ep + (in ? lp : 0) * EP + cc * 8 + 4);
                        if constexpr (KIND == 2) {
                            if (ob[i] >= 0) ax[i] = *reinterpret_cast<const uint4 *>(p.aux + tab[4 * px + 2] + co);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        if (ob[i] < 0) continue;
                        float v[8] = {lo[i][0], lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1], hi[i][2], hi[i][3]};
                        if constexpr (KIND == 1) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float t = v[k] + bias8[k];
                                v[k] = fmaxf(t, t * p.slope);           // LeakyReLU for 0 <= slope <= 1
                            }
                        } else {
                            const unsigned yy[4] = {ax[i].x, ax[i].y, ax[i].z, ax[i].w};
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float a = __uint_as_float((k & 1) ? (yy[k >> 1] & 0xffff0000u) : (yy[k >> 1] << 16));
                                v[k] = a > 0.0f ? v[k] : v[k] * p.slope;
                            }
                        }
                        *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(p.out) + ob[i] + co) = pack8(v);
                    }
                }
            };
            if (fast_kind == 1) fast(std::integral_constant<int, 1>{});
            else fast(std::integral_constant<int, 2>{});
            continue;
        }
#pragma unroll 2
        for (int lp = tid / CCH; lp < ppx_q; lp += PX_PER_STEP) {
            const int px = pbase + lp;
            const long ob = tab[4 * px + 1];
            if (ob < 0 || co >= p.Cout) continue;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(ep + lp * EP + cc * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(ep + lp * EP + cc * 8 + 4);
            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
            v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
            if (has_bias) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += bias8[k];
            }
            if (p.epilogue == YOLO_EPI_BIAS_ADD_LRELU) {
                const uint4 y = *reinterpret_cast<const uint4 *>(p.aux + tab[4 * px + 2] + co);
                const unsigned yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[k] += __uint_as_float((k & 1) ? (yy[k >> 1] & 0xffff0000u) : (yy[k >> 1] << 16));
                    v[k] = v[k] > 0.0f ? v[k] : v[k] * p.slope;
                }
            } else if (p.epilogue == YOLO_EPI_BIAS_LRELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.0f ? v[k] : v[k] * p.slope;
            } else if (p.epilogue == YOLO_EPI_MUL_DLRELU) {
                const uint4 y = *reinterpret_cast<const uint4 *>(p.aux + tab[4 * px + 2] + co);
                const unsigned yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float a = __uint_as_float((k & 1) ? (yy[k >> 1] & 0xffff0000u) : (yy[k >> 1] << 16));
                    v[k] = a > 0.0f ? v[k] : v[k] * p.slope;
                }
            }
            if (p.out_fp32) {
                float *o = reinterpret_cast<float *>(outp) + ob + co;
                if (co + 8 <= p.Cout && ((ob + co) & 3) == 0) {
                    *reinterpret_cast<f32x4 *>(o) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4 *>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (co + k < p.Cout) o[k] = v[k];
                }
            } else {
                *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(p.out) + ob + co) = pack8(v);
            }
        }
    }
#ifdef IGEMM_STAMPS
    // diagnostic build: kernel start | table built | stage 0 visible | K loop done | end, per wave of the first 512 workgroups
    if (p.dbg && lane == 0 && blockIdx.x < 512 && blockIdx.y == 0) {
        tstamp[4] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 5; ++i) p.dbg[((long)blockIdx.x * 8 + wave) * 8 + i] = tstamp[i];
    }
#endif
#undef PSTAMP
}

template <int TCO, int NTILES, int NST, bool CODES = false>
static int pipe_launch(const IgemmParams &p, int splits, hipStream_t s)
{
    using C = PipeCfg<TCO, NTILES, NST>;
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm_pipe_kernel<TCO, NTILES, NST, CODES>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return fail((int)e, "yolo_igemm: hipFuncSetAttribute(%d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
        attr_done[dev] = true;
    }
    IgemmParams q = p;
    q.n_co_tiles = (p.Cout + TCO - 1) / TCO;
    if (p.pool) q.tpx_valid = C::TPX;
    if (q.tpx_valid <= 0 || q.tpx_valid > C::TPX) q.tpx_valid = C::TPX;
    q.n_px_tiles = (int)((p.M - p.px_begin + q.tpx_valid - 1) / q.tpx_valid);
    q.nk = (int)(p.Ktot / C::BK);
    if (p.px_fastest < 0) q.px_fastest = 0;
    q.nk_per_split = (q.nk + splits - 1) / splits;
    const int real_splits = p.slab_stride ? splits : 1;
    hipLaunchKernelGGL((igemm_pipe_kernel<TCO, NTILES, NST, CODES>), dim3(q.n_co_tiles * q.n_px_tiles, real_splits), dim3(C::NTHR), C::LDS_BYTES, s, q);
    return check_launch("yolo_igemm (pipelined)");
}

int igemm_pipe_launch(const IgemmParams &p, int hint, int splits, hipStream_t s)
{
    if (p.stats || p.w_blocked) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d has no BatchNorm statistics epilogue and no blocked weights", hint);
    if (splits > 1 && !p.slab_stride) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d splits K into slabs only (split_slabs = 1)", hint);
    if (p.tap_len % 32) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d needs tap_len %% 32 == 0", hint);
    // the loop runs its K steps in pairs with a static tail: every split needs an even number (>= 4) of K steps
    const long nk_all = p.Ktot / 32;
    if (nk_all % (2 * splits) || nk_all / splits < 4) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d needs an even number (>= 4) of 32-deep K steps per split", hint);
    // the DMA sources are addressed as base + 32-bit byte offset
    const long in_bytes = ((p.M / p.HoWo + 1) * p.in_img_stride + (long)p.KH * p.in_row_stride) * 2, w_bytes = (long)p.Cout * p.Ktot * 2;
    if (p.M >= (1L << 31)) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d indexes fewer than 2^31 output pixels", hint);
    if (in_bytes >= (1L << 32) || w_bytes >= (1L << 32)) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d addresses operands below 4 GB", hint);
    const bool px224 = hint == 16 || hint == 18;
    if (p.pool) {
        // whole row pairs per 112-slot pixel group: rows of 112 (two half rows), 56 or 28 pixels, and tiles that start on an even row
        if (!px224 || !(p.Wo == 112 || p.Wo == 56 || p.Wo == 28) || (p.HoWo / p.Wo) % 2 || p.HoWo % 112 || p.px_begin)
            return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: the pooled epilogue of tile_hint 16 / 18 needs rows of 112, 56 or 28 pixels");
    }
    switch (hint) {
    case 15: return pipe_launch<256, 13, 4>(p, splits, s);
    case 16: return p.pool == 3 ? pipe_launch<256, 14, 4, true>(p, splits, s) : pipe_launch<256, 14, 4>(p, splits, s);
    case 17: return pipe_launch<128, 13, 3>(p, splits, s);
    case 18: return p.pool == 3 ? pipe_launch<128, 14, 3, true>(p, splits, s) : pipe_launch<128, 14, 3>(p, splits, s);
    }
    return fail(YOLO_E_ARG, "yolo_igemm: tile_hint %d is not a pipelined configuration", hint);
}

}  // namespace yolo
