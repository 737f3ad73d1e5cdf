// Implicit-GEMM convolution / Linear kernel for gfx950 (CDNA4): bf16 operands, fp32 MFMA accumulate.
//
//   out[px][co] = epi( sum_{tap, c} in[row(px) + tapoff(tap) + c] * w[co][tap][c] )
//
// im2col-free: the "B" operand rows are gathered straight from the zero-haloed NHWC activation by
// per-lane source addresses of `global_load_lds` (LDS-DMA, 16 B per lane), so a 3x3/pad-1 conv needs
// neither an unfolded buffer nor bounds checks.  The same kernel runs
//   * Conv2d + LeakyReLU(0.1) forward (src/yolo/models.py:47-84, :313-322 of the reference),
//   * the conv data-gradient (weights pre-packed transposed + flipped, epilogue multiplies by the
//     previous LeakyReLU's derivative),
//   * Linear forward / data-gradient of the FC head (models.py:239-245) as a 1x1 conv on a 1x1 image.
//
// Structure (per workgroup of WCO x WPX wave64):
//   tile  TCO output channels x TPX output pixels, K step BK (one tap, BK channels)
//   LDS   NST-stage ring x (TCO + TPX) rows x BK bf16, 16-B slots XOR-swizzled inside each 256-B bank
//         row (conflict-free ds_read_b128 for the 32x32x16 operand maps); the swizzle is applied on the
//         per-lane SOURCE address because an LDS-DMA writes lane-linear.  Loads run NST-1 stages ahead
//         and stay in flight ACROSS the per-step barrier (counted s_waitcnt vmcnt(N) + raw s_barrier).
//   MFMA  v_mfma_f32_32x32x16_bf16, A = weights (rows = co), B = activations (cols = px); each
//         wave owns (TCO/WCO) x (TPX/WPX)
//   configs: 128x128 / 64x128 / 128x64 / 64x64 tiles with 4 waves and 2 stages (2 workgroups per CU),
//         256x128 with 8 waves and 3 stages (1 workgroup per CU, 85 flop/B from L2) for the big layers
//   epilogue  accumulators -> LDS fp32 [px][co] -> bias / LeakyReLU / dLeakyReLU -> 16-B coalesced
//         bf16 stores along the channel axis (full 128/256-B lines per pixel)
//   split-K over blockIdx.y with fp32 atomics (Linear layers: M = batch is tiny, K = 50176)
//   block ids are remapped so that each XCD (private L2) works on a contiguous range of tiles.
#include "igemm_common.h"

namespace yolo {

template <int TCO, int TPX, int BK, int WCO, int WPX, int NST, int MF = MFMA_32x32x16>
struct IgemmCfg {
    static constexpr int NW = WCO * WPX;                     // waves per workgroup
    static constexpr int NTHR = NW * 64;
    static constexpr bool UNEVEN = MF == MFMA_16x16x32_STAGGER_U;
    static constexpr int A_BYTES = TCO * BK * 2;
    // the B stage is rounded up to whole 1-KB pieces per wave (TPX = 208: 13 KB -> 16 KB at BK = 32); the pad rows are
    // loaded from one fixed address and never read
    static constexpr int B_BYTES = (TPX * BK * 2 + 1024 * NW - 1) / (1024 * NW) * (1024 * NW);
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int EP = TCO + 4;                       // fp32 epilogue row pitch (floats)
    static constexpr int NTILES = TPX / 16;                  // 16-pixel MFMA columns of the tile (uneven split)
    static constexpr int NT0 = (NTILES + 1) / 2, NT1 = NTILES / 2;
    static constexpr int PPX = UNEVEN ? NT0 * 16 : TPX / WPX;   // pixels per epilogue pass (one wave column)
    static constexpr int EPI_BYTES = PPX * EP * 4;
    static constexpr int TABLE_BYTES = TPX * 32;             // per pixel: in_base, out_base, aux_base (int64 each, padded to 4)
    static constexpr int RED_BYTES = WCO * WPX * 64 * 16 * 4;   // per-thread BatchNorm partial sums (bn_stats), behind the epilogue slab
    static constexpr int MAIN_BYTES = (NST * STAGE_BYTES > EPI_BYTES + RED_BYTES) ? NST * STAGE_BYTES : EPI_BYTES + RED_BYTES;
    static constexpr int LDS_BYTES = TABLE_BYTES + MAIN_BYTES;
    static constexpr int A_INSTR = A_BYTES / 1024 / NW;      // glds wave-instructions per wave per stage
    static constexpr int B_INSTR = B_BYTES / 1024 / NW;
    static constexpr int LOADS = A_INSTR + B_INSTR;
    static constexpr int FR = MF != MFMA_32x32x16 ? 16 : 32;        // MFMA tile edge
    static constexpr int KS = MF != MFMA_32x32x16 ? 32 : 16;        // K per MFMA
    static constexpr int MT = TCO / WCO / FR, NT = UNEVEN ? NT0 : TPX / WPX / FR;  // MFMA tiles per wave (uneven: of group A)
    static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "stage must split evenly over the waves");
    static_assert(!UNEVEN || (WPX == 2 && TPX % 16 == 0), "uneven split: two pixel groups of 16-pixel columns");
    static_assert(UNEVEN || TPX * BK * 2 == B_BYTES, "only the uneven configuration pads its B stage");
    static_assert(TPX <= NTHR, "one table entry per thread");
};

// VAR 1 (STATS): the epilogue also accumulates BatchNorm's per-channel sums (yolo_igemm_desc.bn_stats); VAR 2 (CODES): the pooled
// epilogue also writes the arg-max codes (pool2 = 3).  Separate instantiations, so that the kernels of the plain path keep their
// exact code and register count: given any more epilogue code the compiler spends up to 256 VGPRs on it and the LDS-light
// configurations of the HBM-bound layers drop from five co-resident workgroups per CU to two.
template <int TCO, int TPX, int BK, int WCO, int WPX, int NST, int MF, int VAR>
__global__ void __launch_bounds__(WCO * WPX * 64, (TCO == 64 && TPX == 128 && BK == 32 && VAR == 0 ? 5 : 2)) igemm_kernel(const IgemmParams p)
{
    using Cfg = IgemmCfg<TCO, TPX, BK, WCO, WPX, NST, MF>;
    constexpr bool STATS = VAR == 1, CODES = VAR == 2;
    constexpr bool M16 = MF != MFMA_32x32x16;
    constexpr bool UNEVEN = Cfg::UNEVEN;
    constexpr bool STG = MF == MFMA_16x16x32_STAGGER || UNEVEN;
    constexpr int FR = Cfg::FR;
    constexpr int MT = Cfg::MT, NT = Cfg::NT, NW = Cfg::NW, NTHR = Cfg::NTHR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long *tab = reinterpret_cast<long *>(smem);
    char *stage_base = smem + Cfg::TABLE_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // uneven split: the pixel group IS the stagger group (waves 0..3 = A = pixel columns [0, NT0), waves 4..7 = B)
    const int wco = UNEVEN ? wave % WCO : wave / WPX, wpx = UNEVEN ? wave / WCO : wave % WPX;
    const int px_lo = UNEVEN ? wpx * Cfg::NT0 * 16 : wpx * (TPX / WPX);   // first tile pixel of this wave's column

    // ---- start skew (8-wave configurations: one workgroup per CU).  With equal tiles every CU reaches its prologue
    // burst and its output stores at the same moment and the memory system serves 256 identical phases at once.
    // Delaying the first-round workgroups of phase 1.. by a fraction of a tile time keeps the CUs out of step for the
    // rest of the launch (later workgroups inherit the CU's offset); measured 2-8 % on the multi-round layers.
    if (NW == 8 && p.skew_phases > 1 && gridDim.x * gridDim.y > 256 && blockIdx.y == 0 && blockIdx.x < 256) {
        const int ph = (blockIdx.x >> 3) % p.skew_phases;
        if (ph) {
            const long t0 = __builtin_amdgcn_s_memtime(), wait = ph * p.skew_cycles;
            while ((long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
        }
    }

    // ---- XCD-aware tile mapping: blocks b, b+8, b+16.. share an XCD; give each XCD a contiguous
    // range of logical tiles (co-tile fastest) so its private L2 sees one activation tile being
    // reused across the co-tiles and a narrow band of weight panels.  Bijective for any grid size.
    const int nwg = p.n_co_tiles * p.n_px_tiles;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int co_tile = p.px_fastest ? bid / p.n_px_tiles : bid % p.n_co_tiles;
    const int px_tile = p.px_fastest ? bid % p.n_px_tiles : bid / p.n_co_tiles;
    const int co0 = co_tile * TCO;
    const int tpv = p.tpx_valid;           // pixels per tile (TPX unless the launch asked for fewer: tile_px)
    const long px0 = p.px_begin + (long)px_tile * tpv;

    // ---- per-pixel address tables (input row base / output base), one pixel per thread.
    // Normal mode: tile = TPX consecutive pixels of the flattened (n, oy, ox) index.
    // Pool mode:   tile = a (TPX/TW) x TW patch of one image, so that every 2x2 pooling window lies
    //              inside one tile; out_base then addresses the POOLED map.
    if (tid < TPX) {
        int n, oy, ox;
        bool valid;
        if (p.pool) {
            const int per_img = p.pool_tiles_x * p.pool_tiles_y;
            n = px_tile / per_img;
            const int t = px_tile - n * per_img;
            const int ty = t / p.pool_tiles_x, tx = t - ty * p.pool_tiles_x;
            oy = ty * (TPX / p.pool_tw) + tid / p.pool_tw;
            ox = tx * p.pool_tw + tid % p.pool_tw;
            valid = oy < p.HoWo / p.Wo && ox < p.Wo;
            if (!valid) { oy = 0; ox = 0; }
        } else {
            long m = px0 + tid;
            valid = m < p.M && tid < tpv;
            if (!valid) m = p.M - 1;
            n = (int)(m / p.HoWo);
            const int rem = (int)(m - (long)n * p.HoWo);
            oy = rem / p.Wo;
            ox = rem - oy * p.Wo;
        }
        tab[4 * tid] = (long)n * p.in_img_stride + (long)(oy * p.stride) * p.in_row_stride + (long)(ox * p.stride) * p.in_px_stride + p.in_off;
        const int qy = p.pool ? oy >> 1 : oy, qx = p.pool ? ox >> 1 : ox;
        tab[4 * tid + 1] = valid ? ((long)n * p.out_img_stride + (long)qy * p.out_row_stride + (long)qx * p.out_px_stride + p.out_off) : -1;
        tab[4 * tid + 2] = (long)n * p.aux_img_stride + (long)oy * p.aux_row_stride + (long)ox * p.aux_px_stride + p.aux_off;
    }
    __syncthreads();

    // ---- LDS-DMA source pointers (fixed rows/chunks per lane; only the K offset moves)
    const bf16_t *a_src[Cfg::A_INSTR];
    const bf16_t *b_src[Cfg::B_INSTR];
    int a_dst[Cfg::A_INSTR], b_dst[Cfg::B_INSTR];
    {
        constexpr int CPR = BK / 8, RPB = 16 / CPR;
#pragma unroll
        for (int i = 0; i < Cfg::A_INSTR; ++i) {
            const int q = i * NW + wave;            // wave-instruction index inside the A tile
            const int pos = q * 64 + lane;          // 16-B slot written by this lane
            const int R = pos >> 4, s = (pos & 15) ^ swz_key<BK, M16>(R);
            const int r = R * RPB + s / CPR, chunk = s % CPR;
            int co = co0 + r;
            if (co >= p.Cout) co = p.Cout - 1;
            a_src[i] = p.w_blocked ? p.w + ((long)co_tile * p.nk * TCO + r) * BK + chunk * 8 : p.w + (long)co * p.Ktot + chunk * 8;
            a_dst[i] = q * 1024;
        }
#pragma unroll
        for (int i = 0; i < Cfg::B_INSTR; ++i) {
            const int q = i * NW + wave;
            const int pos = q * 64 + lane;
            const int R = pos >> 4, s = (pos & 15) ^ swz_key<BK, M16>(R);
            const int r = R * RPB + s / CPR, chunk = s % CPR;
            b_src[i] = r < TPX ? p.in + tab[4 * r] + chunk * 8 : p.in + tab[0];   // pad rows of the B stage: one line, never read
            b_dst[i] = Cfg::A_BYTES + q * 1024;
        }
    }

    // ---- K range of this split.  Single-tap problems (Linear layers) interleave the splits: split y
    // takes K steps y, y+S, y+2S, ... so that at any moment the S workgroups of one output tile stream
    // ADJACENT 16-KB pieces of the weight matrix (DRAM-page friendly) instead of S far-apart ranges.
    const bool interleave = gridDim.y > 1 && p.KH * p.KW == 1;
    const int kstep = interleave ? (int)gridDim.y : 1;
    int kbeg, kend;
    if (interleave) {
        const int mine = (p.nk - (int)blockIdx.y + kstep - 1) / kstep;   // K steps owned by this split
        kbeg = 0;
        kend = mine > 0 ? mine : 0;
    } else {
        kbeg = blockIdx.y * p.nk_per_split;
        kend = min(p.nk, kbeg + p.nk_per_split);
    }
    const int cpt = p.tap_len / BK;  // K iterations per tap
    int tap = interleave ? 0 : kbeg / cpt;
    int c0 = interleave ? (int)blockIdx.y * BK : (kbeg - tap * cpt) * BK;
    int ky = tap / p.KW, kx = tap - ky * p.KW;

    auto stage = [&](int buf, int kiter_local) {
        char *sb = stage_base + buf * Cfg::STAGE_BYTES;
        const int kiter = interleave ? (int)blockIdx.y + kiter_local * kstep : kiter_local;
        const long a_off = (long)kiter * (p.w_blocked ? TCO * BK : BK);
        const long b_off = interleave ? (long)kiter * BK : (long)ky * p.in_row_stride + (long)kx * p.in_px_stride + c0;
        if (p.w_blocked) {
            // the blocked panels of a Linear layer are read exactly once per forward (411 MB for the one behind nn.Flatten): non-temporal policy, so the
            // stream does not push the activations out of L2 / Infinity Cache (94 -> 85-91 us at batch 64)
#pragma unroll
            for (int i = 0; i < Cfg::A_INSTR; ++i) GLDS16_NT(a_src[i] + a_off, sb + a_dst[i]);
        } else {
#pragma unroll
            for (int i = 0; i < Cfg::A_INSTR; ++i) GLDS16(a_src[i] + a_off, sb + a_dst[i]);
        }
#pragma unroll
        for (int i = 0; i < Cfg::B_INSTR; ++i) GLDS16(b_src[i] + b_off, sb + b_dst[i]);
        c0 += BK;
        if (c0 == p.tap_len) {
            c0 = 0;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };

    // ---- fragment read offsets.  32x32x16: lane l reads row (l&31), 16-B chunk (l>>5) of each 16-deep
    // k-step;  16x16x32: row (l&15), chunk (l>>4) of each 32-deep k-step.
    int a_rd[MT], b_rd[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_rd[i] = lds_off<BK, M16>(wco * (TCO / WCO) + i * FR + (lane & (FR - 1)), M16 ? (lane >> 4) : (lane >> 5));
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int row = px_lo + i * FR + (lane & (FR - 1));
        if (UNEVEN && row >= TPX) row = TPX - 1;          // group B has one column fewer: its last slot is never used
        b_rd[i] = Cfg::A_BYTES + lds_off<BK, M16>(row, M16 ? (lane >> 4) : (lane >> 5));
    }

    typedef typename std::conditional<M16, f32x4, f32x16>::type acc_t;
    acc_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < (M16 ? 4 : 16); ++r) acc[i][j][r] = 0.0f;

    // ---- K loop: NST-stage ring, loads NST-1 steps ahead.  The LDS-DMA of later stages stays in
    // flight across the barrier: each wave waits only for ITS loads of the stage about to be read
    // (counted vmcnt), then the raw barrier makes every wave's part of that stage visible and proves
    // that everyone has finished reading the stage that is overwritten next.
    constexpr int KSTEPS = BK / Cfg::KS;
    constexpr int KSH = M16 ? 6 : 5;   // the k-step only touches chunk-index bits that the row part left clear -> XOR
    auto read_frags = [&](int buf, bf16x8(&af)[KSTEPS][MT], bf16x8(&bfr)[KSTEPS][NT], auto ntc) {
        constexpr int NTG = decltype(ntc)::value;      // pixel columns of this wave group (NT, or NT1 for group B of an uneven split)
        const char *sb = stage_base + buf * Cfg::STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
            for (int i = 0; i < MT; ++i) af[ks][i] = *reinterpret_cast<const bf16x8 *>(sb + (a_rd[i] ^ (ks << KSH)));
#pragma unroll
            for (int j = 0; j < NTG; ++j) bfr[ks][j] = *reinterpret_cast<const bf16x8 *>(sb + (b_rd[j] ^ (ks << KSH)));
        }
    };
    auto mfmas = [&](bf16x8(&af)[KSTEPS][MT], bf16x8(&bfr)[KSTEPS][NT], auto ntc) {
        constexpr int NTG = decltype(ntc)::value;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTG; ++j) {
                    if constexpr (M16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
                }
    };

    if constexpr (STG) {
        // ---- staggered two-phase schedule (8 waves, 3-stage ring).  Each K step is split into an L phase
        // (all fragment reads of the step -> registers) and an M phase (all its MFMAs), separated by raw
        // barriers.  Waves 0..3 (group A) and 4..7 (group B) sit pairwise on the same SIMDs and run ONE
        // PHASE APART, so on every SIMD one wave feeds the matrix pipe while its partner reads LDS:
        //     slot:   0     1     2     3     4 ...
        //     A:      L0    M0    L1    M1    L2
        //     B:      -     L0    M0    L1    M1
        // Global->LDS loads of stage s are issued in the L phases (A in slot 2(s-D), B in slot 2(s-D)+1) and each wave
        // drains its own part with a counted vmcnt at the end of slot 2s-1, one barrier before the first
        // reader (A in slot 2s).  Buffer (s mod NST) was last read in slot 2(s-NST)+1, at least one barrier earlier.
        static_assert(!STG || (NST >= 3 && NW == 8), "stagger needs 8 waves and a ring of >= 3 stages");
        constexpr std::integral_constant<int, NT> ntA{};
        constexpr std::integral_constant<int, UNEVEN ? Cfg::NT1 : NT> ntB{};
        // general ring of NST stages (D = NST-1 stages of distance): stage s is issued in slot 2(s-D), must
        // have landed by the end of slot 2s-1; at that wait D-1 younger stages may stay in flight.
        constexpr int D = NST - 1;
        const bool grpB = wave >= NW / 2;
        const int nkk = kend - kbeg;
#pragma unroll
        for (int s0 = 0; s0 < D; ++s0)
            if (s0 < nkk) stage(s0, kbeg + s0);
        auto wait_stage = [&](int newer) {   // newer = younger stages this wave has already issued
            if (newer >= D - 1) wait_vmcnt<(D - 1) * Cfg::LOADS>();
            else if (newer == 1 && D - 1 > 1) wait_vmcnt<Cfg::LOADS>();
            else wait_vmcnt<0>();
        };
        wait_stage(nkk - 1 < D - 1 ? nkk - 1 : D - 1);   // stage 0 landed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 af[KSTEPS][MT], bfr[KSTEPS][NT];
        int rd_buf = 0;
#ifdef IGEMM_STAMPS
        // diagnostic build: where one K iteration of this wave spends its cycles (stamps go to a buffer of their own)
        long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) do { if (p.dbg && it == p.dbg_it) { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
        if (!grpB) {
            int ld_buf = D % NST;
            for (int it = 0; it < nkk; ++it) {
                STAMP(0);
                read_frags(rd_buf, af, bfr, ntA);                                // L(it)   (even slot)
                STAMP(1);
                if (it + D < nkk) stage(ld_buf, kbeg + it + D);
                STAMP(2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                STAMP(3);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                STAMP(4);
                __builtin_amdgcn_s_setprio(1);
                mfmas(af, bfr, ntA);                                             // M(it)   (odd slot)
                __builtin_amdgcn_s_setprio(0);
                STAMP(5);
                { const int left = nkk - 2 - it; wait_stage(left < 0 ? 0 : left); }   // stage it+1 landed
                STAMP(6);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                STAMP(7);
                rd_buf = rd_buf + 1 == NST ? 0 : rd_buf + 1;
                ld_buf = ld_buf + 1 == NST ? 0 : ld_buf + 1;
            }
            __builtin_amdgcn_s_barrier();
        } else {
            __builtin_amdgcn_s_barrier();                                        // slot 0
            __builtin_amdgcn_sched_barrier(0);
            int ld_buf = D % NST;
            for (int it = 0; it < nkk; ++it) {
                STAMP(0);
                read_frags(rd_buf, af, bfr, ntB);                                // L(it)   (odd slot)
                STAMP(1);
                // the group's share of stage it+D is issued here, in the L phase (as group A does in its own): the MFMA pipe
                // belongs to the partner wave now, and the M phase below stays pure MFMA (in-kernel stamps: ~390 cycles of
                // DMA issue in front of the MFMAs left the pipe idle; tools/stamps_igemm.py)
                if (it + D < nkk) stage(ld_buf, kbeg + it + D);
                STAMP(2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                { const int left = nkk - 2 - it; wait_stage(left < 0 ? 0 : left); }   // stage it+1 landed
                STAMP(3);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                STAMP(4);
                STAMP(5);
                __builtin_amdgcn_s_setprio(1);
                mfmas(af, bfr, ntB);                                             // M(it)   (even slot)
                __builtin_amdgcn_s_setprio(0);
                STAMP(6);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                STAMP(7);
                rd_buf = rd_buf + 1 == NST ? 0 : rd_buf + 1;
                ld_buf = ld_buf + 1 == NST ? 0 : ld_buf + 1;
            }
        }
#ifdef IGEMM_STAMPS
        if (p.dbg && lane == 0 && blockIdx.x < 512) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p.dbg[((long)blockIdx.x * 8 + wave) * 8 + i] = stamp[i];
        }
#endif
#undef STAMP
    } else {
    constexpr int D = NST - 1;
#pragma unroll
    for (int s0 = 0; s0 < D; ++s0)
        if (kbeg + s0 < kend) stage(s0, kbeg + s0);
    int cur = 0, nxt = D % NST;
    for (int it = kbeg; it < kend; ++it) {
        if (D >= 2 && it + 1 < kend) wait_vmcnt<(D >= 2 ? (D - 1) * Cfg::LOADS : 0)>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + D < kend) stage(nxt, it + D);
        bf16x8 af[KSTEPS][MT], bfr[KSTEPS][NT];
        const char *sb = stage_base + cur * Cfg::STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
            for (int i = 0; i < MT; ++i) af[0][i] = *reinterpret_cast<const bf16x8 *>(sb + (a_rd[i] ^ (ks << KSH)));
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8 *>(sb + (b_rd[j] ^ (ks << KSH)));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (M16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
                }
        }
        cur = (cur + 1 == NST) ? 0 : cur + 1;
        nxt = (nxt + 1 == NST) ? 0 : nxt + 1;
    }
    }
    __syncthreads();  // all MFMA operand reads done before the stage area is reused for the epilogue

    // ---- epilogue, in WPX passes of PPX = TPX/WPX pixels (keeps the fp32 staging tile small: the 256-wide
    // tiles would not fit otherwise).  Pass q: the waves of pixel column q put their accumulators into LDS as
    // fp32 [px][co]; then ALL threads stream that slab out, 8 channels (16 B of bf16) per thread per step.
    //   32x32 C/D map: col (= pixel) = lane&31, row (= co) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    //   16x16 C/D map: col (= pixel) = lane&15, row (= co) = 4*(lane>>4) + reg
    float *ep = reinterpret_cast<float *>(stage_base);
    constexpr int PPX = Cfg::PPX;
    constexpr int CCH = TCO / 8;             // 8-channel chunks per pixel
    constexpr int PX_PER_STEP = NTHR / CCH;  // pixels covered by the workgroup per step
    const int cc = tid % CCH;
    const int co = co0 + cc * 8;
    const bool has_bias = p.epilogue == YOLO_EPI_BIAS || p.epilogue == YOLO_EPI_BIAS_LRELU || p.epilogue == YOLO_EPI_BIAS_ADD_LRELU;
    float bias8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bias8[k] = has_bias && (co + k < p.Cout) ? p.bias[co + k] : 0.0f;
    const bool split = gridDim.y > 1 && p.slab_stride == 0;
    // split-K into slabs: split y owns the dense fp32 partial tile at out + y * slab_stride and takes the plain store path below
    void *const outp = p.slab_stride ? (void *)(reinterpret_cast<float *>(p.out) + (long)blockIdx.y * p.slab_stride) : p.out;
    // BatchNorm statistics of this thread's outputs (p.stats) live in LDS behind the epilogue slab, not in registers: the
    // accumulators of the later passes are still live here and the 256x128 staggered kernel has none to spare
    float *red = reinterpret_cast<float *>(stage_base + Cfg::EPI_BYTES) + tid * 16;
    if constexpr (STATS) {
#pragma unroll
        for (int k = 0; k < 16; ++k) red[k] = 0.0f;
    }

    // Epilogues that read a second tensor (residual add / LeakyReLU' gate), bf16 output, small tiles (the HBM-bound thin-K layers):
    // the aux vectors of a pass are fetched in front of that pass's LDS staging, so that their HBM latency overlaps the staging
    // and its barriers instead of being paid pixel by pixel inside the store loop (ResNet-50 inference at batch 64: 6.28 -> 5.99 ms;
    // fetching them already in front of the K loop was slower, 6.16 ms -- they then delay the operand stream).
    constexpr int E_CCH = TCO / 8, E_PXS = NTHR / E_CCH, E_PPX = Cfg::PPX;
    constexpr int NIT = (E_PPX + E_PXS - 1) / E_PXS;
    constexpr bool CAN_PREFETCH = NIT <= 4 && !UNEVEN && !STATS && TCO * TPX <= 128 * 128;   // (the big tiles have no registers to spare)
    const bool aux_fast = CAN_PREFETCH && (p.epilogue == YOLO_EPI_BIAS_ADD_LRELU || p.epilogue == YOLO_EPI_MUL_DLRELU) && gridDim.y == 1 && !p.pool &&
                          !p.out_fp32 && co0 + (tid % E_CCH) * 8 + 8 <= p.Cout;
    uint4 axv[CAN_PREFETCH ? NIT : 1];
    long axo[CAN_PREFETCH ? NIT : 1];
    auto aux_fetch = [&](int q) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int lp = tid / E_CCH + it * E_PXS;
            const int px = q * E_PPX + (lp < E_PPX ? lp : 0);
            axo[it] = lp < E_PPX ? tab[4 * px + 1] : -1;
            axv[it] = axo[it] >= 0 ? *reinterpret_cast<const uint4 *>(p.aux + tab[4 * px + 2] + co0 + (tid % E_CCH) * 8) : uint4{0u, 0u, 0u, 0u};
        }
    };

    for (int q = 0; q < WPX; ++q) {
        if (q > 0) __syncthreads();   // the previous slab has been streamed out
        if constexpr (CAN_PREFETCH) {
            if (aux_fast) aux_fetch(q);
        }
        if (wpx == q) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (M16) {
                        const int lp = j * 16 + (lane & 15);
                        const int cob = wco * (TCO / WCO) + i * 16 + 4 * (lane >> 4);
                        f32x4 v = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        *reinterpret_cast<f32x4 *>(ep + lp * Cfg::EP + cob) = v;
                    } else {
                        const int lp = j * 32 + (lane & 31);
                        const int cob = wco * (TCO / WCO) + i * 32 + 4 * (lane >> 5);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                            *reinterpret_cast<f32x4 *>(ep + lp * Cfg::EP + cob + 8 * g) = v;
                        }
                    }
                }
        }
        __syncthreads();
        const int pbase = q * PPX;   // first tile pixel of this slab
        const int ppx_q = UNEVEN && q == 1 ? Cfg::NT1 * 16 : PPX;   // its pixels (group B of an uneven split has one column fewer)

        if (split) {
            // split-K partial tile: fp32 atomics shaped as 256 contiguous bytes per wave-instruction (one
            // dword per lane along the channel axis) -- the fast form of global_atomic_add_f32
            float *o = reinterpret_cast<float *>(p.out);
            for (int e = tid; e < ppx_q * TCO; e += NTHR) {
                const int lp = e / TCO, c = e - lp * TCO;
                const long ob = tab[4 * (pbase + lp) + 1];
                if (ob >= 0 && co0 + c < p.Cout) atomicAdd(o + ob + co0 + c, ep[lp * Cfg::EP + c]);
            }
            continue;
        }
        if (p.pool) {
            // fused MaxPool2d(2,2): max over the window's four LDS rows, then bias + LeakyReLU (monotone, so
            // pool(lrelu(z + b)) == lrelu(max(z) + b)); one 16-B store per pooled pixel and channel chunk.
            // A slab is PPX/pool_tw full rows of the 8 x 16 patch, so every window lies inside one slab.
            const int tw = p.pool_tw, hw = tw >> 1;
            for (int w = tid / CCH; w < PPX / 4; w += PX_PER_STEP) {
                const int l00 = (w / hw) * 2 * tw + (w % hw) * 2;
                const long ob = tab[4 * (pbase + l00) + 1];
                if (ob < 0 || co >= p.Cout) continue;
                float v[8];
                if (CODES && p.pool == 3) {
                    // pool2 = 3 (training): the pooled map + the window position of every maximum (2 bits per channel, one ushort
                    // per 8 channels), compared on the activations as stored (bf16), first maximum in (0,0),(0,1),(1,0),(1,1) order
                    unsigned code = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float *zp = ep + l00 * Cfg::EP + cc * 8 + k;
                        float m = 0.0f;
                        unsigned am = 0;
#pragma unroll
                        for (int w4 = 0; w4 < 4; ++w4) {
                            float t = zp[((w4 >> 1) * tw + (w4 & 1)) * Cfg::EP] + bias8[k];
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
                        const float a = ep[l00 * Cfg::EP + cc * 8 + k], b = ep[(l00 + 1) * Cfg::EP + cc * 8 + k];
                        const float c = ep[(l00 + tw) * Cfg::EP + cc * 8 + k], d = ep[(l00 + tw + 1) * Cfg::EP + cc * 8 + k];
                        float m = fmaxf(fmaxf(a, b), fmaxf(c, d)) + bias8[k];
                        v[k] = (p.epilogue == YOLO_EPI_BIAS_LRELU && m < 0.0f) ? m * p.slope : m;
                    }
                }
                uint4 pk;
                pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                pk.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
                pk.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
                *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(p.out) + ob + co) = pk;
            }
            if (p.pool == 2) {
                // pool2 = 2 (training): the un-pooled activation is written as well, through the aux pointer / strides
                // (table entry 2 = un-pooled address of every tile pixel), so no separate pooling pass reads it back
                bf16_t *full = const_cast<bf16_t *>(p.aux);
                for (int lp = tid / CCH; lp < PPX; lp += PX_PER_STEP) {
                    const int px = pbase + lp;
                    if (tab[4 * px + 1] < 0) continue;
                    if (co >= p.Cout) continue;
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float m = ep[lp * Cfg::EP + cc * 8 + k] + bias8[k];
                        v[k] = (p.epilogue == YOLO_EPI_BIAS_LRELU && m < 0.0f) ? m * p.slope : m;
                    }
                    uint4 pk;
                    pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    pk.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
                    pk.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
                    *reinterpret_cast<uint4 *>(full + tab[4 * px + 2] + co) = pk;
                }
            }
            continue;
        }
        if constexpr (CAN_PREFETCH) {
            if (aux_fast) {
#pragma unroll
                for (int it0 = 0; it0 < NIT; ++it0) {
                    const uint4 av = axv[it0];
                    const long ao = axo[it0];
                    if (ao < 0) continue;
                    const int lp = tid / CCH + it0 * PX_PER_STEP;
                    const f32x4 lo = *reinterpret_cast<const f32x4 *>(ep + lp * Cfg::EP + cc * 8);
                    const f32x4 hi = *reinterpret_cast<const f32x4 *>(ep + lp * Cfg::EP + cc * 8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const unsigned yy[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float a = __uint_as_float((k & 1) ? (yy[k >> 1] & 0xffff0000u) : (yy[k >> 1] << 16));
                        if (p.epilogue == YOLO_EPI_BIAS_ADD_LRELU) {
                            v[k] += bias8[k] + a;
                            v[k] = v[k] > 0.0f ? v[k] : v[k] * p.slope;
                        } else {
                            v[k] = a > 0.0f ? v[k] : v[k] * p.slope;
                        }
                    }
                    uint4 pk;
                    pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    pk.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
                    pk.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
                    *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(p.out) + ao + co) = pk;
                }
                continue;
            }
        }
#pragma unroll 2
        for (int lp = tid / CCH; lp < ppx_q; lp += PX_PER_STEP) {
            const int px = pbase + lp;
            const long ob = tab[4 * px + 1];
            if (ob < 0 || co >= p.Cout) continue;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(ep + lp * Cfg::EP + cc * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(ep + lp * Cfg::EP + cc * 8 + 4);
            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
            v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
            if (has_bias) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += bias8[k];
            }
            if (p.epilogue == YOLO_EPI_BIAS_ADD_LRELU) {
                // residual branch (ResNet bottleneck): out = act(conv + bias + identity)
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
                bf16_t *o = reinterpret_cast<bf16_t *>(p.out) + ob + co;
                uint4 pk;
                pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                pk.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
                pk.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
                *reinterpret_cast<uint4 *>(o) = pk;  // Cout % 8 == 0 is required for bf16 outputs
                if constexpr (STATS) {
                    const unsigned w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float r = __uint_as_float((k & 1) ? (w4[k >> 1] & 0xffff0000u) : (w4[k >> 1] << 16));   // the value as stored
                        red[k] += r;
                        red[8 + k] += r * r;
                    }
                }
            }
        }
    }
    if constexpr (STATS) {
        // fold the threads that share a channel chunk (tid % CCH) through LDS, then one fp64 atomic pair per channel and tile
        // into one of YOLO_BN_ACC_REPLICAS accumulators (same-address fp64 atomics serialise)
        __syncthreads();
        const float *all = reinterpret_cast<const float *>(stage_base + Cfg::EPI_BYTES);
        for (int c = tid; c < TCO; c += NTHR) {
            const int c8 = c >> 3, k = c & 7;
            float a = 0.0f, b = 0.0f;
            for (int j = 0; j < NTHR / CCH; ++j) { a += all[(j * CCH + c8) * 16 + k]; b += all[(j * CCH + c8) * 16 + 8 + k]; }
            if (co0 + c < p.Cout) {
                double *rep = p.stats + (size_t)(blockIdx.x % YOLO_BN_ACC_REPLICAS) * 2 * p.Cout;
                atomicAdd(rep + co0 + c, (double)a);
                atomicAdd(rep + p.Cout + co0 + c, (double)b);
            }
        }
    }
}

template <int TCO, int TPX, int BK, int WCO, int WPX, int NST, int MF, int VAR>
static int launch_impl(const IgemmParams &p, int splits, hipStream_t s)
{
    using Cfg = IgemmCfg<TCO, TPX, BK, WCO, WPX, NST, MF>;
    static bool attr_done[64] = {};   // per device: a function's attributes belong to the device it is loaded on
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm_kernel<TCO, TPX, BK, WCO, WPX, NST, MF, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return fail((int)e, "yolo_igemm: hipFuncSetAttribute(%d B LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
        attr_done[dev] = true;
    }
    IgemmParams q = p;
    q.n_co_tiles = (p.Cout + TCO - 1) / TCO;
    if (p.pool) {
        q.tpx_valid = TPX;
        q.pool_tiles_x = (p.Wo + p.pool_tw - 1) / p.pool_tw;
        q.pool_tiles_y = (p.HoWo / p.Wo + TPX / p.pool_tw - 1) / (TPX / p.pool_tw);
        q.n_px_tiles = (int)(p.M / p.HoWo) * q.pool_tiles_x * q.pool_tiles_y;
    } else {
        if (q.tpx_valid <= 0 || q.tpx_valid > TPX) q.tpx_valid = TPX;
        q.n_px_tiles = (int)((p.M - p.px_begin + q.tpx_valid - 1) / q.tpx_valid);
    }
    q.nk = (int)(p.Ktot / BK);
    // L2 working set: when the weights are much larger than an XCD's 4-MB L2, give each XCD its own weight
    // panel(s) and let it stream the pixel tiles (activations are then read once per XCD instead of the whole
    // weight tensor once per group of pixel tiles)
    if (p.px_fastest < 0) q.px_fastest = 0;   // measured: channel-tiles-fastest is never slower on this network (tile_order overrides)
    q.nk_per_split = (q.nk + splits - 1) / splits;
    // slabs: every split must store its slab (the finishing pass adds all of them), also one whose K range is empty
    const int real_splits = p.slab_stride ? splits : (q.nk + q.nk_per_split - 1) / q.nk_per_split;
    hipLaunchKernelGGL((igemm_kernel<TCO, TPX, BK, WCO, WPX, NST, MF, VAR>), dim3(q.n_co_tiles * q.n_px_tiles, real_splits), dim3(Cfg::NTHR), Cfg::LDS_BYTES, s, q);
    return check_launch("yolo_igemm");
}

template <int TCO, int TPX, int BK, int WCO, int WPX, int NST, int MF = MFMA_32x32x16>
static int launch(const IgemmParams &p, int splits, hipStream_t s)
{
    return p.stats ? launch_impl<TCO, TPX, BK, WCO, WPX, NST, MF, 1>(p, splits, s) : launch_impl<TCO, TPX, BK, WCO, WPX, NST, MF, 0>(p, splits, s);
}

// Finishing pass of a split-K conv: acc fp32 [M][Cout] (dense, summed by the split workgroups' atomics) -> the layer's
// real output with the epilogue the un-split launch would have applied (bias / LeakyReLU / times LeakyReLU'(aux)),
// bf16 into the zero-haloed NHWC buffer.  One thread = one pixel x 8 channels.
__global__ void __launch_bounds__(256) igemm_finish_kernel(const float *__restrict__ acc, const float *__restrict__ bias, const bf16_t *__restrict__ aux,
                                                           bf16_t *__restrict__ out, long M, int slabs, int HoWo, int Wo, int Cout, int epilogue, float slope,
                                                           long out_img, int out_row, int out_px, int out_off, long aux_img, int aux_row, int aux_px, int aux_off)
{
    const int C8 = Cout >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * C8) return;
    const int c = (int)(idx % C8) * 8;
    const long m = idx / C8;
    const int n = (int)(m / HoWo);
    const int rem = (int)(m - (long)n * HoWo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const float4 a = *reinterpret_cast<const float4 *>(acc + m * Cout + c), b = *reinterpret_cast<const float4 *>(acc + m * Cout + c + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    for (int sl = 1; sl < slabs; ++sl) {      // fixed order: the sum does not depend on which split finished first
        const float *as = acc + (long)sl * M * Cout + m * Cout + c;
        const float4 a2 = *reinterpret_cast<const float4 *>(as), b2 = *reinterpret_cast<const float4 *>(as + 4);
        v[0] += a2.x; v[1] += a2.y; v[2] += a2.z; v[3] += a2.w;
        v[4] += b2.x; v[5] += b2.y; v[6] += b2.z; v[7] += b2.w;
    }
    if (epilogue == YOLO_EPI_BIAS || epilogue == YOLO_EPI_BIAS_LRELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bias[c + k];
    }
    if (epilogue == YOLO_EPI_BIAS_LRELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.0f ? v[k] : v[k] * slope;
    } else if (epilogue == YOLO_EPI_MUL_DLRELU) {
        const uint4 y = *reinterpret_cast<const uint4 *>(aux + (long)n * aux_img + (long)oy * aux_row + (long)ox * aux_px + aux_off + c);
        const unsigned yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float t = __uint_as_float((k & 1) ? (yy[k >> 1] & 0xffff0000u) : (yy[k >> 1] << 16));
            v[k] = t > 0.0f ? v[k] : v[k] * slope;
        }
    }
    uint4 o;
    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
    o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4 *>(out + (long)n * out_img + (long)oy * out_row + (long)ox * out_px + out_off + c) = o;
}

}  // namespace yolo

using namespace yolo;

static long *g_dbg = nullptr;
static int g_dbg_it = 0;

namespace yolo {
void debug_stamp_target(long **buf, int *it)
{
    *buf = g_dbg;
    *it = g_dbg_it;
}
}  // namespace yolo

YOLO_API int yolo_debug_stamps(void *buf, int k_iter)
{
    g_dbg = (long *)buf;
    g_dbg_it = k_iter;
#ifdef IGEMM_STAMPS
    return 0;
#else
    return buf ? fail(YOLO_E_UNSUPPORTED, "yolo_debug_stamps: this library was built without -DIGEMM_STAMPS (make diag)") : 0;
#endif
}

YOLO_API int yolo_igemm(const yolo_igemm_desc *d, const void *in, const void *w, const float *bias, const void *aux, void *out, yolo_stream_t stream)
{
    if (!d || !in || !w || !out) return fail(YOLO_E_ARG, "yolo_igemm: null pointer");
    if (d->N <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->KH <= 0 || d->KW <= 0 || d->tap_len <= 0 || d->Cout <= 0 || d->stride <= 0)
        return fail(YOLO_E_ARG, "yolo_igemm: bad descriptor");
    if ((d->epilogue == YOLO_EPI_BIAS || d->epilogue == YOLO_EPI_BIAS_LRELU || d->epilogue == YOLO_EPI_BIAS_ADD_LRELU) && !bias) return fail(YOLO_E_ARG, "yolo_igemm: epilogue needs bias");
    if ((d->epilogue == YOLO_EPI_MUL_DLRELU || d->epilogue == YOLO_EPI_BIAS_ADD_LRELU) && !aux) return fail(YOLO_E_ARG, "yolo_igemm: epilogue needs aux");
    if (d->epilogue < 0 || d->epilogue > YOLO_EPI_BIAS_ADD_LRELU) return fail(YOLO_E_ARG, "yolo_igemm: epilogue %d", d->epilogue);
    const int splits = d->split_k > 1 ? d->split_k : 1;
    if (splits > 1 && (!d->out_fp32 || d->epilogue != YOLO_EPI_NONE)) return fail(YOLO_E_ARG, "yolo_igemm: split_k needs fp32 output and EPI_NONE");
    if (d->split_slabs && splits > 1 && (d->out_off != 0 || (d->Wo > 1 && d->out_px_stride != d->Cout) || (d->Ho > 1 && d->out_row_stride != d->Wo * d->Cout)
                                         || (d->N > 1 && d->out_img_stride != (int64_t)d->Ho * d->Wo * d->Cout)))
        return fail(YOLO_E_ARG, "yolo_igemm: split_slabs needs the dense [N*Ho*Wo][Cout] output addressing");
    if (d->tile_px < 0) return fail(YOLO_E_ARG, "yolo_igemm: tile_px %d", d->tile_px);
    if (!d->out_fp32 && ((d->Cout & 7) || (d->out_off & 7) || (d->out_px_stride & 7) || (d->out_row_stride & 7) || (d->out_img_stride & 7)))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: bf16 output needs Cout and output strides in multiples of 8");
    if ((d->tap_len & 31) || (d->in_off & 7) || (d->in_px_stride & 3) || (d->in_row_stride & 7) || (d->in_img_stride & 7))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tap_len=%d must be a multiple of 32 and input strides 16-B aligned", d->tap_len);

    IgemmParams p{};
    p.in = (const bf16_t *)in; p.w = (const bf16_t *)w; p.bias = bias; p.aux = (const bf16_t *)aux; p.out = out;
    p.M = (long)d->N * d->Ho * d->Wo; p.HoWo = d->Ho * d->Wo; p.Wo = d->Wo;
    if (d->px_begin < 0 || d->px_end < 0 || d->px_begin > p.M || d->px_end > p.M || (d->px_end && d->px_end <= d->px_begin))
        return fail(YOLO_E_ARG, "yolo_igemm: bad pixel range [%ld, %ld) of %ld", (long)d->px_begin, (long)d->px_end, (long)p.M);
    if ((d->px_begin || d->px_end) && d->pool2) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: pixel ranges are not available with pool2");
    p.px_begin = d->px_begin;
    p.tpx_valid = d->tile_px;     // clamped to the configuration's pixel-tile edge at launch
    p.slab_stride = (d->split_slabs && splits > 1) ? (long)d->N * d->Ho * d->Wo * d->Cout : 0;
    if (d->px_end) p.M = d->px_end;   // the kernels bound pixels by p.M
    p.in_img_stride = d->in_img_stride; p.in_row_stride = d->in_row_stride; p.in_px_stride = d->in_px_stride; p.in_off = d->in_off; p.stride = d->stride;
    p.KH = d->KH; p.KW = d->KW; p.tap_len = d->tap_len; p.Cout = d->Cout;
    p.Ktot = (long)d->KH * d->KW * d->tap_len;
    p.out_img_stride = d->out_img_stride; p.out_row_stride = d->out_row_stride; p.out_px_stride = d->out_px_stride; p.out_off = d->out_off;
    p.aux_img_stride = d->aux_img_stride; p.aux_row_stride = d->aux_row_stride; p.aux_px_stride = d->aux_px_stride; p.aux_off = d->aux_off;
    p.epilogue = d->epilogue; p.slope = d->slope; p.out_fp32 = d->out_fp32;
    hipStream_t s = STRM(stream);

    const bool bk64 = (d->tap_len % 64) == 0;
    const bool small_co = d->Cout <= 64 || (d->Cout % 128 != 0 && d->Cout % 64 == 0 && d->Cout < 512);
    const long npx = p.M - p.px_begin;   // pixels of this launch
    const long tiles128 = ((npx + 127) / 128) * ((d->Cout + 127) / 128);
    const int force = d->tile_hint;   // 0 = heuristic; tests / tuning may force a configuration
    p.w_blocked = d->w_blocked;
    p.px_fastest = d->tile_order == 1 ? 0 : (d->tile_order == 2 ? 1 : -1);
    if (d->skew_phases < 0 || d->skew_phases > 64 || d->skew_step < 0 || d->skew_step > (1 << 22)) return fail(YOLO_E_ARG, "yolo_igemm: skew_phases %d / skew_step %d", d->skew_phases, d->skew_step);
    p.skew_phases = d->skew_phases;
    p.skew_cycles = d->skew_step;
    p.stats = (double *)d->bn_stats;
    p.dbg = g_dbg;
    p.dbg_it = g_dbg_it;
    if (p.stats && (d->out_fp32 || splits > 1 || d->pool2 || d->px_begin || d->px_end))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: bn_stats needs a plain bf16 launch (no split_k, pool2, pixel range)");
    p.pool = (d->pool2 == 2 || d->pool2 == 3) ? d->pool2 : (d->pool2 ? 1 : 0);
    if (p.pool == 2 && !aux) return fail(YOLO_E_ARG, "yolo_igemm: pool2 = 2 writes the un-pooled activation through aux (pointer + aux_* strides)");
    if (p.pool == 3 && (!aux || (d->Cout & 7) || (d->out_img_stride & 7) || (d->out_row_stride & 7) || (d->out_px_stride & 7) || (d->out_off & 7)))
        return fail(YOLO_E_ARG, "yolo_igemm: pool2 = 3 writes the arg-max codes through aux (one uint16 per 8 channels at (pooled element address) / 8): "
                                "Cout and the out_* strides must be multiples of 8");
    p.pool_tw = 16;
    if (p.pool && d->tile_px) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_px is not available with pool2");
    if (p.pool && (splits > 1 || d->out_fp32 || (d->Ho & 1) || (d->Wo & 1) || (d->epilogue != YOLO_EPI_BIAS && d->epilogue != YOLO_EPI_BIAS_LRELU)))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: pool2 needs even Ho/Wo, bf16 output, a bias epilogue and no split-K");
    if (force == 20 || force == 21) return igemm_persist_launch(p, force, splits, s);   // persistent loop, epilogue from the accumulator registers (igemm_persist.hip)
    if (p.pool == 3 && !(force >= 15 && force <= 18)) {
        // pooled map + arg-max codes: the two 8 x 16-patch configurations in their own instantiations (VAR 2), whatever the hint
        if (!bk64 || d->w_blocked || p.stats) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: pool2 = 3 needs tap_len %% 64 == 0 (or tile_hint 16 / 18), plain weights, no bn_stats");
        if (small_co) return launch_impl<64, 128, 64, 2, 2, 2, MFMA_16x16x32, 2>(p, splits, s);
        return launch_impl<128, 128, 64, 2, 2, 2, MFMA_16x16x32, 2>(p, splits, s);
    }
    if (d->w_blocked) {
        if (!bk64) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: blocked weights need tap_len %% 64 == 0");
        // batch <= 64: a pure weight stream -- three stages keep two 24-KB loads in flight per workgroup (4.9 vs 4.4 TB/s on FC1)
        return npx <= 64 ? launch<128, 64, 64, 2, 2, 3, MFMA_16x16x32>(p, splits, s) : launch<128, 128, 64, 2, 2, 2, MFMA_16x16x32>(p, splits, s);
    }
    if (force == 12) return launch<256, 256, 32, 2, 4, 4, MFMA_16x16x32_STAGGER>(p, splits, s);
    if (force == 14) {
        if (d->pool2) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint 14 has no pooled epilogue");
        return launch<256, 208, 32, 4, 2, 4, MFMA_16x16x32_STAGGER_U>(p, splits, s);
    }
    if (force >= 15 && force <= 18) return igemm_pipe_launch(p, force, splits, s);     // register-pipelined one-barrier loop (igemm_pipe.hip)
    if (force == 22) return conv_c64_launch(p, s);                                      // 3x3 64 -> 64: weights resident in LDS, patch staged once per tile (conv_c64.hip)
    if (force == 19) return igemm_stream_launch(p, splits, s);                          // streaming 1x1 convolution, thin K (igemm_stream.hip)
    if (force == 13) return launch<256, 128, 32, 4, 2, 4, MFMA_16x16x32_STAGGER>(p, splits, s);
    if (force == 10) return launch<64, 128, 32, 2, 2, 2>(p, splits, s);     // 28 KB of LDS: five workgroups per CU (thin-K 1x1 layers)
    if (!bk64) {
        if (force == 7) return launch<64, 128, 32, 2, 2, 3>(p, splits, s);
        if (force == 9) return launch<64, 128, 32, 2, 2, 4>(p, splits, s);
        if (force == 10) return launch<64, 128, 32, 2, 2, 2>(p, splits, s);
        if (small_co) return launch<64, 128, 32, 2, 2, 3>(p, splits, s);
        return launch<128, 128, 32, 2, 2, 2>(p, splits, s);
    }
    if (force == 8) return launch<64, 128, 64, 2, 2, 3, MFMA_16x16x32>(p, splits, s);
    if (p.pool) {   // pooled epilogue works on 8 x 16 pixel patches = 128-pixel tiles
        if (small_co) return launch<64, 128, 64, 2, 2, 2, MFMA_16x16x32>(p, splits, s);
        return launch<128, 128, 64, 2, 2, 2, MFMA_16x16x32>(p, splits, s);
    }
    if (force == 1) return launch<128, 128, 64, 2, 2, 2>(p, splits, s);
    if (force == 2) return launch<256, 128, 64, 4, 2, 3>(p, splits, s);
    if (force == 3) return launch<128, 64, 64, 2, 2, 2>(p, splits, s);
    if (force == 4) return launch<64, 128, 64, 2, 2, 2>(p, splits, s);
    if (force == 5) return launch<128, 128, 64, 2, 2, 2, MFMA_16x16x32>(p, splits, s);
    if (force == 6) return launch<256, 128, 64, 4, 2, 3, MFMA_16x16x32>(p, splits, s);
    if (force == 11) return launch<256, 128, 64, 4, 2, 3, MFMA_16x16x32_STAGGER>(p, splits, s);
    if (npx <= 64) {
        if (small_co) return launch<64, 64, 64, 2, 2, 2, MFMA_16x16x32>(p, splits, s);
        // split-K at batch <= 64 = a Linear layer streaming its weights: three stages, as for the blocked panels
        return splits > 1 ? launch<128, 64, 64, 2, 2, 3, MFMA_16x16x32>(p, splits, s) : launch<128, 64, 64, 2, 2, 2, MFMA_16x16x32>(p, splits, s);
    }
    if (small_co) return launch<64, 128, 64, 2, 2, 2, MFMA_16x16x32>(p, splits, s);
    // (the 256x128 8-wave 3-stage configuration is built and tested but measured slower than 128x128
    //  on every layer of this network -- it is reachable through tile_hint only)
    // few tiles (7x7 layers): halve the pixel tile to double the number of workgroups
    if (tiles128 * splits < 320 && d->Cout >= 128) return launch<128, 64, 64, 2, 2, 2, MFMA_16x16x32>(p, splits, s);
    return launch<128, 128, 64, 2, 2, 2, MFMA_16x16x32>(p, splits, s);   // measured 3-8 % faster than 32x32x16 on the 3x3 layers
}

YOLO_API int yolo_igemm_finish(const yolo_igemm_desc *d, const float *acc, const float *bias, const void *aux, void *out, yolo_stream_t stream)
{
    if (!d || !acc || !out) return fail(YOLO_E_ARG, "yolo_igemm_finish: null pointer");
    if (d->epilogue != YOLO_EPI_NONE && d->epilogue != YOLO_EPI_BIAS && d->epilogue != YOLO_EPI_BIAS_LRELU && d->epilogue != YOLO_EPI_MUL_DLRELU)
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm_finish: epilogue %d", d->epilogue);
    if ((d->epilogue == YOLO_EPI_BIAS || d->epilogue == YOLO_EPI_BIAS_LRELU) && !bias) return fail(YOLO_E_ARG, "yolo_igemm_finish: epilogue needs bias");
    if (d->epilogue == YOLO_EPI_MUL_DLRELU && !aux) return fail(YOLO_E_ARG, "yolo_igemm_finish: epilogue needs aux");
    if (d->out_fp32 || d->pool2 || (d->Cout & 7) || (d->out_off & 7) || (d->out_px_stride & 7) || (d->out_row_stride & 7) || (d->out_img_stride & 7))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm_finish: bf16 output with Cout and strides in multiples of 8, no pool2");
    const long M = (long)d->N * d->Ho * d->Wo;
    const long total = M * (d->Cout / 8);
    hipLaunchKernelGGL(igemm_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, STRM(stream), acc, bias, (const bf16_t *)aux, (bf16_t *)out, M,
                       (d->split_slabs && d->split_k > 1) ? d->split_k : 1, d->Ho * d->Wo, d->Wo, d->Cout, d->epilogue, d->slope, (long)d->out_img_stride, d->out_row_stride, d->out_px_stride, d->out_off,
                       (long)d->aux_img_stride, d->aux_row_stride, d->aux_px_stride, d->aux_off);
    return check_launch("yolo_igemm_finish");
}
