// Weight-gradient kernel for gfx950, 256 x 256 tile (yolo_wgrad_desc.variant = 5):
//     dw[co][tap][ci] (+)= sum_p dy[p][co] * x[p + tapoff(tap)][ci]
// The product, the pixel-major LDS-DMA staging and the transposing fragment reads (ds_read_b64_tr_b16) are those of
// wgrad.hip.  What differs -- measured reasons in DESIGN.md (round 2):
//   * wave tile 128 co x 64 ci (eight waves as 2 x 4) instead of 64 x 64: 12 transposing reads per 8 MFMAs (32x32x16)
//     instead of 16, and a 256 x 256 workgroup tile halves the bytes staged per MAC;
//   * stages of 32 pixels in a ring of FOUR, LDS-DMA three stages ahead behind counted vmcnt waits, ONE barrier per stage;
//   * register pipelining inside the wave: while the MFMAs of a 16-pixel sub-step run, the fragments of the next sub-step
//     are read into the second register set and the wave's share of the next stage's LDS-DMA is issued between the MFMAs
//     (sched_group_barrier), as in igemm_pipe.hip.
// Columns: Cin >= 256 -> one tap per tile (ci tiles of 256); Cin in {64, 128} -> 256 / Cin adjacent taps per tile.
// Pixel indexing (flat / geometry), the two-segment schedule and the XCD map are shared with wgrad.hip (wgrad_common.h).
#include "wgrad_common.h"

namespace yolo {

namespace wp {
constexpr int TCO = 256, TCI = 256, BP = 32, NST = 4, D = NST - 1, NW = 8, NTHR = NW * 64;
constexpr int ROW = 512;                          // bytes per pixel row of a tile (256 channels)
constexpr int TILE_BYTES = BP * ROW;              // 16 KB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // dy + x
constexpr int LDS_BYTES = NST * STAGE_BYTES;      // 128 KB
constexpr int LOADS = 4;                          // LDS-DMA instructions per wave and stage (2 dy pieces + 2 x pieces)
constexpr int SUB = 16 * ROW;                     // byte distance of the two 16-pixel sub-steps of a stage
constexpr int HI = 4 * ROW;                       // rows +4 of a transposing read pair
}  // namespace wp

typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_p;

// LDS-DMA from inline asm: invisible to the compiler's LDS alias / waitcnt tracking.  Through the builtin, hipcc treats the
// transposing reads as possible aliases of every outstanding LDS-DMA and drains vmcnt(0) in front of them -- the whole
// prefetch ring would collapse to depth 0.  lds = wave-uniform LDS byte address of the wave's 1-KB destination; the waits
// are counted by hand below.
#define WP_DMA16(voff, base, lds) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0")

template <int N>
__device__ __forceinline__ void wp_wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct WPFrag {
    bf16x8 a[4], b[2];
};

// GEO: pixel-geometry indexing (compile-time: a run-time branch inside the stage would split its basic block and undo the interleave)
template <bool GEO>
__global__ void __launch_bounds__(wp::NTHR, 2) wgrad_pipe_kernel(const WgradParams p)
{
    using namespace wp;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave >> 2, wci = wave & 3;           // 2 x 4 waves: 128 co x 64 ci each
#ifdef IGEMM_STAMPS
    long kstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // dbg_it < 0: whole-kernel sections instead of one stage
#define KSTAMP(i) do { if (p.dbg && p.dbg_it < 0) { __builtin_amdgcn_sched_barrier(0); kstamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define KSTAMP(i) do { } while (0)
#endif
    KSTAMP(0);

    const int ntap_tiles = (p.ntaps + p.tile_taps - 1) / p.tile_taps;
    const int nwg = p.n_co_tiles * p.n_ci_tiles * ntap_tiles;
    int bid;
    long pbeg, pend;
    bool atomic;
    int part;
    wgrad_map(p, nwg, bid, pbeg, pend, atomic, part);
    if (pbeg >= pend) return;      // (slab mode: the host sums only the ranges that hold pixels)
    const int co_tile = bid % p.n_co_tiles;
    const int rest = bid / p.n_co_tiles;
    const int ci_tile = rest % p.n_ci_tiles;
    const int tap0 = (rest / p.n_ci_tiles) * p.tile_taps;        // first tap of this tile
    const int co0 = co_tile * TCO, ci0 = ci_tile * TCI;
    const int cpt = p.tile_taps > 1 ? p.Cin / 8 : 32;              // 16-B chunks per tap inside the tile's 32 column chunks
    // columns of this tile that exist in dw[co][tap][ci] (adjacent taps are adjacent columns)
    const int col_lim = p.tile_taps > 1 ? min(TCI, (p.ntaps - tap0) * p.Cin) : min(TCI, p.Cin - ci0);

    // ---- LDS-DMA pieces.  A wave-instruction covers 2 pixel rows x 512 B; slot (row, c') holds data chunk
    // c = c' ^ ((row & 3) << 2): the transposing reads of a 32-lane half then touch 16 distinct 16-B slots of the 256-B bank row.
    // Piece i of this wave = rows 16 i + 2 wave + {0, 1}; the same rows for the dy and the x tile.
    // Addresses are scalar base + per-lane 32-bit BYTE offset (saddr form): in-kernel stamps showed the 64-bit per-row address
    // arithmetic of wgrad.hip's scheme (two 64-bit multiplies per row and operand, quarter-rate v_mul_*_u32) taking 700-1200 of the
    // ~2000 cycles of a stage.  Slots and strides are < 2^24, so the products are 24-bit multiplies.  The x base is moved back by
    // G = |most negative tap offset| so that every offset is >= 0; rows past the pixel range and taps past the last one read a
    // halo slot / a valid tap instead of a zero line (dy is exactly zero on its halo, so such rows add nothing; columns of
    // missing taps are never stored).
    int row_of[2];
    unsigned a_cb[2], b_cb[2];          // constant byte offsets: channel chunk (+ tap offset + G for x)
    const long G = (long)p.pad * p.x_row_stride + (long)p.pad * p.x_px_stride;
    const char *const dyb = reinterpret_cast<const char *>(p.dy);
    const char *const xb = reinterpret_cast<const char *>(p.x) - 2 * G;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pos = (i * NW + wave) * 64 + lane;
        const int row = pos >> 5, cs = pos & 31;
        const int c = cs ^ ((row & 3) << 2);
        row_of[i] = row;
        int ca = co0 / 8 + c;
        if (ca >= p.Cout_ld / 8) ca = p.Cout_ld / 8 - 1;
        a_cb[i] = (unsigned)ca * 16u;
        const int tt = p.tile_taps > 1 ? c / cpt : 0;                 // tap of this chunk inside the tile
        int cb = p.tile_taps > 1 ? c - tt * cpt : ci0 / 8 + c;
        if (cb >= p.Cin_ld / 8) cb = p.Cin_ld / 8 - 1;
        int tap = tap0 + tt;
        if (tap >= p.ntaps) tap = p.ntaps - 1;                        // columns of a missing tap are never stored
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        b_cb[i] = (unsigned)(2 * ((long)(ky - p.pad) * p.x_row_stride + (long)(kx - p.pad) * p.x_px_stride + G) + cb * 16);
    }
    const unsigned dy_sb = (unsigned)p.dy_px_stride * 2u, x_sb = (unsigned)p.x_px_stride * 2u;   // bytes per pixel slot

    // (n0, oy0, ox0) = pixel coordinates of the first row of the NEXT stage to issue (geometry mode), as wgrad.hip
    int n0 = 0, oy0 = 0, ox0 = 0;
    if constexpr (GEO) {
        const long row = pbeg / p.gW;
        ox0 = (int)(pbeg - row * p.gW);
        n0 = (int)(row / p.gH);
        oy0 = (int)(row - (long)n0 * p.gH);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    long pb_next = pbeg;               // first pixel of the next stage to issue
    // byte offsets of the wave's two piece pairs (dy + x rows 16 i + 2 wave + {0, 1}) of the next stage -- cheap VALU work that the
    // scheduler spreads over the MFMA gaps -- and the stage coordinates advanced by one stage
    auto stage_addr = [&](unsigned (&va)[2], unsigned (&vx)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long pr = pb_next + row_of[i];
            unsigned slot;
            if constexpr (GEO) {
                const unsigned a = (unsigned)(ox0 + row_of[i]);
                const unsigned qx = __umulhi(a, p.mW);
                const unsigned b = (unsigned)oy0 + qx;
                const unsigned qy = __umulhi(b, p.mH);
                slot = __umul24((unsigned)n0 + qy, (unsigned)p.g_img) + __umul24(b - __umul24(qy, (unsigned)p.gH), (unsigned)p.g_row)
                       + __umul24(a - __umul24(qx, (unsigned)p.gW), (unsigned)p.g_px) + (unsigned)p.g_off;
            } else {
                slot = (unsigned)pr;
            }
            slot = pr < pend ? slot : 0u;      // slot 0 is a halo slot of the first image: dy is zero there
            va[i] = __umul24(slot, dy_sb) + a_cb[i];
            vx[i] = __umul24(slot, x_sb) + b_cb[i];
        }
        pb_next += BP;
        if constexpr (GEO) {
            const unsigned a = (unsigned)(ox0 + BP);
            const unsigned qx = __umulhi(a, p.mW);
            const unsigned b = (unsigned)oy0 + qx;
            const unsigned qy = __umulhi(b, p.mH);
            ox0 = (int)(a - __umul24(qx, (unsigned)p.gW));
            oy0 = (int)(b - __umul24(qy, (unsigned)p.gH));
            n0 += (int)qy;
        }
    };
    auto issue_dy = [&](int buf, int i, unsigned va) { WP_DMA16(va, dyb, lds0 + buf * STAGE_BYTES + (i * NW + wave) * 1024); };
    auto issue_x = [&](int buf, int i, unsigned vx) { WP_DMA16(vx, xb, lds0 + buf * STAGE_BYTES + TILE_BYTES + (i * NW + wave) * 1024); };
    auto stage_issue = [&](int buf, int i, unsigned va, unsigned vx) {
        issue_dy(buf, i, va);
        issue_x(buf, i, vx);
    };
    auto stage = [&](int buf) {
        unsigned va[2], vx[2];
        stage_addr(va, vx);
        stage_issue(buf, 0, va[0], vx[0]);
        stage_issue(buf, 1, va[1], vx[1]);
    };

    // ---- transposing fragment reads (wgrad.hip): group g = lane >> 4 supplies rows (g >> 1) * 8 + q (+4 for the second read),
    // 16 columns (g & 1) * 16 + 4 pp of a 32-column MFMA operand; q = (lane >> 2) & 3, pp = lane & 3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int a_rd[4], b_rd[2];
    {
        const int row = (g >> 1) * 8 + q;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ca = (wco * 128 + t * 32 + (g & 1) * 16 + 4 * pp) >> 3;
            a_rd[t] = row * ROW + ((ca ^ ((row & 3) << 2)) << 4) + (pp & 1) * 8;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int cb = (wci * 64 + t * 32 + (g & 1) * 16 + 4 * pp) >> 3;
            b_rd[t] = TILE_BYTES + row * ROW + ((cb ^ ((row & 3) << 2)) << 4) + (pp & 1) * 8;
        }
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    // bias gradient rides along: the dy fragments of the (first tap tile, ci-tile 0) workgroups cover every (pixel, co) exactly
    // once over the grid; the branch is taken ONCE per wave, outside the K loop (a per-lane `if` inside would split its blocks)
    const bool do_bias = p.db != nullptr && tap0 == 0 && ci_tile == 0 && wci == 0;
    auto run = [&](auto bias_tag) {
    constexpr bool BIAS = decltype(bias_tag)::value;
    auto rd = [&](int buf, int sub, WPFrag &f) {
        // issue order = the order in which the MFMAs of the sub-step need the operands: a0, b0, b1, a1, a2, a3
        const char *sb = smem + buf * STAGE_BYTES + sub * SUB;
        auto ra = [&](int t) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(sb + a_rd[t]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(sb + a_rd[t] + HI));
            f.a[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        auto rb = [&](int t) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(sb + b_rd[t]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(sb + b_rd[t] + HI));
            f.b[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        ra(0); rb(0); rb(1); ra(1); ra(2); ra(3);
    };
    auto mm = [&](const WPFrag &f, auto loc, auto hic) {          // MFMAs [lo, hi) of the 8 of a sub-step, row-major over (i, j)
        constexpr int LO = decltype(loc)::value, HIX = decltype(hic)::value;
#pragma unroll
        for (int k = LO; k < HIX; ++k) acc[k >> 1][k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[k >> 1], f.b[k & 1], acc[k >> 1][k & 1], 0, 0, 0);
        if constexpr (BIAS && LO == 0) {     // bias gradient: column sums of the dy fragments (the MFMA operand holds 8 pixels of one co per lane)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const s16x4 *h = reinterpret_cast<const s16x4 *>(&f.a[t]);
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum[t] += __uint_as_float(((unsigned)(unsigned short)h[0][e]) << 16) + __uint_as_float(((unsigned)(unsigned short)h[1][e]) << 16);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>;
    using I6 = std::integral_constant<int, 6>;
    using I7 = std::integral_constant<int, 7>;
    using I8 = std::integral_constant<int, 8>;

    // ---- K loop over stages of 32 pixels (two 16-pixel sub-steps).  Fragment set f0 always holds sub-step 0, f1 sub-step 1:
    //   top of stage s: counted vmcnt (my pieces of stage s+1 have landed), barrier (everyone's are visible; buffer (s-1) % 4 is free)
    //   sub-step 0: MFMAs on f0 | reads of (s, sub-step 1) -> f1   | first half of the DMA of stage s+3
    //   sub-step 1: MFMAs on f1 | reads of (s+1, sub-step 0) -> f0 | second half
    const int nst = (int)((pend - pbeg + BP - 1) / BP);
#pragma unroll
    for (int s0 = 0; s0 < D; ++s0)
        if (s0 < nst) stage(s0);
    if (nst >= D) wp_wait_vmcnt<(D - 1) * LOADS>();
    else wp_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    KSTAMP(1);
    WPFrag f0, f1;
    rd(0, 0, f0);
    int cur = 0, nxt = 1, lbuf = D;
    auto adv = [&]() {
        cur = nxt;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
        lbuf = lbuf + 1 == NST ? 0 : lbuf + 1;
    };
#ifdef IGEMM_STAMPS
    long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WSTAMP(i) do { if (p.dbg && it == p.dbg_it) { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif
    int it = 0;
    // One stage = two 16-pixel sub-steps of eight MFMAs.  Each sub-step has a "read" half (four MFMAs with the 12 transposing
    // reads of the next sub-step and the address arithmetic behind them) and a "DMA" half (four MFMAs with two LDS-DMA between
    // them).  The two waves of a SIMD share its matrix pipe and the per-stage barrier starts them in lock step, so group A
    // (waves 0-3) runs read half, DMA half and group B (SWAP) DMA half, read half: one wave's non-MFMA issue slots then face the
    // other's MFMAs instead of its stalls (stamps: each wave is busy ~1070 cycles per stage, 512 of them issuing MFMAs).
    auto step = [&](auto fullc, auto morec, auto swapc) {
        constexpr bool FULL = decltype(fullc)::value, MORE = decltype(morec)::value, SWAP = decltype(swapc)::value;
        WSTAMP(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(1);
        unsigned sa[2], sx[2];
        if constexpr (FULL) stage_addr(sa, sx);
        auto read_half = [&](WPFrag &cur_f, auto loc, auto hic, auto dordc, int rbuf, int rsub, WPFrag &dst) {
            constexpr bool DO = decltype(dordc)::value;
            if constexpr (DO) rd(rbuf, rsub, dst);
            mm(cur_f, loc, hic);
            if constexpr (DO) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    if constexpr (FULL) __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto dma_half = [&](WPFrag &cur_f, auto loc, int i) {
            constexpr int LO = decltype(loc)::value;
            if constexpr (FULL) {
                issue_dy(lbuf, i, sa[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
            mm(cur_f, std::integral_constant<int, LO>{}, std::integral_constant<int, LO + 2>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FULL) {
                issue_x(lbuf, i, sx[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
            mm(cur_f, std::integral_constant<int, LO + 2>{}, std::integral_constant<int, LO + 4>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (!SWAP) {
            read_half(f0, I0{}, I4{}, std::true_type{}, cur, 1, f1);
            WSTAMP(2);
            dma_half(f0, I4{}, 0);
            WSTAMP(3);
            read_half(f1, I0{}, I4{}, std::integral_constant<bool, MORE>{}, nxt, 0, f0);
            WSTAMP(4);
            dma_half(f1, I4{}, 1);
            WSTAMP(5);
        } else {
            dma_half(f0, I0{}, 0);
            WSTAMP(2);
            read_half(f0, I4{}, I8{}, std::true_type{}, cur, 1, f1);
            WSTAMP(3);
            dma_half(f1, I0{}, 1);
            WSTAMP(4);
            read_half(f1, I4{}, I8{}, std::integral_constant<bool, MORE>{}, nxt, 0, f0);
            WSTAMP(5);
        }
        WSTAMP(6);
    };
    auto loop = [&](auto swapc) {
    for (; it + D < nst; ++it) {                      // stages that issue the DMA of stage it + D
        wp_wait_vmcnt<(D - 2) * LOADS>();
        step(std::true_type{}, std::true_type{}, swapc);
        adv();
    }
    for (; it + 1 < nst; ++it) {                      // nothing left to stage
        if (nst - 2 - it >= D - 2) wp_wait_vmcnt<(D - 2) * LOADS>();
        else wp_wait_vmcnt<0>();
        step(std::false_type{}, std::true_type{}, swapc);
        adv();
    }
    wp_wait_vmcnt<0>();
    step(std::false_type{}, std::false_type{}, swapc);       // last stage: no further reads
    };
    if (wave < NW / 2) loop(std::false_type{});
    else loop(std::true_type{});
    KSTAMP(2);
#ifdef IGEMM_STAMPS
    if (p.dbg && p.dbg_it >= 0 && lane == 0 && blockIdx.x < 512) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.dbg[((long)blockIdx.x * 8 + wave) * 8 + i] = stamp[i];
    }
#endif
#undef WSTAMP

    };
    if (do_bias) run(std::true_type{});
    else run(std::false_type{});

    if (do_bias) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float tot = bsum[t] + __shfl_xor(bsum[t], 32, 64);  // the two k-halves of each co row
            const int co = co0 + wco * 128 + t * 32 + (lane & 31);
            if (lane < 32 && co < p.Cout) atomicAdd(p.db + co, tot);
        }
    }

    KSTAMP(3);
    // ---- output through LDS, 64 co rows at a time ([64 co][256 ci] fp32 = 64 KB): accumulator layout of the 32x32 MFMA is
    // row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col = lane & 31
    const long ldw = (long)p.ntaps * p.Cin;
    const long col0 = p.tile_taps > 1 ? (long)tap0 * p.Cin : (long)tap0 * p.Cin + ci0;
    float *ot = reinterpret_cast<float *>(smem);
    const bool vec_ok = !atomic && (p.Cin & 3) == 0 && ((uintptr_t)p.dw & 15) == 0;
    // workgroups that add into the same tile (different pixel ranges) finish together: each starts at a different row block so
    // that their atomics do not queue on the same addresses
    if (p.slabs) {
        // slab mode: the whole 256 x 256 partial tile, dense, as plain 16-B stores (1 KB per row and wave-instruction) -- fp32 atomics
        // top out at 1.5 TB/s chip-wide and cost a workgroup 60-90 k cycles; wgrad_slab_sum_kernel adds the partials in range order
        float *dst = p.slabs + (long)part * (TCO * TCI);
        auto spass = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            __syncthreads();
            if (wco == (h >> 1)) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    constexpr int i0 = (h & 1) * 2;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            ot[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TCI + wci * 64 + j * 32 + (lane & 31)] = acc[i0 + ii][j][r];
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = k * NTHR + tid, row = idx >> 6, c4 = (idx & 63) * 4;
                *reinterpret_cast<float4 *>(dst + (h * 64 + row) * TCI + c4) = *reinterpret_cast<const float4 *>(ot + row * TCI + c4);
            }
        };
        spass(std::integral_constant<int, 0>{});
        spass(std::integral_constant<int, 1>{});
        spass(std::integral_constant<int, 2>{});
        spass(std::integral_constant<int, 3>{});
        return;
    }
    const int rot = (int)((pbeg / BP) % 127);
    auto pass = [&](auto hc) {          // 64 co rows h * 64 .. of the tile; h is static: the accumulators must stay in registers
        constexpr int h = decltype(hc)::value;
        __syncthreads();   // (first pass: every wave is done reading the stage buffers)
        if (wco == (h >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                constexpr int i0 = (h & 1) * 2;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ot[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TCI + wci * 64 + j * 32 + (lane & 31)] = acc[i0 + ii][j][r];
            }
        }
        __syncthreads();
        if (vec_ok) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = k * NTHR + tid, row = idx >> 6, c4 = (idx & 63) * 4;
                const int co = co0 + h * 64 + row;
                if (co < p.Cout && c4 < col_lim)
                    *reinterpret_cast<float4 *>(p.dw + (long)co * ldw + col0 + c4) = *reinterpret_cast<const float4 *>(ot + row * TCI + c4);
            }
        } else {
#pragma unroll 4
            for (int k0 = 0; k0 < 32; ++k0) {
                const int k = (k0 + (rot >> 2)) & 31;
                const int idx = k * NTHR + tid, row = idx >> 8, c = idx & 255;
                const int co = co0 + h * 64 + row;
                if (co < p.Cout && c < col_lim) {
                    float *o = p.dw + (long)co * ldw + col0 + c;
                    const float v = ot[row * TCI + c];
                    if (atomic) atomicAdd(o, v);
                    else *o = v;
                }
            }
        }
    };
    for (int h0 = 0; h0 < 4; ++h0) {
        switch ((h0 + rot) & 3) {
        case 0: pass(std::integral_constant<int, 0>{}); break;
        case 1: pass(std::integral_constant<int, 1>{}); break;
        case 2: pass(std::integral_constant<int, 2>{}); break;
        default: pass(std::integral_constant<int, 3>{}); break;
        }
    }
#ifdef IGEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KSTAMP(4);
    if (p.dbg && p.dbg_it < 0 && lane == 0 && blockIdx.x < 512) {
        kstamp[5] = (pend - pbeg + BP - 1) / BP;
#pragma unroll
        for (int i = 0; i < 8; ++i) p.dbg[((long)blockIdx.x * 8 + wave) * 8 + i] = kstamp[i];
    }
#endif
#undef KSTAMP
}

// Sum of the partial tiles of slab mode into the packed gradient dw[co][tap][ci]: one thread = 4 consecutive columns of one tile row,
// partials added in range order (fixed: the result does not depend on the order in which the workgroups ran).
__global__ void __launch_bounds__(256) wgrad_slab_sum_kernel(const float *__restrict__ slabs, float *__restrict__ dw, int Cout, int Cin, int ntaps, int n_co_tiles,
                                                             int n_ci_tiles, int tile_taps, int main_tiles, int main_split, int tail_split, int main_ranges,
                                                             int tail_ranges)
{
    using namespace wp;
    const int bid = blockIdx.y;                                   // tile
    const int idx = blockIdx.x * 256 + threadIdx.x;               // float4 inside the tile: row = idx / 64
    const int row = idx >> 6, c4 = (idx & 63) * 4;
    const int co_tile = bid % n_co_tiles, rest = bid / n_co_tiles;
    const int ci_tile = rest % n_ci_tiles, tap0 = (rest / n_ci_tiles) * tile_taps;
    const int co = co_tile * TCO + row;
    const int col_lim = tile_taps > 1 ? min(TCI, (ntaps - tap0) * Cin) : min(TCI, Cin - ci_tile * TCI);
    if (co >= Cout || c4 >= col_lim) return;
    const bool tail = bid >= main_tiles;
    const long first = tail ? (long)main_tiles * main_split + (long)(bid - main_tiles) * tail_split : (long)bid * main_split;
    const int n = tail ? tail_ranges : main_ranges;
    const float *src = slabs + first * (TCO * TCI) + row * TCI + c4;
    float4 a = *reinterpret_cast<const float4 *>(src);
    for (int r = 1; r < n; ++r) {
        const float4 b = *reinterpret_cast<const float4 *>(src + (long)r * (TCO * TCI));
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const long col0 = tile_taps > 1 ? (long)tap0 * Cin : (long)tap0 * Cin + ci_tile * TCI;
    *reinterpret_cast<float4 *>(dw + (long)co * ntaps * Cin + col0 + c4) = a;
}

int wgrad_slab_sum_launch(const WgradParams &p, int tiles, int main_ranges, int tail_ranges, hipStream_t s)
{
    hipLaunchKernelGGL(wgrad_slab_sum_kernel, dim3(wp::TCO * wp::TCI / 4 / 256, (unsigned)tiles), dim3(256), 0, s, (const float *)p.slabs, p.dw, p.Cout, p.Cin, p.ntaps,
                       p.n_co_tiles, p.n_ci_tiles, p.tile_taps, p.seg ? p.main_tiles : tiles, p.seg ? p.main_split : main_ranges, p.seg ? p.tail_split : 1,
                       main_ranges, tail_ranges);
    return check_launch("yolo_wgrad (slab sum)");
}

int wgrad_pipe_launch(const WgradParams &p, dim3 grid, hipStream_t s)
{
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)wgrad_pipe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, wp::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)wgrad_pipe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, wp::LDS_BYTES);
        if (e != hipSuccess) return fail((int)e, "yolo_wgrad: hipFuncSetAttribute(%d B LDS): %s", wp::LDS_BYTES, hipGetErrorString(e));
        attr_done[dev] = true;
    }
    WgradParams q = p;
    debug_stamp_target(&q.dbg, &q.dbg_it);
    if (p.gW) hipLaunchKernelGGL(wgrad_pipe_kernel<true>, grid, dim3(wp::NTHR), wp::LDS_BYTES, s, q);
    else hipLaunchKernelGGL(wgrad_pipe_kernel<false>, grid, dim3(wp::NTHR), wp::LDS_BYTES, s, q);
    return check_launch("yolo_wgrad (pipelined)");
}

}  // namespace yolo
