// Weight-gradient kernel for gfx950, 256 x 256 tile, FOUR waves with 128 x 128 wave tiles (yolo_wgrad_desc.variant = 6):
//     dw[co][tap][ci] (+)= sum_p dy[p][co] * x[p + tapoff(tap)][ci]
// Tiles, staging (pixel-major LDS-DMA, ring of four 32-pixel stages, counted vmcnt waits, one barrier per stage), the
// transposing fragment reads, the schedule over (tile, pixel range) and the output are those of wgrad_pipe.hip (variant 5) --
// the same MFMAs on the same operands in the same order per accumulator, so a tile's partial sums are bit-identical.
// What differs is who computes them.  wgrad_pipe.hip runs eight waves of 128 x 64: 24 transposing reads per 16 MFMAs, two waves
// per SIMD that take turns on its matrix pipe and meet at the stage barrier; stamps put a stage at 1450-1600 cycles for 1024
// cycles of MFMA issue.  Here ONE wave per SIMD owns 128 x 128 outputs:
//   * 256 accumulator registers per lane -- with one wave per SIMD the register file gives 512 per lane, the accumulators sit in
//     its upper half (AGPRs) and the 64 fragment registers + addresses in the lower;
//   * 32 transposing reads per 32 MFMAs (a third fewer LDS bytes per MAC), every MFMA gap of the wave takes one read of the next
//     16-pixel sub-step, the LDS-DMA of the stage three ahead or a piece of its address arithmetic: the matrix pipe of a SIMD
//     is fed by one instruction stream and never waits for the other wave's turn;
//   * four waves at the barrier instead of eight.
#include "wgrad_common.h"

namespace yolo {

namespace ww {
constexpr int TCO = 256, TCI = 256, BP = 32, NST = 4, D = NST - 1, NW = 4, NTHR = NW * 64;
constexpr int ROW = 512;                          // bytes per pixel row of a tile (256 channels)
constexpr int TILE_BYTES = BP * ROW;              // 16 KB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // dy + x
constexpr int LDS_BYTES = NST * STAGE_BYTES;      // 128 KB
constexpr int NP = 4;                             // LDS-DMA pieces (2 rows x 512 B) per wave, operand and stage
constexpr int LOADS = 2 * NP;                     // LDS-DMA instructions per wave and stage
constexpr int SUB = 16 * ROW;                     // byte distance of the two 16-pixel sub-steps of a stage
constexpr int HI = 4 * ROW;                       // rows +4 of a transposing read pair
}  // namespace ww

typedef __attribute__((address_space(3))) s16x4 *ww_lds_s16x4_p;

// LDS-DMA from inline asm (wgrad_pipe.hip: through the builtin hipcc drains vmcnt(0) in front of every transposing read)
#ifdef WW_NO_DMA        // ablation build ("what bounds the weight-gradient K loop"): no operand traffic, wrong results
#define WW_DMA16(voff, base, lds) asm volatile("" ::"v"(voff), "s"(base), "s"(lds))
#else
#define WW_DMA16(voff, base, lds) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0")
#endif

template <int N>
__device__ __forceinline__ void ww_wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct WWFrag {
    bf16x8 a[4], b[4];
};

template <bool GEO>
__global__ void __launch_bounds__(ww::NTHR) wgrad_wide_kernel(const WgradParams p)
{
    using namespace ww;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave >> 1, wci = wave & 1;           // 2 x 2 waves: 128 co x 128 ci each
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
    const int col_lim = p.tile_taps > 1 ? min(TCI, (p.ntaps - tap0) * p.Cin) : min(TCI, p.Cin - ci0);

    // ---- LDS-DMA pieces: layout, swizzle and addressing of wgrad_pipe.hip; piece i of this wave = rows 8 i + 2 wave + {0, 1}, a
    // 32-lane half per row.  The row numbers are wave-uniform: the pixel -> slot arithmetic of both rows of a piece (two
    // multiply-high "small divisions" each in geometry mode) runs on the SCALAR unit, a lane only selects its half's slot and
    // multiplies by the pixel strides -- as per-lane arithmetic (wgrad_pipe.hip) it is 32 quarter-rate multiplies per stage, which
    // a single wave per SIMD would have to issue between its own MFMAs.
    const int half = lane >> 5, cs = lane & 31;
    unsigned a_cb[NP], b_cb[NP];          // constant byte offsets: channel chunk (+ tap offset + G for x)
    const long G = (long)p.pad * p.x_row_stride + (long)p.pad * p.x_px_stride;
    const char *const dyb = reinterpret_cast<const char *>(p.dy);
    const char *const xb = reinterpret_cast<const char *>(p.x) - 2 * G;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = (i * NW + wave) * 2 + half;
        const int c = cs ^ ((row & 3) << 2);
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

    int n0 = 0, oy0 = 0, ox0 = 0;       // pixel coordinates of the first row of the NEXT stage to issue (geometry mode)
    if constexpr (GEO) {
        const long row = pbeg / p.gW;
        ox0 = (int)(pbeg - row * p.gW);
        n0 = (int)(row / p.gH);
        oy0 = (int)(row - (long)n0 * p.gH);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    int rel_next = 0;                  // first pixel of the next stage to issue, relative to pbeg
    const int npx = (int)(pend - pbeg);
    const unsigned slot_base = (unsigned)pbeg;     // (flat mode: the slot is the pixel number)
    auto row_slot = [&](int row) -> unsigned {     // wave-uniform
        unsigned slot;
        if constexpr (GEO) {
            const unsigned a = (unsigned)(ox0 + row);
            const unsigned qx = __umulhi(a, p.mW);
            const unsigned b = (unsigned)oy0 + qx;
            const unsigned qy = __umulhi(b, p.mH);
            slot = ((unsigned)n0 + qy) * (unsigned)p.g_img + (b - qy * (unsigned)p.gH) * (unsigned)p.g_row + (a - qx * (unsigned)p.gW) * (unsigned)p.g_px + (unsigned)p.g_off;
        } else {
            slot = slot_base + (unsigned)(rel_next + row);
        }
        return rel_next + row < npx ? slot : 0u;      // slot 0 is a halo slot of the first image: dy is zero there
    };
    unsigned ns[NP][2];                // slots of the rows of the stage to issue next (wave-uniform: SGPRs)
    auto slots_for = [&](int i) {
        const int row = (i * NW + wave) * 2;
        ns[i][0] = row_slot(row);
        ns[i][1] = row_slot(row + 1);
    };
    auto piece_addr = [&](int i, unsigned &va, unsigned &vx) {
        const unsigned slot = half ? ns[i][1] : ns[i][0];
        va = __umul24(slot, dy_sb) + a_cb[i];
        vx = __umul24(slot, x_sb) + b_cb[i];
    };
    auto stage_advance = [&]() {
        rel_next += BP;
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
    auto issue_dy = [&](int buf, int i, unsigned va) { WW_DMA16(va, dyb, lds0 + buf * STAGE_BYTES + (i * NW + wave) * 1024); };
    auto issue_x = [&](int buf, int i, unsigned vx) { WW_DMA16(vx, xb, lds0 + buf * STAGE_BYTES + TILE_BYTES + (i * NW + wave) * 1024); };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            unsigned va, vx;
            slots_for(i);
            piece_addr(i, va, vx);
            issue_dy(buf, i, va);
            issue_x(buf, i, vx);
        }
        stage_advance();
    };

    // ---- transposing fragment reads (wgrad.hip): group g = lane >> 4 supplies rows (g >> 1) * 8 + q (+4 for the second read),
    // 16 columns (g & 1) * 16 + 4 pp of a 32-column MFMA operand; q = (lane >> 2) & 3, pp = lane & 3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int a_rd[4], b_rd[4];
    {
        const int row = (g >> 1) * 8 + q;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ca = (wco * 128 + t * 32 + (g & 1) * 16 + 4 * pp) >> 3;
            a_rd[t] = row * ROW + ((ca ^ ((row & 3) << 2)) << 4) + (pp & 1) * 8;
            const int cb = (wci * 128 + t * 32 + (g & 1) * 16 + 4 * pp) >> 3;
            b_rd[t] = TILE_BYTES + row * ROW + ((cb ^ ((row & 3) << 2)) << 4) + (pp & 1) * 8;
        }
    }

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    // bias gradient rides along (wgrad_pipe.hip): the branch is taken ONCE per wave, outside the K loop
    const bool do_bias = p.db != nullptr && tap0 == 0 && ci_tile == 0 && wci == 0;
    auto run = [&](auto bias_tag) {
    constexpr bool BIAS = decltype(bias_tag)::value;
#define WW_TR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((ww_lds_s16x4_p)(ptr))
    auto ra = [&](const char *sb, int t, WWFrag &f) {
        const s16x4 lo = WW_TR(sb + a_rd[t]);
        const s16x4 hi = WW_TR(sb + a_rd[t] + HI);
        f.a[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto rb = [&](const char *sb, int t, WWFrag &f) {
        const s16x4 lo = WW_TR(sb + b_rd[t]);
        const s16x4 hi = WW_TR(sb + b_rd[t] + HI);
        f.b[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto rd = [&](int buf, int sub, WWFrag &f) {
#ifdef WW_NO_READS      // ablation build: no fragment reads, the MFMAs keep their first operands
        asm volatile("" : "+v"(f.a[0]), "+v"(f.b[0]));
        return;
#endif
        // issue order = the order in which the MFMAs of the sub-step first need the operands (row-major over (i, j))
        const char *sb = smem + buf * STAGE_BYTES + sub * SUB;
        ra(sb, 0, f); rb(sb, 0, f); rb(sb, 1, f); rb(sb, 2, f); rb(sb, 3, f); ra(sb, 1, f); ra(sb, 2, f); ra(sb, 3, f);
    };
    auto mm = [&](const WWFrag &f, auto loc, auto hic) {          // MFMAs [lo, hi) of the 16 of a sub-step, row-major over (i, j)
        constexpr int LO = decltype(loc)::value, HIX = decltype(hic)::value;
#pragma unroll
        for (int k = LO; k < HIX; ++k) {
#ifdef WW_NO_MFMA       // ablation build: operand traffic and fragment reads only, wrong results
            acc[k >> 2][k & 3][0] += __builtin_bit_cast(float, __builtin_shufflevector(f.a[k >> 2], f.b[k & 3], 0, 8));
#else
            acc[k >> 2][k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[k >> 2], f.b[k & 3], acc[k >> 2][k & 3], 0, 0, 0);
#endif
        }
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
    using I8 = std::integral_constant<int, 8>;
    using I10 = std::integral_constant<int, 10>;
    using I12 = std::integral_constant<int, 12>;
    using I16 = std::integral_constant<int, 16>;

    // ---- K loop over stages of 32 pixels (two 16-pixel sub-steps of 16 MFMAs).  f0 always holds sub-step 0, f1 sub-step 1:
    //   top of stage s: counted vmcnt (my pieces of stage s+1 have landed), barrier (everyone's are visible; buffer (s-1) % 4 is free)
    //   sub-step 0: MFMAs on f0 | 16 reads of (s, sub-step 1) -> f1   | pieces 0, 1 of the DMA of stage s+3
    //   sub-step 1: MFMAs on f1 | 16 reads of (s+1, sub-step 0) -> f0 | pieces 2, 3
    // Inside a sub-step: MFMAs 0-7 with the reads (two per gap), MFMAs 8-15 with two pieces (address arithmetic + dy + x DMA).
    const int nst = (int)((pend - pbeg + BP - 1) / BP);
#pragma unroll
    for (int s0 = 0; s0 < D; ++s0)
        if (s0 < nst) stage(s0);
#pragma unroll
    for (int i = 0; i < NP; ++i) slots_for(i);        // of stage D, which the first step of the loop issues
    stage_advance();
    if (nst >= D) ww_wait_vmcnt<(D - 1) * LOADS>();
    else ww_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    KSTAMP(1);
    WWFrag f0 = {}, f1 = {};
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
    auto step = [&](auto fullc, auto morec) {
        constexpr bool FULL = decltype(fullc)::value, MORE = decltype(morec)::value;
        WSTAMP(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(1);
        // byte offsets of the wave's four piece pairs of stage it + D (a select and two multiply-adds per piece, from the slots the
        // previous step left in SGPRs)
        unsigned sa[NP] = {0, 0, 0, 0}, sx[NP] = {0, 0, 0, 0};
        if constexpr (FULL) {
#pragma unroll
            for (int i = 0; i < NP; ++i) piece_addr(i, sa[i], sx[i]);
        }
        auto read_part = [&](WWFrag &cur_f, auto dordc, int rbuf, int rsub, WWFrag &dst) {         // MFMAs 0-7 + the 16 reads, two per gap
            constexpr bool DO = decltype(dordc)::value;
            if constexpr (DO) rd(rbuf, rsub, dst);
            mm(cur_f, I0{}, I8{});
            if constexpr (DO) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // two LDS reads
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // MFMAs 8-15 with two pieces: ONE LDS-DMA per gap (stamps: two behind each other cost ~80 cycles of issue, more than the 32
        // cycles the matrix pipe works on one MFMA)
        auto dma_part = [&](WWFrag &cur_f, int i0) {
            auto one = [&](auto kc) {
                constexpr int K = decltype(kc)::value;
                mm(cur_f, std::integral_constant<int, K>{}, std::integral_constant<int, K + 1>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            one(I8{});
            if constexpr (FULL) { issue_dy(lbuf, i0, sa[i0]); __builtin_amdgcn_sched_barrier(0); }
            one(std::integral_constant<int, 9>{});
            if constexpr (FULL) { issue_x(lbuf, i0, sx[i0]); __builtin_amdgcn_sched_barrier(0); }
            one(I10{});
            if constexpr (FULL) { issue_dy(lbuf, i0 + 1, sa[i0 + 1]); __builtin_amdgcn_sched_barrier(0); }
            one(std::integral_constant<int, 11>{});
            if constexpr (FULL) { issue_x(lbuf, i0 + 1, sx[i0 + 1]); __builtin_amdgcn_sched_barrier(0); }
            // MFMAs 12-15: their gaps take the scalar pixel -> slot arithmetic of two pieces of the stage the NEXT step issues
            if constexpr (FULL) {
                slots_for(i0);
                slots_for(i0 + 1);
                if (i0 == 2) stage_advance();
            }
            mm(cur_f, I12{}, I16{});
            if constexpr (FULL) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x004, GEO ? 24 : 6, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        read_part(f0, std::true_type{}, cur, 1, f1);
        WSTAMP(2);
        dma_part(f0, 0);
        WSTAMP(3);
        read_part(f1, std::integral_constant<bool, MORE>{}, nxt, 0, f0);
        WSTAMP(4);
        dma_part(f1, 2);
        WSTAMP(5);

    };
    for (; it + D < nst; ++it) {                      // stages that issue the DMA of stage it + D
        WSTAMP(7);
#ifdef IGEMM_STAMPS
        if (p.dbg && it == p.dbg_it + 100) { __builtin_amdgcn_sched_barrier(0); stamp[6] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }      // 100 stages later
        if (p.dbg && p.dbg_it >= 100000 && it >= p.dbg_it - 100000 && it < p.dbg_it - 100000 + 8) {      // tops of eight consecutive stages
            const long t = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (it - (p.dbg_it - 100000) == k) stamp[k] = t;
        }
#endif
        ww_wait_vmcnt<(D - 2) * LOADS>();
        step(std::true_type{}, std::true_type{});
        adv();
    }
    for (; it + 1 < nst; ++it) {                      // nothing left to stage
        if (nst - 2 - it >= D - 2) ww_wait_vmcnt<(D - 2) * LOADS>();
        else ww_wait_vmcnt<0>();
        step(std::false_type{}, std::true_type{});
        adv();
    }
    ww_wait_vmcnt<0>();
    step(std::false_type{}, std::false_type{});       // last stage: no further reads
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
    // ---- output through LDS, 128 co rows at a time ([128 co][256 ci] fp32 = 128 KB = the stage ring): accumulator layout of the
    // 32x32 MFMA is row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col = lane & 31
    const long ldw = (long)p.ntaps * p.Cin;
    const long col0 = p.tile_taps > 1 ? (long)tap0 * p.Cin : (long)tap0 * p.Cin + ci0;
    float *ot = reinterpret_cast<float *>(smem);
    const bool vec_ok = !atomic && (p.Cin & 3) == 0 && ((uintptr_t)p.dw & 15) == 0;
    const int rot = (int)((pbeg / BP) % 127);
    float *slab = p.slabs ? p.slabs + (long)part * (TCO * TCI) : nullptr;
    auto pass = [&](auto hc) {          // 128 co rows h * 128 .. of the tile; h is static: the accumulators must stay in registers
        constexpr int h = decltype(hc)::value;
        __syncthreads();   // (first pass: every wave is done reading the stage buffers)
        if (wco == h) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ot[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TCI + wci * 128 + j * 32 + (lane & 31)] = acc[i][j][r];
        }
        __syncthreads();
        if (slab) {
            // slab mode: the partial tile, dense, as plain 16-B stores; wgrad_slab_sum_kernel adds the partials in range order
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                const int idx = k * NTHR + tid, row = idx >> 6, c4 = (idx & 63) * 4;
                *reinterpret_cast<float4 *>(slab + (h * 128 + row) * TCI + c4) = *reinterpret_cast<const float4 *>(ot + row * TCI + c4);
            }
        } else if (vec_ok) {
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                const int idx = k * NTHR + tid, row = idx >> 6, c4 = (idx & 63) * 4;
                const int co = co0 + h * 128 + row;
                if (co < p.Cout && c4 < col_lim)
                    *reinterpret_cast<float4 *>(p.dw + (long)co * ldw + col0 + c4) = *reinterpret_cast<const float4 *>(ot + row * TCI + c4);
            }
        } else {
            // workgroups that add into the same tile (different pixel ranges) finish together: each starts at a different row so
            // that their atomics do not queue on the same addresses
#pragma unroll 4
            for (int k0 = 0; k0 < 128; ++k0) {
                const int row = (k0 + rot) & 127;
                const int co = co0 + h * 128 + row;
                if (co < p.Cout && tid < col_lim) {
                    float *o = p.dw + (long)co * ldw + col0 + tid;
                    const float v = ot[row * TCI + tid];
                    if (atomic) atomicAdd(o, v);
                    else *o = v;
                }
            }
        }
    };
    if (rot & 1) {
        pass(std::integral_constant<int, 1>{});
        pass(std::integral_constant<int, 0>{});
    } else {
        pass(std::integral_constant<int, 0>{});
        pass(std::integral_constant<int, 1>{});
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

int wgrad_wide_launch(const WgradParams &p, dim3 grid, hipStream_t s)
{
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)wgrad_wide_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ww::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)wgrad_wide_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ww::LDS_BYTES);
        if (e != hipSuccess) return fail((int)e, "yolo_wgrad: hipFuncSetAttribute(%d B LDS): %s", ww::LDS_BYTES, hipGetErrorString(e));
        attr_done[dev] = true;
    }
    WgradParams q = p;
    debug_stamp_target(&q.dbg, &q.dbg_it);
    if (p.gW) hipLaunchKernelGGL(wgrad_wide_kernel<true>, grid, dim3(ww::NTHR), ww::LDS_BYTES, s, q);
    else hipLaunchKernelGGL(wgrad_wide_kernel<false>, grid, dim3(ww::NTHR), ww::LDS_BYTES, s, q);
    return check_launch("yolo_wgrad (wide)");
}

}  // namespace yolo
