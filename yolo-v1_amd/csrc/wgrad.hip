// Weight-gradient kernel for gfx950:  dw[co][tap][ci] (+)= sum_p dy[p][co] * x[p + tapoff(tap)][ci]
//
// Both operands are K-strided for this product (the reduction index is the pixel, the contiguous
// axis is the channel), so the tiles are staged pixel-major by LDS-DMA exactly as they lie in HBM
// and the MFMA fragments are fetched with gfx950's transposing LDS read (ds_read_b64_tr_b16): no
// transposed copy of an activation is ever written.
//
// "Flat" pixel indexing: p runs over EVERY pixel slot of the zero-haloed dy buffer (halo included).
// dy is exactly zero on its halo, so those slots add nothing, and for a stride-1 conv the matching
// input pixel of tap (ky,kx) is simply p + (ky-pad)*row_stride + (kx-pad): no div/mod, no bounds
// checks, perfectly contiguous 16-B loads.  The price is (H+2)(W+2)/(HW) extra pixels (1.04x at
// 112x112 .. 1.65x at 7x7).  A stride-2 conv runs on a zero-stuffed dy of the input's geometry.
// Callers keep a guard band of zeros (>= row_stride+1 pixels) in front of and behind every
// activation buffer so that shifted reads of halo slots stay inside the allocation.
//
// Tile: 128 co x 128 ci per (tap), 64 pixels per K step, 4 waves (2x2), v_mfma_f32_32x32x16_bf16.
// Split over pixel ranges (blockIdx.y) with fp32 atomics into the packed gradient.
// Replaces the weight-gradient half of aten convolution_backward / addmm backward for
// src/yolo/models.py:47-84,239-245,313-332.
#include "wgrad_common.h"

namespace yolo {

__device__ uint4 g_zero_line[16];  // 256 B of zeros: source for out-of-range rows

// the same instruction from inline asm, invisible to the compiler's LDS alias / waitcnt tracking (wgrad256_kernel);
// lds = wave-uniform LDS byte address of the wave's 1-KB destination
#define DMA16(gptr, lds) \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds) : "memory", "m0")

constexpr int WG_T = 128;                 // tile edge (co and ci)
constexpr int WG_BP = 64;                 // pixels per K step
constexpr int WG_TILE_BYTES = WG_BP * WG_T * 2;   // 16 KB
constexpr int WG_STAGE_BYTES = 2 * WG_TILE_BYTES; // dy + x
constexpr int WG_LDS_BYTES = 2 * WG_STAGE_BYTES;  // double buffered: 64 KB

typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;

struct WFrag {
    bf16x8 a[2], b[2];
};

// fragments of one 16-pixel sub-step: 4 + 4 transposing reads (rows +4 lie `a_hi` / `b_hi` bytes further)
template <bool BIAS>
__device__ __forceinline__ void wave_load(const char *sb, const int (&a_rd)[2], const int (&b_rd)[2], int a_off, int a_hi, int b_off, int b_hi, WFrag &f,
                                          float (&bsum)[2])
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sb + a_rd[t] + a_off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sb + a_rd[t] + a_off + a_hi));
        f.a[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        if (BIAS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[t] += __uint_as_float(((unsigned)(unsigned short)lo[e]) << 16) + __uint_as_float(((unsigned)(unsigned short)hi[e]) << 16);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sb + b_rd[t] + b_off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sb + b_rd[t] + b_off + b_hi));
        f.b[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
}

// one 64-pixel K step of a wave's 64 x 64 tile, software-pipelined: the reads of sub-step ks+1 are issued in front of
// the MFMAs of sub-step ks.  BIAS is a template flag and the caller branches ONCE per workgroup: a per-lane `if` inside
// this loop splits it into basic blocks and the compiler then waits lgkmcnt(0) in front of every MFMA group.
template <bool BIAS, int A_KS, int A_HI, int B_KS, int B_HI>
__device__ __forceinline__ void wave_step(const char *sb, const int (&a_rd)[2], const int (&b_rd)[2], f32x16 (&acc)[2][2], float (&bsum)[2])
{
    WFrag cur, nxt;
    wave_load<BIAS>(sb, a_rd, b_rd, 0, A_HI, 0, B_HI, cur, bsum);
#pragma unroll
    for (int ks = 0; ks < WG_BP / 16; ++ks) {
        if (ks + 1 < WG_BP / 16) wave_load<BIAS>(sb, a_rd, b_rd, (ks + 1) * A_KS, A_HI, (ks + 1) * B_KS, B_HI, nxt, bsum);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[i], cur.b[j], acc[i][j], 0, 0, 0);
        cur = nxt;
    }
}

__global__ void __launch_bounds__(256, 2) wgrad_kernel(const WgradParams p)
{
    // Two SEPARATE LDS objects, one per pipeline stage, and a K loop unrolled by two so that every
    // access names its stage statically: hipcc then knows the LDS-DMA writes of stage t+1 cannot alias
    // the transposing reads of stage t and does not drain vmcnt(0) in front of the first ds_read
    // (with one array and a runtime stage index it serialised load and compute).
    __shared__ __attribute__((aligned(16))) char bufA[WG_STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char bufB[WG_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave >> 1, wci = wave & 1;

    const int ntap_tiles = p.pair_taps ? (p.ntaps + 1) / 2 : p.ntaps;
    const int nwg = p.n_co_tiles * p.n_ci_tiles * ntap_tiles;
    int bid;
    long pbeg, pend;
    bool atomic;
    int part_unused;
    wgrad_map(p, nwg, bid, pbeg, pend, atomic, part_unused);
    if (pbeg >= pend) return;   // empty range of a rounded-up split (the whole workgroup leaves)
    // co-tile fastest, then ci-tile, then tap: neighbours share the x tile of one tap
    const int co_tile = bid % p.n_co_tiles;
    const int rest = bid / p.n_co_tiles;
    const int ci_tile = rest % p.n_ci_tiles;
    const int tap_tile = rest / p.n_ci_tiles;
    const int tap = p.pair_taps ? 2 * tap_tile : tap_tile;          // first (or only) tap of this tile
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const long tap_off = (long)(ky - p.pad) * p.x_row_stride + (long)(kx - p.pad) * p.x_px_stride;
    const int ky2 = (tap + 1) / p.KW, kx2 = (tap + 1) - ky2 * p.KW;   // pair mode: the tap of tile columns 64..127
    const long tap_off2 = (long)(ky2 - p.pad) * p.x_row_stride + (long)(kx2 - p.pad) * p.x_px_stride;
    const bool tap2_ok = tap + 1 < p.ntaps;
    const int co0 = co_tile * WG_T, ci0 = ci_tile * WG_T;
    // columns of this tile that exist in dw[co][tap][ci]
    const int col_lim = p.pair_taps ? (tap2_ok ? 2 * p.Cin : p.Cin) : p.Cin - ci0;


    // ---- LDS-DMA sources.  A wave-instruction covers 4 pixel rows x 256 B.  Slot (row, c') holds
    // data chunk c = c' ^ ((row&3)<<2): the transposing reads of one 32-lane half then touch 16
    // distinct 16-B slots of the 256-B bank row (conflict-free).
    int row_of[4], a_coff[4], b_coff[4];
    bool b_second[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pos = (i * 4 + wave) * 64 + lane;
        const int row = pos >> 4, cs = pos & 15;
        const int c = cs ^ ((row & 3) << 2);
        row_of[i] = row;
        int ca = co0 / 8 + c, cb = ci0 / 8 + c;
        if (ca >= p.Cout_ld / 8) ca = p.Cout_ld / 8 - 1;
        b_second[i] = false;
        if (p.pair_taps) {                      // chunks 0..7 -> tap, 8..15 -> tap + 1 (Cin == 64: 8 chunks per tap)
            b_second[i] = c >= 8;
            cb = c & 7;
        }
        if (cb >= p.Cin_ld / 8) cb = p.Cin_ld / 8 - 1;
        a_coff[i] = ca * 8;
        b_coff[i] = cb * 8;
    }
    const bf16_t *zline = reinterpret_cast<const bf16_t *>(g_zero_line) + (lane & 15) * 8;

    // slot[i]: buffer slot of the lane's i-th pixel row in the NEXT stage to be issued (-1: past the range -> zeros).
    // With a pixel index the lookups for stage t+1 are issued right behind the LDS-DMA of stage t and have a whole K
    // step to arrive.
    // (n0, oy0, ox0) = pixel coordinates of the first row of the NEXT stage to issue, carried in scalars; a lane's rows lie
    // < 64 pixels further, so their coordinates follow with two multiply-high "small divisions" each -- no index table, no
    // extra memory instruction (a table cost 7-17 % on the 56x56 / 112x112 layers, whichever way it was read).
    int n0 = 0, oy0 = 0, ox0 = 0;
    if (p.gW) {
        const long row = pbeg / p.gW;
        ox0 = (int)(pbeg - row * p.gW);
        n0 = (int)(row / p.gH);
        oy0 = (int)(row - (long)n0 * p.gH);
    }
    auto stage = [&](char *sb, long pb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long pr = pb + row_of[i];
            long slot;
            if (p.gW) {
                const unsigned a = (unsigned)(ox0 + row_of[i]);
                const unsigned qx = __umulhi(a, p.mW);
                const unsigned b = (unsigned)oy0 + qx;
                const unsigned qy = __umulhi(b, p.mH);
                slot = (long)(n0 + (int)qy) * p.g_img + (int)(b - qy * p.gH) * p.g_row + (int)(a - qx * p.gW) * p.g_px + p.g_off;
            } else {
                slot = pr;
            }
            const bool ok = pr < pend;
            const bf16_t *sa = ok ? p.dy + slot * p.dy_px_stride + a_coff[i] : zline;
            const bf16_t *sx = (ok && !(b_second[i] && !tap2_ok)) ? p.x + slot * p.x_px_stride + (b_second[i] ? tap_off2 : tap_off) + b_coff[i] : zline;
            GLDS16(sa, sb + (i * 4 + wave) * 1024);
            GLDS16(sx, sb + WG_TILE_BYTES + (i * 4 + wave) * 1024);
        }
        if (p.gW) {   // advance the scalar coordinates by one stage (64 pixels)
            const unsigned a = (unsigned)(ox0 + WG_BP);
            const unsigned qx = __umulhi(a, p.mW);
            const unsigned b = (unsigned)oy0 + qx;
            const unsigned qy = __umulhi(b, p.mH);
            ox0 = (int)(a - qx * p.gW);
            oy0 = (int)(b - qy * p.gH);
            n0 += (int)qy;
        }
    };

    // ---- transposing fragment reads (see header): group g = lane>>4, q = (lane>>2)&3, pp = lane&3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = (g >> 1) * 8 + q;
        const int ca = (wco * 64 + t * 32 + (g & 1) * 16 + 4 * pp) >> 3;
        const int cb = (wci * 64 + t * 32 + (g & 1) * 16 + 4 * pp) >> 3;
        a_rd[t] = row * 256 + ((ca ^ ((row & 3) << 2)) << 4) + (pp & 1) * 8;
        b_rd[t] = WG_TILE_BYTES + row * 256 + ((cb ^ ((row & 3) << 2)) << 4) + (pp & 1) * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // bias gradient rides along: the dy fragments of the (tap 0, ci-tile 0) workgroups cover every
    // (pixel, co) exactly once over the grid, so summing them costs no extra HBM pass
    const bool do_bias = p.db != nullptr && tap == 0 && ci_tile == 0 && wci == 0;
    float bsum[2] = {0.0f, 0.0f};

    auto run = [&](auto bias_tag) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        stage(bufA, pbeg);
        for (long pb = pbeg; pb < pend; pb += 2 * WG_BP) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // stage A landed for every wave; stage B is free
            if (pb + WG_BP < pend) stage(bufB, pb + WG_BP);
            wave_step<BIAS, 4096, 1024, 4096, 1024>(bufA, a_rd, b_rd, acc, bsum);
            if (pb + WG_BP >= pend) break;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (pb + 2 * WG_BP < pend) stage(bufA, pb + 2 * WG_BP);
            wave_step<BIAS, 4096, 1024, 4096, 1024>(bufB, a_rd, b_rd, acc, bsum);
        }
    };
    if (do_bias) run(std::true_type{});
    else run(std::false_type{});

    if (do_bias) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float tot = bsum[t] + __shfl_xor(bsum[t], 32, 64);  // the two k-halves of each co row
            const int co = co0 + wco * 64 + t * 32 + (lane & 31);
            if (lane < 32 && co < p.Cout) atomicAdd(p.db + co, tot);
        }
    }

    // ---- output through LDS, one half of the co rows at a time ([64 co][128 ci] fp32 = stage A): the accumulator layout
    // (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31) would give 4-B stores / atomics in 128-B pieces; from LDS
    // a wave-instruction covers 1 KB of one dw row as 16-B stores, or 256 contiguous bytes per atomic instruction.
    const long ldw = (long)p.ntaps * p.Cin;
    float *ot = reinterpret_cast<float *>(bufA);
    const bool vec_ok = !atomic && (p.Cin & 3) == 0 && ((uintptr_t)p.dw & 15) == 0;
    float ssq = 0.0f;      // p.sumsq: squares of the values this thread stores (the host requires the vector store path for it)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();   // (first pass: every wave is done reading the stage buffers)
        if (wco == h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ot[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * WG_T + wci * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
        }
        __syncthreads();
        if (vec_ok) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = it * 256 + tid, row = idx >> 5, c4 = (idx & 31) * 4;
                const int co = co0 + h * 64 + row, ci = ci0 + c4;
                if (co < p.Cout && c4 < col_lim) {   // Cin % 4 == 0: a quad is inside or outside as a whole
                    const float4 v = *reinterpret_cast<const float4 *>(ot + row * WG_T + c4);
                    *reinterpret_cast<float4 *>(p.dw + (long)co * ldw + (long)tap * p.Cin + ci) = v;
                    ssq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
            }
        } else {
#pragma unroll 4
            for (int it = 0; it < 32; ++it) {
                const int idx = it * 256 + tid, row = idx >> 7, c = idx & 127;
                const int co = co0 + h * 64 + row, ci = ci0 + c;
                if (co < p.Cout && c < col_lim) {
                    float *o = p.dw + (long)co * ldw + (long)tap * p.Cin + ci;
                    const float v = ot[row * WG_T + c];
                    if (atomic) atomicAdd(o, v);
                    else *o = v;
                }
            }
        }
    }
    if (p.sumsq) {       // wave-uniform branch; one fp64 atomic per workgroup
        double s = (double)ssq;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        __syncthreads();
        double *part = reinterpret_cast<double *>(bufA);
        if (lane == 0) part[wave] = s;
        __syncthreads();
        if (tid == 0) atomicAdd(p.sumsq, part[0] + part[1] + part[2] + part[3]);
    }
}

// 8-wave form of the 128 x 128 kernel: a wave owns 32 co x 64 ci (one A fragment, two B fragments per 16-pixel sub-step),
// four waves per SIMD with two resident workgroups -- more waves to cover the per-step LDS latency and barrier, at 1.5x the
// fragment reads per MFMA (yolo_wgrad_desc.variant = 4).
template <bool BIAS>
__device__ __forceinline__ void wave_step8(const char *sb, const int (&a_rd)[2], const int (&b_rd)[2], f32x16 (&acc)[1][2], float (&bsum)[2])
{
    bf16x8 ca, cb[2], na, nb[2];
    auto load = [&](int ks, bf16x8 &a, bf16x8 (&b)[2]) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sb + a_rd[0] + ks * 4096));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sb + a_rd[0] + ks * 4096 + 1024));
        a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        if (BIAS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[0] += __uint_as_float(((unsigned)(unsigned short)lo[e]) << 16) + __uint_as_float(((unsigned)(unsigned short)hi[e]) << 16);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const s16x4 l2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sb + b_rd[t] + ks * 4096));
            const s16x4 h2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sb + b_rd[t] + ks * 4096 + 1024));
            b[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    load(0, ca, cb);
#pragma unroll
    for (int ks = 0; ks < WG_BP / 16; ++ks) {
        if (ks + 1 < WG_BP / 16) load(ks + 1, na, nb);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca, cb[j], acc[0][j], 0, 0, 0);
        ca = na; cb[0] = nb[0]; cb[1] = nb[1];
    }
}

__global__ void __launch_bounds__(512, 2) wgrad8_kernel(const WgradParams p)
{
    // Two SEPARATE LDS objects, one per pipeline stage, and a K loop unrolled by two so that every
    // access names its stage statically: hipcc then knows the LDS-DMA writes of stage t+1 cannot alias
    // the transposing reads of stage t and does not drain vmcnt(0) in front of the first ds_read
    // (with one array and a runtime stage index it serialised load and compute).
    __shared__ __attribute__((aligned(16))) char bufA[WG_STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char bufB[WG_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave >> 1, wci = wave & 1;

    const int ntap_tiles = p.pair_taps ? (p.ntaps + 1) / 2 : p.ntaps;
    const int nwg = p.n_co_tiles * p.n_ci_tiles * ntap_tiles;
    int bid;
    long pbeg, pend;
    bool atomic;
    int part_unused;
    wgrad_map(p, nwg, bid, pbeg, pend, atomic, part_unused);
    if (pbeg >= pend) return;   // empty range of a rounded-up split (the whole workgroup leaves)
    // co-tile fastest, then ci-tile, then tap: neighbours share the x tile of one tap
    const int co_tile = bid % p.n_co_tiles;
    const int rest = bid / p.n_co_tiles;
    const int ci_tile = rest % p.n_ci_tiles;
    const int tap_tile = rest / p.n_ci_tiles;
    const int tap = p.pair_taps ? 2 * tap_tile : tap_tile;          // first (or only) tap of this tile
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const long tap_off = (long)(ky - p.pad) * p.x_row_stride + (long)(kx - p.pad) * p.x_px_stride;
    const int ky2 = (tap + 1) / p.KW, kx2 = (tap + 1) - ky2 * p.KW;   // pair mode: the tap of tile columns 64..127
    const long tap_off2 = (long)(ky2 - p.pad) * p.x_row_stride + (long)(kx2 - p.pad) * p.x_px_stride;
    const bool tap2_ok = tap + 1 < p.ntaps;
    const int co0 = co_tile * WG_T, ci0 = ci_tile * WG_T;
    // columns of this tile that exist in dw[co][tap][ci]
    const int col_lim = p.pair_taps ? (tap2_ok ? 2 * p.Cin : p.Cin) : p.Cin - ci0;


    // ---- LDS-DMA sources.  A wave-instruction covers 4 pixel rows x 256 B.  Slot (row, c') holds
    // data chunk c = c' ^ ((row&3)<<2): the transposing reads of one 32-lane half then touch 16
    // distinct 16-B slots of the 256-B bank row (conflict-free).
    int row_of[2], a_coff[2], b_coff[2];
    bool b_second[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pos = (i * 8 + wave) * 64 + lane;
        const int row = pos >> 4, cs = pos & 15;
        const int c = cs ^ ((row & 3) << 2);
        row_of[i] = row;
        int ca = co0 / 8 + c, cb = ci0 / 8 + c;
        if (ca >= p.Cout_ld / 8) ca = p.Cout_ld / 8 - 1;
        b_second[i] = false;
        if (p.pair_taps) {                      // chunks 0..7 -> tap, 8..15 -> tap + 1 (Cin == 64: 8 chunks per tap)
            b_second[i] = c >= 8;
            cb = c & 7;
        }
        if (cb >= p.Cin_ld / 8) cb = p.Cin_ld / 8 - 1;
        a_coff[i] = ca * 8;
        b_coff[i] = cb * 8;
    }
    const bf16_t *zline = reinterpret_cast<const bf16_t *>(g_zero_line) + (lane & 15) * 8;

    // slot[i]: buffer slot of the lane's i-th pixel row in the NEXT stage to be issued (-1: past the range -> zeros).
    // With a pixel index the lookups for stage t+1 are issued right behind the LDS-DMA of stage t and have a whole K
    // step to arrive.
    // (n0, oy0, ox0) = pixel coordinates of the first row of the NEXT stage to issue, carried in scalars; a lane's rows lie
    // < 64 pixels further, so their coordinates follow with two multiply-high "small divisions" each -- no index table, no
    // extra memory instruction (a table cost 7-17 % on the 56x56 / 112x112 layers, whichever way it was read).
    int n0 = 0, oy0 = 0, ox0 = 0;
    if (p.gW) {
        const long row = pbeg / p.gW;
        ox0 = (int)(pbeg - row * p.gW);
        n0 = (int)(row / p.gH);
        oy0 = (int)(row - (long)n0 * p.gH);
    }
    auto stage = [&](char *sb, long pb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long pr = pb + row_of[i];
            long slot;
            if (p.gW) {
                const unsigned a = (unsigned)(ox0 + row_of[i]);
                const unsigned qx = __umulhi(a, p.mW);
                const unsigned b = (unsigned)oy0 + qx;
                const unsigned qy = __umulhi(b, p.mH);
                slot = (long)(n0 + (int)qy) * p.g_img + (int)(b - qy * p.gH) * p.g_row + (int)(a - qx * p.gW) * p.g_px + p.g_off;
            } else {
                slot = pr;
            }
            const bool ok = pr < pend;
            const bf16_t *sa = ok ? p.dy + slot * p.dy_px_stride + a_coff[i] : zline;
            const bf16_t *sx = (ok && !(b_second[i] && !tap2_ok)) ? p.x + slot * p.x_px_stride + (b_second[i] ? tap_off2 : tap_off) + b_coff[i] : zline;
            GLDS16(sa, sb + (i * 8 + wave) * 1024);
            GLDS16(sx, sb + WG_TILE_BYTES + (i * 8 + wave) * 1024);
        }
        if (p.gW) {   // advance the scalar coordinates by one stage (64 pixels)
            const unsigned a = (unsigned)(ox0 + WG_BP);
            const unsigned qx = __umulhi(a, p.mW);
            const unsigned b = (unsigned)oy0 + qx;
            const unsigned qy = __umulhi(b, p.mH);
            ox0 = (int)(a - qx * p.gW);
            oy0 = (int)(b - qy * p.gH);
            n0 += (int)qy;
        }
    };

    // ---- transposing fragment reads (see header): group g = lane>>4, q = (lane>>2)&3, pp = lane&3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int a_rd[2], b_rd[2];     // a_rd[1] unused: a wave owns 32 output channels here
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = (g >> 1) * 8 + q;
        const int ca = (wco * 32 + (g & 1) * 16 + 4 * pp) >> 3;
        const int cb = (wci * 64 + t * 32 + (g & 1) * 16 + 4 * pp) >> 3;
        a_rd[t] = row * 256 + ((ca ^ ((row & 3) << 2)) << 4) + (pp & 1) * 8;
        b_rd[t] = WG_TILE_BYTES + row * 256 + ((cb ^ ((row & 3) << 2)) << 4) + (pp & 1) * 8;
    }

    f32x16 acc[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;

    // bias gradient rides along: the dy fragments of the (tap 0, ci-tile 0) workgroups cover every
    // (pixel, co) exactly once over the grid, so summing them costs no extra HBM pass
    const bool do_bias = p.db != nullptr && tap == 0 && ci_tile == 0 && wci == 0;
    float bsum[2] = {0.0f, 0.0f};

    auto run = [&](auto bias_tag) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        stage(bufA, pbeg);
        for (long pb = pbeg; pb < pend; pb += 2 * WG_BP) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // stage A landed for every wave; stage B is free
            if (pb + WG_BP < pend) stage(bufB, pb + WG_BP);
            wave_step8<BIAS>(bufA, a_rd, b_rd, acc, bsum);
            if (pb + WG_BP >= pend) break;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (pb + 2 * WG_BP < pend) stage(bufA, pb + 2 * WG_BP);
            wave_step8<BIAS>(bufB, a_rd, b_rd, acc, bsum);
        }
    };
    if (do_bias) run(std::true_type{});
    else run(std::false_type{});

    if (do_bias) {
#pragma unroll
        for (int t = 0; t < 1; ++t) {
            const float tot = bsum[t] + __shfl_xor(bsum[t], 32, 64);  // the two k-halves of each co row
            const int co = co0 + wco * 32 + (lane & 31);
            if (lane < 32 && co < p.Cout) atomicAdd(p.db + co, tot);
        }
    }

    // ---- output through LDS, one half of the co rows at a time ([64 co][128 ci] fp32 = stage A): the accumulator layout
    // (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31) would give 4-B stores / atomics in 128-B pieces; from LDS
    // a wave-instruction covers 1 KB of one dw row as 16-B stores, or 256 contiguous bytes per atomic instruction.
    const long ldw = (long)p.ntaps * p.Cin;
    float *ot = reinterpret_cast<float *>(bufA);
    const bool vec_ok = !atomic && (p.Cin & 3) == 0 && ((uintptr_t)p.dw & 15) == 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();   // (first pass: every wave is done reading the stage buffers)
        if ((wco >> 1) == h) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ot[((wco & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * WG_T + wci * 64 + j * 32 + (lane & 31)] = acc[0][j][r];
        }
        __syncthreads();
        if (vec_ok) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = it * 512 + tid, row = idx >> 5, c4 = (idx & 31) * 4;
                const int co = co0 + h * 64 + row, ci = ci0 + c4;
                if (co < p.Cout && c4 < col_lim)   // Cin % 4 == 0: a quad is inside or outside as a whole
                    *reinterpret_cast<float4 *>(p.dw + (long)co * ldw + (long)tap * p.Cin + ci) = *reinterpret_cast<const float4 *>(ot + row * WG_T + c4);
            }
        } else {
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int idx = it * 512 + tid, row = idx >> 7, c = idx & 127;
                const int co = co0 + h * 64 + row, ci = ci0 + c;
                if (co < p.Cout && c < col_lim) {
                    float *o = p.dw + (long)co * ldw + (long)tap * p.Cin + ci;
                    const float v = ot[row * WG_T + c];
                    if (atomic) atomicAdd(o, v);
                    else *o = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 256 co x 128 ci variant, 8 waves (4 x 2, the same 64 x 64 wave tile), THREE 48-KB stages.
// The 128 x 128 kernel above keeps at most one 32-KB stage per workgroup in flight (two workgroups per
// CU): a K step computes for ~0.45 us while a stage takes 1.5-2 us to arrive, so the loop runs at the
// LDS-DMA latency, not at the MFMA rate (measured 31 % of peak on the 3x3 layers).  Here two stages
// (96 KB per CU) are in flight behind the one being consumed and every byte staged feeds twice the
// MFMA work; waits are counted (vmcnt(6) = "everything but the newest stage") and the barrier is the
// raw s_barrier, so a wave never drains the queue.  One static LDS array per stage + a loop unrolled by
// three keep every access's stage static (see the note in wgrad_kernel).
constexpr int W2_TCO = 256, W2_TCI = 128;
constexpr int W2_A_BYTES = WG_BP * W2_TCO * 2;     // 32 KB
constexpr int W2_B_BYTES = WG_BP * W2_TCI * 2;     // 16 KB
constexpr int W2_STAGE = W2_A_BYTES + W2_B_BYTES;  // 48 KB

template <bool STG>
__global__ void __launch_bounds__(512, 1) wgrad256_kernel(const WgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 3 stages x 48 KB
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave >> 1, wci = wave & 1;

    const int nwg = p.n_co_tiles * p.n_ci_tiles * p.ntaps;
    int bid;
    long pbeg, pend;
    bool atomic;
    int part_unused;
    wgrad_map(p, nwg, bid, pbeg, pend, atomic, part_unused);
    if (pbeg >= pend) return;   // empty range of a rounded-up split (the whole workgroup leaves)
    const int co_tile = bid % p.n_co_tiles;
    const int rest = bid / p.n_co_tiles;
    const int ci_tile = rest % p.n_ci_tiles;
    const int tap = rest / p.n_ci_tiles;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const long tap_off = (long)(ky - p.pad) * p.x_row_stride + (long)(kx - p.pad) * p.x_px_stride;
    const int co0 = co_tile * W2_TCO, ci0 = ci_tile * W2_TCI;


    // ---- LDS-DMA sources: dy rows are 512 B (2 pixel rows per wave-instruction), x rows 256 B (4 per instruction);
    // slot (row, c') holds chunk c = c' ^ ((row&3)<<2) in both (see wgrad_kernel)
    int a_row[4], a_coff[4], b_row[2], b_coff[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pos = (i * 8 + wave) * 64 + lane;
        const int row = pos >> 5, cs = pos & 31;
        int ca = co0 / 8 + (cs ^ ((row & 3) << 2));
        if (ca >= p.Cout_ld / 8) ca = p.Cout_ld / 8 - 1;
        a_row[i] = row;
        a_coff[i] = ca * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pos = (i * 8 + wave) * 64 + lane;
        const int row = pos >> 4, cs = pos & 15;
        int cb = ci0 / 8 + (cs ^ ((row & 3) << 2));
        if (cb >= p.Cin_ld / 8) cb = p.Cin_ld / 8 - 1;
        b_row[i] = row;
        b_coff[i] = cb * 8;
    }
    const bf16_t *zline = reinterpret_cast<const bf16_t *>(g_zero_line) + (lane & 15) * 8;

    auto stage = [&](int st, long pb) {
        const unsigned sb = lds0 + st * W2_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long pr = pb + a_row[i];
            const bf16_t *sa = pr < pend ? p.dy + pr * p.dy_px_stride + a_coff[i] : zline;
            DMA16(sa, sb + (i * 8 + wave) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long pr = pb + b_row[i];
            const bf16_t *sx = pr < pend ? p.x + pr * p.x_px_stride + tap_off + b_coff[i] : zline;
            DMA16(sx, sb + W2_A_BYTES + (i * 8 + wave) * 1024);
        }
    };

    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = (g >> 1) * 8 + q;
        const int ca = (wco * 64 + t * 32 + (g & 1) * 16 + 4 * pp) >> 3;
        const int cb = (wci * 64 + t * 32 + (g & 1) * 16 + 4 * pp) >> 3;
        a_rd[t] = row * 512 + ((ca ^ ((row & 3) << 2)) << 4) + (pp & 1) * 8;
        b_rd[t] = W2_A_BYTES + row * 256 + ((cb ^ ((row & 3) << 2)) << 4) + (pp & 1) * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const bool do_bias = p.db != nullptr && tap == 0 && ci_tile == 0 && wci == 0;
    float bsum[2] = {0.0f, 0.0f};

    // The LDS-DMA is issued from inline asm (DMA16): hipcc's own waitcnt pass otherwise drains vmcnt(0) in front of the
    // first transposing read of every loop iteration (it cannot bound the age of the stage being read across the
    // back-edge), which empties the pipeline every third step.  So ordering is all explicit here: wait until only the
    // newest stage (6 instructions per wave) may still be in flight, meet the other waves, and fence the compiler.
#define W2_SYNC(more)                                                 \
    do {                                                              \
        if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         \
        __builtin_amdgcn_s_barrier();                                 \
        asm volatile("" ::: "memory");                                \
    } while (0)

    const long nsteps = pbeg < pend ? (pend - pbeg + WG_BP - 1) / WG_BP : 0;
    auto run = [&](auto bias_tag) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        if constexpr (STG) {
            // ---- staggered two-phase schedule (the one of igemm.hip's MFMA_16x16x32_STAGGER): every K step is an L phase (all
            // 32 transposing reads of the step -> registers) and an M phase (its 16 MFMAs) between raw barriers; waves 0..3 and
            // 4..7 sit pairwise on the same SIMDs and run ONE PHASE APART, so one wave feeds the matrix pipe while its partner
            // reads LDS.  Stage s is issued by everybody in slot 2(s-2) and drained (counted vmcnt) at the end of slot 2s-1.
            auto read_all = [&](int st, WFrag (&f)[4]) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wave_load<BIAS>(smem + st * W2_STAGE, a_rd, b_rd, ks * 8192, 2048, ks * 4096, 1024, f[ks], bsum);
            };
            auto mfma_all = [&](WFrag (&f)[4]) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks].a[i], f[ks].b[j], acc[i][j], 0, 0, 0);
            };
            auto wait_stage = [&](long newer) {   // newer = younger stages this wave has already issued
                if (newer >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            };
#define W2_BAR()                                   \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)
            const bool grpB = wave >= 4;
            const long nkk = nsteps;
            if (nkk > 0) stage(0, pbeg);
            if (nkk > 1) stage(1, pbeg + WG_BP);
            wait_stage(nkk - 1 < 1 ? nkk - 1 : 1);
            W2_BAR();
            WFrag f[4];
            int rd = 0;
            if (!grpB) {
                int ld = 2;
                for (long it = 0; it < nkk; ++it) {
                    read_all(rd, f);                                                    // L(it)
                    if (it + 2 < nkk) stage(ld, pbeg + (it + 2) * WG_BP);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    W2_BAR();
                    __builtin_amdgcn_s_setprio(1);
                    mfma_all(f);                                                        // M(it)
                    __builtin_amdgcn_s_setprio(0);
                    { const long left = nkk - 2 - it; wait_stage(left < 0 ? 0 : left); }   // stage it+1 landed
                    W2_BAR();
                    rd = rd + 1 == 3 ? 0 : rd + 1;
                    ld = ld + 1 == 3 ? 0 : ld + 1;
                }
                __builtin_amdgcn_s_barrier();
            } else {
                if (2 < nkk) stage(2, pbeg + 2 * WG_BP);                                // slot 0
                W2_BAR();
                int ld = 0;
                for (long it = 0; it < nkk; ++it) {
                    read_all(rd, f);                                                    // L(it)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    { const long left = nkk - 2 - it; wait_stage(left < 0 ? 0 : left); }
                    W2_BAR();
                    if (it + 3 < nkk) stage(ld, pbeg + (it + 3) * WG_BP);               // M(it)
                    __builtin_amdgcn_s_setprio(1);
                    mfma_all(f);
                    __builtin_amdgcn_s_setprio(0);
                    W2_BAR();
                    rd = rd + 1 == 3 ? 0 : rd + 1;
                    ld = ld + 1 == 3 ? 0 : ld + 1;
                }
            }
#undef W2_BAR
        } else {
        if (nsteps > 0) stage(0, pbeg);
        if (nsteps > 1) stage(1, pbeg + WG_BP);
        for (long t = 0; t < nsteps; t += 3) {
            W2_SYNC(t + 1 < nsteps);
            if (t + 2 < nsteps) stage(2, pbeg + (t + 2) * WG_BP);
            wave_step<BIAS, 8192, 2048, 4096, 1024>(smem, a_rd, b_rd, acc, bsum);
            if (t + 1 >= nsteps) break;
            W2_SYNC(t + 2 < nsteps);
            if (t + 3 < nsteps) stage(0, pbeg + (t + 3) * WG_BP);
            wave_step<BIAS, 8192, 2048, 4096, 1024>(smem + W2_STAGE, a_rd, b_rd, acc, bsum);
            if (t + 2 >= nsteps) break;
            W2_SYNC(t + 3 < nsteps);
            if (t + 4 < nsteps) stage(1, pbeg + (t + 4) * WG_BP);
            wave_step<BIAS, 8192, 2048, 4096, 1024>(smem + 2 * W2_STAGE, a_rd, b_rd, acc, bsum);
        }
        }
    };
    if (do_bias) run(std::true_type{});
    else run(std::false_type{});
#undef W2_SYNC

    if (do_bias) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float tot = bsum[t] + __shfl_xor(bsum[t], 32, 64);
            const int co = co0 + wco * 64 + t * 32 + (lane & 31);
            if (lane < 32 && co < p.Cout) atomicAdd(p.db + co, tot);
        }
    }
    const long ldw = (long)p.ntaps * p.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wci * 64 + j * 32 + (lane & 31);
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co >= p.Cout) continue;
                float *o = p.dw + (long)co * ldw + (long)tap * p.Cin + ci;
                if (atomic) atomicAdd(o, acc[i][j][r]);
                else *o = acc[i][j][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// Weight + bias gradient of the 7x7 / stride-2 / pad-3 stem (3 input channels stored NHWC4, 64 outputs;
// src/yolo/models.py:49).  No channel-contiguous reduction axis exists for Cin = 3, so the generic kernel
// needed a 1.4 GB row-unfolded copy of the input (yolo_im2col_rows); here the unfolding happens in the
// LDS READ ADDRESSES instead:
//   D[co][(ky, kx, c)] += sum_px dy[px][co] * x[2*oy + ky][2*ox + kx][c]
// workgroup = 8 x 16 output pixels of one image.  dy tile [128 px][64 co] and the raw input patch
// [21 rows][40 px][4 ch] are staged by LDS-DMA; both MFMA operands come from ds_read_b64_tr_b16, whose
// per-lane addresses make "row px, 16 consecutive (kx, c) columns" a plain 32-byte run of the patch.
// 4 waves = the 4 blocks of 16 output channels; 14 column blocks (7 ky x 2 halves of the kx-padded-to-8
// row) x 4 pixel blocks of 32 -> 56 v_mfma_f32_16x16x32_bf16 per wave and tile.  A workgroup walks over
// many tiles (double-buffered) and leaves ONE partial [64][7][8][4] (+64 bias sums) in a scratch buffer;
// stem_wgrad_reduce_kernel sums the partials in a fixed order (deterministic, no atomics) straight into
// the OIHW gradient.  HBM-bound: dy (411 MB at batch 64) is read exactly once.
constexpr int ST_TH = 8, ST_TW = 16;           // tile of output pixels
constexpr int ST_DY_BYTES = ST_TH * ST_TW * 64 * 2;   // 16 KB
constexpr int ST_PW = 40, ST_PH = 21;          // patch pitch (pixels) and rows: 2*8+5, 2*16+5 (+kx pad) <= 40
constexpr int ST_X_BYTES = 7 * 1024;           // 7 DMA wave-instructions >= 21*40*8 = 6720 B
constexpr int ST_STAGE = ST_DY_BYTES + ST_X_BYTES;
constexpr int ST_COLS = 7 * 8 * 4;             // 224 packed columns per output channel
constexpr int ST_PART = 64 * ST_COLS + 64;     // floats per workgroup partial (weights + bias)

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct StemWgradParams {
    const bf16_t *x;       // NHWC4, halo 3
    const bf16_t *dy;      // NHWC 64 channels, any halo (dy_off = offset of pixel (0,0))
    float *part;
    int N, tiles_x, tiles_y, ntiles;
    long x_img_stride, dy_img_stride;
    int x_row_stride, dy_row_stride, dy_off;
    // POOLED: `dy` is the layer's un-pooled ACTIVATION and dpool the gradient of the 2x2-pooled map; the gradient tile is
    // rebuilt on the fly (arg-max of each window gets dpool * LeakyReLU'), exactly as yolo_maxpool2_bwd_lrelu would write it
    const bf16_t *dpool;
    long dp_img_stride;
    int dp_row_stride, dp_off;
    float slope;
    // CODES (MODE 2): `dy` is the POOLED activation (geometry of dpool) and codes the window positions of the maxima that
    // yolo_conv_stem7_fwd(pool2 = 3) left: uint16 per (pooled pixel, 8 channels) at (pooled element address) / 8
    const unsigned short *codes;
};

template <int MODE>     // 0: dy is the gradient; 1: rebuilt from the un-pooled activation + dpool; 2: from the pooled activation + codes + dpool
__global__ void __launch_bounds__(256, 3) stem_wgrad_kernel(const StemWgradParams p)
{
    constexpr bool POOLED = MODE != 0;
    __shared__ __attribute__((aligned(16))) char bufA[ST_STAGE];
    __shared__ __attribute__((aligned(16))) char bufB[ST_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- DMA sources.  dy slot (k, c') holds 16-B chunk c = c' ^ s(k), s(k) = 2*bit1(k) + 4*bit3(k): the 32 lanes
    // of a transposing read (4 rows x 2 row-groups x 4 column quads) then cover all 64 banks once.
    int dy_off[4], x_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = (i * 4 + wave) * 64 + lane;
        const int k = slot >> 3, cs = slot & 7;
        const int c = cs ^ ((((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2));
        dy_off[i] = (k >> 4) * p.dy_row_stride + (k & 15) * 64 + c * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int slot = (i * 4 + wave) * 64 + lane;          // 16-B slots of the patch: 20 per row
        if (slot >= ST_PH * (ST_PW / 2)) slot = ST_PH * (ST_PW / 2) - 1;
        x_off[i] = (slot / (ST_PW / 2)) * p.x_row_stride + (slot % (ST_PW / 2)) * 8;
    }
    auto stage = [&](char *sb, int tile) {
        const int tx = tile % p.tiles_x, r = tile / p.tiles_x;
        const int ty = r % p.tiles_y, n = r / p.tiles_y;
        const bf16_t *dyb = p.dy + (long)n * p.dy_img_stride + (long)(ty * ST_TH) * p.dy_row_stride + tx * ST_TW * 64 + p.dy_off;
        const bf16_t *xb = p.x + (long)n * p.x_img_stride + (long)(ty * ST_TH * 2) * p.x_row_stride + tx * ST_TW * 2 * 4;
        if (!POOLED) {
#pragma unroll
            for (int i = 0; i < 4; ++i) GLDS16(dyb + dy_off[i], sb + (i * 4 + wave) * 1024);
        }
        GLDS16(xb + x_off[0], sb + ST_DY_BYTES + wave * 1024);
        if (wave < 3) GLDS16(xb + x_off[1], sb + ST_DY_BYTES + (4 + wave) * 1024);
    };
    // POOLED: thread = (2x2 window w of the 8x16 tile, 8-channel chunk c): registers hold the window's four activation
    // vectors and the pooled gradient of the tile fetched one iteration ahead
    const int pw = tid >> 3, pc = tid & 7, pwy = pw >> 3, pwx = pw & 7;
    uint4 ry[4], rg;
    unsigned rcode = 0;
    auto load_regs = [&](int tile) {
        const int tx = tile % p.tiles_x, r = tile / p.tiles_x;
        const int ty = r % p.tiles_y, n = r / p.tiles_y;
        const long pa = (long)n * p.dp_img_stride + (long)(ty * (ST_TH / 2) + pwy) * p.dp_row_stride + (tx * (ST_TW / 2) + pwx) * 64 + p.dp_off + pc * 8;
        if constexpr (MODE == 2) {
            ry[0] = *reinterpret_cast<const uint4 *>(p.dy + pa);       // pooled activation: same geometry as dpool
            rcode = p.codes[pa >> 3];
        } else {
            const bf16_t *yb = p.dy + (long)n * p.dy_img_stride + (long)(ty * ST_TH + 2 * pwy) * p.dy_row_stride + (tx * ST_TW + 2 * pwx) * 64 + p.dy_off + pc * 8;
            ry[0] = *reinterpret_cast<const uint4 *>(yb);
            ry[1] = *reinterpret_cast<const uint4 *>(yb + 64);
            ry[2] = *reinterpret_cast<const uint4 *>(yb + p.dy_row_stride);
            ry[3] = *reinterpret_cast<const uint4 *>(yb + p.dy_row_stride + 64);
        }
        rg = *reinterpret_cast<const uint4 *>(p.dpool + pa);
    };
    auto build = [&](char *sb) {
        const unsigned yv[4][4] = {{ry[0].x, ry[0].y, ry[0].z, ry[0].w}, {ry[1].x, ry[1].y, ry[1].z, ry[1].w}, {ry[2].x, ry[2].y, ry[2].z, ry[2].w},
                                   {ry[3].x, ry[3].y, ry[3].z, ry[3].w}};
        const unsigned gv[4] = {rg.x, rg.y, rg.z, rg.w};
        unsigned o[4][4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {           // the two bf16 of a dword
                float v[4], gg;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = hf ? __uint_as_float(yv[k][d] & 0xffff0000u) : __uint_as_float(yv[k][d] << 16);
                gg = hf ? __uint_as_float(gv[d] & 0xffff0000u) : __uint_as_float(gv[d] << 16);
                int am = 0;
                float m = v[0];
                if constexpr (MODE == 2) {
                    am = (int)((rcode >> (2 * (2 * d + hf))) & 3u);     // v[0] = the pooled activation = the maximum itself
                } else {
                    if (v[1] > m) { m = v[1]; am = 1; }    // first maximum in (0,0),(0,1),(1,0),(1,1) order, like pool.hip
                    if (v[2] > m) { m = v[2]; am = 2; }
                    if (v[3] > m) { m = v[3]; am = 3; }
                }
                const unsigned bits = (unsigned)f32_to_bf16(gg * (m > 0.0f ? 1.0f : p.slope));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned w16 = am == k ? bits : 0u;
                    o[k][d] = hf ? (o[k][d] | (w16 << 16)) : w16;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int px = (2 * pwy + (k >> 1)) * ST_TW + 2 * pwx + (k & 1);
            const int cs = pc ^ ((((px >> 1) & 1) << 1) | (((px >> 3) & 1) << 2));
            *reinterpret_cast<uint4 *>(sb + px * 128 + (cs << 4)) = uint4{o[k][0], o[k][1], o[k][2], o[k][3]};
        }
    };

    // ---- fragment addresses: group g = lane>>4 (pixel sub-block of 8), q = (lane>>2)&3 (row), pp = lane&3 (column quad)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int a_lo[4], a_hi[4], b_lo[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int k0 = kb * 32 + g * 8 + q, k1 = k0 + 4;
        const int ch = wave * 2 + (pp >> 1);
        a_lo[kb] = k0 * 128 + ((ch ^ ((((k0 >> 1) & 1) << 1) | (((k0 >> 3) & 1) << 2))) << 4) + (pp & 1) * 8;
        a_hi[kb] = k1 * 128 + ((ch ^ ((((k1 >> 1) & 1) << 1) | (((k1 >> 3) & 1) << 2))) << 4) + (pp & 1) * 8;
        b_lo[kb] = ST_DY_BYTES + ((2 * (k0 >> 4)) * ST_PW + 2 * (k0 & 15) + pp) * 8;   // pixel k0+4 lies 4 columns = 64 B further
    }

    f32x4 acc[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float bsum = 0.0f;

    typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
    auto compute = [&](const char *sb) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sb + a_lo[kb]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sb + a_hi[kb]));
            const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum += __uint_as_float(((unsigned)(unsigned short)lo[e]) << 16) + __uint_as_float(((unsigned)(unsigned short)hi[e]) << 16);
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const char *bp = sb + b_lo[kb] + (ky * ST_PW + 4 * h) * 8;
                    const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bp));
                    const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bp + 64));
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[ky * 2 + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[ky * 2 + h], 0, 0, 0);
                }
        }
    };

    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile < p.ntiles) {
        stage(bufA, tile);
        if (POOLED) load_regs(tile);
    }
    while (tile < p.ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (POOLED) build(bufA);
        __syncthreads();
        if (tile + G < p.ntiles) {
            stage(bufB, tile + G);
            if (POOLED) load_regs(tile + G);
        }
        compute(bufA);
        tile += G;
        if (tile >= p.ntiles) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (POOLED) build(bufB);
        __syncthreads();
        if (tile + G < p.ntiles) {
            stage(bufA, tile + G);
            if (POOLED) load_regs(tile + G);
        }
        compute(bufB);
        tile += G;
    }

    // ---- partial: D row (co) = 4*(lane>>4) + reg, column = lane & 15 of block (ky, h)
    float *out = p.part + (long)blockIdx.x * ST_PART;
#pragma unroll
    for (int nb = 0; nb < 14; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(wave * 16 + 4 * g + r) * ST_COLS + nb * 16 + (lane & 15)] = acc[nb][r];
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < 16) out[64 * ST_COLS + wave * 16 + lane] = bsum;
}

// dw[co][c][ky][kx] = sum_g part[g][co][ky][kx][c]  (c < 3, kx < 7), db[co] = sum_g part[g][64*224 + co];
// workgroup = 16 outputs x 16 slices of the partial list
__global__ void __launch_bounds__(256) stem_wgrad_reduce_kernel(const float *__restrict__ part, int G, float *__restrict__ dw, float *__restrict__ db)
{
    __shared__ float red[16][17];
    const int j = blockIdx.x * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
    float s = 0.0f;
    if (j < ST_PART)
        for (int gi = sl; gi < G; gi += 16) s += part[(long)gi * ST_PART + j];
    red[sl][threadIdx.x & 15] = s;
    __syncthreads();
    if (sl == 0 && j < ST_PART) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        if (j >= 64 * ST_COLS) {
            if (db) db[j - 64 * ST_COLS] = t;
        } else {
            const int c = j & 3, kx = (j >> 2) & 7, ky = (j >> 5) % 7, co = j / ST_COLS;
            if (c < 3 && kx < 7) dw[((co * 3 + c) * 7 + ky) * 7 + kx] = t;
        }
    }
}

// db[c] += sum_p dy[p][c]  (bias gradient).  HBM-bound column sum: a workgroup covers w <= 256
// 8-channel chunks (one 16-B load per thread) x R = 256/w pixel rows per pass, so every wave reads
// whole contiguous pixel rows whatever the channel count; LDS reduction over R, one fp32 atomic per
// channel per workgroup.
__global__ void __launch_bounds__(256) colsum_kernel(const bf16_t *__restrict__ dy, long P, int px_stride, int C, int w, long p_per_blk, float *__restrict__ db)
{
    const int nchunks = (C + 7) >> 3;
    const int R = 256 / w;
    const int col = threadIdx.x % w, r = threadIdx.x / w;
    const int c8 = blockIdx.x * w + col;
    // grid-stride over pixel rows: all workgroups sweep the buffer together (DRAM-page friendly),
    // 4 independent 16-B loads in flight per thread
    const long step = (long)gridDim.y * R;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto acc8 = [&](const uint4 &v) {
        s[0] += __uint_as_float(v.x << 16); s[1] += __uint_as_float(v.x & 0xffff0000u);
        s[2] += __uint_as_float(v.y << 16); s[3] += __uint_as_float(v.y & 0xffff0000u);
        s[4] += __uint_as_float(v.z << 16); s[5] += __uint_as_float(v.z & 0xffff0000u);
        s[6] += __uint_as_float(v.w << 16); s[7] += __uint_as_float(v.w & 0xffff0000u);
    };
    if (r < R && c8 < nchunks) {
        const bf16_t *base = dy + c8 * 8;
        long pr = (long)blockIdx.y * R + r;
        for (; pr + 3 * step < P; pr += 4 * step) {
            const uint4 v0 = *reinterpret_cast<const uint4 *>(base + pr * px_stride);
            const uint4 v1 = *reinterpret_cast<const uint4 *>(base + (pr + step) * px_stride);
            const uint4 v2 = *reinterpret_cast<const uint4 *>(base + (pr + 2 * step) * px_stride);
            const uint4 v3 = *reinterpret_cast<const uint4 *>(base + (pr + 3 * step) * px_stride);
            acc8(v0); acc8(v1); acc8(v2); acc8(v3);
        }
        for (; pr < P; pr += step) acc8(*reinterpret_cast<const uint4 *>(base + pr * px_stride));
    }
    __shared__ float red[256][9];
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = s[k];
    __syncthreads();
    if (r == 0 && c8 < nchunks) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = 0.0f;
            for (int rr = 0; rr < R; ++rr) t += red[rr * w + col][k];
            if (c8 * 8 + k < C) atomicAdd(db + c8 * 8 + k, t);
        }
    }
}

}  // namespace yolo

using namespace yolo;

static int wgrad_stem7_impl(const void *x_nhwc4, const void *dy, int N, int Ho, int Wo, long x_img_stride, int x_row_stride, long dy_img_stride, int dy_row_stride,
                            int dy_off, const void *dpool, long dp_img_stride, int dp_row_stride, int dp_off, float slope, float *dw_oihw, float *db, float *scratch,
                            long scratch_elems, yolo_stream_t stream, const void *codes);

YOLO_API int yolo_wgrad_stem7(const void *x_nhwc4, const void *dy, int N, int Ho, int Wo, long x_img_stride, int x_row_stride, long dy_img_stride,
                              int dy_row_stride, int dy_off, float *dw_oihw, float *db, float *scratch, long scratch_elems, yolo_stream_t stream)
{
    return wgrad_stem7_impl(x_nhwc4, dy, N, Ho, Wo, x_img_stride, x_row_stride, dy_img_stride, dy_row_stride, dy_off, nullptr, 0, 0, 0, 1.0f, dw_oihw, db, scratch,
                            scratch_elems, stream, nullptr);
}

YOLO_API int yolo_wgrad_stem7_pooled(const void *x_nhwc4, const void *y_full, int N, int Ho, int Wo, long x_img_stride, int x_row_stride, long y_img_stride,
                                     int y_row_stride, int y_off, const void *dpool, long dp_img_stride, int dp_row_stride, int dp_off, float slope, float *dw_oihw,
                                     float *db, float *scratch, long scratch_elems, yolo_stream_t stream)
{
    if (!dpool || (dp_row_stride & 7) || (dp_img_stride & 7) || (dp_off & 7) || ((uintptr_t)dpool & 15))
        return fail(YOLO_E_ARG, "yolo_wgrad_stem7_pooled: dpool must be a 16-B aligned NHWC buffer with strides in multiples of 8 elements");
    return wgrad_stem7_impl(x_nhwc4, y_full, N, Ho, Wo, x_img_stride, x_row_stride, y_img_stride, y_row_stride, y_off, dpool, dp_img_stride, dp_row_stride, dp_off, slope,
                            dw_oihw, db, scratch, scratch_elems, stream, nullptr);
}

YOLO_API int yolo_wgrad_stem7_codes(const void *x_nhwc4, const void *y_pooled, const void *codes, int N, int Ho, int Wo, long x_img_stride, int x_row_stride,
                                    const void *dpool, long dp_img_stride, int dp_row_stride, int dp_off, float slope, float *dw_oihw, float *db, float *scratch,
                                    long scratch_elems, yolo_stream_t stream)
{
    if (!dpool || !codes || !y_pooled || (dp_row_stride & 7) || (dp_img_stride & 7) || (dp_off & 7) || (((uintptr_t)dpool | (uintptr_t)y_pooled) & 15))
        return fail(YOLO_E_ARG, "yolo_wgrad_stem7_codes: y_pooled / dpool must be 16-B aligned NHWC buffers of one geometry with strides in multiples of 8 elements");
    return wgrad_stem7_impl(x_nhwc4, y_pooled, N, Ho, Wo, x_img_stride, x_row_stride, dp_img_stride, dp_row_stride, dp_off, dpool, dp_img_stride, dp_row_stride, dp_off,
                            slope, dw_oihw, db, scratch, scratch_elems, stream, codes);
}

static int wgrad_stem7_impl(const void *x_nhwc4, const void *dy, int N, int Ho, int Wo, long x_img_stride, int x_row_stride, long dy_img_stride, int dy_row_stride,
                            int dy_off, const void *dpool, long dp_img_stride, int dp_row_stride, int dp_off, float slope, float *dw_oihw, float *db, float *scratch,
                            long scratch_elems, yolo_stream_t stream, const void *codes)
{
    if (!x_nhwc4 || !dy || !dw_oihw || !scratch || N <= 0 || Ho <= 0 || Wo <= 0) return fail(YOLO_E_ARG, "yolo_wgrad_stem7: bad argument");
    if ((Ho % ST_TH) || (Wo % ST_TW)) return fail(YOLO_E_UNSUPPORTED, "yolo_wgrad_stem7: output %dx%d is not a multiple of %dx%d (use yolo_im2col_rows + yolo_wgrad)", Ho, Wo, ST_TH, ST_TW);
    if ((x_row_stride & 7) || (x_img_stride & 7) || (dy_row_stride & 7) || (dy_img_stride & 7) || (dy_off & 7) || ((uintptr_t)x_nhwc4 & 15) || ((uintptr_t)dy & 15))
        return fail(YOLO_E_UNSUPPORTED, "yolo_wgrad_stem7: strides must be multiples of 8 elements and pointers 16-B aligned");
    StemWgradParams p{};
    p.x = (const bf16_t *)x_nhwc4; p.dy = (const bf16_t *)dy; p.part = scratch;
    p.N = N; p.tiles_x = Wo / ST_TW; p.tiles_y = Ho / ST_TH;
    const long nt = (long)N * p.tiles_x * p.tiles_y;
    if (nt > 0x7fffffffL) return fail(YOLO_E_UNSUPPORTED, "yolo_wgrad_stem7: too many tiles");
    p.ntiles = (int)nt;
    p.x_img_stride = x_img_stride; p.dy_img_stride = dy_img_stride;
    p.x_row_stride = x_row_stride; p.dy_row_stride = dy_row_stride; p.dy_off = dy_off;
    p.dpool = (const bf16_t *)dpool; p.dp_img_stride = dp_img_stride; p.dp_row_stride = dp_row_stride; p.dp_off = dp_off; p.slope = slope;
    p.codes = (const unsigned short *)codes;
    long G = std::min<long>(nt, 768);
    G = std::min<long>(G, scratch_elems / ST_PART);
    if (G < 1) return fail(YOLO_E_ARG, "yolo_wgrad_stem7: scratch must hold at least %d floats", ST_PART);
    if (dpool && codes) hipLaunchKernelGGL(stem_wgrad_kernel<2>, dim3((unsigned)G), dim3(256), 0, STRM(stream), p);
    else if (dpool) hipLaunchKernelGGL(stem_wgrad_kernel<1>, dim3((unsigned)G), dim3(256), 0, STRM(stream), p);
    else hipLaunchKernelGGL(stem_wgrad_kernel<0>, dim3((unsigned)G), dim3(256), 0, STRM(stream), p);
    if (int rc = check_launch("yolo_wgrad_stem7")) return rc;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((ST_PART + 15) / 16), dim3(256), 0, STRM(stream), (const float *)scratch, (int)G, dw_oihw, db);
    return check_launch("yolo_wgrad_stem7(reduce)");
}

static int wgrad_run(const yolo_wgrad_desc *d, const void *x, const void *dy, float *dw, float *db, yolo_stream_t stream, long *query);

YOLO_API int yolo_wgrad(const yolo_wgrad_desc *d, const void *x, const void *dy, float *dw, float *db, yolo_stream_t stream)
{
    return wgrad_run(d, x, dy, dw, db, stream, nullptr);
}

YOLO_API int yolo_wgrad_slab_floats(const yolo_wgrad_desc *d, long *floats)
{
    if (!floats) return fail(YOLO_E_ARG, "yolo_wgrad_slab_floats: null pointer");
    *floats = 0;
    static float dummy[4];      // the schedule does not depend on the operand addresses; nothing is launched
    return wgrad_run(d, dummy, dummy, dummy, nullptr, nullptr, floats);
}

static int wgrad_run(const yolo_wgrad_desc *d, const void *x, const void *dy, float *dw, float *db, yolo_stream_t stream, long *query)
{
    if (!d || !x || !dy || (!dw && !db)) return fail(YOLO_E_ARG, "yolo_wgrad: null pointer");
    if (d->P <= 0 || d->Cout <= 0 || d->Cin <= 0 || d->KH <= 0 || d->KW <= 0 || d->split < 0) return fail(YOLO_E_ARG, "yolo_wgrad: bad descriptor");
    if ((d->dy_px_stride & 7) || (d->x_px_stride & 7) || (d->x_row_stride & 7) || d->dy_px_stride < d->Cout || d->x_px_stride < d->Cin)
        return fail(YOLO_E_UNSUPPORTED, "yolo_wgrad: pixel strides must be multiples of 8 elements and cover the channels");
    hipStream_t s = STRM(stream);
    if (dw) {
        WgradParams p{};
        p.x = (const bf16_t *)x; p.dy = (const bf16_t *)dy; p.dw = dw; p.db = db;
        p.P = d->P;
        if (d->geo_W < 0 || d->geo_H < 0 || (d->geo_W > 0 && (d->geo_H <= 0 || d->geo_W > 65536 || d->geo_H > 65536)))
            return fail(YOLO_E_ARG, "yolo_wgrad: bad pixel geometry %d x %d", d->geo_W, d->geo_H);
        if (d->geo_W > 0) {
            if (d->variant == 2 || d->variant == 3) return fail(YOLO_E_UNSUPPORTED, "yolo_wgrad: the 256x128 variants have no pixel-geometry mode");
            if (d->P % ((long)d->geo_W * d->geo_H)) return fail(YOLO_E_ARG, "yolo_wgrad: P = %ld is not a whole number of %d x %d images", (long)d->P, d->geo_W, d->geo_H);
            p.gW = d->geo_W; p.gH = d->geo_H;
            p.g_img = d->geo_img_slots; p.g_row = d->geo_row_slots; p.g_px = d->geo_px_slots; p.g_off = d->geo_slot0;
            p.mW = (unsigned)(((1ull << 32) + d->geo_W - 1) / d->geo_W);   // 2^32 for W = 1 wraps to 0: handled by W == 1 -> q = a
            p.mH = (unsigned)(((1ull << 32) + d->geo_H - 1) / d->geo_H);
            if (d->geo_W == 1 || d->geo_H == 1) return fail(YOLO_E_UNSUPPORTED, "yolo_wgrad: pixel geometry needs W, H >= 2 (use flat indexing)");
        }
        p.dy_px_stride = d->dy_px_stride; p.x_px_stride = d->x_px_stride;
        p.Cout = d->Cout; p.Cin = d->Cin;
        p.Cout_ld = (int)((std::min<long>(d->dy_px_stride, ((long)d->Cout + 7) & ~7L)));
        p.Cin_ld = (int)((std::min<long>(d->x_px_stride, ((long)d->Cin + 7) & ~7L)));
        p.KH = d->KH; p.KW = d->KW; p.pad = d->pad; p.x_row_stride = d->x_row_stride;
        p.n_ci_tiles = (d->Cin + WG_T - 1) / WG_T;
        p.ntaps = d->KH * d->KW;
        // Cin == 64 with several taps (the 64 -> 192 3x3 layer): two taps per 128-column tile instead of half-empty tiles
        p.pair_taps = ((d->variant <= 1 || d->variant == 4) && d->Cin == 64 && p.ntaps > 1 && d->x_px_stride >= 64) ? 1 : 0;
        const long steps_total = (d->P + WG_BP - 1) / WG_BP;
        // kernel variant: 256-wide co tiles (8 waves, 3 stages) only on request (d->variant == 2): measured no faster
        // than two co-resident 128 x 128 workgroups on any layer of the model
        const bool big = d->variant == 2 || d->variant == 3;   // 3: + staggered two-phase schedule
        const bool pipe = d->variant == 5 || d->variant == 6;   // 256 x 256 tile, register-pipelined loop (wgrad_pipe.hip: eight waves of
                                                                // 128 x 64; wgrad_wide.hip (6): four waves of 128 x 128)
        p.n_co_tiles = big ? (d->Cout + W2_TCO - 1) / W2_TCO : (d->Cout + WG_T - 1) / WG_T;
        int tiles = p.n_co_tiles * p.n_ci_tiles * (p.pair_taps ? (p.ntaps + 1) / 2 : p.ntaps);
        p.tile_taps = 1;
        if (pipe) {
            // 32-bit byte offsets from the operand bases, 24-bit slot arithmetic; a pixel range that ends inside a 32-pixel stage
            // reads halo slot 0 for the missing rows, which a Linear layer's operands (no halo) do not have
            const long max_slot = d->geo_W > 0 ? (d->P / ((long)d->geo_W * d->geo_H)) * (long)d->geo_img_slots + d->geo_slot0 : d->P;
            if (max_slot >= (1L << 24) || max_slot * (long)std::max(d->dy_px_stride, d->x_px_stride) * 2 + 4 * (long)d->pad * (d->x_row_stride + d->x_px_stride) >= (1L << 32)
                || d->dy_px_stride >= (1 << 23) || d->x_px_stride >= (1 << 23))
                return fail(YOLO_E_UNSUPPORTED, "yolo_wgrad: variant 5 addresses operands below 4 GB with fewer than 2^24 pixel slots");
            if (d->geo_W == 0 && d->pad == 0 && d->KH * d->KW == 1 && (d->P % 32))
                return fail(YOLO_E_UNSUPPORTED, "yolo_wgrad: variant 5 needs zero-haloed operands or P %% 32 == 0");
            p.tile_taps = (d->Cin < 256 && 256 % d->Cin == 0 && p.ntaps > 1 && d->x_px_stride >= d->Cin) ? 256 / d->Cin : 1;
            p.n_co_tiles = (d->Cout + 255) / 256;
            p.n_ci_tiles = p.tile_taps > 1 ? 1 : (d->Cin + 255) / 256;
            tiles = p.n_co_tiles * p.n_ci_tiles * ((p.ntaps + p.tile_taps - 1) / p.tile_taps);
        }
        dim3 grid;
        if (d->split > 0) {
            // uniform split, 2-D grid (tiles x ranges)
            long per = (d->P + d->split - 1) / d->split;
            per = (per + WG_BP - 1) / WG_BP * WG_BP;
            const int splits = (int)((d->P + per - 1) / per);
            p.p_per_split = per;
            p.atomic = (splits > 1 || d->accumulate) ? 1 : 0;
            grid = dim3((unsigned)tiles, (unsigned)splits);
        } else {
            // split == 0: two-segment schedule (see WgradParams), dw is accumulated.  Cost model in K steps: a round of
            // workgroups costs its K steps + ~30 steps' worth of prologue and atomic epilogue (fitted on the 3x3 layers).
            const int slots = (big || pipe) ? WG_SLOTS / 2 : WG_SLOTS;
            const double E = 30.0;
            double best = 1e30;
            int bs = 1, bt = 1;
            for (int sp = 1; sp <= slots; sp *= 2) {
                if (sp > 1 && steps_total / sp < 4) break;
                const int tpr = slots / sp;
                const int mt = tiles / tpr * tpr, tt = tiles - mt;
                long ts = 1;
                if (tt > 0) ts = std::max<long>(sp, std::min<long>(slots / tt, std::max<long>(1, steps_total / 4)));
                const double cost = (double)(mt / tpr) * ((double)((steps_total + sp - 1) / sp) + E) + (tt > 0 ? (double)((steps_total + ts - 1) / ts) + E : 0.0);
                if (cost < best) { best = cost; bs = sp; bt = (int)ts; }
            }
            const int tpr = slots / bs;
            p.seg = 1;
            p.slots = slots;
            p.main_split = bs;
            p.main_tiles = tiles / tpr * tpr;
            p.tail_tiles = tiles - p.main_tiles;
            p.tail_split = p.tail_tiles > 0 ? bt : 1;
            p.per_main = ((d->P + bs - 1) / bs + WG_BP - 1) / WG_BP * WG_BP;
            p.per_tail = ((d->P + p.tail_split - 1) / p.tail_split + WG_BP - 1) / WG_BP * WG_BP;
            // a segment whose tiles are reduced by ONE workgroup stores; only split tiles need atomics (dw arrives zero-filled
            // or holds the value to add to: accumulate forces atomics everywhere)
            p.atomic = d->accumulate ? 3 : ((bs > 1 ? 1 : 0) | (p.tail_split > 1 ? 2 : 0));
            const long nblk = (long)p.main_tiles * bs + (long)p.tail_tiles * p.tail_split;
            grid = dim3((unsigned)((nblk + slots - 1) / slots * slots));   // whole groups of `slots` ids for the XCD map
        }
        // slab mode (variant 5): partial tiles as plain stores + a fixed-order sum instead of fp32 atomics
        int main_ranges = 1, tail_ranges = 1;
        if (d->slabs || query) {
            long need = 0;
            if (pipe) {
                long nparts;
                if (d->split > 0) {
                    nparts = (long)tiles * grid.y;
                    main_ranges = (int)grid.y;
                } else {
                    nparts = (long)p.main_tiles * p.main_split + (long)p.tail_tiles * p.tail_split;
                    main_ranges = (int)std::min<long>(p.main_split, (d->P + p.per_main - 1) / p.per_main);
                    tail_ranges = (int)std::min<long>(p.tail_split, (d->P + p.per_tail - 1) / p.per_tail);
                }
                need = nparts * 256 * 256;
            }
            if (query) {
                *query = need;
                return 0;
            }
            if (!pipe || d->accumulate || (d->Cin & 3) || (((uintptr_t)dw | (uintptr_t)d->slabs) & 15))
                return fail(YOLO_E_UNSUPPORTED, "yolo_wgrad: slabs need variant 5, accumulate = 0, Cin %% 4 == 0 and 16-B aligned dw / slabs");
            if (d->slab_floats < need) return fail(YOLO_E_ARG, "yolo_wgrad: slabs hold %ld floats, this launch needs %ld (yolo_wgrad_slab_floats)", (long)d->slab_floats, need);
            p.slabs = d->slabs;
            p.atomic = 0;
        }
        if (d->dw_sumsq && (pipe || big || d->variant == 4 || p.atomic || (d->Cin & 3) || ((uintptr_t)dw & 15)))
            return fail(YOLO_E_UNSUPPORTED, "yolo_wgrad: dw_sumsq needs the 128 x 128 kernel storing every tile from one workgroup (split 1, no accumulate, Cin %% 4 == 0)");
        if (pipe) {
            if (int rc = d->variant == 6 ? wgrad_wide_launch(p, grid, s) : wgrad_pipe_launch(p, grid, s)) return rc;
            if (p.slabs) {
                if (int rc = wgrad_slab_sum_launch(p, tiles, main_ranges, tail_ranges, s)) return rc;
            }
        } else if (big) {
            static bool lds_ok = false;
            if (!lds_ok) {
                hipError_t e = hipFuncSetAttribute((const void *)wgrad256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * W2_STAGE);
                if (e == hipSuccess) e = hipFuncSetAttribute((const void *)wgrad256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * W2_STAGE);
                if (e != hipSuccess) return fail((int)e, "yolo_wgrad: hipFuncSetAttribute(%d B LDS): %s", 3 * W2_STAGE, hipGetErrorString(e));
                lds_ok = true;
            }
            if (d->variant == 3) hipLaunchKernelGGL(wgrad256_kernel<true>, grid, dim3(512), 3 * W2_STAGE, s, p);
            else hipLaunchKernelGGL(wgrad256_kernel<false>, grid, dim3(512), 3 * W2_STAGE, s, p);
        } else if (d->variant == 4) {
            hipLaunchKernelGGL(wgrad8_kernel, grid, dim3(512), 0, s, p);
        } else {
            p.sumsq = d->dw_sumsq;
            hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, s, p);
        }
        if (int rc = check_launch("yolo_wgrad")) return rc;
    }
    if (db && !dw) {
        const int nchunks = (d->Cout + 7) / 8;
        const int w = nchunks < 256 ? nchunks : 256;
        const int R = 256 / w;
        const int gx = (nchunks + w - 1) / w;
        long gy = d->P / ((long)R * 32);
        if (gy > 2048 / gx) gy = 2048 / gx;
        if (gy < 1) gy = 1;
        const long per = (d->P + gy - 1) / gy;
        hipLaunchKernelGGL(colsum_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, s, (const bf16_t *)dy, (long)d->P, d->dy_px_stride, d->Cout, w, per, db);
        if (int rc = check_launch("yolo_wgrad(bias)")) return rc;
    }
    return 0;
}
