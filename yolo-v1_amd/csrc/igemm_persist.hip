// Persistent register-pipelined implicit-GEMM kernels for gfx950 (tile_hint 20 / 21): the K loop of igemm_pipe.hip (256 channels x
// 16 * NTILES pixel slots, BK = 32, four LDS stages, ONE barrier per K step, two register sets of fragments) run as ONE software
// pipeline over ALL the tiles of a workgroup:
//
//   * one workgroup per CU walks tiles b, b + G, b + 2G, .. ; the LDS-DMA of the next tile's first three stages is issued under the
//     last three K steps of the current tile, the fragments of its step 0 are read under the current tile's last MFMAs -- the first
//     stage's HBM / L2 latency (4-6 k cycles per tile in igemm_pipe, in-kernel stamps) is paid once per workgroup, not once per tile;
//   * the per-pixel address table of tile t + 1 is built under tile t's K loop (two table buffers; divisions by multiply-high);
//   * the EPILOGUE RUNS FROM THE ACCUMULATOR REGISTERS: in the 16x16x32 result layout a lane holds 4 consecutive channels of one
//     pixel, so bias / LeakyReLU / LeakyReLU' / bf16 packing happen in place and each accumulator tile leaves as one 8-byte store
//     per lane (32 contiguous bytes per pixel and instruction, 128 B per pixel over a wave's four channel tiles; L2 merges them into
//     whole lines).  No LDS slab, no second pass, no barrier: ~2-3 k cycles per tile against 13-15 k of the LDS epilogue, and the
//     stores drain under the next tile's K loop (the counted vmcnt waits of its first two steps leave them in flight);
//   * hint 21 (224-pixel tiles of whole row pairs) fuses MaxPool2d(2,2): the address table puts the four pixels of a pooling window
//     on four NEIGHBOURING LANES of one accumulator column, so the window maximum is two DPP quad-permute max operations per value;
//     lane e of a quad then stores channel tile e of the pooled pixel -- again one 8-byte store per lane, a quarter of the bytes.
//
// Pixel slots beyond a tile's valid pixels compute (and store) the true values of the pixels they would hold in the flat order --
// identical bits written twice -- so that every wave issues the SAME number of stores in every epilogue: the vmcnt arithmetic of
// the first two K steps behind an epilogue depends on it.
// Reference arithmetic: nn.Conv2d + nn.LeakyReLU(0.1) (+ nn.MaxPool2d(2,2)), src/yolo/models.py:47-84.
#include "igemm_common.h"
#include <atomic>
#include <mutex>

namespace yolo {

template <int NTILES_>
struct PersistCfg {
    static constexpr int TCO = 256, NTILES = NTILES_, NST = 4, BK = 32;
    static constexpr int TPX = 16 * NTILES;
    static constexpr int WCO = 4, NW = 8, NTHR = 512;
    static constexpr int NT0 = 7, NT1 = NTILES - 7;
    static constexpr int A_BYTES = TCO * BK * 2;
    static constexpr int B_BYTES = 256 * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_PIECES = A_BYTES / 1024 / NW, B_PIECES = B_BYTES / 1024 / NW;
    static constexpr int LOADS = A_PIECES + B_PIECES;
    static constexpr int TABLE_BYTES = 256 * 16;            // one table: 16 B per pixel slot {in, out, aux byte offsets, -}
    static constexpr int STASH_BYTES = NW * 64 * 4;         // per-wave bias stash
    static constexpr int MAIL_BYTES = 64;                   // tile numbers drawn from the queue, wave 0 -> all waves
    static constexpr int LDS_BYTES = 2 * TABLE_BYTES + STASH_BYTES + MAIL_BYTES + NST * STAGE_BYTES;
    static constexpr int D = NST - 1;
    static_assert(NT1 >= 1 && NT1 <= NT0, "pixel group B holds 1 .. 7 columns");
};

__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned magic, unsigned shift)
{
    return magic ? (__umulhi(n, magic) >> shift) : n;       // magic 0: divisor 1
}

// The lane id, recomputed where it is used: everything the per-tile code derives from it (table slots, stash and output offsets) would
// otherwise be hoisted out of the tile loop as loop invariants and -- at ~250 live registers in the K loop -- spilled to scratch, whose
// reloads wait for vmcnt(0), i.e. for the previous tile's stores and the next tile's stages.  (volatile: not hoisted, not merged)
__device__ __forceinline__ int fresh_lane()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// Maximum over the four lanes of a quad, for the four values of one accumulator tile: v_max_f32 with a DPP operand (quad_perm
// [1,0,3,2], then [2,3,0,1]) -- two instructions per value.  (Through fmaxf + __builtin_amdgcn_update_dpp hipcc emits a v_mov_dpp
// and two canonicalising self-maxima per step: six.)  The four chains are interleaved, so a DPP operand is always at least three
// instructions old (the ISA asks for two wait states between a VALU write and a DPP read of the same register); the leading s_nop
// covers whatever instruction wrote the inputs.
__device__ __forceinline__ f32x4 quad_max4(f32x4 v)
{
    float a0, a1, a2, a3;
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    return f32x4{a0, a1, a2, a3};
}

// minimum over the four lanes of a quad of four small unsigned keys (the first lane holding a maximum)
__device__ __forceinline__ void quad_min4(unsigned (&k)[4])
{
    asm volatile("s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_u32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_u32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_u32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_u32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_u32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]));
}

__device__ __forceinline__ void store1(const void *base, unsigned voff, unsigned byte)
{
    asm volatile("global_store_byte %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(byte), "s"(base) : "memory");
}

// two floats -> packed bf16 (round to nearest even, NaN stays NaN): ONE v_cvt_pk_bf16_f32 (casting the floats one by one costs a
// conversion each plus a shift and an or)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// 8-byte store: scalar base + 32-bit byte offset per lane + immediate.  Inline asm: exactly one VMEM instruction per call, whatever
// the optimiser thinks (the waits behind an epilogue count them).
template <int IMM>
__device__ __forceinline__ void store8(const void *base, unsigned voff, unsigned lo, unsigned hi)
{
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 d = {lo, hi};
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(d), "s"(base), "n"(IMM) : "memory");
}

// DGRAD: the instantiation whose epilogue multiplies by LeakyReLU'(aux) (its 56 registers of aux vectors stay out of the others)
template <int IMM>
__device__ __forceinline__ void store16(const void *base, unsigned voff, unsigned a, unsigned b, unsigned c, unsigned d)
{
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 v = {a, b, c, d};
    // (s_nop: a store of more than 64 bits reads its data registers for a cycle after issue, and the compiler, which pads that hazard
    // for its own stores, does not look into inline assembly: the next VALU instruction could overwrite them)
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(base), "n"(IMM) : "memory");
}

// v_permlane16_swap_b32 x, y: rows (16 lanes) 1 and 3 of x change places with rows 0 and 2 of y -- for the four registers of two
// accumulator tiles at once.  Inline assembly: through __builtin_amdgcn_permlane16_swap hipcc (ROCm 7.2) used the instruction's FIRST
// result for both outputs once the inputs were dead behind it (seen in the ISA; half of every 16-byte store was wrong).
__device__ __forceinline__ void row_swap4(f32x4 &x, f32x4 &y)
{
    float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3];
    // (s_nop in front: a VALU write of an operand -- the compiler's register copies -- must be two wait states old; behind: the same for
    // the consumers; the compiler pads neither around inline assembly)
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %4\n\t"
                 "v_permlane16_swap_b32 %1, %5\n\t"
                 "v_permlane16_swap_b32 %2, %6\n\t"
                 "v_permlane16_swap_b32 %3, %7\n\t"
                 "s_nop 1"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
    x = f32x4{x0, x1, x2, x3};
    y = f32x4{y0, y1, y2, y3};
}

// CODES (with POOL): training's pooled map + 2-bit arg-max codes (pool2 = 3), in its own instantiation
// SLAB: split-K -- a "tile" is (channel tile, pixel tile, K range); every range STORES its partial tile, fp32, densely into slab
// `range` of the output (yolo_igemm_finish adds the slabs in fixed order and applies the layer's epilogue: deterministic).  The deep-K,
// few-pixel layers (7x7 maps: 64 tiles of 256 x 196 for 256 CUs) fill the chip that way.
// MT: 16-channel accumulator tiles per wave (4; 3 for Cout = 192 -- the pooled epilogues only)
template <int NTILES, bool POOL, int DGRAD, bool CODES = false, bool SLAB = false, int MT = 4>
__global__ void __launch_bounds__(512, 2) igemm_persist_kernel(const IgemmParams p)
{
    using C = PersistCfg<NTILES>;
    constexpr int TPX = C::TPX, BK = C::BK, NW = C::NW, WCO = C::WCO, NT0 = C::NT0, NT1 = C::NT1, NST = C::NST;
    constexpr int A_BYTES = C::A_BYTES, STAGE_BYTES = C::STAGE_BYTES, LOADS = C::LOADS, D = C::D;
    constexpr int WCH = MT * 16;          // channels per wave (64; 48 with MT = 3: 192-channel layers without a quarter of the MFMAs on padding)
    constexpr int TCO = WCO * WCH;        // channels per tile (the stage still holds 256 weight rows: the surplus rows are clamped duplicates)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned *tab = reinterpret_cast<unsigned *>(smem);                         // [2][256][4]
    float *stash = reinterpret_cast<float *>(smem + 2 * C::TABLE_BYTES);        // [NW][64]
    int *mail = reinterpret_cast<int *>(smem + 2 * C::TABLE_BYTES + C::STASH_BYTES);     // [2] (read and written between barriers)
    char *stage_base = smem + 2 * C::TABLE_BYTES + C::STASH_BYTES + C::MAIL_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % WCO, grp = wave / WCO;
    const int px_lo = grp * NT0 * 16;

#ifdef IGEMM_STAMPS
    // diagnostic build: [0] kernel start | [1] stage 0 visible | [2] / [3] end of tile 0's K loop / epilogue | [4] / [5] the same of tile 1 |
    // [6] end of the last tile's K loop | [7] kernel end -- per wave of the first 512 workgroups (tools/stamps_persist.py)
    long tstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tstamp[0] = __builtin_amdgcn_s_memtime();
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); tstamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
    const int S = SLAB ? p.pool_tiles_x : 1;              // K ranges (the launch passes the count in a field the pooled epilogue of igemm.hip owns)
    const int T = p.n_co_tiles * p.n_px_tiles * S;
    const int G = gridDim.x;
    const int tpv = p.tpx_valid;
    // Tiles are handed out per XCD: workgroups b and b + 8 share an XCD (and its L2), so label x = b & 7 owns one contiguous eighth of
    // the tile order (igemm.hip's bijective map); a workgroup's FIRST tile is static (number b >> 3 of its label's range), every further
    // one is drawn from the label's counter in device memory (p.tile_ctr[x], zero at launch).  Static rounds (tile b, b + G, ..) assume
    // that all G workgroups are resident at once; when something else holds CUs -- RCCL's all-reduce kernels beside the backward pass, a
    // background optimizer pass -- the workgroups that start late would begin their rounds when the others have finished theirs, up to
    // twice the time.  Drawn tiles make the launch work-conserving: whoever runs takes the next tile.
    const int xl = (int)blockIdx.x & 7;
    const int cnt_x = (T >> 3) + (xl < (T & 7) ? 1 : 0);       // tiles of this label
    const int nwg_x = (G - xl + 7) >> 3;                       // workgroups of this label = its statically assigned tiles
    unsigned *const tile_ctr = p.tile_ctr + xl;

    // tile j of this label -> (first channel, first pixel, K range)
    auto tile_of = [&](int j, int &co0, unsigned &px0, int &ks) {
        const int q = T >> 3, r = T & 7;
        int bid = (xl < r ? xl * (q + 1) : r * (q + 1) + (xl - r) * q) + j;
        ks = 0;
        if constexpr (SLAB) {            // the K ranges of one output tile are neighbours: they run at the same time and share its operands in L2
            ks = bid % S;
            bid /= S;
        }
        const int co_tile = p.px_fastest ? bid / p.n_px_tiles : bid % p.n_co_tiles;
        const int px_tile = p.px_fastest ? bid % p.n_px_tiles : bid / p.n_co_tiles;
        co0 = co_tile * TCO;
        px0 = (unsigned)px_tile * (unsigned)tpv;
    };

    // address table of one tile: byte offsets of the pixel's input row base / output / aux element 0
    auto build_table = [&](int buf, unsigned px0) {
        const int tid = wave * 64 + fresh_lane();
        if (tid < TPX) {
            unsigned m;
            if constexpr (POOL) {
                // slot = 16 * column + 4 * window-in-column + element: the four pixels of a 2x2 window are four neighbouring lanes
                const unsigned W = (unsigned)(tid >> 4) * 4 + (unsigned)((tid >> 2) & 3), e = (unsigned)tid & 3;
                const unsigned hw = (unsigned)p.Wo >> 1;
                const unsigned wy = fast_div(W, p.div_hw2_magic, p.div_hw2_shift), wx = W - wy * hw;
                m = px0 + (2 * wy + (e >> 1)) * (unsigned)p.Wo + 2 * wx + (e & 1);
            } else {
                m = px0 + (unsigned)tid;
            }
            if (m >= (unsigned)p.M) m = (unsigned)p.M - 1;           // past the end: the last pixel once more (same bits)
            const unsigned n = fast_div(m, p.div_hw_magic, p.div_hw_shift);
            const unsigned rem = m - n * (unsigned)p.HoWo;
            const unsigned oy = fast_div(rem, p.div_w_magic, p.div_w_shift), ox = rem - oy * (unsigned)p.Wo;
            const long in_e = (long)n * p.in_img_stride + (long)(oy * p.stride) * p.in_row_stride + (long)(ox * p.stride) * p.in_px_stride + p.in_off;
            const unsigned qy = POOL ? oy >> 1 : oy, qx = POOL ? ox >> 1 : ox;
            const long out_e = (long)n * p.out_img_stride + (long)qy * p.out_row_stride + (long)qx * p.out_px_stride + p.out_off;
            const long aux_e = (long)n * p.aux_img_stride + (long)oy * p.aux_row_stride + (long)ox * p.aux_px_stride + p.aux_off;
            uint4 ent;
            ent.x = (unsigned)(in_e * 2);
            ent.y = (unsigned)(out_e * (SLAB ? 4 : 2));
            ent.z = (unsigned)(aux_e * 2);
            ent.w = 0;
            *reinterpret_cast<uint4 *>(tab + (buf * 256 + tid) * 4) = ent;
        }
    };

    // LDS-DMA source lanes (inverse swizzle, igemm.hip): weight rows relative to the tile's first channel, pixel rows from the table
    unsigned a_voff[C::A_PIECES], b_voff[C::B_PIECES];
#pragma unroll
    for (int i = 0; i < C::A_PIECES; ++i) {
        const int pos = (i * NW + wave) * 64 + lane;
        const int R = pos >> 4, s = (pos & 15) ^ swz_key<BK, true>(R);
        int r = R * 4 + s / 4;
        const int chunk = s % 4;
        if (r >= p.Cout) r = p.Cout - 1;                     // (a single ragged channel tile; several tiles: Cout % 256 == 0, host-checked)
        a_voff[i] = (unsigned)(((long)r * p.Ktot + chunk * 8) * 2);
    }
    // (recomputed per tile rather than kept: the K loop runs at ~250 live registers)
    auto load_b_voff = [&](int buf) {
        const int lane = fresh_lane();
#pragma unroll
        for (int i = 0; i < C::B_PIECES; ++i) {
            const int pos = (i * NW + wave) * 64 + lane;
            const int R = pos >> 4, s = (pos & 15) ^ swz_key<BK, true>(R);
            const int r = R * 4 + s / 4;
            b_voff[i] = tab[(buf * 256 + (r < TPX ? r : 0)) * 4] + (unsigned)((s % 4) * 16);
        }
    };

    const int nk = SLAB ? p.nk_per_split : p.nk;          // K steps per tile
    const int cpt = p.tap_len / BK;
    // scalar staging state of the tile being STAGED (the current tile, or the next one in a tile's last three steps)
    unsigned a_soff = 0;          // byte offset of the step's weight columns + of the tile's first weight row
    int c0 = 0, ky = 0, kx = 0;

    auto stage = [&](int buf) {
        char *sb = stage_base + buf * STAGE_BYTES + wave * 1024;
        const unsigned b_soff = (unsigned)((ky * p.in_row_stride + kx * p.in_px_stride + c0) * 2);
        const char *wb = reinterpret_cast<const char *>(p.w) + a_soff;
        const char *xb = reinterpret_cast<const char *>(p.in) + b_soff;
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wb + (unsigned long)a_voff[i]),
                                             (__attribute__((address_space(3))) void *)(sb + i * NW * 1024), 16, 0, 0);
#ifdef PERSIST_NO_B      // ablation build: what the K loop does without the pixel operand's LDS-DMA; wrong results
        constexpr int NB = 0;
        (void)xb;
#elif defined(PERSIST_HALF_B)
        const int NB = (kx == 1) ? C::B_PIECES : 0;          // ablation: the pixel operand staged for one tap column in three
#else
        constexpr int NB = C::B_PIECES;
#endif
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xb + (unsigned long)b_voff[i]),
                                             (__attribute__((address_space(3))) void *)(sb + A_BYTES + i * NW * 1024), 16, 0, 0);
        a_soff += BK * 2;
        c0 += BK;
        const int w0 = c0 == p.tap_len;
        c0 = w0 ? 0 : c0;
        kx += w0;
        const int w1 = kx == p.KW;
        kx = w1 ? 0 : kx;
        ky += w1;
    };
    auto stage_reset = [&](int co0, int ks) {
        const int kbeg = ks * nk;                            // first K step of the range (0 without split-K)
        a_soff = (unsigned)((long)co0 * p.Ktot * 2) + (unsigned)kbeg * (BK * 2);
        const int tap = kbeg / cpt;
        c0 = (kbeg - tap * cpt) * BK;
        ky = tap / p.KW;
        kx = tap - ky * p.KW;
    };

    // fragment read offsets: 16 rows further = 4 bank rows (1 KB) further with the same swizzle key -> one base register per operand,
    // the tiles as immediate offsets
    const int a_rd0 = lds_off<BK, true>(wco * WCH + (lane & 15), lane >> 4);
    const int b_rd0 = A_BYTES + lds_off<BK, true>(px_lo + (lane & 15), lane >> 4);

    f32x4 acc[MT][NT0];      // (never zero-filled: the MFMAs of a tile's step 0 take the constant 0 as their C operand)

    // The queue.  draw(): wave 0's lane 0 adds 1 to the label's counter (a returning atomic from inline assembly: the compiler does not
    // know it is in flight, so it inserts no wait; the result register is valid once a vmcnt wait has covered the instruction -- every
    // K step's counted wait does, two steps later at the latest).  publish(): the same lane puts the drawn number into the LDS mailbox;
    // the other waves read it behind a barrier.  Numbers >= cnt_x mean "no tile left".
    unsigned drawn = 0;
    auto draw = [&]() {
        if (wave == 0) {
            const unsigned l = fresh_lane();
            if (l == 0) {       // (the lane number doubles as the zero offset; the register that holds the addend takes the result)
                unsigned d = 1u;
                // (s_nop 4: the counter's address may have been reloaded from an SGPR spill lane by the v_readlane right in front of
                // this block, and hipcc pads no hazard into inline assembly -- 5 wait states from a VALU write of an SGPR to a vector
                // memory instruction that reads it; tools/check_asm_hazards.py checks the ISA for this)
                asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %0, %2 sc0" : "+v"(d) : "v"(l), "s"(tile_ctr) : "memory");
                drawn = d;
            }
        }
    };
    auto publish = [&](int slot) {
        if (wave == 0) {
            if (fresh_lane() == 0) mail[slot] = nwg_x + (int)drawn;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };

    // ---- prologue: table + bias stash of the first tile, its first D stages in flight, stage 0 landed and visible; the second tile's number
    int j_cur = (int)blockIdx.x >> 3;
    draw();
    int co0_cur, ks_cur;
    unsigned px0_cur;
    tile_of(j_cur, co0_cur, px0_cur, ks_cur);
    build_table(0, px0_cur);
    const bool has_bias = p.epilogue == YOLO_EPI_BIAS || p.epilogue == YOLO_EPI_BIAS_LRELU || p.epilogue == YOLO_EPI_BIAS_ADD_LRELU;
    auto load_stash = [&](int co0) {
        const int lane = fresh_lane();
        const int ch = co0 + wco * WCH + lane;
        stash[wave * 64 + lane] = (has_bias && lane < WCH && ch < p.Cout) ? p.bias[ch] : 0.0f;
    };
    load_stash(co0_cur);
    __syncthreads();
    load_b_voff(0);
    stage_reset(co0_cur, ks_cur);
#pragma unroll
    for (int s0 = 0; s0 < D; ++s0) stage(s0);
    wait_vmcnt<(D - 1) * LOADS>();        // stage 0 has landed -- and the draw, which is older than every stage load, has returned
    publish(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int j_next = __builtin_amdgcn_readfirstlane(mail[0]);
    PSTAMP(1);

    auto run = [&](auto ntc) {
        constexpr int NTG = decltype(ntc)::value;
        bf16x8 a0[MT], b0[NT0], a1[MT], b1[NT0];
        auto rd = [&](int buf, bf16x8(&af)[MT], bf16x8(&bfr)[NT0]) {
            const char *sb = stage_base + buf * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(sb + a_rd0 + i * 1024);
#pragma unroll
            for (int j = 0; j < NTG; ++j) bfr[j] = *reinterpret_cast<const bf16x8 *>(sb + b_rd0 + j * 1024);
        };
        auto mm = [&](auto zc, bf16x8(&af)[MT], bf16x8(&bfr)[NT0]) {
            constexpr bool Z = decltype(zc)::value;           // a tile's first step: C = 0 (inline constant), no accumulator to clear
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTG; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], Z ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : acc[i][j], 0, 0, 0);
        };
        using ZeroC = std::integral_constant<bool, true>;
        using AccC = std::integral_constant<bool, false>;
        constexpr int NRD = MT + NTG, NMF = MT * NTG;
        auto body = [&](auto zc, int rbuf, int lbuf, bf16x8(&ca)[MT], bf16x8(&cb)[NT0], bf16x8(&na)[MT], bf16x8(&nb)[NT0]) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            rd(rbuf, na, nb);
            stage(lbuf);
            mm(zc, ca, cb);
#pragma unroll
            for (int k = 0; k < NRD; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            constexpr int PER = (NMF - NRD) / (LOADS + 1);
#pragma unroll
            for (int k = 0; k < LOADS; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD - LOADS * PER, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        // a tile's LAST step when another tile follows: stages that tile's third stage, reads nothing -- the fragments of the next tile's
        // step 0 are read behind the epilogue (they would occupy 44 registers across it; the read costs ~200 cycles per tile)
        auto last_step = [&](int lbuf, bf16x8(&ca)[MT], bf16x8(&cb)[NT0]) {
            wait_vmcnt<(D - 2) * LOADS>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stage(lbuf);
            mm(AccC{}, ca, cb);
            constexpr int PER = NMF / (LOADS + 1);
#pragma unroll
            for (int k = 0; k < LOADS; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - LOADS * PER, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        // steady-state step: stage k+1 has landed when all but the youngest LOADS VMEM operations are done
        auto step = [&](int rbuf, int lbuf, bf16x8(&ca)[MT], bf16x8(&cb)[NT0], bf16x8(&na)[MT], bf16x8(&nb)[NT0]) {
            wait_vmcnt<(D - 2) * LOADS>();
            body(AccC{}, rbuf, lbuf, ca, cb, na, nb);
        };
        // the first two steps behind an epilogue: its NS stores sit between the stages in the (in-order) VMEM queue and may stay in flight
        // (nst = channel tiles of this wave that exist: a ragged last channel tile issues fewer stores -- a wave-uniform number)
        auto step_x = [&](auto zc, int nst, int rbuf, int lbuf, bf16x8(&ca)[MT], bf16x8(&cb)[NT0], bf16x8(&na)[MT], bf16x8(&nb)[NT0]) {
            // stores of the last epilogue: pooled NTG (one per column), else NTG per PAIR of channel tiles that exists
            if (nst == 0) wait_vmcnt<(D - 2) * LOADS>();
            else if (SLAB) wait_vmcnt<(D - 2) * LOADS + MT * NTG>();      // (one 16-byte fp32 store per accumulator tile)
            else if (POOL ? !CODES : nst <= 2) wait_vmcnt<(D - 2) * LOADS + NTG>();
            else wait_vmcnt<(D - 2) * LOADS + 2 * NTG>();       // (pooled map + code bytes, or two pairs of channel tiles)
            body(zc, rbuf, lbuf, ca, cb, na, nb);
        };
        // ---- epilogue of one tile, from the accumulator registers (lane: pixel column lane & 15, channels 4 * (lane >> 4) + r)
        auto epilogue = [&](int tb, int co0, int nvi, int ks) {
            // (the accumulators were written by the MFMAs just issued: the compiler's hazard handling does not look into the inline
            // assembly that reads them first -- give the matrix pipe its worst-case drain time)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            const int lane = fresh_lane();
            const int g4 = (lane >> 4) * 4;
            const unsigned *tb_ = tab + tb * 1024;
            const float *st = stash + wave * 64;
            const float slope = p.slope;
            if constexpr (SLAB) {
                // the partial tile as it stands: a lane's four fp32 values are four consecutive channels of its pixel = one 16-byte store
                const float *slab = reinterpret_cast<const float *>(p.out) + (long)ks * p.slab_stride;
                const unsigned chb = (unsigned)((co0 + wco * WCH + g4) * 4);
#pragma unroll
                for (int j = 0; j < NTG; ++j) {
                    const unsigned ob = tb_[(px_lo + j * 16 + (lane & 15)) * 4 + 1] + chb;
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const unsigned w0 = __float_as_uint(acc[i][j][0]), w1 = __float_as_uint(acc[i][j][1]), w2 = __float_as_uint(acc[i][j][2]), w3 = __float_as_uint(acc[i][j][3]);
                        if (i == 0) store16<0>(slab, ob, w0, w1, w2, w3);
                        else if (i == 1) store16<64>(slab, ob, w0, w1, w2, w3);
                        else if (i == 2) store16<128>(slab, ob, w0, w1, w2, w3);
                        else store16<192>(slab, ob, w0, w1, w2, w3);
                    }
                }
                (void)nvi; (void)st; (void)slope;
            } else if constexpr (POOL) {
                const int e = lane & 3;
                if constexpr (CODES) {
                    // pool2 = 3: besides the pooled map, WHICH window element was the maximum (2 bits per channel; a lane's four channels
                    // are one byte of the uint16 that holds eight) -- all the backward pass needs of the un-pooled activation.  As in the
                    // LDS epilogue of igemm_pipe.hip the comparison runs on the activations AS THEY WOULD BE STORED (bias, LeakyReLU,
                    // rounded to bf16) and the first maximum in window order (0,0), (0,1), (1,0), (1,1) = lanes 0 .. 3 of the quad wins.
                    f32x4 bsa[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) bsa[i] = *reinterpret_cast<const f32x4 *>(st + i * 16 + g4);
                    const unsigned chb = (unsigned)((co0 + wco * WCH + e * 16 + g4) * 2);
                    const unsigned cdb = (unsigned)((co0 + wco * WCH + e * 16) >> 2) + (unsigned)(lane >> 4);       // byte of this lane's code
                    const bool lrelu = p.epilogue == YOLO_EPI_BIAS_LRELU;
#pragma unroll
                    for (int j = 0; j < NTG; ++j) {
                        const unsigned obp = tb_[(px_lo + j * 16 + (lane & 15)) * 4 + 1];
                        f32x4 sel = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        unsigned code = 0;
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            f32x4 t;
#pragma unroll
                            for (int r = 0; r < 4; r += 2) {
                                float t0 = acc[i][j][r] + bsa[i][r], t1 = acc[i][j][r + 1] + bsa[i][r + 1];
                                t0 = (lrelu && t0 < 0.0f) ? t0 * slope : t0;
                                t1 = (lrelu && t1 < 0.0f) ? t1 * slope : t1;
                                const unsigned pk = pack_bf16x2(t0, t1);
                                t[r] = __uint_as_float(pk << 16);
                                t[r + 1] = __uint_as_float(pk & 0xffff0000u);
                            }
                            const f32x4 m = quad_max4(t);
                            unsigned key[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) key[r] = (t[r] == m[r]) ? (unsigned)e : 4u;
                            quad_min4(key);
                            const unsigned c4 = key[0] | (key[1] << 2) | (key[2] << 4) | (key[3] << 6);
                            code = (e == i) ? c4 : code;
#pragma unroll
                            for (int r = 0; r < 4; ++r) sel[r] = (e == i) ? m[r] : sel[r];
                        }
                        const unsigned lo = pack_bf16x2(sel[0], sel[1]), hi = pack_bf16x2(sel[2], sel[3]);      // (exact: already bf16 values)
                        if (e < nvi) {
                            store8<0>(p.out, obp + chb, lo, hi);
                            store1(p.aux, (obp >> 3) + cdb, code);
                        }
                    }
                } else {
                // bias of the channel tile this lane stores (tile e of the wave's four)
                const f32x4 bs = *reinterpret_cast<const f32x4 *>(st + e * 16 + g4);
                const unsigned chb = (unsigned)((co0 + wco * WCH + e * 16 + g4) * 2);
#pragma unroll
                for (int j = 0; j < NTG; ++j) {
                    const unsigned ob = tb_[(px_lo + j * 16 + (lane & 15)) * 4 + 1] + chb;
                    f32x4 sel = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const f32x4 m = quad_max4(acc[i][j]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) sel[r] = (e == i) ? m[r] : sel[r];
                    }
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = sel[r] + bs[r];
                        v[r] = (p.epilogue == YOLO_EPI_BIAS_LRELU && t < 0.0f) ? t * slope : t;
                    }
                    const unsigned lo = pack_bf16x2(v[0], v[1]), hi = pack_bf16x2(v[2], v[3]);
                    if (e < nvi) store8<0>(p.out, ob, lo, hi);        // (issued once per column by every wave with nvi > 0)
                }
                }
            } else {
                // Two channel tiles (i0, i0 + 1) at a time: after swapping rows 1 / 3 of tile i0's registers with rows 0 / 2 of tile
                // i0 + 1's, a lane of row g holds EIGHT consecutive channels -- of tile i0 + (g & 1), channels 8 * (g >> 1) .. + 7 -- of
                // its pixel: 16-byte stores (64 contiguous bytes per pixel and instruction), half as many as 8-byte ones; the 8-byte
                // form took 9-13 k cycles per tile, bound by store issue (in-kernel stamps).
                const int g = lane >> 4;
                const int cl = (g & 1) * 16 + (g >> 1) * 8;                 // first channel of this lane inside a pair of tiles
                f32x4 bs[2][2];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    bs[pr][0] = *reinterpret_cast<const f32x4 *>(st + pr * 32 + cl);
                    bs[pr][1] = *reinterpret_cast<const f32x4 *>(st + pr * 32 + cl + 4);
                }
                const unsigned chb = (unsigned)((co0 + wco * WCH + cl) * 2);
                const int my_tile = g & 1;                                   // + 2 * pair: the channel tile this lane stores
                const bool lrelu = p.epilogue == YOLO_EPI_BIAS_LRELU;
#pragma unroll
                for (int j = 0; j < NTG; ++j) {
                    const unsigned slot4 = (unsigned)(px_lo + j * 16 + (lane & 15)) * 4;
                    const unsigned ob = tb_[slot4 + 1] + chb;
                    uint4 ax[2];
                    if constexpr (DGRAD) {
                        const char *ap = reinterpret_cast<const char *>(p.aux) + (unsigned long)(tb_[slot4 + 2] + chb);
                        ax[0] = *reinterpret_cast<const uint4 *>(ap);
                        ax[1] = *reinterpret_cast<const uint4 *>(ap + 64);
                    }
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        f32x4 x = acc[2 * pr][j], y = acc[2 * pr + 1][j];
                        row_swap4(x, y);
                        float v[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
                        if constexpr (DGRAD == 1) {
                            const unsigned yy[4] = {ax[pr].x, ax[pr].y, ax[pr].z, ax[pr].w};
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float a = __uint_as_float((k & 1) ? (yy[k >> 1] & 0xffff0000u) : (yy[k >> 1] << 16));
                                v[k] = a > 0.0f ? v[k] : v[k] * slope;
                            }
                        } else if constexpr (DGRAD == 2) {
                            const unsigned yy[4] = {ax[pr].x, ax[pr].y, ax[pr].z, ax[pr].w};
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                float t = v[k] + bs[pr][k >> 2][k & 3];
                                t += __uint_as_float((k & 1) ? (yy[k >> 1] & 0xffff0000u) : (yy[k >> 1] << 16));
                                v[k] = t > 0.0f ? t : t * slope;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float t = v[k] + bs[pr][k >> 2][k & 3];
                                v[k] = lrelu ? fmaxf(t, t * slope) : t;      // LeakyReLU for 0 <= slope <= 1 (host-checked)
                            }
                        }
                        const unsigned w0 = pack_bf16x2(v[0], v[1]), w1 = pack_bf16x2(v[2], v[3]), w2 = pack_bf16x2(v[4], v[5]), w3 = pack_bf16x2(v[6], v[7]);
                        if (2 * pr + my_tile < nvi) {        // (ragged last channel tile: issued by the wave iff 2 * pr < nvi)
                            if (pr == 0) store16<0>(p.out, ob, w0, w1, w2, w3);
                            else store16<64>(p.out, ob, w0, w1, w2, w3);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };

        rd(0, a0, b0);
        int rbuf = 1, lbuf = D;
        auto adv = [&]() {
            rbuf = rbuf + 1 == NST ? 0 : rbuf + 1;
            lbuf = lbuf + 1 == NST ? 0 : lbuf + 1;
        };
        // ONE loop body for every tile, no branch with MFMAs in its arms (at ~250 live registers hipcc answers such a branch by
        // spilling the accumulators around it): behind the workgroup's last tile the three boundary steps stage that tile's own first
        // stages once more -- 96 KB nobody reads; the kernel drains them before it ends.
        int nst = 0;                                        // stores of the previous epilogue, in channel tiles (0: none yet)
        // (a loop on a limit that the body lowers, not `do .. while (has_next)`: with the uncounted form hipcc keeps ~40 more registers
        // live across the K loop and spills)
        bool has_next;
        int n_lim = cnt_x;
        for (int ti = 0; ti < n_lim;) {
            has_next = (unsigned)j_next < (unsigned)cnt_x;
            int co0_next, ks_next;
            unsigned px0_next;
            tile_of(has_next ? j_next : j_cur, co0_next, px0_next, ks_next);
            step_x(ZeroC{}, nst, rbuf, lbuf, a0, b0, a1, b1);        // step 0
            adv();
            // every wave is past the previous tile's epilogue (barrier of step 0): its table buffer is free for the next tile;
            // and the tile after that is requested from the queue now -- its number is needed a whole tile from here
            draw();
            build_table((ti + 1) & 1, px0_next);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the table is in LDS before this wave arrives at the next barrier
            step_x(AccC{}, nst, rbuf, lbuf, a1, b1, a0, b0);         // step 1
            adv();
            for (int it = 2; it + 4 < nk; it += 2) {
                step(rbuf, lbuf, a0, b0, a1, b1);
                adv();
                step(rbuf, lbuf, a1, b1, a0, b0);
                adv();
            }
            step(rbuf, lbuf, a0, b0, a1, b1);               // step nk-4: stages the tile's last stage
            adv();
            publish((ti + 1) & 1);                          // (the draw of step 0 is covered by the waits of steps 1 .. nk-4)
            // the next tile's first D stages ride under this tile's last D steps
            load_b_voff((ti + 1) & 1);
            stage_reset(co0_next, ks_next);
            step(rbuf, lbuf, a1, b1, a0, b0);               // step nk-3
            adv();
            step(rbuf, lbuf, a0, b0, a1, b1);               // step nk-2
            adv();
            last_step(lbuf, a1, b1);                        // step nk-1
            adv();
#ifdef IGEMM_STAMPS
            if (ti == 0) PSTAMP(2);
            if (ti == 1) PSTAMP(4);
            if (!has_next) PSTAMP(6);
#endif
            // channel tiles (of 16) this wave holds inside Cout: 4 but for a ragged last tile (Cout % 16 == 0, host-checked)
            nst = __builtin_amdgcn_readfirstlane(min(MT, max(0, (p.Cout - co0_cur - wco * WCH) >> 4)));
            epilogue(ti & 1, co0_cur, nst, ks_cur);
            ks_cur = ks_next;
#ifdef IGEMM_STAMPS
            if (ti == 0) PSTAMP(3);
            if (ti == 1) PSTAMP(5);
#endif
            // fragments of the next tile's step 0: its stage landed (wait of step nk-1) and is visible (barrier of step nk-1)
            // (behind the last tile: of the restaged tile, unused)
            rd(rbuf == 0 ? NST - 1 : rbuf - 1, a0, b0);
            if (co0_next != co0_cur) load_stash(co0_next);  // (the same wave writes and reads its stash: in order)
            co0_cur = co0_next;
            j_cur = j_next;
            j_next = __builtin_amdgcn_readfirstlane(mail[(ti + 1) & 1]);     // published in front of the barrier of step nk-3
            ++ti;
            if (!has_next) n_lim = ti;
        }
        wait_vmcnt<0>();                                    // no LDS-DMA may outlive the workgroup (and the last draw has returned)
        // the label's last workgroup to leave puts both counters back to zero for the next launch that uses this slot: nobody of the
        // label draws any more (every workgroup's draws have returned before it counts itself out)
        if (threadIdx.x == 0) {
            const unsigned gone = atomicAdd(p.tile_ctr + 8 + xl, 1u);
            if (gone == (unsigned)nwg_x - 1u) {
                atomicExch(p.tile_ctr + xl, 0u);
                atomicExch(p.tile_ctr + 8 + xl, 0u);
            }
        }
    };
    if (grp == 0) run(std::integral_constant<int, NT0>{});
    else run(std::integral_constant<int, NT1>{});
#ifdef IGEMM_STAMPS
    if (p.dbg && lane == 0 && blockIdx.x < 512) {
        tstamp[7] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 8; ++i) p.dbg[((long)blockIdx.x * 8 + wave) * 8 + i] = tstamp[i];
    }
#endif
#undef PSTAMP
}

static void magic_u31(unsigned d, unsigned &magic, unsigned &shift)
{
    // floor(n / d) = umulhi(n, magic) >> shift for every n < 2^31 (Granlund & Montgomery; magic = ceil(2^(31+l) / d), l = ceil(log2 d))
    if (d <= 1) { magic = 0; shift = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned long long num = 1ull << (31 + l);
    magic = (unsigned)((num + d - 1) / d);
    shift = l - 1;
}

// Tile counters: a ring of slots (16 words each: 8 draw counters + 8 leave counters, one pair per XCD label), one slot per launch so that
// launches on different streams never share counters; a slot is zero again when its kernel has finished (the kernel resets it), and comes
// round after RING launches.  Allocated and cleared once per device.
static unsigned *persist_counters(int dev)
{
    constexpr unsigned RING = 1024;
    static unsigned *ring[64] = {};
    static std::atomic<unsigned> next[64];
    static std::mutex mu;
    if (!ring[dev]) {
        std::lock_guard<std::mutex> g(mu);
        if (!ring[dev]) {
            unsigned *r = nullptr;
            if (hipMalloc(&r, RING * 16 * sizeof(unsigned)) != hipSuccess) return nullptr;
            // (hipMemset on device memory may return before it has run, and the launch streams do not wait for the null stream)
            if (hipMemset(r, 0, RING * 16 * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
            ring[dev] = r;
        }
    }
    return ring[dev] + 16u * (next[dev].fetch_add(1u, std::memory_order_relaxed) % RING);
}

template <int NTILES, bool POOL, int DGRAD, bool CODES = false, bool SLAB = false, int MT = 4>
static int persist_launch(const IgemmParams &p, int splits, hipStream_t s)
{
    using C = PersistCfg<NTILES>;
    static bool attr_done[64] = {};
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm_persist_kernel<NTILES, POOL, DGRAD, CODES, SLAB, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return fail((int)e, "yolo_igemm: hipFuncSetAttribute(%d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
        attr_done[dev] = true;
    }
    IgemmParams q = p;
    q.n_co_tiles = (p.Cout + 64 * MT - 1) / (64 * MT);
    if (POOL) q.tpx_valid = C::TPX;
    if (q.tpx_valid <= 0 || q.tpx_valid > C::TPX) q.tpx_valid = C::TPX;
    q.n_px_tiles = (int)((p.M + q.tpx_valid - 1) / q.tpx_valid);
    q.nk = (int)(p.Ktot / C::BK);
    if (p.px_fastest < 0) q.px_fastest = 0;
    magic_u31((unsigned)p.HoWo, q.div_hw_magic, q.div_hw_shift);
    magic_u31((unsigned)p.Wo, q.div_w_magic, q.div_w_shift);
    magic_u31((unsigned)p.Wo / 2, q.div_hw2_magic, q.div_hw2_shift);
    q.nk_per_split = q.nk / (SLAB ? splits : 1);
    const long tiles = (long)q.n_co_tiles * q.n_px_tiles * (SLAB ? splits : 1);
    const int G = (int)std::min<long>(tiles, cus[dev]);
    q.pool_tiles_x = SLAB ? splits : 1;
    q.tile_ctr = persist_counters(dev);
    if (!q.tile_ctr) return fail(2, "yolo_igemm: cannot allocate the tile counters");
    hipLaunchKernelGGL((igemm_persist_kernel<NTILES, POOL, DGRAD, CODES, SLAB, MT>), dim3(G), dim3(C::NTHR), C::LDS_BYTES, s, q);
    return check_launch("yolo_igemm (persistent)");
}

int igemm_persist_launch(const IgemmParams &p, int hint, int splits, hipStream_t s)
{
    const bool slab = splits > 1;
    if (p.stats || p.w_blocked || p.px_begin || (slab ? (!p.slab_stride || !p.out_fp32 || p.epilogue != YOLO_EPI_NONE || p.pool) : (p.slab_stride || p.out_fp32)))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d is a bf16 launch, or split-K into fp32 slabs (split_slabs = 1, epilogue NONE); no bn_stats, blocked weights, pixel range", hint);
    if (p.tap_len % 32) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d needs tap_len %% 32 == 0", hint);
    const long nk_all = p.Ktot / 32;
    if (nk_all % splits) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d splits K into equal ranges", hint);
    const long nk = nk_all / splits;
    if (nk % 2 || nk < 6) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d needs an even number (>= 6) of 32-deep K steps per range", hint);
    if (p.epilogue < YOLO_EPI_NONE || p.epilogue > YOLO_EPI_BIAS_ADD_LRELU) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d has no epilogue %d", hint, p.epilogue);
    if (!(p.slope >= 0.0f && p.slope <= 1.0f)) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d needs 0 <= slope <= 1", hint);
    if (p.Cout > 256 && p.Cout % 256) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d needs Cout <= 256 or Cout %% 256 == 0", hint);
    if (p.Cout % 16) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d needs Cout %% 16 == 0", hint);
    if (p.M >= (1L << 31)) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d indexes fewer than 2^31 output pixels", hint);
    // every operand is addressed as base + 32-bit byte offset
    const long n_img = p.M / p.HoWo + 1;
    const long in_bytes = (n_img * p.in_img_stride + (long)p.KH * p.in_row_stride) * 2, w_bytes = (long)p.Cout * p.Ktot * 2;
    const long out_bytes = (n_img * p.out_img_stride + p.out_off) * (slab ? 4 : 2), aux_bytes = p.aux ? (n_img * p.aux_img_stride + p.aux_off) * 2 : 0;
    if (in_bytes >= (1L << 32) || w_bytes >= (1L << 32) || out_bytes >= (1L << 32) || aux_bytes >= (1L << 32))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d addresses operands below 4 GB", hint);
    if ((p.out_off | p.out_px_stride | p.out_row_stride) & 3 || (p.out_img_stride & 3))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d stores 8-byte pieces: output strides in multiples of 4 elements", hint);
    if ((p.epilogue == YOLO_EPI_MUL_DLRELU || p.epilogue == YOLO_EPI_BIAS_ADD_LRELU) && ((p.aux_off | p.aux_px_stride | p.aux_row_stride) & 3 || (p.aux_img_stride & 3)))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d reads aux in 8-byte pieces", hint);
    if (p.pool) {
        if (hint != 21 || (p.pool != 1 && p.pool != 3) || !(p.Wo == 112 || p.Wo == 56 || p.Wo == 28) || (p.HoWo / p.Wo) % 2 || p.M % 224
            || (p.epilogue != YOLO_EPI_BIAS && p.epilogue != YOLO_EPI_BIAS_LRELU))
            return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: the pooled epilogue of tile_hint 21 needs pool2 = 1 or 3, rows of 112, 56 or 28 pixels and whole 224-pixel tiles");
        if (p.pool == 3 && ((p.out_off | p.out_px_stride | p.out_row_stride) & 7 || (p.out_img_stride & 7) || (p.Cout & 7)))
            return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: pool2 = 3 needs Cout and the output strides in multiples of 8");
        if (p.Cout > 128 && p.Cout <= 192)        // three 16-channel tiles per wave: no MFMA spent on 64 channels of padding
            return p.pool == 3 ? persist_launch<14, true, 0, true, false, 3>(p, 1, s) : persist_launch<14, true, 0, false, false, 3>(p, 1, s);
        return p.pool == 3 ? persist_launch<14, true, 0, true>(p, 1, s) : persist_launch<14, true, 0>(p, 1, s);
    }
    if (slab && p.Cout % 256) return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint %d splits K only for Cout %% 256 == 0", hint);
    if (slab) return hint == 20 ? persist_launch<13, false, 0, false, true>(p, splits, s) : persist_launch<14, false, 0, false, true>(p, splits, s);
    const int dg = p.epilogue == YOLO_EPI_MUL_DLRELU ? 1 : (p.epilogue == YOLO_EPI_BIAS_ADD_LRELU ? 2 : 0);
    switch (hint) {
    case 20: return dg == 1 ? persist_launch<13, false, 1>(p, 1, s) : (dg == 2 ? persist_launch<13, false, 2>(p, 1, s) : persist_launch<13, false, 0>(p, 1, s));
    case 21: return dg == 1 ? persist_launch<14, false, 1>(p, 1, s) : (dg == 2 ? persist_launch<14, false, 2>(p, 1, s) : persist_launch<14, false, 0>(p, 1, s));
    }
    return fail(YOLO_E_ARG, "yolo_igemm: tile_hint %d is not a persistent configuration", hint);
}

}  // namespace yolo
