// Pieces shared by the weight-gradient kernels (wgrad.hip, wgrad_pipe.hip): launch parameters and the workgroup -> (tile,
// pixel range) map of the two-segment schedule.
#pragma once
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace yolo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;


struct WgradParams {
    const bf16_t *x;
    const bf16_t *dy;
    float *dw;
    float *db;              // optional: bias gradient, accumulated by the (tap 0, ci-tile 0) workgroups
    double *sumsq;          // optional (128 x 128 kernel, tiles stored by ONE workgroup): += sum of squares of the dw this launch writes
    float *slabs;           // optional (wgrad_pipe.hip): every workgroup STORES its 256 x 256 partial tile at slabs + part * 65536 (part = its
                            // (tile, pixel range) number, see wgrad_map) instead of adding it to dw; wgrad_slab_sum_kernel adds them up in range order
    long P;                 // pixels to reduce over
    // pixel p -> slot in the dy / x buffers.  gW == 0: p is the slot ("flat" indexing).  Else p = (n*gH + oy)*gW + ox and
    // slot = n*g_img + oy*g_row + ox*g_px + g_off (interior pixels of a zero-haloed buffer, optionally every 2nd one);
    // mW, mH = ceil(2^32 / gW), ceil(2^32 / gH) turn the small divisions into v_mul_hi_u32
    int gW, gH, g_img, g_row, g_px, g_off;
    unsigned mW, mH;
    long p_per_split;
    int dy_px_stride, x_px_stride;
    int Cout, Cin, Cout_ld, Cin_ld;  // logical sizes and loadable (multiple-of-8) widths
    int KH, KW, pad;
    long x_row_stride;
    int n_co_tiles, n_ci_tiles, ntaps;
    int pair_taps;          // Cin == 64 (128 x 128 kernel): a ci-tile holds TWO taps (columns 0..63 / 64..127 are adjacent in
                            // dw[co][tap][ci]), so half the tile is not wasted on padding; tile index = pair index then
    long *dbg;              // diagnostic builds (-DIGEMM_STAMPS): s_memtime stamps of stage dbg_it (yolo_debug_stamps)
    int dbg_it;
    int tile_taps;          // wgrad_pipe (256-column tiles): taps per ci-tile = 256 / Cin when Cin < 256 divides 256 (columns of adjacent
                            // taps are adjacent in dw[co][tap][ci]), else 1
    int atomic;             // uniform split: accumulate with atomics.  Two-segment schedule: bit 0 = main segment, bit 1 = tail
    // two-segment schedule (seg = 1, 1-D grid): the first main_tiles tiles are split into main_split pixel ranges and fill
    // whole rounds of the chip's 512 workgroup slots; the remaining tail_tiles (< 512 / main_split) tiles are split finer
    // (tail_split ranges) so that they fill one more, shorter round instead of leaving most CUs idle for a full-length one
    int seg, slots;         // slots: resident workgroups on the chip (512 for the 128x128 kernel, 256 for the 256x128 ones)
    int main_tiles, main_split, tail_tiles, tail_split;
    long per_main, per_tail;
};

constexpr int WG_SLOTS = 512;   // 256 CUs x 2 resident workgroups (64 KB of LDS each)

// workgroup -> (logical tile id, first pixel, pixels).  Hardware hands consecutive workgroup ids to the 8 XCDs round-robin;
// within every group of 512 ids an XCD gets 64 CONSECUTIVE logical workgroups: same pixel range, neighbouring tiles
// (co fastest), so the dy / x rows they stream are shared through that XCD's L2.
__device__ __forceinline__ void wgrad_map(const WgradParams &p, int nwg, int &bid, long &pbeg, long &pend, bool &atomic, int &part)
{
    atomic = p.atomic & 1;
    part = 0;
    if (p.seg) {
        const int id = blockIdx.x, per_xcd = p.slots >> 3;
        const int L = id / p.slots * p.slots + (id & 7) * per_xcd + ((id % p.slots) >> 3);
        const int nmain = p.main_tiles * p.main_split;
        int range;
        long per;
        if (L < nmain) {
            const int tpr = p.slots / p.main_split;           // tiles per round
            const int round = L / p.slots, within = L % p.slots;
            range = within / tpr;
            bid = round * tpr + within % tpr;
            per = p.per_main;
            part = bid * p.main_split + range;
        } else {
            const int t = L - nmain, tt = max(p.tail_tiles, 1);
            range = p.tail_tiles > 0 ? t / tt : p.tail_split;   // no tail: past every range -> empty
            bid = p.main_tiles + t % tt;
            per = p.per_tail;
            atomic = (p.atomic >> 1) & 1;
            part = nmain + (t % tt) * p.tail_split + range;
        }
        pbeg = (long)range * per;
        pend = min(p.P, pbeg + per);
        if (bid >= nwg) pend = pbeg;   // ids past the last logical workgroup (grid rounded up to the XCD map)
    } else {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        pbeg = (long)blockIdx.y * p.p_per_split;
        pend = min(p.P, pbeg + p.p_per_split);
        part = bid * (int)gridDim.y + (int)blockIdx.y;
    }
}

#define GLDS16(gptr, lptr) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// wgrad_pipe.hip: 256 x 256 tile, wave tile 128 x 64, register-pipelined one-barrier loop (yolo_wgrad_desc.variant = 5)
int wgrad_pipe_launch(const WgradParams &p, dim3 grid, hipStream_t s);
// wgrad_wide.hip: the same tiles and schedule, four waves of 128 x 128 with the accumulators in AGPRs (yolo_wgrad_desc.variant = 6)
int wgrad_wide_launch(const WgradParams &p, dim3 grid, hipStream_t s);
// ... slab mode: sum of the partial tiles into the packed gradient (main_ranges / tail_ranges = pixel ranges that hold pixels)
int wgrad_slab_sum_launch(const WgradParams &p, int tiles, int main_ranges, int tail_ranges, hipStream_t s);
// igemm.hip: the debug buffer of yolo_debug_stamps (diagnostic builds)
void debug_stamp_target(long **buf, int *it);

}  // namespace yolo
