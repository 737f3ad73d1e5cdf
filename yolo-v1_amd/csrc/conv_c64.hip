// 3x3 / stride-1 / pad-1 convolution with 64 input and 64 output channels (+ bias, LeakyReLU / ReLU) for gfx950 -- yolo_igemm_desc.tile_hint = 22.
// The three 3x3 convs of ResNet-50's first stage (src/yolo/models.py:131-176: torchvision's layer1 bottlenecks at 112 x 112 for a 448 x 448 input,
// BatchNorm folded in inference) are 59 GFLOP over 206 MB each.  Through the tiled implicit GEMM they run at 500 TFLOP/s: a 64-channel output fills a
// quarter of a 256-channel tile or, with 64 x 128 tiles, stages (64 + 128) x 64 B per 0.26 MFLOP -- the K loop is bound by the issue of its LDS-DMA
// instructions.  Here nothing is staged per K step:
//   * the WHOLE weight panel (64 x 576 bf16 = 72 KB) sits in LDS for the life of the workgroup, in MFMA-fragment order;
//   * a workgroup walks 16 x 16-pixel output tiles; the input patch of a tile (18 x 18 pixels x 128 B = 40.5 KB, double-buffered) is fetched ONCE by
//     41 LDS-DMA instructions and the nine taps read it at shifted addresses -- 1.27 fetched bytes per input byte instead of 9;
//   * wave w owns output rows 2w, 2w + 1 of the tile x all 64 channels: 18 K blocks (tap-major, the implicit GEMM's order, so the fp32 sums are
//     bit-identical) x 8 MFMAs (v_mfma_f32_16x16x32_bf16), operands by ds_read_b128 from conflict-free images (below), no barrier inside a tile;
//   * outputs go through the dead patch buffer to become 16-byte coalesced stores.
// LDS: 72 KB weights + 2 x 41 KB patches = 154 KB of the 160: one workgroup of eight waves per CU.
#include "igemm_common.h"

namespace yolo {

namespace c64 {
constexpr int NW = 8, NTHR = NW * 64;
constexpr int TH = 16, TW = 16;                   // output tile
constexpr int PW = TW + 2, PH = TH + 2;           // patch
constexpr int PATCH_PX = PW * PH;                 // 324
constexpr int PIECES = (PATCH_PX + 7) / 8;        // LDS-DMA instructions per patch: 8 pixels x 128 B each (41; the last one half empty)
constexpr int PATCH_BYTES = PIECES * 1024;        // 41,984
constexpr int W_BYTES = 64 * 576 * 2;             // 73,728
constexpr int PPW = (PIECES + NW - 1) / NW;       // pieces per wave (6; the tail waves' last piece repeats piece 40)
constexpr int OP = 64 + 8;                        // bf16 pitch of the output staging tile (144 B: 16-B aligned, bank-spread)
constexpr int LDS_BYTES = W_BYTES + 2 * PATCH_BYTES;
static_assert(TH * TW * OP * 2 <= PATCH_BYTES, "the output tile is staged in the dead patch buffer");
}  // namespace c64

// LDS-DMA from inline asm: through the builtin hipcc treats every LDS read as a possible alias of an LDS-DMA in flight and drains vmcnt(0) in front of
// it -- the next tile's patch would be waited for before the current tile's first operand read.  The waits are counted by hand below.
#define C64_DMA16(voff, base, lds) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0")

typedef __attribute__((ext_vector_type(8))) __bf16 c64_bf16x8;
typedef __attribute__((ext_vector_type(4))) float c64_f32x4;

// position of 16-byte chunk c (8 channels) of patch pixel pi inside the pixel's 128 bytes.  A ds_read_b128 is served in four groups of 16 lanes
// that are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and likewise above (MI355X_MICROARCH.md, LDS): for the 16x16x32 operand map (lane = pixel
// n16 = l & 15, chunk l >> 4) a group is eight pixels at chunk e and the other eight at chunk e + 1.  With 128-byte pixels the 16-B slot modulo 256 B
// is 8 * (pi & 1) + position, so the eight even and the eight odd pixels of a group each need eight distinct positions whatever the tap shift --
// c ^ (2 * ((pi >> 1) & 3)) does that (checked exhaustively over all alignments; c ^ ((pi >> 1) & 7) does not).
__device__ __forceinline__ int c64_pos(int pi, int c)
{
    return c ^ (2 * ((pi >> 1) & 3));
}

// STATS: BatchNorm's per-channel sum / sum of squares of the stored (bf16-rounded) outputs as well (yolo_igemm_desc.bn_stats, epilogue NONE: the raw conv of the
// ResNet trunk in training mode).  In the store phase a thread always handles the same 8-channel chunk (tid & 7), so its sixteen partial sums live in registers
// for the workgroup's whole walk; one fold through LDS and 128 fp64 atomics per workgroup at the end.
template <bool STATS>
__global__ void __launch_bounds__(c64::NTHR) conv_c64_kernel(const IgemmParams p, int tiles_x, int tiles_y, int ntiles)
{
    using namespace c64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *wl = smem;                                  // weights: chunk (kblk, kg, co) at ((kblk * 4 + kg) * 64 + co) * 16
    char *patch = smem + W_BYTES;                     // [2][PATCH_BYTES]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kb = lane >> 4;

    // ---- LDS-DMA source offsets of this lane's pieces (the same for every tile): piece q covers patch pixels 8 q .. 8 q + 7, lane l = pixel 8 q + (l >> 3),
    // position l & 7, which holds chunk c with c64_pos(pi, c) = l & 7 (the XOR is its own inverse)
    unsigned src_off[PPW];
    int piece_of[PPW];                  // (wave-uniform)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int q = i * NW + wave;
        if (q >= PIECES) q = PIECES - 1;
        piece_of[i] = q;
        int pi = q * 8 + (lane >> 3);
        const int pos = lane & 7;
        const int c = c64_pos(pi, pos);
        if (pi >= PATCH_PX) pi = PATCH_PX - 1;          // (the last piece's upper half: a valid address, LDS bytes nobody reads)
        const int pr = pi / PW, pc = pi - pr * PW;
        src_off[i] = (unsigned)((pr * p.in_row_stride + pc * p.in_px_stride + c * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    auto stage = [&](int buf, int tile) {
        const int tx = tile % tiles_x, r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const char *xb = reinterpret_cast<const char *>(p.in + (long)n * p.in_img_stride + (long)(ty * TH) * p.in_row_stride + (long)(tx * TW) * p.in_px_stride + p.in_off);
        const unsigned sb = lds0 + W_BYTES + buf * PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < PPW; ++i) C64_DMA16(src_off[i], xb, sb + piece_of[i] * 1024);
    };

    int tile = blockIdx.x;
    if (tile < ntiles) stage(0, tile);

    // ---- weights -> LDS (once): W[co][k], k = tap * 64 + ci
    for (int c = tid; c < 64 * 72; c += NTHR) {
        const int co = c & 63, kc = c >> 6;             // kc = kblk * 4 + kg
        *reinterpret_cast<uint4 *>(wl + ((long)kc * 64 + co) * 16) = *reinterpret_cast<const uint4 *>(p.w + (long)co * 576 + kc * 8);
    }
    // bias of the 16 accumulator rows this lane owns: co = mt * 16 + 4 * kb + r
    const bool has_bias = p.epilogue == YOLO_EPI_BIAS || p.epilogue == YOLO_EPI_BIAS_LRELU;
    const float slope = p.epilogue == YOLO_EPI_BIAS_LRELU ? p.slope : 1.0f;
    float bias_r[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[mt][r] = has_bias ? p.bias[mt * 16 + 4 * kb + r] : 0.0f;

    // ---- operand addresses (tile-invariant): A fragment of (kblk, mt): 16 B at ((kblk * 4 + kb) * 64 + mt * 16 + n16) * 16;
    // B fragment of (row rr of the wave, tap, channel block cb): patch pixel (2 wave + rr + ky, n16 + kx), chunk cb * 4 + kb
    const int a_base = (kb * 64 + n16) * 16;
    int b_addr[2][9][2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int pi = (2 * wave + rr + ky) * PW + n16 + kx;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) b_addr[rr][tap][cb] = pi * 128 + c64_pos(pi, cb * 4 + kb) * 16;
        }

    float red[16] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
        const int buf = it & 1;
        const char *sb = patch + buf * PATCH_BYTES;
        // The patch of this tile has landed: its pieces are older than the previous tile's four stores per lane, which may stay in flight (the VMEM
        // counter retires in order).  Then everyone's pieces are visible, and every wave is past the previous tile's store phase, which read the
        // OTHER buffer -- the next patch may go there.
        if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TH * TW * 8 / NTHR) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (tile + (int)gridDim.x < ntiles) stage(buf ^ 1, tile + gridDim.x);

        c64_f32x4 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) acc[mt][rr] = c64_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // fragments of K block k + 1 are read under the MFMAs of block k (two register sets): one ds_read_b128 behind each of the first six MFMAs
        c64_bf16x8 af[2][4], bf[2][2];
        auto rd = [&](int kblk, int set) {
            const int tap = kblk >> 1, cb = kblk & 1;
            // issue order = the order in which the block's MFMAs (mt-major) first need the operands: a0, b0, b1, a1, a2, a3
            auto ra = [&](int mt) { af[set][mt] = *reinterpret_cast<const c64_bf16x8 *>(wl + kblk * 4096 + a_base + mt * 256); };
            auto rb = [&](int rr) { bf[set][rr] = *reinterpret_cast<const c64_bf16x8 *>(sb + b_addr[rr][tap][cb]); };
            ra(0); rb(0); rb(1); ra(1); ra(2); ra(3);
        };
        rd(0, 0);
#pragma unroll
        for (int kblk = 0; kblk < 18; ++kblk) {
            const int set = kblk & 1;
            if (kblk + 1 < 18) rd(kblk + 1, set ^ 1);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) acc[mt][rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[set][mt], bf[set][rr], acc[mt][rr], 0, 0, 0);
            if (kblk + 1 < 18) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);          // (the scheduler must not pull the next block's MFMAs -- and with them the waits on these reads -- up here)
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();          // every wave is done with the patch (its reads fed the MFMAs): the buffer becomes the output staging tile [256 px][64 co] bf16
        __builtin_amdgcn_sched_barrier(0);
        bf16_t *otile = reinterpret_cast<bf16_t *>(patch + buf * PATCH_BYTES);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int px = (2 * wave + rr) * TW + n16;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[mt][rr][r] + bias_r[mt][r];
                    v[r] = v[r] > 0.0f ? v[r] : v[r] * slope;
                }
                uint2 o;
                o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                *reinterpret_cast<uint2 *>(otile + px * OP + mt * 16 + 4 * kb) = o;      // D row (co) = 4 kb + r, column (pixel) = n16
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // coalesced stores: 16 B (8 channels) per lane, 8 lanes per pixel.  (Measured against 8-byte stores straight from the accumulators, which save the
        // two barriers and the LDS round trip: 63 vs 66 us per layer -- the narrow stores cost more than they save.)
        const int tx = tile % tiles_x, r0 = tile / tiles_x;
        const int ty = r0 % tiles_y, n = r0 / tiles_y;
        bf16_t *ob = reinterpret_cast<bf16_t *>(p.out) + (long)n * p.out_img_stride + (long)(ty * TH) * p.out_row_stride + (long)(tx * TW) * p.out_px_stride + p.out_off;
#pragma unroll
        for (int j = 0; j < TH * TW * 8 / NTHR; ++j) {
            const int q = j * NTHR + tid, px = q >> 3, c8 = q & 7;
            const int oy = px / TW, ox = px - oy * TW;
            const uint4 pk = *reinterpret_cast<const uint4 *>(otile + px * OP + c8 * 8);
            *reinterpret_cast<uint4 *>(ob + (long)oy * p.out_row_stride + ox * p.out_px_stride + c8 * 8) = pk;
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
    if constexpr (STATS) {
        // fold the 64 threads of each channel chunk (tid & 7) through LDS, then one fp64 atomic pair per channel into one of YOLO_BN_ACC_REPLICAS accumulators
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float *all = reinterpret_cast<float *>(patch);
#pragma unroll
        for (int k = 0; k < 16; ++k) all[tid * 16 + k] = red[k];
        __syncthreads();
        if (tid < 64) {
            const int c8 = tid >> 3, k = tid & 7;
            float a = 0.0f, b = 0.0f;
            for (int j = 0; j < NTHR / 8; ++j) { a += all[(j * 8 + c8) * 16 + k]; b += all[(j * 8 + c8) * 16 + 8 + k]; }
            double *rep = p.stats + (size_t)(blockIdx.x % YOLO_BN_ACC_REPLICAS) * 2 * 64;
            atomicAdd(rep + tid, (double)a);
            atomicAdd(rep + 64 + tid, (double)b);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup
}

int conv_c64_launch(const IgemmParams &p, hipStream_t s)
{
    using namespace c64;
    const int Ho = p.HoWo / p.Wo, Wo = p.Wo;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.tap_len != 64 || p.Cout != 64 || p.out_fp32 || p.pool || p.w_blocked || p.px_begin != 0 ||
        (p.epilogue != YOLO_EPI_NONE && p.epilogue != YOLO_EPI_BIAS && p.epilogue != YOLO_EPI_BIAS_LRELU) || (Ho % TH) || (Wo % TW) || (p.in_px_stride & 7) ||
        (p.in_row_stride & 7) || (p.in_img_stride & 7) || (p.in_off & 7) || (p.out_px_stride & 7) || (p.out_row_stride & 7) || (p.out_img_stride & 7) || (p.out_off & 7))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint 22 is the 3x3 / stride-1 conv of 64 -> 64 channels on maps of (16k) x (16k) pixels, bf16 out, 16-B aligned strides");
    static bool attr_done[64] = {};
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_c64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_c64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return fail((int)e, "yolo_igemm: hipFuncSetAttribute(%d B LDS): %s", LDS_BYTES, hipGetErrorString(e));
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
        attr_done[dev] = true;
    }
    const int tiles_x = Wo / TW, tiles_y = Ho / TH;
    const long ntiles = (long)(p.M / p.HoWo) * tiles_x * tiles_y;
    const int G = (int)std::min<long>(ntiles, cus[dev]);
    if (p.stats) hipLaunchKernelGGL(conv_c64_kernel<true>, dim3(G), dim3(NTHR), LDS_BYTES, s, p, tiles_x, tiles_y, (int)ntiles);
    else hipLaunchKernelGGL(conv_c64_kernel<false>, dim3(G), dim3(NTHR), LDS_BYTES, s, p, tiles_x, tiles_y, (int)ntiles);
    return check_launch("yolo_igemm (3x3 64 -> 64)");
}

}  // namespace yolo
