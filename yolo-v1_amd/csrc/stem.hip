// Forward of the 7x7 / stride-2 / pad-3 stem convolution (3 input channels stored NHWC4 with halo 3, 64 output
// channels) + bias + LeakyReLU (+ MaxPool2d(2,2)) for gfx950.  Replaces aten convolution + leaky_relu (+ max_pool2d) for
// src/yolo/models.py:49-51 (and the ResNet stem conv1+bn1+relu with the BatchNorm folded, slope 0).
//
// The generic implicit GEMM (igemm.hip) gathers, for every output pixel and every ky, the 32-element run
// (kx 0..7, c 0..3) of the input row straight from L2: neighbouring pixels' runs overlap 4x (stride 2 pixels = 16 B of
// a 64-B run), so the stem moved 448 B per output pixel through the L2 -> LDS path (1.44 GB per batch-64 forward) and
// ran at the gather rate, not at the MFMA or HBM rate.  Here a workgroup stages the RAW input patch of an 8 x 16 output
// tile once (21 x 40 pixels x 8 B = 6.7 KB for 128 pixels: 52 B per output pixel) and the overlap is resolved by the
// LDS read addresses: the B fragment of v_mfma_f32_16x16x32_bf16 for pixel (r, n) and k-block kb of tap row ky is the
// 16-byte run at patch pixel (2r + ky, 2n + 2kb) -- one ds_read_b128, conflict-free (16 lanes = 256 contiguous bytes).
// The whole weight panel (64 x 7 x 32 bf16 = 28 KB) lives in REGISTERS (112 VGPRs of A fragments per lane, loaded once
// per persistent workgroup); wave w computes output rows 2w, 2w+1 of the tile x all 64 channels: 56 MFMAs per tile.
// Measured at batch 64 (tools/time_stem.py): 0.13 ms with the fused pool (generic kernel: 0.20), of which the MFMA phase
// alone is 0.09 (ablation): 56 MFMAs per wave and tile do not amortise the per-tile LDS latency and epilogue VALU at two
// waves per SIMD -- the next step would be a 16 x 16 tile.
// Epilogue: accumulators + bias -> LeakyReLU -> bf16 LDS tile [128 px][64 co] -> (2x2 max of the rounded values, which is
// what a separate MaxPool2d over the stored activation computes) -> 16-B coalesced stores.
#include <algorithm>

#include "common.h"

namespace yolo {

typedef __attribute__((ext_vector_type(8))) __bf16 sbf16x8;
typedef __attribute__((ext_vector_type(4))) float sf32x4;

#define STEM_GLDS16(gptr, lptr) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

constexpr int SF_TH = 8, SF_TW = 16;            // output tile
constexpr int SF_PW = 40, SF_PH = 21;           // patch pitch (pixels) / rows
constexpr int SF_X_BYTES = 7 * 1024;            // 7 LDS-DMA wave-instructions >= 21 * 40 * 8 B
constexpr int SF_OP = 64 + 8;                   // bf16 pitch of the output staging tile (144 B: 16-B aligned, bank-spread)
constexpr int SF_OUT_BYTES = SF_TH * SF_TW * SF_OP * 2;

struct StemFwdParams {
    const bf16_t *x;        // NHWC4, halo 3
    const bf16_t *w;        // [64][7][8][4] packed bf16 (kx 7 and c 3 are zero)
    const float *bias;
    bf16_t *out;
    int tiles_x, tiles_y, ntiles;
    long x_img_stride, out_img_stride;
    int x_row_stride, out_row_stride, out_off;
    int pool;
    float slope;
    bf16_t *full;           // pool mode only, optional: the un-pooled activation as well (training keeps it for the backward pass)
    unsigned short *codes;  // pool mode only, optional (pool2 = 3): arg-max codes instead, uint16 per (pooled pixel, 8 channels) at (pooled address) / 8
    long full_img_stride;
    int full_row_stride, full_off;
    // F32IN: the input is the caller's NCHW fp32 tensor (3 planes of H x W): the patch is converted to NHWC4 bf16 on its way into LDS,
    // so the separate layout pass (154 MB read + 103 MB written + 103 MB read again at batch 64) disappears
    const float *x32;
    int H, W;
};

template <bool F32IN>
__global__ void __launch_bounds__(256, 2) stem_fwd_kernel(const StemFwdParams p)
{
    __shared__ __attribute__((aligned(16))) char patchA[SF_X_BYTES];
    __shared__ __attribute__((aligned(16))) char patchB[SF_X_BYTES];
    __shared__ __attribute__((aligned(16))) bf16_t otileA[SF_TH * SF_TW * SF_OP];
    __shared__ __attribute__((aligned(16))) bf16_t otileB[SF_TH * SF_TW * SF_OP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kb = lane >> 4;

    // ---- weights -> registers: A[m = co][k]: lane holds co = mb*16 + n16, k = kb*8 .. +7 of tap row ky
    sbf16x8 afrag[4][7];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
            afrag[mb][ky] = *reinterpret_cast<const sbf16x8 *>(p.w + ((mb * 16 + n16) * 7 + ky) * 32 + kb * 8);
    // bias of the 16 accumulator rows this lane owns: co = mb*16 + 4*kb + r
    float bias_r[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[mb][r] = p.bias[mb * 16 + 4 * kb + r];

    // ---- patch staging (same slots as stem_wgrad_kernel): 20 16-B slots per patch row
    int x_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int slot = (i * 4 + wave) * 64 + lane;
        if (slot >= SF_PH * (SF_PW / 2)) slot = SF_PH * (SF_PW / 2) - 1;
        x_off[i] = (slot / (SF_PW / 2)) * p.x_row_stride + (slot % (SF_PW / 2)) * 8;
    }
    auto stage = [&](char *sb, int tile) {
        const int tx = tile % p.tiles_x, r = tile / p.tiles_x;
        const int ty = r % p.tiles_y, n = r / p.tiles_y;
        const bf16_t *xb = p.x + (long)n * p.x_img_stride + (long)(ty * SF_TH * 2) * p.x_row_stride + tx * SF_TW * 2 * 4;
        STEM_GLDS16(xb + x_off[0], sb + wave * 1024);
        if (wave < 3) STEM_GLDS16(xb + x_off[1], sb + (4 + wave) * 1024);
    };

    // F32IN: the 21 x 40-pixel patch of a tile = 21 rows x 11 aligned float4 x 3 planes (the patch starts one pixel behind a multiple of
    // four; W % 4 == 0, so a quad lies entirely inside or outside the image).  A thread takes up to three (row, plane, quad) items:
    // loaded into registers one tile ahead, written to LDS as the bf16 channel c of four NHWC4 pixels after the MFMA phase of the current
    // tile (channel 3 is zeroed once); pixels outside the image are the conv's zero padding.
    constexpr int F32_ITEMS = SF_PH * 3 * 11, F32_PER_THREAD = (F32_ITEMS + 255) / 256;
    float4 xr[F32_PER_THREAD];
    if constexpr (F32IN) {
        for (int i = tid; i < SF_X_BYTES / 16; i += 256) {
            *reinterpret_cast<uint4 *>(patchA + i * 16) = uint4{0u, 0u, 0u, 0u};
            *reinterpret_cast<uint4 *>(patchB + i * 16) = uint4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
    }
    auto f32_load = [&](int tile) {
        const int tx = tile % p.tiles_x, r = tile / p.tiles_x;
        const int ty = r % p.tiles_y, n = r / p.tiles_y;
        const float *xb = p.x32 + (long)n * 3 * p.H * p.W;
#pragma unroll
        for (int j = 0; j < F32_PER_THREAD; ++j) {
            const int item = j * 256 + tid;
            const int q = item % 11, rc = item / 11, c = rc % 3, pr = rc / 3;
            const int y = ty * SF_TH * 2 - 3 + pr, x = tx * SF_TW * 2 - 4 + 4 * q;
            const bool in = item < F32_ITEMS && y >= 0 && y < p.H && x >= 0 && x < p.W;
            xr[j] = in ? *reinterpret_cast<const float4 *>(xb + ((long)c * p.H + y) * p.W + x) : float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    auto f32_store = [&](char *sb) {
#pragma unroll
        for (int j = 0; j < F32_PER_THREAD; ++j) {
            const int item = j * 256 + tid;
            if (item >= F32_ITEMS) continue;
            const int q = item % 11, rc = item / 11, c = rc % 3, pr = rc / 3;
            const float v[4] = {xr[j].x, xr[j].y, xr[j].z, xr[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ppx = 4 * q - 1 + e;                      // patch pixel 0 = image x = 32 tx - 3
                if (ppx >= 0 && ppx < SF_PW) *reinterpret_cast<bf16_t *>(sb + (pr * SF_PW + ppx) * 8 + c * 2) = f32_to_bf16(v[e]);
            }
        }
    };

    // B fragment address of (local row rr of the wave, ky): pixel (2*(2*wave+rr) + ky, 2*n16 + 2*kb)
    const int b_base = ((4 * wave) * SF_PW + 2 * n16 + 2 * kb) * 8;

    auto mfma_phase = [&](const char *sb, bf16_t *otile) {
        sf32x4 acc[4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) acc[mb][rr] = sf32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            sbf16x8 bfr[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) bfr[rr] = *reinterpret_cast<const sbf16x8 *>(sb + b_base + ((2 * rr + ky) * SF_PW) * 8);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) acc[mb][rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[mb][ky], bfr[rr], acc[mb][rr], 0, 0, 0);
        }
        // ---- accumulators (+ bias, LeakyReLU when not pooling) -> otile[px][co] bf16; D row (co) = 4*kb + r, col (px) = n16
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int px = (2 * wave + rr) * SF_TW + n16;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[mb][rr][r] + bias_r[mb][r];
                    v[r] = v[r] > 0.0f ? v[r] : v[r] * p.slope;
                }
                uint2 o;
                o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                *reinterpret_cast<uint2 *>(otile + px * SF_OP + mb * 16 + 4 * kb) = o;
            }
        }
    };
    // The stores of a tile are issued one iteration LATER, from the other staging tile: at the per-tile wait (vmcnt(0) --
    // gfx9 counts stores there too) everything outstanding is then a full tile old; storing right after the MFMAs made
    // every tile wait out its own store latency.
    auto store_phase = [&](const bf16_t *otile, int tile) {
        // ---- coalesced stores: 16 B (8 channels) per lane
        const int tx = tile % p.tiles_x, r0 = tile / p.tiles_x;
        const int ty = r0 % p.tiles_y, n = r0 / p.tiles_y;
        bf16_t *ob = p.out + (long)n * p.out_img_stride + p.out_off;
        if (p.pool) {
            // 4 x 8 pooled pixels x 8 chunks = 256 lanes, one pass
            const int c8 = tid & 7, q = tid >> 3, qy = q >> 3, qx = q & 7;
            const bf16_t *s0 = otile + ((2 * qy) * SF_TW + 2 * qx) * SF_OP + c8 * 8;
            const uint4 a = *reinterpret_cast<const uint4 *>(s0), b = *reinterpret_cast<const uint4 *>(s0 + SF_OP);
            const uint4 c = *reinterpret_cast<const uint4 *>(s0 + SF_TW * SF_OP), d = *reinterpret_cast<const uint4 *>(s0 + SF_TW * SF_OP + SF_OP);
            const unsigned va[4] = {a.x, a.y, a.z, a.w}, vb[4] = {b.x, b.y, b.z, b.w}, vc[4] = {c.x, c.y, c.z, c.w}, vd[4] = {d.x, d.y, d.z, d.w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // bf16 max per half-word through the fp32 ordering
                const float lo = fmaxf(fmaxf(__uint_as_float(va[k] << 16), __uint_as_float(vb[k] << 16)), fmaxf(__uint_as_float(vc[k] << 16), __uint_as_float(vd[k] << 16)));
                const float hi = fmaxf(fmaxf(__uint_as_float(va[k] & 0xffff0000u), __uint_as_float(vb[k] & 0xffff0000u)),
                                       fmaxf(__uint_as_float(vc[k] & 0xffff0000u), __uint_as_float(vd[k] & 0xffff0000u)));
                o[k] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
            }
            const long pa = (long)(ty * (SF_TH / 2) + qy) * p.out_row_stride + (tx * (SF_TW / 2) + qx) * 64 + c8 * 8;
            *reinterpret_cast<uint4 *>(ob + pa) = uint4{o[0], o[1], o[2], o[3]};
            if (p.codes) {
                // window position of the first maximum (order (0,0),(0,1),(1,0),(1,1)), 2 bits per channel: with the pooled map, all the
                // backward pass needs of the un-pooled activation (yolo_wgrad_stem7_pooled codes form)
                unsigned code = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const float w0 = hf ? __uint_as_float(va[k] & 0xffff0000u) : __uint_as_float(va[k] << 16);
                        const float w1 = hf ? __uint_as_float(vb[k] & 0xffff0000u) : __uint_as_float(vb[k] << 16);
                        const float w2 = hf ? __uint_as_float(vc[k] & 0xffff0000u) : __uint_as_float(vc[k] << 16);
                        const float w3 = hf ? __uint_as_float(vd[k] & 0xffff0000u) : __uint_as_float(vd[k] << 16);
                        unsigned am = 0;
                        float m = w0;
                        if (w1 > m) { m = w1; am = 1; }
                        if (w2 > m) { m = w2; am = 2; }
                        if (w3 > m) { m = w3; am = 3; }
                        code |= am << (2 * (2 * k + hf));
                    }
                }
                p.codes[((long)n * p.out_img_stride + p.out_off + pa) >> 3] = (unsigned short)code;
            }
            if (p.full) {
                bf16_t *fb = p.full + (long)n * p.full_img_stride + p.full_off;
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int idx = pass * 256 + tid, cc = idx & 7, px = idx >> 3;
                    const uint4 v = *reinterpret_cast<const uint4 *>(otile + px * SF_OP + cc * 8);
                    *reinterpret_cast<uint4 *>(fb + (long)(ty * SF_TH + (px >> 4)) * p.full_row_stride + (tx * SF_TW + (px & 15)) * 64 + cc * 8) = v;
                }
            }
        } else {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int idx = pass * 256 + tid, c8 = idx & 7, px = idx >> 3;
                const int py = px >> 4, pxx = px & 15;
                const uint4 v = *reinterpret_cast<const uint4 *>(otile + px * SF_OP + c8 * 8);
                *reinterpret_cast<uint4 *>(ob + (long)(ty * SF_TH + py) * p.out_row_stride + (tx * SF_TW + pxx) * 64 + c8 * 8) = v;
            }
        }
    };

    const int G = gridDim.x;
    int tile = blockIdx.x, prev = -1;
    if (tile < p.ntiles) {
        if constexpr (F32IN) {
            f32_load(tile);
            f32_store(patchA);
        } else {
            stage(patchA, tile);
        }
    }
    while (tile < p.ntiles) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // patch A landed, otile B complete (tile prev), otile A free
        asm volatile("" ::: "memory");
        if (tile + G < p.ntiles) {
            if constexpr (F32IN) f32_load(tile + G);
            else stage(patchB, tile + G);
        }
        if (prev >= 0) store_phase(otileB, prev);
        mfma_phase(patchA, otileA);
        if constexpr (F32IN) {
            if (tile + G < p.ntiles) f32_store(patchB);      // (patch B was last read before this iteration's barrier)
        }
        prev = tile;
        tile += G;
        if (tile >= p.ntiles) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            store_phase(otileA, prev);
            break;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tile + G < p.ntiles) {
            if constexpr (F32IN) f32_load(tile + G);
            else stage(patchA, tile + G);
        }
        store_phase(otileA, prev);
        mfma_phase(patchB, otileB);
        if constexpr (F32IN) {
            if (tile + G < p.ntiles) f32_store(patchA);
        }
        prev = tile;
        tile += G;
        if (tile >= p.ntiles) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            store_phase(otileB, prev);
        }
    }
}

}  // namespace yolo

using namespace yolo;

YOLO_API int yolo_conv_stem7_fwd(const void *x_nhwc4, const void *w_packed, const float *bias, int N, int Ho, int Wo, long x_img_stride, int x_row_stride,
                                 float slope, int pool2, void *out, long out_img_stride, int out_row_stride, int out_off, void *out_full, long full_img_stride,
                                 int full_row_stride, int full_off, yolo_stream_t stream)
{
    if (pool2 == 3 && !out_full) return fail(YOLO_E_ARG, "yolo_conv_stem7_fwd: pool2 = 3 writes the arg-max codes through out_full");
    if (out_full && pool2 != 3 && (!pool2 || (full_row_stride & 7) || (full_img_stride & 7) || (full_off & 7) || ((uintptr_t)out_full & 15)))
        return fail(YOLO_E_ARG, "yolo_conv_stem7_fwd: out_full needs pool2 = 1 and strides in multiples of 8 elements");
    if (!x_nhwc4 || !w_packed || !bias || !out || N <= 0 || Ho <= 0 || Wo <= 0) return fail(YOLO_E_ARG, "yolo_conv_stem7_fwd: bad argument");
    if ((Ho % SF_TH) || (Wo % SF_TW)) return fail(YOLO_E_UNSUPPORTED, "yolo_conv_stem7_fwd: output %dx%d is not a multiple of %dx%d (use yolo_igemm)", Ho, Wo, SF_TH, SF_TW);
    if ((x_row_stride & 7) || (x_img_stride & 7) || (out_row_stride & 7) || (out_img_stride & 7) || (out_off & 7) || ((uintptr_t)x_nhwc4 & 15) || ((uintptr_t)out & 15) ||
        ((uintptr_t)w_packed & 15))
        return fail(YOLO_E_UNSUPPORTED, "yolo_conv_stem7_fwd: strides must be multiples of 8 elements and pointers 16-B aligned");
    StemFwdParams p{};
    p.x = (const bf16_t *)x_nhwc4; p.w = (const bf16_t *)w_packed; p.bias = bias; p.out = (bf16_t *)out;
    p.tiles_x = Wo / SF_TW; p.tiles_y = Ho / SF_TH;
    const long nt = (long)N * p.tiles_x * p.tiles_y;
    if (nt > 0x7fffffffL) return fail(YOLO_E_UNSUPPORTED, "yolo_conv_stem7_fwd: too many tiles");
    p.ntiles = (int)nt;
    p.x_img_stride = x_img_stride; p.out_img_stride = out_img_stride;
    p.x_row_stride = x_row_stride; p.out_row_stride = out_row_stride; p.out_off = out_off;
    p.pool = pool2 ? 1 : 0;
    p.slope = slope;
    if (pool2 == 3) {
        p.codes = (unsigned short *)out_full;
    } else {
        p.full = (bf16_t *)out_full; p.full_img_stride = full_img_stride; p.full_row_stride = full_row_stride; p.full_off = full_off;
    }
    // persistent workgroups, 2 resident per CU (200 VGPRs): 512 .. 2048 measure the same
    const long G = std::min<long>(nt, 1024);
    hipLaunchKernelGGL(stem_fwd_kernel<false>, dim3((unsigned)G), dim3(256), 0, STRM(stream), p);
    return check_launch("yolo_conv_stem7_fwd");
}

YOLO_API int yolo_conv_stem7_fwd_f32(const float *x_nchw, const void *w_packed, const float *bias, int N, int H, int W, float slope, int pool2, void *out,
                                     long out_img_stride, int out_row_stride, int out_off, void *out_full, long full_img_stride, int full_row_stride, int full_off,
                                     yolo_stream_t stream)
{
    const int Ho = H / 2, Wo = W / 2;
    if (!x_nchw || !w_packed || !bias || !out || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return fail(YOLO_E_ARG, "yolo_conv_stem7_fwd_f32: bad argument");
    if (pool2 == 3 && !out_full) return fail(YOLO_E_ARG, "yolo_conv_stem7_fwd_f32: pool2 = 3 writes the arg-max codes through out_full");
    if (out_full && pool2 != 3 && (!pool2 || (full_row_stride & 7) || (full_img_stride & 7) || (full_off & 7) || ((uintptr_t)out_full & 15)))
        return fail(YOLO_E_ARG, "yolo_conv_stem7_fwd_f32: out_full needs pool2 = 1 and strides in multiples of 8 elements");
    if ((Ho % SF_TH) || (Wo % SF_TW)) return fail(YOLO_E_UNSUPPORTED, "yolo_conv_stem7_fwd_f32: output %dx%d is not a multiple of %dx%d", Ho, Wo, SF_TH, SF_TW);
    if ((uintptr_t)x_nchw & 15) return fail(YOLO_E_UNSUPPORTED, "yolo_conv_stem7_fwd_f32: the input must be 16-B aligned (rows are read as float4)");
    if ((out_row_stride & 7) || (out_img_stride & 7) || (out_off & 7) || ((uintptr_t)out & 15) || ((uintptr_t)w_packed & 15))
        return fail(YOLO_E_UNSUPPORTED, "yolo_conv_stem7_fwd_f32: strides must be multiples of 8 elements and pointers 16-B aligned");
    StemFwdParams p{};
    p.x32 = x_nchw; p.H = H; p.W = W;
    p.w = (const bf16_t *)w_packed; p.bias = bias; p.out = (bf16_t *)out;
    p.tiles_x = Wo / SF_TW; p.tiles_y = Ho / SF_TH;
    const long nt = (long)N * p.tiles_x * p.tiles_y;
    if (nt > 0x7fffffffL) return fail(YOLO_E_UNSUPPORTED, "yolo_conv_stem7_fwd_f32: too many tiles");
    p.ntiles = (int)nt;
    p.out_img_stride = out_img_stride; p.out_row_stride = out_row_stride; p.out_off = out_off;
    p.pool = pool2 ? 1 : 0;
    p.slope = slope;
    if (pool2 == 3) {
        p.codes = (unsigned short *)out_full;
    } else {
        p.full = (bf16_t *)out_full; p.full_img_stride = full_img_stride; p.full_row_stride = full_row_stride; p.full_off = full_off;
    }
    const long G = std::min<long>(nt, 1024);
    hipLaunchKernelGGL(stem_fwd_kernel<true>, dim3((unsigned)G), dim3(256), 0, STRM(stream), p);
    return check_launch("yolo_conv_stem7_fwd_f32");
}
