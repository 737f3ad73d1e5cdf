// Streaming 1x1 convolution for gfx950 (yolo_igemm_desc.tile_hint = 19): thin-K pointwise layers (K = Cin <= 256: the 64 / 128 / 256-
// channel 1x1 convs of a ResNet bottleneck, with or without the residual add) are HBM-bound -- per output pixel they move
// 2 K + 2 Cout (+ 2 Cout residual) bytes for 2 K Cout FLOPs -- and the tiled kernels of igemm.hip reach 55-70 % of the HBM rate on
// them: every tile is a chain of dependent round trips (address table, operand fetch, residual fetch, store) that only co-resident
// workgroups overlap.  Here nothing is tiled through LDS and no barrier follows the prologue:
//   * the weight panel of the workgroup (<= 256 co x K, <= 64 KB) is copied to LDS once, in MFMA-fragment order;
//   * every wave then walks groups of 16 pixels on its own: the activation fragments of the NEXT group (16 B per lane and 32 channels,
//     straight from global memory: a lane's 8 consecutive k ARE contiguous in NHWC) and the residual vectors of the current one are in
//     flight while the current group's MFMAs (v_mfma_f32_16x16x32_bf16) and stores run;
//   * the accumulators go through a wave-private 4-KB LDS patch, 64 channels at a time, to become 16-B stores / residual loads.
// Same arithmetic as the tiled kernels: fp32 accumulation over k in ascending 32-blocks, epilogue on the fp32 value, one bf16 rounding.
#include "igemm_common.h"

namespace yolo {

namespace is {
constexpr int NW = 4, NTHR = NW * 64;
constexpr int EP_BYTES = 16 * 64 * 4;           // wave-private epilogue patch: 16 px x 64 co fp32
}  // namespace is

// TCO: channels per workgroup (64, 128 or 256); KS = K / 32 (2, 4, 6, 8 or 16); STATS: BatchNorm's per-channel sum / sum of squares of the
// stored (bf16-rounded) outputs as well (yolo_igemm_desc.bn_stats, epilogue NONE): in the epilogue patch lane l reads channel l of
// the 16 pixels and keeps the two sums in registers for the wave's whole pixel stream; one fp64 atomic pair per channel and wave
template <int TCO, int KS, bool STATS>
__global__ void __launch_bounds__(is::NTHR, 2) igemm_stream_kernel(const IgemmParams p)
{
    using namespace is;
    constexpr int K = KS * 32, NCT = TCO / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_co_tiles = p.n_co_tiles;
    // workgroup -> (co tile, pixel stream): the co tiles of one pixel stream get ids that are equal mod 8, i.e. the same XCD -- they read
    // the same activations at about the same time, through one L2 (the host rounds the stream count up to a multiple of 8)
    const int nwg = gridDim.x / n_co_tiles;
    const int co_tile = (blockIdx.x >> 3) % n_co_tiles, wg = (blockIdx.x & 7) + 8 * (int)(blockIdx.x / (8 * n_co_tiles));
    const int co0 = co_tile * TCO;

    // ---- weight panel -> LDS in fragment order: 16-B chunk (s, kg, co) = W[co0 + co][32 s + 8 kg ..] at ((s * 4 + kg) * TCO + co) * 16,
    // so the 16 lanes of a fragment row group read 256 contiguous bytes (no bank conflicts)
    char *wl = smem;
    for (int c = tid; c < TCO * KS * 4; c += NTHR) {
        const int co = c % TCO, skg = c / TCO;
        *reinterpret_cast<uint4 *>(wl + ((long)skg * TCO + co) * 16) = *reinterpret_cast<const uint4 *>(p.w + (long)(co0 + co) * K + skg * 8);
    }
    float *ep = reinterpret_cast<float *>(smem + TCO * K * 2 + wave * EP_BYTES);
    __syncthreads();

    const int px = lane & 15, kg = lane >> 4;
    // epilogue role of a lane: pixel lane >> 2 of the group, 16-channel segment lane & 3 of the current 64-channel chunk
    const int epx = lane >> 2, eq = lane & 3;
    const long ngroups = p.M >> 4;
    const bool has_bias = p.epilogue == YOLO_EPI_BIAS || p.epilogue == YOLO_EPI_BIAS_LRELU || p.epilogue == YOLO_EPI_BIAS_ADD_LRELU;
    const bool has_aux = p.epilogue == YOLO_EPI_BIAS_ADD_LRELU || p.epilogue == YOLO_EPI_MUL_DLRELU;

    auto pixel = [&](long m, long &in_o, long &out_o, long &aux_o) {
        const unsigned mu = (unsigned)m;
        const int n = (int)(mu / (unsigned)p.HoWo);
        const int rem = (int)(mu - (unsigned)n * (unsigned)p.HoWo);
        const int oy = (int)((unsigned)rem / (unsigned)p.Wo), ox = rem - oy * p.Wo;
        in_o = (long)n * p.in_img_stride + (long)(oy * p.stride) * p.in_row_stride + (long)(ox * p.stride) * p.in_px_stride + p.in_off;
        out_o = (long)n * p.out_img_stride + (long)oy * p.out_row_stride + (long)ox * p.out_px_stride + p.out_off;
        aux_o = (long)n * p.aux_img_stride + (long)oy * p.aux_row_stride + (long)ox * p.aux_px_stride + p.aux_off;
    };
    auto load_b = [&](long g, bf16x8 (&b)[KS]) {
        long io, oo, ao;
        pixel(g * 16 + px, io, oo, ao);
#pragma unroll
        for (int s = 0; s < KS; ++s) b[s] = *reinterpret_cast<const bf16x8 *>(p.in + io + s * 32 + kg * 8);
    };

    long g = (long)wg * NW + wave;
    const long gstep = (long)nwg * NW;
    // K <= 256: the next group's fragments are prefetched into a second register set; deeper K (512): one set of 64 registers, the
    // 16 loads of a group and the other waves of the CU cover the latency
    constexpr bool PREFETCH = KS <= 8;
    bf16x8 bcur[KS], bnxt[PREFETCH ? KS : 1];
    float ssum[TCO / 64], ssq[TCO / 64];
#pragma unroll
    for (int c = 0; c < TCO / 64; ++c) ssum[c] = ssq[c] = 0.0f;
    if (PREFETCH && g < ngroups) load_b(g, bcur);
    for (; g < ngroups; g += gstep) {
        if constexpr (PREFETCH) {
            if (g + gstep < ngroups) load_b(g + gstep, bnxt);
        } else {
            load_b(g, bcur);
        }
        // output / residual addresses of this lane's epilogue pixel; the residual vectors of all chunks are fetched now
        long io, oo, ao;
        pixel(g * 16 + epx, io, oo, ao);
        uint4 ax[TCO / 64][2];
        if (has_aux) {
#pragma unroll
            for (int c = 0; c < TCO / 64; ++c) {
                const bf16_t *a = p.aux + ao + co0 + c * 64 + eq * 16;
                ax[c][0] = *reinterpret_cast<const uint4 *>(a);
                ax[c][1] = *reinterpret_cast<const uint4 *>(a + 8);
            }
        }
        f32x4 acc[NCT];
#pragma unroll
        for (int i = 0; i < NCT; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(wl + ((long)(s * 4 + kg) * TCO + i * 16 + px) * 16);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bcur[s], acc[i], 0, 0, 0);
            }
        }
        // ---- epilogue, 64 channels at a time through the wave's LDS patch: D row (co) = 4 * (lane >> 4) + r, col (pixel) = lane & 15
#pragma unroll
        for (int c = 0; c < TCO / 64; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(ep + px * 64 + i * 16 + 4 * kg) = acc[c * 4 + i];
            // (same wave writes and reads: LDS operations of a wave complete in order)
            if constexpr (STATS) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float r = __uint_as_float((unsigned)f32_to_bf16(ep[q * 64 + lane]) << 16);     // the value as stored
                    ssum[c] += r;
                    ssq[c] += r * r;
                }
            }
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(ep + epx * 64 + eq * 16 + q * 4);
                v[q * 4] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
            }
            const int co = co0 + c * 64 + eq * 16;
            if (has_bias) {
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] += p.bias[co + k];
            }
            if (has_aux) {
                const unsigned yy[8] = {ax[c][0].x, ax[c][0].y, ax[c][0].z, ax[c][0].w, ax[c][1].x, ax[c][1].y, ax[c][1].z, ax[c][1].w};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float a = __uint_as_float((k & 1) ? (yy[k >> 1] & 0xffff0000u) : (yy[k >> 1] << 16));
                    if (p.epilogue == YOLO_EPI_BIAS_ADD_LRELU) {
                        v[k] += a;
                        v[k] = v[k] > 0.0f ? v[k] : v[k] * p.slope;
                    } else {
                        v[k] = a > 0.0f ? v[k] : v[k] * p.slope;
                    }
                }
            } else if (p.epilogue == YOLO_EPI_BIAS_LRELU) {
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = v[k] > 0.0f ? v[k] : v[k] * p.slope;
            }
            uint4 o0, o1;
            o0.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            o0.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            o0.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
            o0.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
            o1.x = (unsigned)f32_to_bf16(v[8]) | ((unsigned)f32_to_bf16(v[9]) << 16);
            o1.y = (unsigned)f32_to_bf16(v[10]) | ((unsigned)f32_to_bf16(v[11]) << 16);
            o1.z = (unsigned)f32_to_bf16(v[12]) | ((unsigned)f32_to_bf16(v[13]) << 16);
            o1.w = (unsigned)f32_to_bf16(v[14]) | ((unsigned)f32_to_bf16(v[15]) << 16);
            bf16_t *o = reinterpret_cast<bf16_t *>(p.out) + oo + co;
            *reinterpret_cast<uint4 *>(o) = o0;
            *reinterpret_cast<uint4 *>(o + 8) = o1;
        }
        if constexpr (PREFETCH) {
#pragma unroll
            for (int s = 0; s < KS; ++s) bcur[s] = bnxt[s];
        }
    }
    if constexpr (STATS) {
        double *rep = p.stats + (size_t)(blockIdx.x % YOLO_BN_ACC_REPLICAS) * 2 * p.Cout;
#pragma unroll
        for (int c = 0; c < TCO / 64; ++c) {
            atomicAdd(rep + co0 + c * 64 + lane, (double)ssum[c]);
            atomicAdd(rep + p.Cout + co0 + c * 64 + lane, (double)ssq[c]);
        }
    }
}

template <int TCO, int KS, bool STATS>
static int stream_launch_impl(const IgemmParams &p, hipStream_t s)
{
    constexpr int LDS = TCO * KS * 32 * 2 + is::NW * is::EP_BYTES;
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm_stream_kernel<TCO, KS, STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return fail((int)e, "yolo_igemm: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
        attr_done[dev] = true;
    }
    IgemmParams q = p;
    q.n_co_tiles = p.Cout / TCO;
    // persistent workgroups: as many as fit (LDS-bound), each wave walks its own pixel groups
    const int per_cu = std::max(1, std::min(4, (160 * 1024) / LDS));
    long nwg = (long)256 * per_cu / q.n_co_tiles;
    const long need = ((p.M >> 4) + is::NW - 1) / is::NW;
    if (nwg > need) nwg = need;
    nwg = (std::max<long>(nwg, 1) + 7) / 8 * 8;
    hipLaunchKernelGGL((igemm_stream_kernel<TCO, KS, STATS>), dim3((unsigned)(nwg * q.n_co_tiles)), dim3(is::NTHR), LDS, s, q);
    return check_launch("yolo_igemm (streaming 1x1)");
}

template <int TCO, int KS>
static int stream_launch(const IgemmParams &p, hipStream_t s)
{
    return p.stats ? stream_launch_impl<TCO, KS, true>(p, s) : stream_launch_impl<TCO, KS, false>(p, s);
}

int igemm_stream_launch(const IgemmParams &p, int splits, hipStream_t s)
{
    const int K = p.tap_len;
    if (p.KH != 1 || p.KW != 1 || p.stride < 1 || (K != 64 && K != 128 && K != 192 && K != 256 && K != 512) || (p.Cout % 64) || p.out_fp32 || p.pool || p.w_blocked || splits > 1 ||
        p.slab_stride || p.px_begin || (p.M & 15) || p.M >= (1L << 31) || (p.stats && p.epilogue != YOLO_EPI_NONE))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint 19 (streaming 1x1) takes 1x1 convs with 64 / 128 / 192 / 256 / 512 input channels, Cout %% 64 == 0, "
                                        "bf16 output, M %% 16 == 0, no pool / split-K / pixel range; bn_stats with epilogue NONE only");
    if ((p.in_px_stride & 7) || (p.in_row_stride & 7) || (p.in_img_stride & 7) || (p.in_off & 7) || (p.out_px_stride & 7) || (p.out_row_stride & 7) ||
        (p.out_img_stride & 7) || (p.out_off & 7) || (p.aux_px_stride & 7) || (p.aux_row_stride & 7) || (p.aux_img_stride & 7) || (p.aux_off & 7))
        return fail(YOLO_E_UNSUPPORTED, "yolo_igemm: tile_hint 19 needs strides and offsets in multiples of 8 elements");
    // channels per workgroup: the panel must fit 64 KB of LDS (K = 192 / 256: 128 channels, K = 512: 64)
    const int tco = (p.Cout % 256 == 0 && K <= 128) ? 256 : ((p.Cout % 128 == 0 && K <= 256) ? 128 : 64);
    switch (K) {
    case 64: return tco == 256 ? stream_launch<256, 2>(p, s) : (tco == 128 ? stream_launch<128, 2>(p, s) : stream_launch<64, 2>(p, s));
    case 128: return tco == 256 ? stream_launch<256, 4>(p, s) : (tco == 128 ? stream_launch<128, 4>(p, s) : stream_launch<64, 4>(p, s));
    case 192: return tco == 128 ? stream_launch<128, 6>(p, s) : stream_launch<64, 6>(p, s);
    case 256: return tco == 128 ? stream_launch<128, 8>(p, s) : stream_launch<64, 8>(p, s);
    default: return stream_launch<64, 16>(p, s);
    }
}

}  // namespace yolo
