// Image preprocessing on the device for gfx950: uint8 HWC RGB -> Resize (Pillow BILINEAR, bit-exact) -> ToTensor (/255)
// -> Normalize(mean, std) -> the stem's input format (zero-haloed NHWC4 bf16) and / or NCHW fp32.
// Replaces `YOLOInference.transform` / the eval transform of the reference (src/yolo/inference.py:58-66,
// src/yolo/dataset.py:224-233: torchvision Resize -> PIL.Image.resize(BILINEAR), ToTensor, Normalize), which run on the
// host per image and ship 2.4 MB of fp32 per image over PCIe; here the host ships the decoded uint8 image (3 B/pixel).
//
// Resize is Pillow's two-pass 8-bit resampling (libImaging/Resample.c): horizontal pass into a uint8 intermediate, then
// vertical pass; 22-bit fixed-point triangle coefficients whose support grows with the down-scaling factor; the
// coefficient tables are computed on the host (yolo/preprocess.py) exactly as Pillow does and passed in.  Integer
// arithmetic throughout -> bit-identical to Pillow; the normalisation is three correctly rounded fp32 operations, as
// torch performs them.  HBM/latency-bound byte work: one thread per output pixel (3 channels).
#include "common.h"

namespace yolo {

constexpr int PP_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int acc)
{
    int v = acc >> PP_BITS;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    return (unsigned char)v;
}

// src [N][Hs][Ws][3] -> tmp [N][Hs][Wo][3]
__global__ void __launch_bounds__(256) resize_h_u8_kernel(const unsigned char *__restrict__ src, int N, int Hs, int Ws, int Wo, const int *__restrict__ bounds,
                                                          const int *__restrict__ coef, int ksize, unsigned char *__restrict__ tmp)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * Hs * Wo;
    if (idx >= total) return;
    const int xx = (int)(idx % Wo);
    const long row = idx / Wo;                       // n*Hs + y
    const int x0 = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int *k = coef + (long)xx * ksize;
    const unsigned char *s = src + (row * Ws + x0) * 3;
    int a0 = 1 << (PP_BITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < cnt; ++x) {
        const int w = k[x];
        a0 += (int)s[3 * x] * w;
        a1 += (int)s[3 * x + 1] * w;
        a2 += (int)s[3 * x + 2] * w;
    }
    unsigned char *o = tmp + idx * 3;
    o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
}

// in [N][Hin][Wo][3] uint8 -(optional vertical pass)-> [N][Ho][Wo][3] -> normalise -> NHWC4 bf16 (halo) and / or NCHW fp32
__global__ void __launch_bounds__(256) resize_v_norm_kernel(const unsigned char *__restrict__ in, int N, int Hin, int Ho, int Wo, const int *__restrict__ bounds,
                                                            const int *__restrict__ coef, int ksize, float m0, float m1, float m2, float s0, float s1, float s2,
                                                            bf16_t *__restrict__ out4, int halo, float *__restrict__ out_nchw)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * Ho * Wo;
    if (idx >= total) return;
    const int xx = (int)(idx % Wo);
    const int yy = (int)((idx / Wo) % Ho);
    const int n = (int)(idx / ((long)Wo * Ho));
    int v0, v1, v2;
    if (bounds) {
        const int y0 = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const int *k = coef + (long)yy * ksize;
        const unsigned char *s = in + (((long)n * Hin + y0) * Wo + xx) * 3;
        int a0 = 1 << (PP_BITS - 1), a1 = a0, a2 = a0;
        for (int y = 0; y < cnt; ++y) {
            const int w = k[y];
            a0 += (int)s[0] * w; a1 += (int)s[1] * w; a2 += (int)s[2] * w;
            s += (long)Wo * 3;
        }
        v0 = clip8(a0); v1 = clip8(a1); v2 = clip8(a2);
    } else {
        const unsigned char *s = in + (((long)n * Hin + yy) * Wo + xx) * 3;
        v0 = s[0]; v1 = s[1]; v2 = s[2];
    }
    // ToTensor: uint8 -> fp32 / 255 ; Normalize: (x - mean) / std  (each operation correctly rounded, no contraction)
    const float f0 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v0, 255.0f), m0), s0);
    const float f1 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v1, 255.0f), m1), s1);
    const float f2 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v2, 255.0f), m2), s2);
    if (out4) {
        const int Hp = Ho + 2 * halo, Wp = Wo + 2 * halo;
        uint2 o;
        o.x = (unsigned)f32_to_bf16(f0) | ((unsigned)f32_to_bf16(f1) << 16);
        o.y = (unsigned)f32_to_bf16(f2);
        *reinterpret_cast<uint2 *>(out4 + (((long)n * Hp + yy + halo) * Wp + xx + halo) * 4) = o;
    }
    if (out_nchw) {
        const long plane = (long)Ho * Wo;
        float *o = out_nchw + (long)n * 3 * plane + (long)yy * Wo + xx;
        o[0] = f0; o[plane] = f1; o[2 * plane] = f2;
    }
}

}  // namespace yolo

using namespace yolo;

YOLO_API int yolo_preprocess_u8(const unsigned char *src, int N, int Hs, int Ws, int Ho, int Wo, const int *hbounds, const int *hcoef, int hk, const int *vbounds,
                                const int *vcoef, int vk, unsigned char *tmp, const float *mean3, const float *std3, void *out_nhwc4, int halo, float *out_nchw,
                                yolo_stream_t stream)
{
    if (!src || !mean3 || !std3 || (!out_nhwc4 && !out_nchw) || N <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || halo < 0)
        return fail(YOLO_E_ARG, "yolo_preprocess_u8: bad argument");
    const bool need_h = Ws != Wo, need_v = Hs != Ho;
    if (need_h && (!hbounds || !hcoef || hk <= 0 || !tmp)) return fail(YOLO_E_ARG, "yolo_preprocess_u8: width %d -> %d needs the horizontal tables and tmp", Ws, Wo);
    if (need_v && (!vbounds || !vcoef || vk <= 0)) return fail(YOLO_E_ARG, "yolo_preprocess_u8: height %d -> %d needs the vertical tables", Hs, Ho);
    if (std3[0] == 0.0f || std3[1] == 0.0f || std3[2] == 0.0f) return fail(YOLO_E_ARG, "yolo_preprocess_u8: zero std");
    hipStream_t s = STRM(stream);
    const unsigned char *stage1 = src;
    if (need_h) {
        const long total = (long)N * Hs * Wo;
        hipLaunchKernelGGL(resize_h_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, N, Hs, Ws, Wo, hbounds, hcoef, hk, tmp);
        if (int rc = check_launch("yolo_preprocess_u8(horizontal)")) return rc;
        stage1 = tmp;
    }
    const long total = (long)N * Ho * Wo;
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, stage1, N, Hs, Ho, Wo, need_v ? vbounds : nullptr, vcoef, vk, mean3[0],
                       mean3[1], mean3[2], std3[0], std3[1], std3[2], (bf16_t *)out_nhwc4, halo, out_nchw);
    return check_launch("yolo_preprocess_u8");
}
