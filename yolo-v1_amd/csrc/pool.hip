// MaxPool2d(2,2) forward and backward on zero-haloed NHWC bf16 (HBM-bound: 16 B per lane).
// Replaces aten max_pool2d_with_indices{,_backward} for src/yolo/models.py:51,55,65,72; the
// backward recomputes the arg-max from the stored (post-LeakyReLU) activation instead of keeping
// an index tensor, and applies the LeakyReLU derivative of the layer in front of the pool in the
// same pass (sign(lrelu(z)) == sign(z)).
#include "common.h"

namespace yolo {

__device__ __forceinline__ void unpack8(const uint4 &v, float f[8])
{
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float f[8])
{
    uint4 o;
    o.x = (unsigned)f32_to_bf16(f[0]) | ((unsigned)f32_to_bf16(f[1]) << 16);
    o.y = (unsigned)f32_to_bf16(f[2]) | ((unsigned)f32_to_bf16(f[3]) << 16);
    o.z = (unsigned)f32_to_bf16(f[4]) | ((unsigned)f32_to_bf16(f[5]) << 16);
    o.w = (unsigned)f32_to_bf16(f[6]) | ((unsigned)f32_to_bf16(f[7]) << 16);
    return o;
}

// one thread = one pooled pixel x 8 channels
__global__ void __launch_bounds__(256) maxpool2_fwd_kernel(const bf16_t *__restrict__ x, int N, int H, int W, int C, int hi, int ho,
                                                           bf16_t *__restrict__ y)
{
    const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    const int ox = (int)((idx / C8) % Wo);
    const int oy = (int)((idx / ((long)C8 * Wo)) % Ho);
    const int n = (int)(idx / ((long)C8 * Wo * Ho));
    const int Wp = W + 2 * hi, Hp = H + 2 * hi;
    const bf16_t *p = x + (((long)n * Hp + 2 * oy + hi) * Wp + 2 * ox + hi) * C + c8 * 8;
    float a[8], b[8], c[8], d[8], m[8];
    unpack8(*reinterpret_cast<const uint4 *>(p), a);
    unpack8(*reinterpret_cast<const uint4 *>(p + C), b);
    unpack8(*reinterpret_cast<const uint4 *>(p + (long)Wp * C), c);
    unpack8(*reinterpret_cast<const uint4 *>(p + (long)Wp * C + C), d);
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(c[k], d[k]));
    const int Wop = Wo + 2 * ho, Hop = Ho + 2 * ho;
    *reinterpret_cast<uint4 *>(y + (((long)n * Hop + oy + ho) * Wop + ox + ho) * C + c8 * 8) = pack8(m);
}

// MaxPool2d(3, stride 2, pad 1) on a non-negative input (zero halo == -inf padding): one thread = one
// output pixel x 8 channels, 9 x 16-B loads
__global__ void __launch_bounds__(256) maxpool3s2_fwd_kernel(const bf16_t *__restrict__ x, int N, int H, int W, int C, int hi, int ho,
                                                             bf16_t *__restrict__ y)
{
    const int C8 = C >> 3, Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * Ho * Wo * C8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    const int ox = (int)((idx / C8) % Wo);
    const int oy = (int)((idx / ((long)C8 * Wo)) % Ho);
    const int n = (int)(idx / ((long)C8 * Wo * Ho));
    const int Wp = W + 2 * hi, Hp = H + 2 * hi;
    // window rows 2*oy-1 .. 2*oy+1 in logical coordinates -> +hi in the buffer (hi >= 1)
    const bf16_t *p = x + (((long)n * Hp + 2 * oy - 1 + hi) * Wp + 2 * ox - 1 + hi) * C + c8 * 8;
    float m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4 *>(p + ((long)dy * Wp + dx) * C), v);
#pragma unroll
            for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], v[k]);
        }
    const int Wop = Wo + 2 * ho, Hop = Ho + 2 * ho;
    *reinterpret_cast<uint4 *>(y + (((long)n * Hop + oy + ho) * Wop + ox + ho) * C + c8 * 8) = pack8(m);
}

// one thread = one pooled pixel x 8 channels; writes the four un-pooled gradient pixels
__global__ void __launch_bounds__(256) maxpool2_bwd_kernel(const bf16_t *__restrict__ yfull, const bf16_t *__restrict__ dpool, int N, int H,
                                                           int W, int C, int hi, int ho, float slope, bf16_t *__restrict__ dz)
{
    const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    const int ox = (int)((idx / C8) % Wo);
    const int oy = (int)((idx / ((long)C8 * Wo)) % Ho);
    const int n = (int)(idx / ((long)C8 * Wo * Ho));
    const int Wp = W + 2 * hi, Hp = H + 2 * hi;
    const int Wop = Wo + 2 * ho, Hop = Ho + 2 * ho;
    const long off = (((long)n * Hp + 2 * oy + hi) * Wp + 2 * ox + hi) * C + c8 * 8;
    const bf16_t *p = yfull + off;
    float v[4][8], g[8], o[4][8];
    unpack8(*reinterpret_cast<const uint4 *>(p), v[0]);
    unpack8(*reinterpret_cast<const uint4 *>(p + C), v[1]);
    unpack8(*reinterpret_cast<const uint4 *>(p + (long)Wp * C), v[2]);
    unpack8(*reinterpret_cast<const uint4 *>(p + (long)Wp * C + C), v[3]);
    unpack8(*reinterpret_cast<const uint4 *>(dpool + (((long)n * Hop + oy + ho) * Wop + ox + ho) * C + c8 * 8), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int am = 0;
        float m = v[0][k];
        if (v[1][k] > m) { m = v[1][k]; am = 1; }  // first maximum in (0,0),(0,1),(1,0),(1,1) order
        if (v[2][k] > m) { m = v[2][k]; am = 2; }
        if (v[3][k] > m) { m = v[3][k]; am = 3; }
        const float gv = g[k] * (m > 0.0f ? 1.0f : slope);
        o[0][k] = am == 0 ? gv : 0.0f;
        o[1][k] = am == 1 ? gv : 0.0f;
        o[2][k] = am == 2 ? gv : 0.0f;
        o[3][k] = am == 3 ? gv : 0.0f;
    }
    bf16_t *q = dz + off;
    *reinterpret_cast<uint4 *>(q) = pack8(o[0]);
    *reinterpret_cast<uint4 *>(q + C) = pack8(o[1]);
    *reinterpret_cast<uint4 *>(q + (long)Wp * C) = pack8(o[2]);
    *reinterpret_cast<uint4 *>(q + (long)Wp * C + C) = pack8(o[3]);
}

// the same from the POOLED activation and the arg-max codes a fused conv + pool epilogue left (yolo_igemm pool2 = 3,
// yolo_conv_stem7_fwd): one uint16 per (pooled pixel, 8 channels), 2 bits per channel = window position of the first maximum
__global__ void __launch_bounds__(256) maxpool2_bwd_codes_kernel(const bf16_t *__restrict__ ypool, const unsigned short *__restrict__ codes,
                                                                 const bf16_t *__restrict__ dpool, int N, int H, int W, int C, int hi, int ho, float slope,
                                                                 bf16_t *__restrict__ dz)
{
    const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    const int ox = (int)((idx / C8) % Wo);
    const int oy = (int)((idx / ((long)C8 * Wo)) % Ho);
    const int n = (int)(idx / ((long)C8 * Wo * Ho));
    const int Wp = W + 2 * hi, Hp = H + 2 * hi;
    const int Wop = Wo + 2 * ho, Hop = Ho + 2 * ho;
    const long off = (((long)n * Hp + 2 * oy + hi) * Wp + 2 * ox + hi) * C + c8 * 8;
    const long poff = (((long)n * Hop + oy + ho) * Wop + ox + ho) * C + c8 * 8;
    float y[8], g[8], o[4][8];
    unpack8(*reinterpret_cast<const uint4 *>(ypool + poff), y);
    unpack8(*reinterpret_cast<const uint4 *>(dpool + poff), g);
    const unsigned code = codes[poff >> 3];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int am = (code >> (2 * k)) & 3;
        const float gv = g[k] * (y[k] > 0.0f ? 1.0f : slope);
        o[0][k] = am == 0 ? gv : 0.0f;
        o[1][k] = am == 1 ? gv : 0.0f;
        o[2][k] = am == 2 ? gv : 0.0f;
        o[3][k] = am == 3 ? gv : 0.0f;
    }
    bf16_t *q = dz + off;
    *reinterpret_cast<uint4 *>(q) = pack8(o[0]);
    *reinterpret_cast<uint4 *>(q + C) = pack8(o[1]);
    *reinterpret_cast<uint4 *>(q + (long)Wp * C) = pack8(o[2]);
    *reinterpret_cast<uint4 *>(q + (long)Wp * C + C) = pack8(o[3]);
}

}  // namespace yolo

using namespace yolo;

static int pool_args(const yolo_pool_desc *d, const char *who)
{
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->in_halo < 0 || d->out_halo < 0) return fail(YOLO_E_ARG, "%s: bad descriptor", who);
    if ((d->C & 7) || (d->H & 1) || (d->W & 1)) return fail(YOLO_E_UNSUPPORTED, "%s: C=%d must be a multiple of 8 and H,W=%d,%d even", who, d->C, d->H, d->W);
    return 0;
}

YOLO_API int yolo_maxpool2_fwd(const yolo_pool_desc *d, const void *x, void *y, yolo_stream_t stream)
{
    if (int rc = pool_args(d, "yolo_maxpool2_fwd")) return rc;
    if (!x || !y) return fail(YOLO_E_ARG, "yolo_maxpool2_fwd: null pointer");
    const long total = (long)d->N * (d->H / 2) * (d->W / 2) * (d->C / 8);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, STRM(stream), (const bf16_t *)x, d->N, d->H, d->W, d->C,
                       d->in_halo, d->out_halo, (bf16_t *)y);
    return check_launch("yolo_maxpool2_fwd");
}

YOLO_API int yolo_maxpool2_bwd_lrelu(const yolo_pool_desc *d, const void *yfull, const void *dpool, float slope, void *dz, yolo_stream_t stream)
{
    if (int rc = pool_args(d, "yolo_maxpool2_bwd_lrelu")) return rc;
    if (!yfull || !dpool || !dz) return fail(YOLO_E_ARG, "yolo_maxpool2_bwd_lrelu: null pointer");
    const long total = (long)d->N * (d->H / 2) * (d->W / 2) * (d->C / 8);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, STRM(stream), (const bf16_t *)yfull, (const bf16_t *)dpool,
                       d->N, d->H, d->W, d->C, d->in_halo, d->out_halo, slope, (bf16_t *)dz);
    return check_launch("yolo_maxpool2_bwd_lrelu");
}

YOLO_API int yolo_maxpool2_bwd_codes(const yolo_pool_desc *d, const void *ypool, const void *codes, const void *dpool, float slope, void *dz,
                                     yolo_stream_t stream)
{
    if (int rc = pool_args(d, "yolo_maxpool2_bwd_codes")) return rc;
    if (!ypool || !codes || !dpool || !dz) return fail(YOLO_E_ARG, "yolo_maxpool2_bwd_codes: null pointer");
    const long total = (long)d->N * (d->H / 2) * (d->W / 2) * (d->C / 8);
    hipLaunchKernelGGL(maxpool2_bwd_codes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, STRM(stream), (const bf16_t *)ypool,
                       (const unsigned short *)codes, (const bf16_t *)dpool, d->N, d->H, d->W, d->C, d->in_halo, d->out_halo, slope, (bf16_t *)dz);
    return check_launch("yolo_maxpool2_bwd_codes");
}

namespace yolo {

// MaxPool2d(3, stride 2, pad 1) backward, gather form (deterministic, no atomics): one thread = one 2x2 block of INPUT
// pixels (rows 2i, 2i+1; columns 2j, 2j+1) x 8 channels.  Exactly four windows touch the block -- (i, j), (i, j+1),
// (i+1, j), (i+1, j+1) -- and for each the thread re-reads the window, finds its arg-max (the FIRST maximum in row-major
// order over the in-range positions, as aten's max_pool2d_with_indices keeps it) and, when that lies inside the block,
// adds the window's gradient to that pixel: 9 loads per input pixel instead of ~20 for a thread per pixel.
__global__ void __launch_bounds__(256) maxpool3s2_bwd_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ dy, int N, int H, int W, int C, int hi,
                                                             int ho, int hd, bf16_t *__restrict__ dx)
{
    const int C8 = C >> 3, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    const long total = (long)N * Hb * Wb * C8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    const int j = (int)((idx / C8) % Wb);
    const int i = (int)((idx / ((long)C8 * Wb)) % Hb);
    const int n = (int)(idx / ((long)C8 * Wb * Hb));
    const int Wp = W + 2 * hi, Hp = H + 2 * hi, Wop = Wo + 2 * ho, Hop = Ho + 2 * ho, Wdp = W + 2 * hd, Hdp = H + 2 * hd;
    const bf16_t *xb = x + ((long)n * Hp * Wp) * C + c8 * 8;
    float acc[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[a][b][k] = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = i + a;
        if (oy >= Ho) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ox = j + b;
            if (ox >= Wo) continue;
            // arg-max of window (oy, ox): position code ky*3+kx of the first maximum, per channel
            float best[8];
            int arg[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { best[k] = -3.4e38f; arg[k] = -1; }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = 2 * oy - 1 + ky;
                if (yy < 0 || yy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = 2 * ox - 1 + kx;
                    if (xx < 0 || xx >= W) continue;
                    float v[8];
                    unpack8(*reinterpret_cast<const uint4 *>(xb + ((long)(yy + hi) * Wp + xx + hi) * C), v);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (v[k] > best[k]) { best[k] = v[k]; arg[k] = ky * 3 + kx; }
                }
            }
            float g[8];
            unpack8(*reinterpret_cast<const uint4 *>(dy + (((long)n * Hop + oy + ho) * Wop + ox + ho) * C + c8 * 8), g);
            // block pixel (r, c) = input (2i + r, 2j + c) sits at window position (ky, kx) = (2i + r - 2oy + 1, 2j + c - 2ox + 1)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int ky = r - 2 * a + 1, kx = c - 2 * b + 1;
                    if (ky < 0 || ky > 2 || kx < 0 || kx > 2) continue;
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[r][c][k] += arg[k] == ky * 3 + kx ? g[k] : 0.0f;
                }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int h = 2 * i + r, w = 2 * j + c;
            if (h < H && w < W) *reinterpret_cast<uint4 *>(dx + (((long)n * Hdp + h + hd) * Wdp + w + hd) * C + c8 * 8) = pack8(acc[r][c]);
        }
}

}  // namespace yolo

YOLO_API int yolo_maxpool3s2_bwd(const yolo_pool_desc *d, const void *x, const void *dy, void *dx, int dx_halo, yolo_stream_t stream)
{
    if (!d || !x || !dy || !dx || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->in_halo < 0 || d->out_halo < 0 || dx_halo < 0)
        return fail(YOLO_E_ARG, "yolo_maxpool3s2_bwd: bad argument");
    if (d->C & 7) return fail(YOLO_E_UNSUPPORTED, "yolo_maxpool3s2_bwd: C must be a multiple of 8");
    const long total = (long)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2) * (d->C / 8);
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, STRM(stream), (const bf16_t *)x, (const bf16_t *)dy, d->N, d->H,
                       d->W, d->C, d->in_halo, d->out_halo, dx_halo, (bf16_t *)dx);
    return check_launch("yolo_maxpool3s2_bwd");
}

YOLO_API int yolo_maxpool3s2_fwd(const yolo_pool_desc *d, const void *x, void *y, yolo_stream_t stream)
{
    if (!d || !x || !y || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->in_halo < 1 || d->out_halo < 0) return fail(YOLO_E_ARG, "yolo_maxpool3s2_fwd: bad argument (input halo >= 1)");
    if (d->C & 7) return fail(YOLO_E_UNSUPPORTED, "yolo_maxpool3s2_fwd: C must be a multiple of 8");
    const int Ho = (d->H - 1) / 2 + 1, Wo = (d->W - 1) / 2 + 1;
    const long total = (long)d->N * Ho * Wo * (d->C / 8);
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, STRM(stream), (const bf16_t *)x, d->N, d->H, d->W, d->C,
                       d->in_halo, d->out_halo, (bf16_t *)y);
    return check_launch("yolo_maxpool3s2_fwd");
}
