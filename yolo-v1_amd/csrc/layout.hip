// Layout / precision conversion kernels at the boundary between PyTorch-layout fp32 tensors
// (NCHW activations, OIHW / [O][K] weights: the state_dict contract of SURVEY.md 8b) and the
// network's internal format (zero-haloed NHWC bf16 activations, K-contiguous bf16 weight panels).
// All HBM-bound; every kernel moves 8-16 B per lane on its contiguous side.
#include "common.h"

namespace yolo {

// ---- NCHW fp32 -> haloed NHWC bf16 -------------------------------------------------------------
// small C (the 3-channel image): one thread per pixel, reads are coalesced along W per plane,
// one 8-byte store per pixel (Cpad == 4).
__global__ void nchw_to_nhwc4_kernel(const float *__restrict__ x, int N, int C, int H, int W, bf16_t *__restrict__ y, int lo, int hi)
{
    const long total = (long)N * H * W;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % W);
    const int h = (int)((idx / W) % H);
    const int n = (int)(idx / ((long)W * H));
    const int Wp = W + lo + hi, Hp = H + lo + hi;
    unsigned short v[4] = {0, 0, 0, 0};
    for (int c = 0; c < C; ++c) v[c] = f32_to_bf16(x[(((long)n * C + c) * H + h) * W + w]);
    uint2 o;
    o.x = (unsigned)v[0] | ((unsigned)v[1] << 16);
    o.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
    *reinterpret_cast<uint2 *>(y + (((long)n * Hp + h + lo) * Wp + w + lo) * 4) = o;
}

// general C: 32(c) x 32(w) tile through LDS.  grid = (ceil(W/32), ceil(C/32), N*H)
__global__ void __launch_bounds__(256) nchw_to_nhwc_tile_kernel(const float *__restrict__ x, int N, int C, int H, int W,
                                                                bf16_t *__restrict__ y, int Cpad, int lo, int hi)
{
    __shared__ float t[32][33];
    const int w0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int n = blockIdx.z / H, h = blockIdx.z % H;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, w = w0 + tx;
        t[k][tx] = (c < C && w < W) ? x[(((long)n * C + c) * H + h) * W + w] : 0.0f;
    }
    __syncthreads();
    const int Wp = W + lo + hi, Hp = H + lo + hi;
    for (int k = ty; k < 32; k += 8) {
        const int w = w0 + k, c = c0 + tx;
        if (w < W && c < Cpad) y[(((long)n * Hp + h + lo) * Wp + w + lo) * Cpad + c] = f32_to_bf16(t[tx][k]);
    }
}

// ---- haloed NHWC bf16 -> NCHW fp32 ------------------------------------------------------------
template <typename TO>
__global__ void __launch_bounds__(256) nhwc_to_nchw_tile_kernel(const bf16_t *__restrict__ x, int N, int C, int H, int W, int halo,
                                                                TO *__restrict__ y)
{
    __shared__ float t[32][33];
    const int w0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int n = blockIdx.z / H, h = blockIdx.z % H;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int Wp = W + 2 * halo, Hp = H + 2 * halo;
    for (int k = ty; k < 32; k += 8) {
        const int w = w0 + k, c = c0 + tx;
        t[k][tx] = (w < W && c < C) ? bf16_to_f32(x[(((long)n * Hp + h + halo) * Wp + w + halo) * C + c]) : 0.0f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, w = w0 + tx;
        if (c < C && w < W) {
            if constexpr (sizeof(TO) == 4) y[(((long)n * C + c) * H + h) * W + w] = t[tx][k];
            else y[(((long)n * C + c) * H + h) * W + w] = f32_to_bf16(t[tx][k]);
        }
    }
}

// ---- conv weights ------------------------------------------------------------------------------
// forward panel  wf[co][ky][kx][ci]  (K-contiguous per output channel; kx/ci zero-padded)
__global__ void pack_conv_fwd_kernel(const float *__restrict__ w, int Cout, int Cin, int KH, int KW, int Cinp, int KWp, bf16_t *__restrict__ wf)
{
    const long total = (long)Cout * KH * KWp * Cinp;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ci = (int)(idx % Cinp);
    const int kx = (int)((idx / Cinp) % KWp);
    const int ky = (int)((idx / ((long)Cinp * KWp)) % KH);
    const int co = (int)(idx / ((long)Cinp * KWp * KH));
    float v = 0.0f;
    if (ci < Cin && kx < KW) v = w[(((long)co * Cin + ci) * KH + ky) * KW + kx];
    wf[idx] = f32_to_bf16(v);
}
// data-gradient panel  wd[ci][ky][kx][co] = w[co][ci][KH-1-ky][KW-1-kx]
__global__ void pack_conv_dgrad_kernel(const float *__restrict__ w, int Cout, int Cin, int KH, int KW, bf16_t *__restrict__ wd)
{
    const long total = (long)Cin * KH * KW * Cout;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co = (int)(idx % Cout);
    const int kx = (int)((idx / Cout) % KW);
    const int ky = (int)((idx / ((long)Cout * KW)) % KH);
    const int ci = (int)(idx / ((long)Cout * KW * KH));
    wd[idx] = f32_to_bf16(w[(((long)co * Cin + ci) * KH + (KH - 1 - ky)) * KW + (KW - 1 - kx)]);
}

// packed fp32 gradient [co][tap][ci] -> OIHW
__global__ void unpack_conv_wgrad_kernel(const float *__restrict__ dwp, int Cout, int Cin, int KH, int KW, int Cinp, int KWp, float *__restrict__ dw, int accumulate)
{
    const long total = (long)Cout * Cin * KH * KW;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kx = (int)(idx % KW);
    const int ky = (int)((idx / KW) % KH);
    const int ci = (int)((idx / ((long)KW * KH)) % Cin);
    const int co = (int)(idx / ((long)KW * KH * Cin));
    const float v = dwp[(((long)co * KH + ky) * KWp + kx) * Cinp + ci];
    dw[idx] = accumulate ? dw[idx] + v : v;
}

// ---- Linear weights ----------------------------------------------------------------------------
// w[o][c*HW + hw] fp32 -> wf[o][hw*C + c] bf16.  One workgroup per (o, 64-channel slab): the slab
// (64*HW contiguous floats) is read coalesced into LDS and written back as HW runs of 64 bf16.
__global__ void __launch_bounds__(256) pack_fc_kernel(const float *__restrict__ w, int O, int C, int HW, bf16_t *__restrict__ wf)
{
    extern __shared__ float slab[];  // [64][HW] as stored
    const int o = blockIdx.y, c0 = blockIdx.x * 64;
    const int nc = min(64, C - c0);
    const float *src = w + (long)o * C * HW + (long)c0 * HW;
    for (int k = threadIdx.x; k < nc * HW; k += 256) slab[k] = src[k];
    __syncthreads();
    bf16_t *dst = wf + (long)o * C * HW;
    for (int k = threadIdx.x; k < nc * HW; k += 256) {
        const int hw = k / nc, c = k - hw * nc;
        dst[(long)hw * C + c0 + c] = f32_to_bf16(slab[c * HW + hw]);
    }
}

// w[o][k] fp32 -> bf16 panels [o/128][k/64][128][64] (rows >= O zero): every LDS stage of yolo_igemm then reads
// one contiguous 16-KB run of a Linear layer's weight stream instead of 128 rows that lie K*2 bytes apart
__global__ void pack_fc_blocked_kernel(const float *__restrict__ w, int O, long K, bf16_t *__restrict__ wb)
{
    const long nk = K / 64;
    const long total8 = (long)((O + 127) / 128) * nk * 128 * 8;  // 8-element groups
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total8) return;
    const int c8 = (int)(idx & 7);
    const int r = (int)((idx >> 3) & 127);
    const long kt = (idx >> 10) % nk;
    const long ct = (idx >> 10) / nk;
    const long o = ct * 128 + r;
    uint4 out = {0u, 0u, 0u, 0u};
    if (o < O) {
        const float *src = w + o * K + kt * 64 + c8 * 8;
        const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
        out.x = (unsigned)f32_to_bf16(a.x) | ((unsigned)f32_to_bf16(a.y) << 16);
        out.y = (unsigned)f32_to_bf16(a.z) | ((unsigned)f32_to_bf16(a.w) << 16);
        out.z = (unsigned)f32_to_bf16(b.x) | ((unsigned)f32_to_bf16(b.y) << 16);
        out.w = (unsigned)f32_to_bf16(b.z) | ((unsigned)f32_to_bf16(b.w) << 16);
    }
    *reinterpret_cast<uint4 *>(wb + idx * 8) = out;
}

// generic tiled transposes: [R][Cc] -> [Cc][R]
template <typename TI, typename TO, typename CV>
__device__ __forceinline__ void transpose_tile(const TI *__restrict__ x, long R, long Cc, TO *__restrict__ y, long ld, CV cv)
{
    __shared__ TO t[64][65];
    const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    for (int k = ty; k < 64; k += 4) {
        const long r = r0 + k, c = c0 + tx;
        if (r < R && c < Cc) t[k][tx] = cv(x[r * Cc + c]);
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        const long c = c0 + k, r = r0 + tx;
        if (r < R && c < Cc) y[c * ld + r] = t[tx][k];
    }
}
__global__ void __launch_bounds__(256) transpose_bf16_kernel(const bf16_t *__restrict__ x, long R, long Cc, bf16_t *__restrict__ y)
{
    transpose_tile(x, R, Cc, y, R, [](bf16_t v) { return v; });
}
__global__ void __launch_bounds__(256) transpose_f32_bf16_kernel(const float *__restrict__ x, long R, long Cc, bf16_t *__restrict__ y, long ld)
{
    transpose_tile(x, R, Cc, y, ld, [](float v) { return f32_to_bf16(v); });
}

// ---- first-layer weight-gradient operand: rows of KH x 32 input elements per output pixel -------
// xcol[n][oy+ho][ox+ho][ky*seg + j] = x[n][oy*stride + ky][ (ox*stride)*px + j ], j < seg  (haloed output geometry)
__global__ void im2col_rows_kernel(const bf16_t *__restrict__ x, long x_img_stride, int x_row_stride, int x_px_stride, int stride, int KH, int seg,
                                   int N, int Ho, int Wo, int ho, bf16_t *__restrict__ xcol)
{
    const int spp = KH * seg / 8;  // 16-B chunks per pixel
    const long total = (long)N * Ho * Wo * spp;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)(idx % spp);
    const long px = idx / spp;
    const int ox = (int)(px % Wo);
    const int oy = (int)((px / Wo) % Ho);
    const int n = (int)(px / ((long)Wo * Ho));
    const int ky = ch / (seg / 8), part = ch % (seg / 8);
    const uint4 v = *reinterpret_cast<const uint4 *>(x + (long)n * x_img_stride + (long)(oy * stride + ky) * x_row_stride + (long)(ox * stride) * x_px_stride + part * 8);
    const int Wop = Wo + 2 * ho, Hop = Ho + 2 * ho;
    *reinterpret_cast<uint4 *>(xcol + (((long)n * Hop + oy + ho) * Wop + ox + ho) * (KH * seg) + ch * 8) = v;
}

// ---- casts / row epilogue ----------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float *__restrict__ x, long n, bf16_t *__restrict__ y)
{
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const float4 a = *reinterpret_cast<const float4 *>(x + i), b = *reinterpret_cast<const float4 *>(x + i + 4);
        uint4 o;
        o.x = (unsigned)f32_to_bf16(a.x) | ((unsigned)f32_to_bf16(a.y) << 16);
        o.y = (unsigned)f32_to_bf16(a.z) | ((unsigned)f32_to_bf16(a.w) << 16);
        o.z = (unsigned)f32_to_bf16(b.x) | ((unsigned)f32_to_bf16(b.y) << 16);
        o.w = (unsigned)f32_to_bf16(b.z) | ((unsigned)f32_to_bf16(b.w) << 16);
        *reinterpret_cast<uint4 *>(y + i) = o;
    } else {
        for (long k = i; k < n; ++k) y[k] = f32_to_bf16(x[k]);
    }
}
__global__ void cast_bf16_f32_kernel(const bf16_t *__restrict__ x, long n, float *__restrict__ y)
{
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(x + i);
        float4 a, b;
        a.x = __uint_as_float(v.x << 16); a.y = __uint_as_float(v.x & 0xffff0000u);
        a.z = __uint_as_float(v.y << 16); a.w = __uint_as_float(v.y & 0xffff0000u);
        b.x = __uint_as_float(v.z << 16); b.y = __uint_as_float(v.z & 0xffff0000u);
        b.z = __uint_as_float(v.w << 16); b.w = __uint_as_float(v.w & 0xffff0000u);
        *reinterpret_cast<float4 *>(y + i) = a;
        *reinterpret_cast<float4 *>(y + i + 4) = b;
    } else {
        for (long k = i; k < n; ++k) y[k] = bf16_to_f32(x[k]);
    }
}
__global__ void bias_lrelu_rows_kernel(const float *__restrict__ x, const float *__restrict__ bias, int R, int Cc, float slope,
                                       bf16_t *__restrict__ yb, float *__restrict__ yf)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * Cc) return;
    const int c = (int)(idx % Cc);
    float v = x[idx] + (bias ? bias[c] : 0.0f);
    v = v > 0.0f ? v : v * slope;
    if (yb) yb[idx] = f32_to_bf16(v);
    if (yf) yf[idx] = v;
}
// y[r][c] = bf16( x[r][c] * (mask ? mask[r][c] * scale : 1) * (act ? (act[r][c] > 0 ? 1 : slope) : 1) ), zero-padded to ld columns
__global__ void scale_rows_kernel(const float *__restrict__ x, const unsigned char *__restrict__ mask, float scale, const bf16_t *__restrict__ act,
                                  float slope, int R, int Cc, int ld, bf16_t *__restrict__ y)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * ld) return;
    const int c = (int)(idx % ld);
    const long r = idx / ld;
    float v = 0.0f;
    if (c < Cc) {
        const long src = r * Cc + c;
        v = x[src];
        if (mask) v *= mask[src] ? scale : 0.0f;
        if (act) v *= bf16_to_f32(act[src]) > 0.0f ? 1.0f : slope;
    }
    y[idx] = f32_to_bf16(v);
}
// y = bf16( x * (mask ? scale : 0) ) elementwise on bf16 (dropout forward)
__global__ void dropout_bf16_kernel(const bf16_t *__restrict__ x, const unsigned char *__restrict__ mask, float scale, long n, bf16_t *__restrict__ y)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    y[idx] = mask[idx] ? f32_to_bf16(bf16_to_f32(x[idx]) * scale) : (bf16_t)0;
}

}  // namespace yolo

using namespace yolo;

static inline unsigned nblk(long n, int per) { return (unsigned)((n + per - 1) / per); }

YOLO_API int yolo_nchw_f32_to_nhwc_bf16(const float *x, int N, int C, int H, int W, void *y, int Cpad, int halo_lo, int halo_hi, yolo_stream_t stream)
{
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C || halo_lo < 0 || halo_hi < 0) return fail(YOLO_E_ARG, "yolo_nchw_f32_to_nhwc_bf16: bad argument");
    if (Cpad == 4 && C <= 4) {
        hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(nblk((long)N * H * W, 256)), dim3(256), 0, STRM(stream), x, N, C, H, W, (bf16_t *)y, halo_lo, halo_hi);
    } else {
        if ((long)N * H > 65535) return fail(YOLO_E_UNSUPPORTED, "yolo_nchw_f32_to_nhwc_bf16: N*H=%ld > 65535", (long)N * H);
        hipLaunchKernelGGL(nchw_to_nhwc_tile_kernel, dim3(nblk(W, 32), nblk(Cpad, 32), N * H), dim3(256), 0, STRM(stream), x, N, C, H, W, (bf16_t *)y, Cpad, halo_lo, halo_hi);
    }
    return check_launch("yolo_nchw_f32_to_nhwc_bf16");
}

YOLO_API int yolo_nhwc_bf16_to_nchw_f32(const void *x, int N, int C, int H, int W, int halo, float *y, yolo_stream_t stream)
{
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || halo < 0) return fail(YOLO_E_ARG, "yolo_nhwc_bf16_to_nchw_f32: bad argument");
    if ((long)N * H > 65535) return fail(YOLO_E_UNSUPPORTED, "yolo_nhwc_bf16_to_nchw_f32: N*H=%ld > 65535", (long)N * H);
    hipLaunchKernelGGL(nhwc_to_nchw_tile_kernel<float>, dim3(nblk(W, 32), nblk(C, 32), N * H), dim3(256), 0, STRM(stream), (const bf16_t *)x, N, C, H, W, halo, y);
    return check_launch("yolo_nhwc_bf16_to_nchw_f32");
}

YOLO_API int yolo_nhwc_bf16_to_nchw_bf16(const void *x, int N, int C, int H, int W, int halo, void *y, yolo_stream_t stream)
{
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || halo < 0) return fail(YOLO_E_ARG, "yolo_nhwc_bf16_to_nchw_bf16: bad argument");
    if ((long)N * H > 65535) return fail(YOLO_E_UNSUPPORTED, "yolo_nhwc_bf16_to_nchw_bf16: N*H=%ld > 65535", (long)N * H);
    hipLaunchKernelGGL(nhwc_to_nchw_tile_kernel<bf16_t>, dim3(nblk(W, 32), nblk(C, 32), N * H), dim3(256), 0, STRM(stream), (const bf16_t *)x, N, C, H, W, halo, (bf16_t *)y);
    return check_launch("yolo_nhwc_bf16_to_nchw_bf16");
}

YOLO_API int yolo_pack_conv_weight(const float *w, int Cout, int Cin, int KH, int KW, int Cinp, int KWp, void *wf, void *wd, yolo_stream_t stream)
{
    if (!w || (!wf && !wd) || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || Cinp < Cin || KWp < KW) return fail(YOLO_E_ARG, "yolo_pack_conv_weight: bad argument");
    if (wf) hipLaunchKernelGGL(pack_conv_fwd_kernel, dim3(nblk((long)Cout * KH * KWp * Cinp, 256)), dim3(256), 0, STRM(stream), w, Cout, Cin, KH, KW, Cinp, KWp, (bf16_t *)wf);
    if (wd) hipLaunchKernelGGL(pack_conv_dgrad_kernel, dim3(nblk((long)Cout * KH * KW * Cin, 256)), dim3(256), 0, STRM(stream), w, Cout, Cin, KH, KW, (bf16_t *)wd);
    return check_launch("yolo_pack_conv_weight");
}

YOLO_API int yolo_pack_fc_weight(const float *w, int O, int C, int HW, void *wf, void *wt, yolo_stream_t stream)
{
    if (!w || !wf || O <= 0 || C <= 0 || HW <= 0) return fail(YOLO_E_ARG, "yolo_pack_fc_weight: bad argument");
    if (HW == 1) {
        hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(nblk((long)O * C, 256 * 8)), dim3(256), 0, STRM(stream), w, (long)O * C, (bf16_t *)wf);
    } else {
        if (O > 65535 || (size_t)64 * HW * 4 > 64 * 1024) return fail(YOLO_E_UNSUPPORTED, "yolo_pack_fc_weight: O=%d > 65535 or HW=%d > 256", O, HW);
        hipLaunchKernelGGL(pack_fc_kernel, dim3(nblk(C, 64), O), dim3(256), (size_t)64 * HW * 4, STRM(stream), w, O, C, HW, (bf16_t *)wf);
    }
    if (wt) {
        const long K = (long)C * HW;
        hipLaunchKernelGGL(transpose_bf16_kernel, dim3(nblk(K, 64), nblk(O, 64)), dim3(256), 0, STRM(stream), (const bf16_t *)wf, (long)O, K, (bf16_t *)wt);
    }
    return check_launch("yolo_pack_fc_weight");
}

YOLO_API int yolo_unpack_conv_wgrad(const float *dwp, int Cout, int Cin, int KH, int KW, int Cinp, int KWp, float *dw, int accumulate, yolo_stream_t stream)
{
    if (!dwp || !dw || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || Cinp < Cin || KWp < KW) return fail(YOLO_E_ARG, "yolo_unpack_conv_wgrad: bad argument");
    hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3(nblk((long)Cout * Cin * KH * KW, 256)), dim3(256), 0, STRM(stream), dwp, Cout, Cin, KH, KW, Cinp, KWp, dw, accumulate);
    return check_launch("yolo_unpack_conv_wgrad");
}

YOLO_API int yolo_transpose_f32_to_bf16(const float *x, int R, int Cc, void *y, int ld, yolo_stream_t stream)
{
    if (!x || !y || R <= 0 || Cc <= 0 || ld < R) return fail(YOLO_E_ARG, "yolo_transpose_f32_to_bf16: bad argument");
    hipLaunchKernelGGL(transpose_f32_bf16_kernel, dim3(nblk(Cc, 64), nblk(R, 64)), dim3(256), 0, STRM(stream), x, (long)R, (long)Cc, (bf16_t *)y, (long)ld);
    return check_launch("yolo_transpose_f32_to_bf16");
}

YOLO_API int yolo_cast_f32_to_bf16(const float *x, long n, void *y, yolo_stream_t stream)
{
    if (!x || !y || n < 0) return fail(YOLO_E_ARG, "yolo_cast_f32_to_bf16: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(nblk(n, 256 * 8)), dim3(256), 0, STRM(stream), x, n, (bf16_t *)y);
    return check_launch("yolo_cast_f32_to_bf16");
}

YOLO_API int yolo_cast_bf16_to_f32(const void *x, long n, float *y, yolo_stream_t stream)
{
    if (!x || !y || n < 0) return fail(YOLO_E_ARG, "yolo_cast_bf16_to_f32: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(nblk(n, 256 * 8)), dim3(256), 0, STRM(stream), (const bf16_t *)x, n, y);
    return check_launch("yolo_cast_bf16_to_f32");
}

YOLO_API int yolo_bias_lrelu_rows(const float *x, const float *bias, int R, int Cc, float slope, void *yb, float *yf, yolo_stream_t stream)
{
    if (!x || (!yb && !yf) || R <= 0 || Cc <= 0) return fail(YOLO_E_ARG, "yolo_bias_lrelu_rows: bad argument");
    hipLaunchKernelGGL(bias_lrelu_rows_kernel, dim3(nblk((long)R * Cc, 256)), dim3(256), 0, STRM(stream), x, bias, R, Cc, slope, (bf16_t *)yb, yf);
    return check_launch("yolo_bias_lrelu_rows");
}

YOLO_API int yolo_scale_rows_to_bf16(const float *x, const unsigned char *mask, float scale, const void *act, float slope, int R, int Cc, int ld, void *y,
                                     yolo_stream_t stream)
{
    if (!x || !y || R <= 0 || Cc <= 0 || ld < Cc) return fail(YOLO_E_ARG, "yolo_scale_rows_to_bf16: bad argument");
    hipLaunchKernelGGL(scale_rows_kernel, dim3(nblk((long)R * ld, 256)), dim3(256), 0, STRM(stream), x, mask, scale, (const bf16_t *)act, slope, R, Cc, ld, (bf16_t *)y);
    return check_launch("yolo_scale_rows_to_bf16");
}

YOLO_API int yolo_dropout_bf16(const void *x, const unsigned char *mask, float scale, long n, void *y, yolo_stream_t stream)
{
    if (!x || !mask || !y || n < 0) return fail(YOLO_E_ARG, "yolo_dropout_bf16: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_bf16_kernel, dim3(nblk(n, 256)), dim3(256), 0, STRM(stream), (const bf16_t *)x, mask, scale, n, (bf16_t *)y);
    return check_launch("yolo_dropout_bf16");
}

YOLO_API int yolo_im2col_rows(const void *x, long x_img_stride, int x_row_stride, int x_px_stride, int stride, int KH, int seg, int N, int Ho, int Wo,
                              int out_halo, void *xcol, yolo_stream_t stream)
{
    if (!x || !xcol || N <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || seg <= 0 || (seg & 7) || stride <= 0 || out_halo < 0) return fail(YOLO_E_ARG, "yolo_im2col_rows: bad argument");
    const long total = (long)N * Ho * Wo * (KH * seg / 8);
    hipLaunchKernelGGL(im2col_rows_kernel, dim3(nblk(total, 256)), dim3(256), 0, STRM(stream), (const bf16_t *)x, x_img_stride, x_row_stride, x_px_stride, stride, KH, seg,
                       N, Ho, Wo, out_halo, (bf16_t *)xcol);
    return check_launch("yolo_im2col_rows");
}

YOLO_API int yolo_pack_fc_weight_blocked(const float *w, int O, long K, void *wb, yolo_stream_t stream)
{
    if (!w || !wb || O <= 0 || K <= 0 || (K & 63)) return fail(YOLO_E_ARG, "yolo_pack_fc_weight_blocked: bad argument (K must be a multiple of 64)");
    const long total8 = (long)((O + 127) / 128) * (K / 64) * 128 * 8;
    hipLaunchKernelGGL(pack_fc_blocked_kernel, dim3(nblk(total8, 256)), dim3(256), 0, STRM(stream), w, O, K, (bf16_t *)wb);
    return check_launch("yolo_pack_fc_weight_blocked");
}
