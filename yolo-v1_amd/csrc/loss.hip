// Fused YOLOv1 loss forward + backward for gfx950.
//
// Replaces YOLOLoss.forward / compute_iou (src/yolo/loss.py:87-212 of mattiaskvist/yolo-v1) and the
// ~35 aten kernels + 5 host syncs autograd runs for it: one wavefront per image computes, per
// cell, the responsible box (IoU arg-max), the four partial sums and d(total)/d(pred) -- including
// the gradient through the IoU confidence target, which the reference does not detach
// (loss.py:111,123,144) -- then a second single-workgroup kernel reduces the per-image partials in
// a fixed order (deterministic, no atomics).
//
// fp32 per-cell arithmetic in the reference's order (compiled with -ffp-contract=off), fp64 sums.
#include "common.h"

namespace yolo {

#pragma clang fp contract(off)

#define LOSS_MAX_B 8

struct IouCtx { float iou, inter, U, iw, ih, gx1, gx2, gy1, gy2, dwp, dhp; };

__device__ __forceinline__ float maxgrad(float a, float b) { return a > b ? 1.0f : (a == b ? 0.5f : 0.0f); }
__device__ __forceinline__ float mingrad(float a, float b) { return a < b ? 1.0f : (a == b ? 0.5f : 0.0f); }

// loss.py:191-212
__device__ __forceinline__ void iou_fwd(const float *p, const float *t, IouCtx &c)
{
    const float x1 = p[0] - p[2] / 2, y1 = p[1] - p[3] / 2, x2 = p[0] + p[2] / 2, y2 = p[1] + p[3] / 2;
    const float tx1 = t[0] - t[2] / 2, ty1 = t[1] - t[3] / 2, tx2 = t[0] + t[2] / 2, ty2 = t[1] + t[3] / 2;
    const float ix1 = x1 > tx1 ? x1 : tx1, iy1 = y1 > ty1 ? y1 : ty1;
    const float ix2 = x2 < tx2 ? x2 : tx2, iy2 = y2 < ty2 ? y2 : ty2;
    const float dw = ix2 - ix1, dh = iy2 - iy1;
    c.iw = dw < 0 ? 0.0f : dw;
    c.ih = dh < 0 ? 0.0f : dh;
    c.dwp = dw >= 0 ? 1.0f : 0.0f;  // clamp(min=0) passes the gradient where input >= 0
    c.dhp = dh >= 0 ? 1.0f : 0.0f;
    c.inter = c.iw * c.ih;
    const float a1 = p[2] * p[3], a2 = t[2] * t[3];
    const float uni = a1 + a2 - c.inter;
    c.U = uni + 1e-6f;
    c.iou = c.inter / c.U;
    c.gx1 = maxgrad(x1, tx1);
    c.gy1 = maxgrad(y1, ty1);
    c.gx2 = mingrad(x2, tx2);
    c.gy2 = mingrad(y2, ty2);
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// grid = N, block = 64 (one wavefront per image)
__global__ void __launch_bounds__(64) loss_cells_kernel(const float *__restrict__ pred, const float *__restrict__ tgt, int S, int B, int C,
                                                        float lc, float ln, float invN, float *__restrict__ dpred,
                                                        double *__restrict__ work)
{
    const int D = B * 5 + C;
    const int ncell = S * S;
    const int img = blockIdx.x;
    double s_coord = 0, s_obj = 0, s_noobj = 0, s_cls = 0, s_err = 0;
    for (int cell = threadIdx.x; cell < ncell; cell += 64) {
        const float *p = pred + ((size_t)img * ncell + cell) * D;
        const float *t = tgt + ((size_t)img * ncell + cell) * D;
        float *g = dpred ? dpred + ((size_t)img * ncell + cell) * D : nullptr;
        // loss.py:98-102 -- the slice 4::5 runs over ALL D channels
        bool obj = false;
        int idx = 0;
        for (int k = 0, ch = 4; ch < D; ch += 5, ++k)
            if (t[ch] > 0.0f) { if (!obj) idx = k; obj = true; }
        if (g)
            for (int k = 0; k < D; ++k) g[k] = 0.0f;
        if (!obj) {
            for (int b = 0; b < B; ++b) {
                const float c = p[b * 5 + 4];
                s_noobj += (double)(c * c);
                if (g) g[b * 5 + 4] = ln * 2.0f * c * invN;
            }
            continue;
        }
        if (idx >= B) { s_err += 1.0; continue; }
        const float *tb = t + idx * 5;
        IouCtx best_ctx;
        int best = 0;
        iou_fwd(p, tb, best_ctx);
        for (int b = 1; b < B; ++b) {
            IouCtx c;
            iou_fwd(p + b * 5, tb, c);
            if (c.iou > best_ctx.iou) { best_ctx = c; best = b; }  // argmax: first maximum
        }
        const float *pr = p + best * 5;
        const IouCtx &c = best_ctx;
        const float dx = pr[0] - tb[0], dy = pr[1] - tb[1];
        const float cw = pr[2] < 1e-6f ? 1e-6f : pr[2], chh = pr[3] < 1e-6f ? 1e-6f : pr[3];
        const float ctw = tb[2] < 1e-6f ? 1e-6f : tb[2], cth = tb[3] < 1e-6f ? 1e-6f : tb[3];
        const float sw = sqrtf(cw), sh = sqrtf(chh);
        const float ew = sw - sqrtf(ctw), eh = sh - sqrtf(cth);
        s_coord += (double)(dx * dx) + (double)(dy * dy) + (double)(ew * ew) + (double)(eh * eh);
        const float ec = pr[4] - c.iou;
        s_obj += (double)(ec * ec);
        for (int b = 0; b < B; ++b)
            if (b != best) {
                const float cc = p[b * 5 + 4];
                s_noobj += (double)(cc * cc);
                if (g) g[b * 5 + 4] = ln * 2.0f * cc * invN;
            }
        for (int k = 0; k < C; ++k) {
            const float e = p[B * 5 + k] - t[B * 5 + k];
            s_cls += (double)(e * e);
            if (g) g[B * 5 + k] = 2.0f * e * invN;
        }
        if (g) {
            float *gr = g + best * 5;
            const float g_iou = -2.0f * ec;
            const float inv_u = 1.0f / c.U;
            const float d_inter = g_iou * (inv_u + c.inter * inv_u * inv_u);
            const float d_area = -g_iou * c.inter * inv_u * inv_u;
            const float d_iw = d_inter * c.ih * c.dwp;
            const float d_ih = d_inter * c.iw * c.dhp;
            const float d_x1 = -d_iw * c.gx1, d_x2 = d_iw * c.gx2;
            const float d_y1 = -d_ih * c.gy1, d_y2 = d_ih * c.gy2;
            float gx = d_x1 + d_x2, gy = d_y1 + d_y2;
            float gw = 0.5f * (d_x2 - d_x1) + d_area * pr[3];
            float gh = 0.5f * (d_y2 - d_y1) + d_area * pr[2];
            gx += lc * 2.0f * dx;
            gy += lc * 2.0f * dy;
            if (pr[2] >= 1e-6f) gw += lc * 2.0f * ew * (0.5f / sw);
            if (pr[3] >= 1e-6f) gh += lc * 2.0f * eh * (0.5f / sh);
            gr[0] = gx * invN; gr[1] = gy * invN; gr[2] = gw * invN; gr[3] = gh * invN;
            gr[4] = 2.0f * ec * invN;
        }
    }
    s_coord = wave_sum(s_coord);
    s_obj = wave_sum(s_obj);
    s_noobj = wave_sum(s_noobj);
    s_cls = wave_sum(s_cls);
    s_err = wave_sum(s_err);
    if (threadIdx.x == 0) {
        double *w = work + (size_t)img * 8;
        w[0] = s_coord; w[1] = s_obj; w[2] = s_noobj; w[3] = s_cls; w[4] = s_err;
    }
}

// single workgroup: fixed-order tree over the per-image partials
__global__ void __launch_bounds__(256) loss_reduce_kernel(const double *__restrict__ work, int N, float lc, float ln, float *__restrict__ out)
{
    __shared__ double sm[5][256];
    double acc[5] = {0, 0, 0, 0, 0};
    for (int n = threadIdx.x; n < N; n += 256)
        for (int k = 0; k < 5; ++k) acc[k] += work[(size_t)n * 8 + k];
    for (int k = 0; k < 5; ++k) sm[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int k = 0; k < 5; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float coord = (float)((double)lc * sm[0][0]);
        const float cobj = (float)sm[1][0];
        const float cno = (float)((double)ln * sm[2][0]);
        const float cls = (float)sm[3][0];
        const float fN = (float)N;
        out[0] = (coord + cobj + cno + cls) / fN;  // loss.py:162
        out[1] = coord / fN;
        out[2] = cobj / fN;
        out[3] = cno / fN;
        out[4] = cls / fN;
        out[5] = sm[4][0] > 0 ? 1.0f : 0.0f;
        out[6] = 0.0f;
        out[7] = 0.0f;
    }
}

__global__ void loss_iou_kernel(const float *__restrict__ b1, const float *__restrict__ b2, long n, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    IouCtx c;
    iou_fwd(b1 + 4 * i, b2 + 4 * i, c);
    out[i] = c.iou;
}

}  // namespace yolo

using namespace yolo;

YOLO_API int yolo_loss_fwd_bwd(const float *pred, const float *tgt, int N, int S, int B, int C, float lambda_coord, float lambda_noobj,
                               float *out, float *dpred, double *work, yolo_stream_t stream)
{
    if (!pred || !tgt || !out || !work || N <= 0 || S <= 0 || B <= 0 || C < 0) return fail(YOLO_E_ARG, "yolo_loss_fwd_bwd: bad argument");
    if (B > LOSS_MAX_B || S * S > 1024) return fail(YOLO_E_UNSUPPORTED, "yolo_loss_fwd_bwd: B=%d > %d or S*S=%d > 1024", B, LOSS_MAX_B, S * S);
    hipLaunchKernelGGL(loss_cells_kernel, dim3(N), dim3(64), 0, STRM(stream), pred, tgt, S, B, C, lambda_coord, lambda_noobj, 1.0f / (float)N, dpred, work);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, STRM(stream), work, N, lambda_coord, lambda_noobj, out);
    return check_launch("yolo_loss_fwd_bwd");
}

YOLO_API int yolo_loss_iou(const float *boxes1, const float *boxes2, long n, float *out, yolo_stream_t stream)
{
    if (!boxes1 || !boxes2 || !out || n < 0) return fail(YOLO_E_ARG, "yolo_loss_iou: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(loss_iou_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STRM(stream), boxes1, boxes2, n, out);
    return check_launch("yolo_loss_iou");
}
