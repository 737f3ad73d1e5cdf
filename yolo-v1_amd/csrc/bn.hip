// BatchNorm2d in TRAINING mode (batch statistics) on zero-haloed NHWC bf16, forward only, for gfx950.
// The reference trains its default model with a FROZEN ResNet-50 backbone whose BatchNorm layers nevertheless run in
// training mode -- train_epoch calls model.train() on the whole model (src/yolo/training/trainer.py:49), so they
// normalise with the batch statistics and update their running statistics.  Such a layer cannot be
// folded into the convolution; it becomes three HBM-bound passes over the conv output z:
//   bn_stats    : per-channel sum and sum of squares over the N*H*W interior pixels (fp32 partials per thread,
//                 fp64 atomics per workgroup)
//   bn_finalize : mean, biased variance, scale = gamma / sqrt(var + eps), shift = beta - mean * scale; running_mean /
//                 running_var (unbiased) updated with `momentum` exactly as aten does; accumulators cleared
//   bn_apply    : y = [relu]( z * scale + shift [+ residual] ), in place or into a second buffer, 16 B per lane
// and, for a TRAINABLE trunk (the reference's default run: ResNetBackbone(freeze=False), src/train.py:144), the backward:
//   bn_bwd_reduce   : per channel sum(dy') and sum(dy' * xhat), dy' = dy * [y > 0] when a ReLU follows, xhat = (z - mean) * invstd
//   bn_bwd_finalize : dgamma, dbeta; coefficients of the apply pass
//   bn_bwd_apply    : dz = gamma * invstd * (dy' - mean(dy') - xhat * mean(dy' * xhat)), written with caller strides (a
//                     stride-2 conv wants it zero-stuffed on its input grid); dy' optionally stored back over dy
// Replaces aten batch_norm (training=True) + relu + the residual add of torchvision's Bottleneck
// (src/yolo/models.py:154-176 via torchvision.models.resnet50).
#include "common.h"

namespace yolo {

__device__ __forceinline__ void bn_unpack8(const uint4 &v, float f[8])
{
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

// (n, y, x) of pixel p0, p0 + stride, p0 + 2*stride, ... without a division per pixel
struct PixIter {
    int n, y, x, sn, sy, sx, H, W;
    __device__ __forceinline__ PixIter(long p0, long stride, int H_, int W_) : H(H_), W(W_)
    {
        const long hw = (long)H_ * W_;
        n = (int)(p0 / hw);
        int r = (int)(p0 - (long)n * hw);
        y = r / W_;
        x = r - y * W_;
        sn = (int)(stride / hw);
        r = (int)(stride - (long)sn * hw);
        sy = r / W_;
        sx = r - sy * W_;
    }
    __device__ __forceinline__ void next()
    {
        x += sx;
        if (x >= W) { x -= W; ++y; }
        y += sy;
        if (y >= H) { y -= H; ++n; }
        n += sn;
    }
    __device__ __forceinline__ long off(int halo, int C) const { return (((long)n * (H + 2 * halo) + y + halo) * (W + 2 * halo) + x + halo) * C; }
};

// thread = (channel group of 8, pixel lane); grid-stride over the interior pixels, four independent loads in flight
__global__ void __launch_bounds__(256) bn_stats_kernel(const bf16_t *__restrict__ z, int N, int H, int W, int C, int halo, double *__restrict__ acc)
{
    const int C8 = C >> 3;
    const int gpb = C8 < 256 ? C8 : 256;          // channel groups per workgroup
    const int ppb = 256 / gpb;                    // pixel lanes per workgroup
    const int cg = blockIdx.x * gpb + threadIdx.x % gpb, pl = threadIdx.x / gpb;
    const long P = (long)N * H * W;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ss[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (cg < C8 && pl < ppb) {
        const long stride = (long)gridDim.y * ppb;
        long p = (long)blockIdx.y * ppb + pl;
        PixIter it(p < P ? p : 0, stride, H, W);
        for (; p + 3 * stride < P; p += 4 * stride) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { v[u] = *reinterpret_cast<const uint4 *>(z + it.off(halo, C) + cg * 8); it.next(); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                bn_unpack8(v[u], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s[k] += f[k]; ss[k] += f[k] * f[k]; }
            }
        }
        for (; p < P; p += stride) {
            float f[8];
            bn_unpack8(*reinterpret_cast<const uint4 *>(z + it.off(halo, C) + cg * 8), f);
            it.next();
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += f[k]; ss[k] += f[k] * f[k]; }
        }
    }
    __shared__ float red[2][256][9];
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[0][threadIdx.x][k] = s[k]; red[1][threadIdx.x][k] = ss[k]; }
    __syncthreads();
    if (pl == 0 && cg < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double a = 0.0, b = 0.0;
            for (int r = 0; r < ppb; ++r) { a += red[0][r * gpb + threadIdx.x][k]; b += red[1][r * gpb + threadIdx.x][k]; }
            double *rep = acc + (size_t)(blockIdx.y % YOLO_BN_ACC_REPLICAS) * 2 * C;   // same-address fp64 atomics cost ~60-150 ns each: spread them
            atomicAdd(rep + cg * 8 + k, a);
            atomicAdd(rep + C + cg * 8 + k, b);
        }
    }
}

__global__ void bn_finalize_kernel(double *__restrict__ acc, int C, double count, const float *__restrict__ gamma, const float *__restrict__ beta, double eps,
                                   double momentum, float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ scale,
                                   float *__restrict__ shift, float *__restrict__ save_mean_invstd, int frozen)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (frozen) {          // eval() mode with gradients: the running statistics ARE the statistics (aten batch_norm(training=False)); nothing is updated
        mean = (double)running_mean[c];
        var = (double)running_var[c];
    } else {
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < YOLO_BN_ACC_REPLICAS; ++r) {
            s1 += acc[(size_t)r * 2 * C + c];
            s2 += acc[(size_t)r * 2 * C + C + c];
            acc[(size_t)r * 2 * C + c] = 0.0;          // ready for the next layer
            acc[(size_t)r * 2 * C + C + c] = 0.0;
        }
        mean = s1 / count;
        var = s2 / count - mean * mean;
    }
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + eps);
    const double sc = (double)gamma[c] * invstd;
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - mean * sc);
    if (save_mean_invstd) {     // [mean | invstd | scale | shift]: the backward recomputes a ReLU mask from z with the forward's own scale / shift
        save_mean_invstd[c] = (float)mean; save_mean_invstd[C + c] = (float)invstd;
        save_mean_invstd[2 * C + c] = scale[c]; save_mean_invstd[3 * C + c] = shift[c];
    }
    if (running_mean && !frozen) {
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

__global__ void __launch_bounds__(256) bn_apply_kernel(bf16_t *__restrict__ z, int N, int H, int W, int C, int halo, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, const bf16_t *__restrict__ residual, int res_halo, int relu,
                                                       bf16_t *__restrict__ out, int out_halo)
{
    const int C8 = C >> 3;
    const long total = (long)N * H * W * C8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % C8);
    const long p = idx / C8;
    const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((long)W * H));
    const int Hp = H + 2 * halo, Wp = W + 2 * halo;
    bf16_t *q = z + (((long)n * Hp + y + halo) * Wp + x + halo) * C + cg * 8;
    float f[8], r[8];
    bn_unpack8(*reinterpret_cast<const uint4 *>(q), f);
    if (residual) {
        const int Hr = H + 2 * res_halo, Wr = W + 2 * res_halo;
        bn_unpack8(*reinterpret_cast<const uint4 *>(residual + (((long)n * Hr + y + res_halo) * Wr + x + res_halo) * C + cg * 8), r);
    }
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        float a = fmaf(f[k], scale[cg * 8 + k], shift[cg * 8 + k]), b = fmaf(f[k + 1], scale[cg * 8 + k + 1], shift[cg * 8 + k + 1]);
        if (residual) { a += r[k]; b += r[k + 1]; }
        if (relu) { a = a > 0.0f ? a : 0.0f; b = b > 0.0f ? b : 0.0f; }
        o[k >> 1] = (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);
    }
    if (out) {
        const int Ho = H + 2 * out_halo, Wo = W + 2 * out_halo;
        q = out + (((long)n * Ho + y + out_halo) * Wo + x + out_halo) * C + cg * 8;
    }
    *reinterpret_cast<uint4 *>(q) = uint4{o[0], o[1], o[2], o[3]};
}

// ---- backward.  dy' = dy (masked by y > 0 when the unit ends in a ReLU); all of dy, y, z are [N][H+2h][W+2h][C].
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const bf16_t *__restrict__ dy, int dy_halo, const bf16_t *__restrict__ yact, int y_halo,
                                                            const bf16_t *__restrict__ z, int z_halo, int N, int H, int W, int C,
                                                            const float *__restrict__ mean_invstd, double *__restrict__ acc, int mask_from_z)
{
    const int C8 = C >> 3;
    const int gpb = C8 < 256 ? C8 : 256;
    const int ppb = 256 / gpb;
    const int cg = blockIdx.x * gpb + threadIdx.x % gpb, pl = threadIdx.x / gpb;
    const long P = (long)N * H * W;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ss[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (cg < C8 && pl < ppb) {
        float mu[8], is[8], sc[8], sh[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mu[k] = mean_invstd[cg * 8 + k]; is[k] = mean_invstd[C + cg * 8 + k];
            sc[k] = mask_from_z ? mean_invstd[2 * C + cg * 8 + k] : 0.0f; sh[k] = mask_from_z ? mean_invstd[3 * C + cg * 8 + k] : 0.0f;
        }
        const long stride = (long)gridDim.y * ppb;
        long p = (long)blockIdx.y * ppb + pl;
        PixIter it(p < P ? p : 0, stride, H, W);
        auto fold = [&](const uint4 &vg, const uint4 &vz, const uint4 &va) {
            float g[8], zz[8], a[8];
            bn_unpack8(vg, g);
            bn_unpack8(vz, zz);
            if (yact) {
                bn_unpack8(va, a);
#pragma unroll
                for (int k = 0; k < 8; ++k) g[k] = a[k] > 0.0f ? g[k] : 0.0f;
            } else if (mask_from_z) {
#pragma unroll
                for (int k = 0; k < 8; ++k) g[k] = fmaf(zz[k], sc[k], sh[k]) > 0.0f ? g[k] : 0.0f;     // the forward's own expression
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += g[k]; ss[k] += g[k] * ((zz[k] - mu[k]) * is[k]); }
        };
        for (; p + 1 * stride < P; p += 2 * stride) {      // two pixels = six independent 16-B loads in flight
            uint4 vg[2], vz[2], va[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                vg[u] = *reinterpret_cast<const uint4 *>(dy + it.off(dy_halo, C) + cg * 8);
                vz[u] = *reinterpret_cast<const uint4 *>(z + it.off(z_halo, C) + cg * 8);
                va[u] = yact ? *reinterpret_cast<const uint4 *>(yact + it.off(y_halo, C) + cg * 8) : uint4{0, 0, 0, 0};
                it.next();
            }
            fold(vg[0], vz[0], va[0]);
            fold(vg[1], vz[1], va[1]);
        }
        for (; p < P; p += stride) {
            const uint4 vg = *reinterpret_cast<const uint4 *>(dy + it.off(dy_halo, C) + cg * 8);
            const uint4 vz = *reinterpret_cast<const uint4 *>(z + it.off(z_halo, C) + cg * 8);
            const uint4 va = yact ? *reinterpret_cast<const uint4 *>(yact + it.off(y_halo, C) + cg * 8) : uint4{0, 0, 0, 0};
            it.next();
            fold(vg, vz, va);
        }
    }
    __shared__ float red[2][256][9];
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[0][threadIdx.x][k] = s[k]; red[1][threadIdx.x][k] = ss[k]; }
    __syncthreads();
    if (pl == 0 && cg < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double a = 0.0, b = 0.0;
            for (int r = 0; r < ppb; ++r) { a += red[0][r * gpb + threadIdx.x][k]; b += red[1][r * gpb + threadIdx.x][k]; }
            double *rep = acc + (size_t)(blockIdx.y % YOLO_BN_ACC_REPLICAS) * 2 * C;   // same-address fp64 atomics cost ~60-150 ns each: spread them
            atomicAdd(rep + cg * 8 + k, a);
            atomicAdd(rep + C + cg * 8 + k, b);
        }
    }
}

__global__ void bn_bwd_finalize_kernel(double *__restrict__ acc, int C, double count, const float *__restrict__ gamma, const float *__restrict__ mean_invstd,
                                       float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ coef, int frozen)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < YOLO_BN_ACC_REPLICAS; ++r) {
        s1 += acc[(size_t)r * 2 * C + c];
        s2 += acc[(size_t)r * 2 * C + C + c];
        acc[(size_t)r * 2 * C + c] = 0.0;
        acc[(size_t)r * 2 * C + C + c] = 0.0;
    }
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    coef[c] = gamma[c] * mean_invstd[C + c];
    // frozen statistics (eval mode): mean and variance do not depend on z, so dz = gamma * invstd * dy' -- the two batch terms drop out
    coef[C + c] = frozen ? 0.0f : (float)(s1 / count);
    coef[2 * C + c] = frozen ? 0.0f : (float)(s2 / count);
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(bf16_t *__restrict__ dy, int dy_halo, const bf16_t *__restrict__ yact, int y_halo,
                                                           const bf16_t *__restrict__ z, int z_halo, int N, int H, int W, int C,
                                                           const float *__restrict__ mean_invstd, const float *__restrict__ coef, bf16_t *__restrict__ dz,
                                                           long dz_img, long dz_row, long dz_px, long dz_off, int store_masked, int mask_from_z)
{
    const int C8 = C >> 3;
    const long total = (long)N * H * W * C8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % C8);
    const long p = idx / C8;
    const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((long)W * H));
    bf16_t *gq = dy + (((long)n * (H + 2 * dy_halo) + y + dy_halo) * (W + 2 * dy_halo) + x + dy_halo) * C + cg * 8;
    float g[8], zz[8], a[8];
    bn_unpack8(*reinterpret_cast<const uint4 *>(gq), g);
    bn_unpack8(*reinterpret_cast<const uint4 *>(z + (((long)n * (H + 2 * z_halo) + y + z_halo) * (W + 2 * z_halo) + x + z_halo) * C + cg * 8), zz);
    if (yact) {
        bn_unpack8(*reinterpret_cast<const uint4 *>(yact + (((long)n * (H + 2 * y_halo) + y + y_halo) * (W + 2 * y_halo) + x + y_halo) * C + cg * 8), a);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = a[k] > 0.0f ? g[k] : 0.0f;
    } else if (mask_from_z) {
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = fmaf(zz[k], mean_invstd[2 * C + cg * 8 + k], mean_invstd[3 * C + cg * 8 + k]) > 0.0f ? g[k] : 0.0f;
    }
    unsigned o[4], m[4];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const int c = cg * 8 + k;
        const float x0 = (zz[k] - mean_invstd[c]) * mean_invstd[C + c], x1 = (zz[k + 1] - mean_invstd[c + 1]) * mean_invstd[C + c + 1];
        const float d0 = coef[c] * (g[k] - coef[C + c] - x0 * coef[2 * C + c]);
        const float d1 = coef[c + 1] * (g[k + 1] - coef[C + c + 1] - x1 * coef[2 * C + c + 1]);
        o[k >> 1] = (unsigned)f32_to_bf16(d0) | ((unsigned)f32_to_bf16(d1) << 16);
        m[k >> 1] = (unsigned)f32_to_bf16(g[k]) | ((unsigned)f32_to_bf16(g[k + 1]) << 16);
    }
    *reinterpret_cast<uint4 *>(dz + (long)n * dz_img + (long)y * dz_row + (long)x * dz_px + dz_off + cg * 8) = uint4{o[0], o[1], o[2], o[3]};
    if (store_masked) *reinterpret_cast<uint4 *>(gq) = uint4{m[0], m[1], m[2], m[3]};
}

}  // namespace yolo

using namespace yolo;

YOLO_API int yolo_batchnorm_train_fwd(void *z, int N, int H, int W, int C, int halo, const float *gamma, const float *beta, double eps, double momentum,
                                      float *running_mean, float *running_var, const void *residual, int residual_halo, int relu, double *acc2c,
                                      float *scale_shift, void *out, int out_halo, float *save_mean_invstd, int stats_ready, yolo_stream_t stream)
{
    if (!z || !gamma || !beta || !acc2c || !scale_shift || N <= 0 || H <= 0 || W <= 0 || C <= 0 || halo < 0 || residual_halo < 0 || out_halo < 0)
        return fail(YOLO_E_ARG, "yolo_batchnorm_train_fwd: bad argument");
    if (C & 7) return fail(YOLO_E_UNSUPPORTED, "yolo_batchnorm_train_fwd: C = %d must be a multiple of 8", C);
    if ((running_mean == nullptr) != (running_var == nullptr)) return fail(YOLO_E_ARG, "yolo_batchnorm_train_fwd: running_mean and running_var go together");
    if (stats_ready < 0 || stats_ready > 2 || (stats_ready == 2 && !running_mean))
        return fail(YOLO_E_ARG, "yolo_batchnorm_train_fwd: stats_ready is 0, 1 or 2 (2 = normalise with the running statistics, which must be given)");
    hipStream_t s = STRM(stream);
    const int C8 = C / 8, gpb = C8 < 256 ? C8 : 256, ppb = 256 / gpb;
    const long P = (long)N * H * W;
    long gy = (P + (long)ppb * 32 - 1) / ((long)ppb * 32);     // ~32 pixels per thread
    if (gy > 2048) gy = 2048;
    if (gy < 1) gy = 1;
    if (!stats_ready) {
        hipLaunchKernelGGL(bn_stats_kernel, dim3((C8 + gpb - 1) / gpb, (unsigned)gy), dim3(256), 0, s, (const bf16_t *)z, N, H, W, C, halo, acc2c);
        if (int rc = check_launch("yolo_batchnorm_train_fwd(stats)")) return rc;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, acc2c, C, (double)P, gamma, beta, eps, momentum, running_mean, running_var, scale_shift,
                       scale_shift + C, save_mean_invstd, stats_ready == 2 ? 1 : 0);
    if (int rc = check_launch("yolo_batchnorm_train_fwd(finalize)")) return rc;
    const long total = P * C8;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (bf16_t *)z, N, H, W, C, halo, scale_shift, scale_shift + C,
                       (const bf16_t *)residual, residual_halo, relu, (bf16_t *)out, out_halo);
    return check_launch("yolo_batchnorm_train_fwd(apply)");
}

YOLO_API int yolo_batchnorm_bwd(void *dy, int dy_halo, const void *y, int y_halo, const void *z, int z_halo, int N, int H, int W, int C, const float *gamma,
                                const float *mean_invstd, void *dz, long dz_img_stride, long dz_row_stride, long dz_px_stride, long dz_off, int store_masked_dy,
                                int relu_from_z, float *dgamma, float *dbeta, double *acc2c, float *coef3c, yolo_stream_t stream)
{
    const int frozen = (relu_from_z >> 1) & 1;         // bit 1: the forward normalised with running statistics (stats_ready = 2)
    relu_from_z &= 1;
    if (!dy || !z || !gamma || !mean_invstd || !dz || !dgamma || !dbeta || !acc2c || !coef3c || N <= 0 || H <= 0 || W <= 0 || C <= 0 || dy_halo < 0 || y_halo < 0 ||
        z_halo < 0 || (relu_from_z && y))
        return fail(YOLO_E_ARG, "yolo_batchnorm_bwd: bad argument (relu_from_z excludes y)");
    if ((C & 7) || (dz_img_stride & 7) || (dz_row_stride & 7) || (dz_px_stride & 7) || (dz_off & 7))
        return fail(YOLO_E_UNSUPPORTED, "yolo_batchnorm_bwd: C = %d and the dz strides must be multiples of 8", C);
    hipStream_t s = STRM(stream);
    const int C8 = C / 8, gpb = C8 < 256 ? C8 : 256, ppb = 256 / gpb;
    const long P = (long)N * H * W;
    long gy = (P + (long)ppb * 32 - 1) / ((long)ppb * 32);
    if (gy > 2048) gy = 2048;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((C8 + gpb - 1) / gpb, (unsigned)gy), dim3(256), 0, s, (const bf16_t *)dy, dy_halo, (const bf16_t *)y, y_halo,
                       (const bf16_t *)z, z_halo, N, H, W, C, mean_invstd, acc2c, relu_from_z);
    if (int rc = check_launch("yolo_batchnorm_bwd(reduce)")) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, acc2c, C, (double)P, gamma, mean_invstd, dgamma, dbeta, coef3c, frozen);
    if (int rc = check_launch("yolo_batchnorm_bwd(finalize)")) return rc;
    const long total = P * C8;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (bf16_t *)dy, dy_halo, (const bf16_t *)y, y_halo, (const bf16_t *)z,
                       z_halo, N, H, W, C, mean_invstd, coef3c, (bf16_t *)dz, dz_img_stride, dz_row_stride, dz_px_stride, dz_off, store_masked_dy, relu_from_z);
    return check_launch("yolo_batchnorm_bwd(apply)");
}
